"""CPU-side checks (run without a GPU): the C-ABI library loads and exports every symbol include/mi355x_asr.h declares,
the ctypes structures mirror the C structs, the drop-in classes keep the reference's constructor / state-dict / typing
contract, flat parameter storage + packing plans are consistent, LR schedule and config plumbing behave like the
reference.  No kernel is launched here."""
import ctypes
import os
import re
import subprocess

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from nemo_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "mi355x_asr.h")).read()
    declared = set(re.findall(r"\b(mi355x_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    lib = ctypes.CDLL(_lib.LIB_PATH)
    missing = [s for s in sorted(declared) if not hasattr(lib, s)]
    assert not missing, missing
    assert declared == set(_lib.EXPORTED_SYMBOLS), declared ^ set(_lib.EXPORTED_SYMBOLS)
    assert _lib.version().startswith("mi355x_asr")


def test_struct_layouts_match_the_header():
    """compile a tiny C program against the header and compare sizeof/offsetof with the ctypes mirrors"""
    from nemo_amd._lib import GemmDesc, PackEntry
    src = r'''
#include <stdio.h>
#include <stddef.h>
#include "mi355x_asr.h"
int main(){
  printf("%zu %zu %zu %zu %zu %zu %zu\n", sizeof(mi355x_gemm_desc), offsetof(mi355x_gemm_desc, c_col_stride),
         offsetof(mi355x_gemm_desc, bias), offsetof(mi355x_gemm_desc, aux_in), offsetof(mi355x_gemm_desc, drop_key),
         offsetof(mi355x_gemm_desc, row_len), offsetof(mi355x_gemm_desc, colsum_out));
  printf("%zu\n", offsetof(mi355x_gemm_desc, colsum_stride));
  printf("%zu %zu\n", sizeof(mi355x_pack_entry), offsetof(mi355x_pack_entry, tile_begin));
  return 0; }'''
    import tempfile
    with tempfile.TemporaryDirectory() as tmp:
        c = os.path.join(tmp, "t.c")
        open(c, "w").write(src)
        exe = os.path.join(tmp, "t")
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), c, "-o", exe])
        out = subprocess.check_output([exe]).decode().split()
    vals = list(map(int, out))
    G = GemmDesc
    assert vals[:7] == [ctypes.sizeof(G), G.c_col_stride.offset, G.bias.offset, G.aux_in.offset, G.drop_key.offset,
                        G.row_len.offset, G.colsum_out.offset]
    assert vals[7] == G.colsum_stride.offset
    assert vals[8:] == [ctypes.sizeof(PackEntry), PackEntry.tile_begin.offset]


def test_invalid_arguments_are_rejected_without_a_gpu():
    """argument validation happens before any launch: rc 1 -> ValueError (the reference raises ValueError for bad shapes)"""
    from nemo_amd import _lib
    d = _lib.GemmDesc()  # all-zero descriptor
    assert _lib.lib.mi355x_gemm(ctypes.byref(d), None) == 1
    with pytest.raises(ValueError):
        _lib.check(1, "gemm")
    with pytest.raises(RuntimeError):
        _lib.check(1000 + 98, "gemm")
    assert _lib.lib.mi355x_ctc_loss(None, None, None, None, None, None, None, None, 1, 1, 1, 1, 0, 1.0, 1, None) == 1
    assert _lib.lib.mi355x_layernorm_fwd(None, 0, None, None, None, 0, None, None, 4, 6, 1e-5, None) == 1


def test_cpu_tensors_fail_loudly():
    from nemo_amd import ops
    x = torch.zeros(4, 8)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.layernorm_fwd(x, torch.ones(8), torch.zeros(8), x.clone(), torch.zeros(4), torch.zeros(4), 4, 8)


def test_encoder_state_dict_is_the_reference_abi(golden_dir):
    """keys / shapes identical to the reference ConformerEncoder's state_dict (fixture written by the reference classes)"""
    from nemo_amd.modules import ConformerEncoder, ConvASRDecoder
    z = np.load(os.path.join(golden_dir, "ref_tiny_model.npz"))
    ref = {k[len("P/encoder."):]: z[k].shape for k in z.files if k.startswith("P/encoder.")}
    enc = ConformerEncoder(feat_in=80, n_layers=2, d_model=32, n_heads=4)
    mine = {k: tuple(v.shape) for k, v in enc.state_dict().items()}
    assert mine == {k: tuple(v) for k, v in ref.items()}
    dec = ConvASRDecoder(feat_in=32, num_classes=16)
    assert {k: tuple(v.shape) for k, v in dec.state_dict().items()} == {
        "decoder_layers.0.weight": (17, 32, 1), "decoder_layers.0.bias": (17,)}
    assert dec.num_classes_with_blank == 17 and dec._feat_in == 32 and enc._feat_out == 32 and enc.subsampling_factor == 4


def test_preprocessor_buffers_and_seq_len(golden_dir):
    from nemo_amd.modules import AudioToMelSpectrogramPreprocessor
    z = np.load(os.path.join(golden_dir, "ref_mel_b3.npz"))
    p = AudioToMelSpectrogramPreprocessor(sample_rate=16000, window_size=0.025, window_stride=0.01, features=80, n_fft=512,
                                          pad_to=0, dither=1e-5)
    sd = p.state_dict()
    assert set(sd) == {"featurizer.window", "featurizer.fb"}  # persistent buffers of the reference (features.py:327,344)
    assert np.array_equal(sd["featurizer.fb"].numpy(), z["fb"])
    assert np.allclose(sd["featurizer.window"].numpy(), z["window"], atol=1e-7)
    assert p.featurizer.dither == 1e-5 and p.featurizer.pad_to == 0 and p._sample_rate == 16000
    # get_seq_len: features.py:413-417 / test_asr_filterbankfeatures_seq_len.py
    assert p.featurizer.get_seq_len(torch.from_numpy(z["audio_len"])).tolist() == z["mel_len"].tolist()
    with pytest.raises(ValueError):
        AudioToMelSpectrogramPreprocessor(window_size=0.02, n_window_size=320)


def test_typecheck_contract():
    from nemo_amd.core import typecheck
    from nemo_amd.modules import ConvASRDecoder, CTCLoss
    dec = ConvASRDecoder(feat_in=8, num_classes=4)
    with pytest.raises(TypeError, match="kwargs only"):
        dec(torch.zeros(1, 8, 3))
    with pytest.raises(TypeError, match="no corresponding input_type"):
        dec(encoder_outputs=torch.zeros(1, 8, 3))
    with pytest.raises(TypeError, match="shape mismatch"):
        dec(encoder_output=torch.zeros(8, 3))
    with pytest.raises(ValueError):
        CTCLoss(num_classes=4, reduction="bogus")
    assert list(dec.input_types) == ["encoder_output"] and list(dec.output_types) == ["logprobs"]


def test_unsupported_configurations_raise():
    from nemo_amd.modules import ConformerEncoder
    for kw in (dict(subsampling="vggnet"), dict(subsampling="striding", subsampling_factor=8), dict(self_attention_model="abs_pos"), dict(conv_norm_type="instance_norm"),
               dict(reduction="pooling"), dict(self_attention_model="rel_pos_local_attn", att_context_size=[6, 6], global_tokens=2)):
        with pytest.raises(NotImplementedError):
            ConformerEncoder(feat_in=80, n_layers=1, d_model=32, n_heads=4, **kw)
    # causal down-sampling (CausalConv2D: two zero rows / columns in front, one behind) changes the sampling grid: 80 -> 41 -> 21 bins
    enc = ConformerEncoder(feat_in=80, n_layers=1, d_model=32, n_heads=4, causal_downsampling=True)
    assert enc.pre_encode._pad == 2 and enc.pre_encode._feat_after == 21 and tuple(enc.pre_encode.out.weight.shape) == (32, 32 * 21)
    assert [int(x) for x in enc._lens(torch.tensor([101, 40, 0]), 2)[-1]] == [26, 11, 1]   # floor(n / 2) + 1, twice
    enc = ConformerEncoder(feat_in=80, n_layers=1, d_model=32, n_heads=4, self_attention_model="rel_pos_local_attn", att_context_size=[6, 6])
    assert enc.att_context_style == "regular" and enc.att_context_size == [6, 6]
    # options that ARE implemented are accepted and validated like the reference (conformer_encoder.py:863-894)
    enc = ConformerEncoder(feat_in=80, n_layers=2, d_model=32, n_heads=4, att_context_size=[128, 0], feat_out=16, stochastic_depth_drop_prob=0.5)
    assert enc.att_context_size == [128, 0] and enc._feat_out == 16 and enc.layer_drop_probs == [0.0, 0.5] and not enc._flash_ok()
    with pytest.raises(ValueError):
        ConformerEncoder(feat_in=80, n_layers=1, d_model=32, n_heads=4, att_context_size=[[8, 3], [4, 1]], att_context_probs=[0.5, 0.6])
    with pytest.raises(ValueError):
        ConformerEncoder(feat_in=80, n_layers=2, d_model=32, n_heads=4, stochastic_depth_drop_prob=1.0)
    enc = ConformerEncoder(feat_in=80, n_layers=1, d_model=32, n_heads=4, conv_kernel_size=9, conv_norm_type="layer_norm", conv_context_size="causal")
    assert enc.conv_context_size == [8, 0] and enc.conv_pad_left == 8 and isinstance(enc.layers[0].conv.batch_norm, torch.nn.LayerNorm)
    with pytest.raises(ValueError):
        ConformerEncoder(feat_in=80, n_layers=1, d_model=32, n_heads=4, conv_kernel_size=9, conv_context_size=[5, 2])


def test_from_config_dict_resolves_reference_targets():
    from nemo_amd.core import Serialization
    from nemo_amd.modules import ConformerEncoder
    enc = Serialization.from_config_dict({"_target_": "nemo.collections.asr.modules.ConformerEncoder", "feat_in": 80,
                                          "n_layers": 1, "d_model": 32, "n_heads": 4})
    assert isinstance(enc, ConformerEncoder) and enc.to_config_dict()["d_model"] == 32
    with pytest.raises(ValueError):
        Serialization.from_config_dict({"_target_": "os.system"})


def test_noam_annealing_matches_reference_formula():
    from nemo_amd.optim import NoamAnnealing
    s = NoamAnnealing(base_lr=2.0, d_model=512, warmup_steps=10000, min_lr=1e-6)
    for step in (1, 10, 9999, 10000, 10001, 400000):
        ref = 2.0 * 512 ** -0.5 * min(step ** -0.5, step * 10000 ** -1.5)
        if step > 10000:
            ref = max(ref, 1e-6)
        assert abs(s.lr_at(step) - ref) < 1e-12
    assert s.step() == s.lr_at(1)


def test_flat_params_alias_parameters_and_grads():
    from nemo_amd.flat import FlatParams
    m = torch.nn.Sequential(torch.nn.Linear(5, 7), torch.nn.LayerNorm(7))
    before = {n: p.detach().clone() for n, p in m.named_parameters()}
    fp = FlatParams(m)
    fp.ensure()
    for n, p in m.named_parameters():
        off, num = fp.offsets[n]
        assert off % 64 == 0 and torch.equal(p.detach(), before[n])
        assert p.data_ptr() == fp.flat.data_ptr() + 4 * off and p.grad.data_ptr() == fp.grad.data_ptr() + 4 * off
    fp.flat.zero_()
    assert all(float(p.abs().sum()) == 0 for p in m.parameters())
    s, e = fp.range_of("1.")
    assert (s, e) == (fp.offsets["1.weight"][0], fp.offsets["1.bias"][0] + 64)
    gen = fp.generation
    m.to(torch.float32)
    fp.ensure()
    assert fp.generation == gen  # still valid: no rebuild


def test_model_config_and_nemo_file_roundtrip(tmp_path):
    from nemo_amd.core import load_nemo
    from nemo_amd.models import EncDecCTCModel, conformer_ctc_config
    cfg = conformer_ctc_config("small", vocab_size=16, n_layers=1, d_model=32)
    m = EncDecCTCModel(cfg)
    assert m.loss.blank == 16 and m.decoder._feat_in == 32
    path = str(tmp_path / "x.nemo")
    m.save_to(path)
    cfg2, sd = load_nemo(path)
    assert cfg2["encoder"]["d_model"] == 32 and set(sd) == set(m.state_dict())
    m2 = EncDecCTCModel.restore_from(path)
    for k, v in m.state_dict().items():
        assert torch.equal(v, m2.state_dict()[k]), k
    # the recipe's spec_augment section instantiates the drop-in SpectrogramAugmentation (ctc_models.py:86-89)
    aug = conformer_ctc_config("small", vocab_size=16, n_layers=1, d_model=32)
    aug["spec_augment"] = {"_target_": "nemo.collections.asr.modules.SpectrogramAugmentation", "freq_masks": 2,
                           "time_masks": 10, "freq_width": 27, "time_width": 0.05}
    m3 = EncDecCTCModel(aug)
    assert type(m3.spec_augmentation).__name__ == "SpectrogramAugmentation" and m3.spec_augmentation.time_masks == 10
    assert set(m3.state_dict()) == set(m.state_dict())  # no parameters / buffers of its own


# ---------------------------------------------------------------------------------------------- SpecAugment host logic
def test_specaug_module_draws_the_reference_masks(golden_dir):
    """the drop-in module's mask parameters (device-agnostic host code, here on CPU tensors) reproduce the cells the
    reference classes masked at the same seeds -- vectorised (torch generator), legacy and cut-out (python rng) modes"""
    import random
    import numpy as np
    from nemo_amd.modules import SpectrogramAugmentation
    from oracle import specaug_ref as SR
    z = np.load(os.path.join(golden_dir, "ref_specaug.npz"))
    x, length = torch.from_numpy(z["x"]), torch.from_numpy(z["length"])
    B, F, T = x.shape

    def masked(groups):
        y = x
        for rects, value in groups:
            y = SR.apply_rects(y, rects, value)
        return (y != x).numpy()

    def want(name):
        return np.unpackbits(z[name + "_mask"])[: x.numel()].reshape(tuple(x.shape)).astype(bool)

    torch.manual_seed(2024)
    m = SpectrogramAugmentation(freq_masks=2, time_masks=10, freq_width=27, time_width=0.05)
    assert np.array_equal(masked(m.mask_rects(B, F, T, length, "cpu")), want("vec_adaptive"))
    torch.manual_seed(2025)
    m = SpectrogramAugmentation(freq_masks=3, time_masks=4, freq_width=15, time_width=40, mask_value=-1.5)
    assert np.array_equal(masked(m.mask_rects(B, F, T, length, "cpu")), want("vec_int"))
    m = SpectrogramAugmentation(freq_masks=2, time_masks=5, freq_width=27, time_width=0.05, rng=random.Random(7),
                                use_vectorized_spec_augment=False)
    assert np.array_equal(masked(m.mask_rects(B, F, T, length, "cpu")), want("legacy"))
    m = SpectrogramAugmentation(freq_masks=1, time_masks=2, freq_width=10, time_width=25, rect_masks=5, rect_time=60,
                                rect_freq=20, rng=random.Random(11), use_vectorized_spec_augment=False)
    assert np.array_equal(masked(m.mask_rects(B, F, T, length, "cpu")), want("cutout_then_legacy"))
    with pytest.raises(ValueError):
        SpectrogramAugmentation(time_masks=1, time_width=1.5)


# ---------------------------------------------------------------------------------------------- flat buffers / launch planning
def test_flat_params_tail_region_and_ranges():
    """FlatParams(tail=...): selected parameters are laid out contiguously AFTER all others (equal strides across layers ->
    one batched GEMM writes all their gradients), `range_of` skips them, `tail_range` covers exactly them."""
    from nemo_amd.flat import FlatParams

    class Layer(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.a = torch.nn.Linear(8, 16)
            self.special = torch.nn.Linear(16, 16, bias=False)
            self.b = torch.nn.Linear(16, 8)

    net = torch.nn.ModuleDict({"layers": torch.nn.ModuleList([Layer() for _ in range(3)]), "head": torch.nn.Linear(8, 4)})
    ref = {k: v.detach().clone() for k, v in net.state_dict().items()}
    fp = FlatParams(net, tail=lambda n: n.endswith("special.weight"))
    fp.build("cpu")
    assert fp.is_valid()
    for k, v in net.state_dict().items():            # values and names untouched (state-dict ABI)
        assert torch.equal(v, ref[k])
    offs = [fp.offsets[f"layers.{i}.special.weight"][0] for i in range(3)]
    assert offs[1] - offs[0] == offs[2] - offs[1] == 256            # equally spaced, 64-element aligned
    lo, hi = fp.tail_range()
    assert lo == offs[0] and hi == offs[2] + 256 and hi == fp.flat.numel()
    others_end = max(o + n for k, (o, n) in fp.offsets.items() if not k.endswith("special.weight"))
    assert others_end <= lo
    for i in range(3):                                # a layer's range is contiguous and excludes the tail parameters
        s0, s1 = fp.range_of(f"layers.{i}.")
        assert s1 <= lo and s1 - s0 == sum(-(-n // 64) * 64 for k, (o, n) in fp.offsets.items()
                                            if k.startswith(f"layers.{i}.") and not k.endswith("special.weight"))
    # gradients are views of ONE buffer too
    net["layers"][1].special.weight.grad.fill_(2.0)
    assert torch.all(fp.grad[offs[1]: offs[1] + 256] == 2.0)


def test_splitk_chooser_fills_rounds_of_the_cus():
    from nemo_amd.modules.conformer_encoder import ConformerEncoder
    f = ConformerEncoder._splitk
    for tiles, K in [(184, 16032), (32, 16032), (72, 320640), (8, 16032), (16, 16032), (300, 16032), (1, 64 * 40)]:
        c = f(tiles, K)
        nk = (K + 63) // 64
        assert 1 <= c <= 16 and (c == 1 or nk // c >= 16)
        blocks = tiles * c
        eff = blocks / (-(-blocks // 256) * 256)
        assert eff >= 0.70 or c == 1 or nk // (c + 1) < 16, (tiles, K, c, eff)
    assert f(184, 16032) == 4 and f(32, 16032) == 8 and f(72, 320640) == 7
    assert f(160, 16032, strided_c=True) == 1      # column-strided outputs: a single round, no extra atomic passes



def test_change_vocabulary_rebuilds_decoder_loss_and_decoding():
    """EncDecCTCModel.change_vocabulary (ctc_models.py:190-262): new decoder over the new alphabet, blank = its length, the
    encoder untouched, dataset configs updated; same vocabulary is a no-op; an empty one is a ValueError"""
    from nemo_amd.models import EncDecCTCModel, conformer_ctc_config
    cfg = conformer_ctc_config("small", vocab_size=5, d_model=32, n_heads=2, n_layers=1)
    cfg["decoder"]["vocabulary"] = list("abcde")
    cfg["train_ds"] = {"manifest_filepath": None, "labels": list("abcde")}
    model = EncDecCTCModel(cfg)
    enc_keys = {k: v.clone() for k, v in model.encoder.state_dict().items()}
    old_dec = model.decoder
    model.change_vocabulary(list("abcde"))
    assert model.decoder is old_dec
    with pytest.raises(ValueError):
        model.change_vocabulary([])
    new_vocab = [" ", "x", "y", "z", "q", "r", "s"]
    model.change_vocabulary(new_vocab)
    assert model.decoder is not old_dec and model.decoder.vocabulary == new_vocab
    assert model.decoder.num_classes_with_blank == 8 and model.loss.blank == 7
    assert tuple(model.decoder.state_dict()["decoder_layers.0.weight"].shape) == (8, 32, 1)
    assert all(torch.equal(v, model.encoder.state_dict()[k]) for k, v in enc_keys.items())
    assert model._cfg["decoder"]["num_classes"] == 7 and model._cfg["train_ds"]["labels"] == new_vocab
    assert model.wer.decoding.blank_id == 7 and model.wer.decoding.vocabulary == new_vocab
    assert model._optimizer is None


def _train_spm(tmp_path, name, vocab_size, seed_words):
    import sentencepiece as spm
    corpus = tmp_path / f"{name}.txt"
    rs = np.random.RandomState(len(seed_words))
    with open(corpus, "w") as f:
        for _ in range(400):
            f.write(" ".join(rs.choice(seed_words, size=rs.randint(3, 9))) + "\n")
    d = tmp_path / name
    d.mkdir()
    spm.SentencePieceTrainer.train(input=str(corpus), model_prefix=str(d / "tokenizer"), vocab_size=vocab_size,
                                   model_type="bpe", character_coverage=1.0, bos_id=-1, eos_id=-1, minloglevel=2)
    return str(d)


def test_bpe_model_builds_from_a_tokenizer_dir_and_carries_it_in_the_nemo_file(tmp_path):
    """EncDecCTCModelBPE (ctc_bpe_models.py:39-110): decoder vocabulary = the SentencePiece pieces in id order, placeholder
    num_classes replaced, BPE dataset selected, tokenizer packed into / restored from the .nemo archive as an artifact
    (modelPT.register_artifact), change_vocabulary with a new tokenizer directory"""
    import tarfile
    from nemo_amd.core import resolve_target
    from nemo_amd.models import EncDecCTCModelBPE, conformer_ctc_config
    words = ["speech", "recognition", "conformer", "attention", "the", "a", "of", "spectrogram", "frame", "token"]
    tok_dir = _train_spm(tmp_path, "tok32", 32, words)
    cfg = conformer_ctc_config("small", vocab_size=-1, d_model=32, n_heads=2, n_layers=1)
    cfg["decoder"]["num_classes"] = -1
    with pytest.raises(ValueError):
        EncDecCTCModelBPE(cfg)
    cfg["tokenizer"] = {"dir": tok_dir, "type": "bpe"}
    assert resolve_target("nemo.collections.asr.models.EncDecCTCModelBPE") is EncDecCTCModelBPE
    model = EncDecCTCModelBPE(cfg)
    pieces = model.tokenizer.vocab
    assert len(pieces) == 32 and model.decoder.vocabulary == pieces
    assert model.decoder.num_classes_with_blank == 33 and model.loss.blank == 32
    ids = model.tokenizer.text_to_ids("the conformer attention")
    assert model.wer.decoding.ids_to_text(ids) == "the conformer attention"
    # the BPE dataset is chosen and tokenises the transcripts
    import json, wave
    with wave.open(str(tmp_path / "u.wav"), "wb") as f:
        f.setnchannels(1); f.setsampwidth(2); f.setframerate(16000); f.writeframes(np.zeros(1600, dtype=np.int16).tobytes())
    with open(tmp_path / "m.json", "w") as f:
        f.write(json.dumps(dict(audio_filepath="u.wav", duration=0.1, text="the frame of a token")) + "\n")
    dl = model.setup_training_data(dict(manifest_filepath=str(tmp_path / "m.json"), batch_size=1, shuffle=False))
    _, _, tok, tl = next(iter(dl))
    assert tok[0, : int(tl[0])].tolist() == model.tokenizer.text_to_ids("the frame of a token")
    # data-set kinds / augmentations the input side does not provide are refused by name (they change what is trained on)
    for bad in (dict(is_tarred=True), dict(use_lhotse=True), dict(is_concat=True), dict(augmentor=dict(speed=dict(prob=0.5)))):
        with pytest.raises(NotImplementedError, match=list(bad)[0]):
            model.setup_training_data(dict(manifest_filepath=str(tmp_path / "m.json"), batch_size=1, shuffle=False, **bad))
    model.setup_training_data(dict(manifest_filepath=str(tmp_path / "m.json"), batch_size=1, shuffle=False, is_tarred=False, augmentor=None))
    # .nemo: three members, the tokenizer among them under a content-hash name; restore resolves it again
    path = str(tmp_path / "bpe.nemo")
    model.save_to(path)
    with tarfile.open(path) as tar:
        names = [m.name for m in tar.getmembers()]
    assert len(names) == 3 and any(n.endswith("_tokenizer.model") for n in names)
    m2 = EncDecCTCModelBPE.restore_from(path)
    assert m2.tokenizer.vocab == pieces and m2._cfg["tokenizer"]["model_path"] != cfg["tokenizer"].get("model_path")
    assert all(torch.equal(v, m2.state_dict()[k]) for k, v in model.state_dict().items())
    # a new tokenizer: decoder / loss / decoding follow, the encoder stays
    enc = {k: v.clone() for k, v in model.encoder.state_dict().items()}
    model.change_vocabulary(_train_spm(tmp_path, "tok48", 48, words + ["transducer", "gradient"]))
    assert model.decoder.num_classes_with_blank == 49 and model.loss.blank == 48 and len(model.tokenizer.vocab) == 48
    assert all(torch.equal(v, model.encoder.state_dict()[k]) for k, v in enc.items())


def test_trainable_ranges_skip_frozen_parameters():
    """torch.optim.AdamW skips parameters without a gradient; the fused optimizer steps FlatParams.trainable_ranges() only:
    frozen parameters (requires_grad=False / .freeze()) are excluded, adjacent trainable ones merge into one launch"""
    from nemo_amd.flat import ALIGN, FlatParams
    m = torch.nn.Sequential(torch.nn.Linear(70, 3), torch.nn.Linear(3, 5), torch.nn.Linear(5, 2))
    fp = FlatParams(m)
    fp.build()
    total = fp.flat.numel()
    assert fp.trainable_ranges() == [(0, total)]            # nothing frozen: one range, one launch
    m[1].weight.requires_grad_(False)
    r = fp.trainable_ranges()
    lo, n = fp.offsets["1.weight"]
    hi = lo + (n + ALIGN - 1) // ALIGN * ALIGN
    assert r == [(0, lo), (hi, total)]
    for p in m.parameters():
        p.requires_grad_(False)
    assert fp.trainable_ranges() == []                      # fully frozen module: no update, no weight decay


def test_gzip_compressed_nemo_archives_load(tmp_path):
    """older .nemo checkpoints are tar.gz (save_restore_connector.py:684-694 falls back from 'r:' to 'r:gz')"""
    import io, tarfile, yaml
    from nemo_amd.core import MODEL_CONFIG_YAML, MODEL_WEIGHTS, load_nemo
    cfg = {"sample_rate": 16000, "encoder": {"d_model": 8}}
    sd = {"w": torch.arange(6, dtype=torch.float32).view(2, 3)}
    for mode, name in (("w:gz", "old.nemo"), ("w:", "new.nemo")):
        path = str(tmp_path / name)
        with tarfile.open(path, mode) as tar:
            for member, blob in ((MODEL_CONFIG_YAML, yaml.safe_dump(cfg).encode()), (MODEL_WEIGHTS, None)):
                if blob is None:
                    b = io.BytesIO(); torch.save(sd, b); blob = b.getvalue()
                ti = tarfile.TarInfo("./" + member); ti.size = len(blob)
                tar.addfile(ti, io.BytesIO(blob))
        c2, s2 = load_nemo(path)
        assert c2 == cfg and torch.equal(s2["w"], sd["w"])


def test_scheduler_follows_the_lightning_order():
    """Lightning calls scheduler.step() AFTER optimizer.step(): optimizer step n runs with lr(max(1, n-1)) of the Noam
    formula (the _LRScheduler constructor leaves last_epoch at 0, lr_scheduler.py:518-576 clamps the step to >= 1)"""
    from nemo_amd.optim import NoamAnnealing
    s = NoamAnnealing(2.0, d_model=512, warmup_steps=1000, min_lr=1e-6)
    used = []
    for _ in range(4):          # what fit_step does: read, (optimizer step), advance
        used.append(s.get_last_lr())
        s.step()
    assert used == [s.lr_at(1), s.lr_at(1), s.lr_at(2), s.lr_at(3)]


def test_dw_striding_encoder_state_dict_is_the_reference_abi(golden_dir):
    """FastConformer / Squeezeformer sub-sampling ('dw_striding'): parameter names and shapes of the drop-in encoder equal the
    reference ConformerEncoder's (the fixture holds the reference state-dict, tests/golden/ref_fastconformer_tiny.npz)"""
    from nemo_amd.modules import ConformerEncoder
    z = np.load(os.path.join(golden_dir, "ref_fastconformer_tiny.npz"))
    enc = ConformerEncoder(feat_in=40, n_layers=2, d_model=32, subsampling="dw_striding", subsampling_factor=8,
                           subsampling_conv_channels=16, n_heads=4, conv_kernel_size=9)
    ours = {k: tuple(v.shape) for k, v in enc.state_dict().items()}
    ref = {k[2:]: tuple(z[k].shape) for k in z.files if k.startswith("P.") and "pos_enc" not in k}
    assert ours == ref, (set(ours) ^ set(ref), [k for k in ours if k in ref and ours[k] != ref[k]])
    with pytest.raises(NotImplementedError):
        ConformerEncoder(feat_in=40, n_layers=1, d_model=32, subsampling="dw_striding", subsampling_factor=2)


def test_config_loader_interpolation_yaml12_floats_missing_and_overrides():
    """nemo_amd.config on an inline document (runs everywhere; the reference's own recipe files are loaded by
    tests/test_boundary_reference.py where the reference tree exists)"""
    from nemo_amd.config import MissingMandatoryValue, apply_overrides, load_config, missing_keys, resolve, select
    text = """
name: demo
model:
  sample_rate: 16000
  train_ds:
    manifest_filepath: ???
    sample_rate: ${model.sample_rate}
    batch_size: 8
  preprocessor: {features: 80, window_size: 0.025}
  encoder: {feat_in: "${model.preprocessor.features}", d_model: 512, label: "d${model.encoder.d_model}_f${model.preprocessor.features}"}
  optim: {lr: 2.0, weight_decay: 1e-3, sched: {d_model: "${model.encoder.d_model}", min_lr: 1e-6, warmup_ratio: null}}
trainer: {devices: -1, precision: 32}
"""
    c = load_config(text=text)
    m = c["model"]
    assert m["train_ds"]["sample_rate"] == 16000 and m["encoder"]["feat_in"] == 80 and m["optim"]["sched"]["d_model"] == 512
    assert m["encoder"]["label"] == "d512_f80"                       # interpolation inside a string
    assert m["optim"]["weight_decay"] == 1e-3 and isinstance(m["optim"]["weight_decay"], float)   # YAML 1.2: no dot needed
    assert m["optim"]["sched"]["min_lr"] == 1e-6 and m["optim"]["sched"]["warmup_ratio"] is None
    assert c["trainer"]["devices"] == -1 and isinstance(c["trainer"]["precision"], int)
    assert missing_keys(c) == ["model.train_ds.manifest_filepath"]
    with pytest.raises(MissingMandatoryValue):
        select(c, "model.train_ds.manifest_filepath", throw_on_missing=True)
    c2 = load_config(text=text, overrides=["model.train_ds.manifest_filepath=/data/train.json", "model.encoder.d_model=256",
                                           "+trainer.fast_dev_run=True", "~model.preprocessor.window_size", "model.optim.lr=1e-3"])
    assert missing_keys(c2) == [] and c2["model"]["optim"]["sched"]["d_model"] == 256 and c2["model"]["encoder"]["label"] == "d256_f80"
    assert c2["trainer"]["fast_dev_run"] is True and "window_size" not in c2["model"]["preprocessor"]
    assert c2["model"]["optim"]["lr"] == 1e-3
    with pytest.raises(KeyError):
        load_config(text=text, overrides=["model.encoder.no_such=1"])      # Hydra: adding a key needs '+'
    with pytest.raises(RecursionError):
        resolve({"a": "${b}", "b": "${a}"})
    assert apply_overrides({"a": {"b": 1}}, ["a.b=[1, 2]"])["a"]["b"] == [1, 2]


def test_lr_schedules_of_the_neighbouring_recipes():
    """NoamHoldAnnealing (Squeezeformer recipe) and CosineAnnealing (FastConformer recipes) against the closed forms of
    nemo/core/optim/lr_scheduler.py:153-228, 387-414, 429-435"""
    import math
    from nemo_amd.optim import CosineAnnealing, NoamHoldAnnealing
    s = NoamHoldAnnealing(1.5e-3, warmup_steps=100, hold_steps=400, decay_rate=1.0, min_lr=1e-5)
    assert s.lr_at(0) == pytest.approx(1.5e-3 * 1 / 101) and s.lr_at(50) == pytest.approx(1.5e-3 * 51 / 101)
    assert s.lr_at(100) == pytest.approx(1.5e-3) and s.lr_at(499) == 1.5e-3          # hold: [warmup, warmup + hold)
    assert s.lr_at(600) == pytest.approx(1.5e-3 * 100 / (600 - 400))                  # lr * warmup^r / (step - hold)^r
    assert s.lr_at(10 ** 7) == 1e-5
    c = CosineAnnealing(1e-3, max_steps=1000, warmup_steps=100, min_lr=1e-4)
    assert c.lr_at(10) == pytest.approx(1e-3 * 11 / 101)
    assert c.lr_at(550) == pytest.approx(1e-4 + 9e-4 * 0.5 * (1 + math.cos(math.pi * 450 / 900)))
    assert c.lr_at(1000) == pytest.approx(1e-4) and c.lr_at(2000) == 1e-4
    lrs = [c.step() for _ in range(3)]
    assert lrs == [c.lr_at(1), c.lr_at(2), c.lr_at(3)] and c.get_last_lr() == lrs[-1]


def test_bench_refuses_a_multi_gpu_job_it_cannot_place():
    """`python bench.py --gpus N` launches its own ranks; with fewer than N GPUs visible (none in this container) it must exit
    non-zero WITHOUT a JSON line -- never a silent N = 1 measurement under an N > 1 flag; under a launcher, a world size that
    disagrees with --gpus is refused the same way"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if torch.cuda.is_available() and torch.cuda.device_count() >= 8:
        pytest.skip("an 8-GPU node can place this job")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "BENCH_DEVICE")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode != 0 and "{" not in r.stdout and "refusing" in r.stderr, (r.returncode, r.stdout[-200:], r.stderr[-300:])
    env.update(WORLD_SIZE="2", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "4", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode != 0 and "{" not in r.stdout and "WORLD_SIZE=2" in r.stderr, (r.returncode, r.stdout[-200:], r.stderr[-300:])


def test_step_scoped_arena_bookkeeping():
    """nemo_amd.arena.Arena on host tensors: the first cycle only measures (everything comes from torch.empty), the next rewind
    sizes the buffer to 1.25 x the demand, tensors are 256-byte aligned views handed out front to back, a cycle that outgrows the
    buffer overflows into torch.empty and the buffer grows at the NEXT rewind (the outgrown one stays alive for recorded graphs)"""
    from nemo_amd.arena import Arena
    dev = torch.device("cpu")
    a = Arena("t")
    a.rewind(dev)
    x = a.take((3, 5), torch.float32, dev)
    assert a.buf is None and x.shape == (3, 5) and a.need == 256
    a.take((1000,), torch.bfloat16, dev)
    need1 = a.need
    a.rewind(dev)
    assert a.buf is not None and a.buf.numel() >= int(need1 * 1.25) and a.gen == 2 and a.off == 0
    base = a.buf.data_ptr()
    t1 = a.take((3, 5), torch.float32, dev)
    t2 = a.take((1000,), torch.bfloat16, dev)
    assert t1.data_ptr() == base and t2.data_ptr() == base + 256 and t2.dtype == torch.bfloat16 and t2.numel() == 1000
    t1.fill_(1.0); t2.fill_(2.0)
    assert float(t1.sum()) == 15.0 and float(t2.float().sum()) == 2000.0  # disjoint storage
    big = a.take((a.buf.numel(),), torch.uint8, dev)  # does not fit behind t1 / t2: overflow path
    assert not (base <= big.data_ptr() < base + a.buf.numel())
    old = a.buf
    a.rewind(dev)
    assert a.buf is not old and a.retired and a.retired[-1] is old and a.buf.numel() > old.numel()

