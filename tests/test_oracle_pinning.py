"""Pins the oracle (oracle/*.py, CPU) before anything trusts it:
  * CTC restatement  vs the warp-ctc known-answer vectors of the reference's tests/collections/asr/k2/test_ctc.py
  * torch ctc_loss (the arithmetic the reference path actually runs, losses/ctc.py:77) vs the same vectors
  * mel / encoder / loss restatement vs fixtures produced by the reference's own source files (oracle/make_golden.py)
  * (build container only) restatement vs the reference files executed live through oracle/ref_shim.py
"""
import json
import os

import numpy as np
import pytest
import torch

from oracle import conformer_ref as R
from oracle import ctc_ref


def _load_ka(golden_dir):
    with open(os.path.join(golden_dir, "ctc_known_answers.json")) as f:
        return json.load(f)


@pytest.mark.parametrize("name", ["test_case_small", "test_case_small_blank_last", "test_case_big_tensor"])
def test_ctc_restatement_matches_known_answers(golden_dir, name):
    case = _load_ka(golden_dir)[name]
    acts = np.array(case["acts"])
    B, T, C = acts.shape
    labels = case["labels"]
    nll, gx = ctc_ref.ctc_loss_and_grad_wrt_logits(acts, labels, [T] * B, [len(l) for l in labels], case["blank"])
    # tolerances of the reference test: rtol 1e-6 on cost, atol 1e-6 on grads (rtol 1e-3 for big_tensor)
    assert np.allclose(nll if B > 1 else nll.sum(), case["expected_costs"], rtol=1e-6)
    assert np.allclose(gx, np.array(case["expected_grads"]), atol=1e-6, rtol=1e-3)


@pytest.mark.parametrize("name", ["test_case_small", "test_case_small_blank_last", "test_case_big_tensor"])
def test_torch_ctc_matches_known_answers(golden_dir, name):
    case = _load_ka(golden_dir)[name]
    acts = torch.tensor(case["acts"], dtype=torch.float32, requires_grad=True)
    B, T, C = acts.shape
    labels = torch.tensor(case["labels"])
    logp = torch.log_softmax(acts, -1)
    loss, per = R.ctc_loss_mean_batch(logp, labels, torch.full((B,), T), torch.tensor([len(l) for l in case["labels"]]),
                                      case["blank"])
    per.sum().backward()
    assert np.allclose(per.detach().numpy(), case["expected_costs"], rtol=1e-5)
    assert np.allclose(acts.grad.numpy(), np.array(case["expected_grads"]), atol=2e-6, rtol=1e-3)


def test_ctc_restatement_matches_torch_random():
    rng = np.random.RandomState(0)
    B, T, C, U = 3, 30, 7, 8
    acts = rng.randn(B, T, C)
    tgt = rng.randint(0, C - 1, size=(B, U))
    tgt[0, 3] = tgt[0, 2]  # repeated label
    in_len = np.array([30, 22, 17])
    tl = np.array([8, 5, 8])
    nll, g = ctc_ref.ctc_loss_and_grad_wrt_logits(acts, tgt, in_len, tl, blank=C - 1)
    a = torch.tensor(acts, dtype=torch.float64, requires_grad=True)
    _, per = R.ctc_loss_mean_batch(torch.log_softmax(a, -1), torch.tensor(tgt), torch.tensor(in_len), torch.tensor(tl), C - 1)
    per.sum().backward()
    assert np.allclose(nll, per.detach().numpy(), rtol=1e-9)
    assert np.allclose(g, a.grad.numpy(), atol=1e-9)


def test_ctc_infeasible_zero_infinity():
    acts = np.random.RandomState(1).randn(1, 3, 4)
    nll, g = ctc_ref.ctc_loss_and_grad_wrt_logits(acts, [[0, 0, 1]], [3], [3], blank=3)  # needs >= 4 frames
    assert nll[0] == 0.0 and np.all(g == 0)


def test_mel_restatement_matches_reference_fixture(golden_dir):
    z = np.load(os.path.join(golden_dir, "ref_mel_b3.npz"))
    assert np.array_equal(R.mel_filterbank(), z["fb"][0]), "Slaney filterbank restatement != reference buffer"
    assert np.allclose(R.hann_window_sym(400).numpy(), z["window"], atol=1e-7)
    mel, mel_len = R.log_mel_features(torch.from_numpy(z["audio"]), torch.from_numpy(z["audio_len"]))
    assert mel.shape == z["mel"].shape
    assert np.array_equal(mel_len.numpy(), z["mel_len"])
    # north_star tolerance: 1e-3 relative fp32 -- the restatement is far inside it
    assert np.allclose(mel.numpy(), z["mel"], rtol=1e-3, atol=2e-4)
    assert np.abs(mel.numpy() - z["mel"]).max() < 2e-4


def test_mel_filterbank_against_an_independent_third_party_implementation(golden_dir):
    """librosa (the reference's dependency, features.py:338-344) is not in this image, and the fixture's `fb` came out of a shim
    that restates librosa.filters.mel with the published Slaney formula -- a restatement on both sides (round-3 review).  The image
    does hold one INDEPENDENT implementation of exactly that function: `transformers.audio_utils.mel_filter_bank(norm='slaney',
    mel_scale='slaney')` (written by other authors to reproduce librosa's filters for the Whisper / Wav2Vec2 feature extractors
    and tested there against librosa).  Oracle, fixture and product builder must all agree with it to float32 rounding."""
    # (the reference shim of other tests in this process registers a spec-less stand-in for `librosa`; transformers probes
    #  importlib.util.find_spec("librosa") at import time, which raises on such a module -- keep the stand-in out of its sight)
    import sys
    hidden = {k: sys.modules.pop(k) for k in list(sys.modules) if k == "librosa" or k.startswith("librosa.")}
    try:
        hf = pytest.importorskip("transformers.audio_utils")
    finally:
        sys.modules.update(hidden)
    from nemo_amd.modules.audio_preprocessing import slaney_mel_filterbank
    for (sr, n_fft, n_mels, fmin, fmax) in [(16000, 512, 80, 0.0, 8000.0), (16000, 512, 64, 0.0, 8000.0), (16000, 512, 80, 20.0, 7600.0),
                                            (8000, 256, 40, 0.0, 4000.0)]:
        want = hf.mel_filter_bank(num_frequency_bins=n_fft // 2 + 1, num_mel_filters=n_mels, min_frequency=fmin, max_frequency=fmax,
                                  sampling_rate=sr, norm="slaney", mel_scale="slaney").T  # [n_mels, bins], float64
        got = np.asarray(slaney_mel_filterbank(sr, n_fft, n_mels, fmin, fmax, "slaney"), dtype=np.float64)
        assert got.shape == want.shape
        assert np.abs(got - want).max() <= 2e-9 + 1e-7 * np.abs(want).max(), (sr, n_fft, n_mels, np.abs(got - want).max())
    z = np.load(os.path.join(golden_dir, "ref_mel_b3.npz"))
    want = hf.mel_filter_bank(num_frequency_bins=257, num_mel_filters=80, min_frequency=0.0, max_frequency=8000.0, sampling_rate=16000,
                              norm="slaney", mel_scale="slaney").T
    assert np.abs(z["fb"][0].astype(np.float64) - want).max() <= 2e-9          # the reference-run fixture's buffer
    assert np.abs(R.mel_filterbank().astype(np.float64) - want).max() <= 2e-9  # the oracle


@pytest.mark.parametrize("mode", ["eval", "train"])
def test_model_restatement_matches_reference_fixture(golden_dir, mode):
    z = np.load(os.path.join(golden_dir, "ref_tiny_model.npz"))
    cfg = R.ConformerCfg(d_model=32, n_heads=4, n_layers=2, vocab=16, dropout=0, dropout_att=0, dropout_pre_encoder=0)
    P = {k[2:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("P/")}
    for k in R.trainable_keys(P):
        P[k] = P[k].clone().requires_grad_(True)
    out = R.model_forward(P, cfg, torch.from_numpy(z["audio"]), torch.from_numpy(z["audio_len"]),
                          torch.from_numpy(z["tokens"]), torch.from_numpy(z["token_len"]),
                          train=False, bn_training=(mode == "train"))
    assert np.array_equal(out["enc_len"].numpy(), z[f"{mode}/enc_len"])
    assert np.allclose(out["enc"].detach().numpy(), z[f"{mode}/enc"], atol=2e-5)
    assert np.allclose(out["logp"].detach().numpy(), z[f"{mode}/logp"], atol=2e-5)
    assert abs(out["loss"].item() - float(z[f"{mode}/loss"])) <= 1e-5 * abs(float(z[f"{mode}/loss"]))
    out["loss"].backward()
    for k in R.trainable_keys(P):
        ref = z[f"{mode}/grad/{k}"]
        scale = max(np.abs(ref).max(), 1e-4)
        assert np.abs(P[k].grad.numpy() - ref).max() <= 1e-3 * scale + 1e-5, k  # analytically-zero grads (dw bias under BN, k bias) are pure rounding noise


def _encoder_option_cases():
    from oracle.make_golden import ENCODER_OPTION_CASES
    return ENCODER_OPTION_CASES


def _option_cfg(kw):
    kw = dict(kw)
    ccs = kw.pop("conv_context_size", None)
    if ccs == "causal":
        ccs = (8, 0)  # [kernel - 1, 0], conformer_encoder.py:902-903
    return R.ConformerCfg(d_model=32, n_heads=4, n_layers=2, vocab=16, conv_kernel=9, dropout=0, dropout_att=0, dropout_pre_encoder=0,
                          att_context_size=tuple(kw.pop("att_context_size", (-1, -1))), att_context_style=kw.pop("att_context_style", "regular"),
                          conv_norm_type=kw.pop("conv_norm_type", "batch_norm"), conv_context_size=tuple(ccs) if ccs else None)


@pytest.mark.parametrize("case", ["att_regular_8_4", "att_regular_left_6", "att_chunked_8_3", "conv_layer_norm", "conv_causal",
                                  "conv_context_6_2", "interctc_l0_l1", "streaming_recipe"])
def test_encoder_option_restatements_match_the_reference_fixture(golden_dir, case):
    """Oracle first (the kernels for these options come after it): limited / chunked attention context, LayerNorm in the conv
    module, causal / asymmetric depthwise padding and the InterCTC loss assembly of oracle/conformer_ref.py against the reference's
    own ConformerEncoder run with the same options (oracle/make_golden.py: make_encoder_options_fixture, train-mode BatchNorm
    statistics, dropout 0): encoder output, lengths, loss, the InterCTC parts and a set of gradients from every block."""
    from oracle.make_golden import OPTION_GRADS
    kw, inter = _encoder_option_cases()[case]
    z = np.load(os.path.join(golden_dir, "ref_tiny_model.npz"))
    zo = np.load(os.path.join(golden_dir, "ref_encoder_options.npz"))
    P = {k[2:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("P/")}
    for k in zo.files:
        if k.startswith(case + "/P/"):
            P[k[len(case) + 3:]] = torch.from_numpy(zo[k])
    for k in R.trainable_keys(P):
        P[k] = P[k].clone().requires_grad_(True)
    out = R.model_forward(P, _option_cfg(kw), torch.from_numpy(z["audio"]), torch.from_numpy(z["audio_len"]),
                          torch.from_numpy(z["tokens"]), torch.from_numpy(z["token_len"]), train=False, bn_training=True,
                          interctc=inter)
    assert np.array_equal(out["enc_len"].numpy(), zo[f"{case}/enc_len"])
    assert np.allclose(out["enc"].detach().numpy(), zo[f"{case}/enc"], atol=2e-5)
    assert abs(out["loss"].item() - float(zo[f"{case}/loss"])) <= 1e-5 * abs(float(zo[f"{case}/loss"]))
    if inter:
        assert abs(out["final_loss"].item() - float(zo[f"{case}/final_loss"])) <= 1e-5 * abs(float(zo[f"{case}/final_loss"]))
        for l in inter[0]:
            a, b = out[f"inter_ctc_loss_l{l}"].item(), float(zo[f"{case}/inter_ctc_loss_l{l}"])
            assert abs(a - b) <= 1e-5 * abs(b), (l, a, b)
    out["loss"].backward()
    for k in OPTION_GRADS:
        ref = zo[f"{case}/grad/{k}"]
        scale = max(np.abs(ref).max(), 1e-4)
        assert np.abs(P[k].grad.numpy() - ref).max() <= 1e-3 * scale + 1e-5, k
    # the options change the result: every case differs from the default encoder's output on the same parameters
    if kw:
        P0 = {k: v.detach() for k, v in P.items()}
        base = R.model_forward(P0, _option_cfg({}), torch.from_numpy(z["audio"]), torch.from_numpy(z["audio_len"]),
                               torch.from_numpy(z["tokens"]), torch.from_numpy(z["token_len"]), train=False, bn_training=True)
        assert (base["enc"] - out["enc"].detach()).abs().max() > 1e-3


def test_context_mask_is_the_reference_mask_for_every_style():
    """R.context_mask against a literal transcription of the reference's torch.triu / tril / chunk arithmetic
    (ConformerEncoder._create_masks, conformer_encoder.py:794-823) over a grid of sizes"""
    for T in (1, 5, 17):
        for style in ("regular", "chunked_limited"):
            for left in (-1, 0, 3, 8, 12):
                for right in (-1, 0, 3):
                    if style == "chunked_limited" and right >= 0 and left > 0 and left % (right + 1):
                        continue
                    m = torch.ones(T, T, dtype=torch.bool)
                    if style == "regular":
                        if left >= 0: m = m.triu(diagonal=-left)
                        if right >= 0: m = m.tril(diagonal=right)
                    elif right == -1:
                        if left >= 0: m = m.triu(diagonal=-left)
                    else:
                        cs = right + 1
                        lc = left // cs if left >= 0 else 10000
                        ci = torch.div(torch.arange(T, dtype=torch.int), cs, rounding_mode="trunc")
                        d = ci.unsqueeze(1) - ci.unsqueeze(0)
                        m = m & (d <= lc) & (d >= 0)
                    got = R.context_mask(R.ConformerCfg(att_context_size=(left, right), att_context_style=style), T)
                    assert torch.equal(got, m), (T, style, left, right)


@pytest.mark.parametrize("lengths", [[16000, 12000, 8123], [16000, 16000, 16000], [16000, 2400, 9000], [15000, 14840, 14680]])
def test_packed_token_restatement_reproduces_the_padded_reference_path(golden_dir, lengths):
    """oracle/packed_ref.py (the layers on the valid frames of a ragged batch only, BatchNorm statistics completed analytically for
    the padded frames: halo frames computed, the rest = the depthwise bias) against oracle/conformer_ref.py's padded computation --
    itself pinned to the reference on this very model (test_model_restatement_matches_reference_fixture): encoder output on every
    valid frame, CTC loss and EVERY gradient, train-mode BatchNorm; utterances shorter than / within the 15-frame halo of the
    longest included, and a batch without padding."""
    from oracle import packed_ref as PK
    z = np.load(os.path.join(golden_dir, "ref_tiny_model.npz"))
    cfg = R.ConformerCfg(d_model=32, n_heads=4, n_layers=2, vocab=16, dropout=0, dropout_att=0, dropout_pre_encoder=0)
    audio, tok = torch.from_numpy(z["audio"]), torch.from_numpy(z["tokens"])
    alen, tl = torch.tensor(lengths), torch.tensor([3, 2, 3])
    with torch.no_grad():
        mel, mel_len = R.log_mel_features(audio, alen, n_mels=cfg.feat_in)
    grads = []
    for packed in (False, True):
        P = {k[2:]: torch.from_numpy(z[k]).clone() for k in z.files if k.startswith("P/")}
        for k in R.trainable_keys(P):
            P[k].requires_grad_(True)
        if packed:
            enc, enc_len = PK.encoder_forward_packed(P, cfg, mel, mel_len, bn_training=True, pfx="encoder.")
        else:
            enc, enc_len = R.encoder_forward(P, cfg, mel, mel_len, train=False, bn_training=True, pfx="encoder.")
        logp = R.decoder_forward(P, enc, "decoder.decoder_layers.0.", cfg)
        loss, _ = R.ctc_loss_mean_batch(logp, tok, enc_len, tl, cfg.vocab)
        loss.backward()
        grads.append((enc.detach(), enc_len, loss.item(), {k: P[k].grad.clone() for k in R.trainable_keys(P)}))
    (e0, n0, l0, g0), (e1, n1, l1, g1) = grads
    assert torch.equal(n0, n1)
    for b, n in enumerate(n0.tolist()):
        assert (e0[b, :, :n] - e1[b, :, :n]).abs().max() <= 2e-5, b
    assert abs(l0 - l1) <= 1e-5 * abs(l0)
    for k in g0:
        scale = max(g0[k].abs().max().item(), 1e-4)
        assert (g0[k] - g1[k]).abs().max().item() <= 1e-3 * scale + 1e-5, k


@pytest.mark.parametrize("case", ["feat_out", "bypass", "sd_uniform", "sd_linear", "causal_striding", "streaming_fastconformer",
                                  "local_attn"])
def test_encoder_structure_restatements_match_the_reference_fixture(golden_dir, case):
    """what the reference's own encoder tests exercise (tests/collections/asr/test_conformer_encoder.py:24-199: stochastic depth,
    bypass_pre_encode with feat_out and a LayerNorm conv module), as VALUES: oracle/conformer_ref.py against the reference's
    ConformerEncoder (oracle/make_golden.py: make_encoder_structure_fixture) -- output, lengths and the gradients of a random
    probe; stochastic depth draws its decisions from torch's global generator at the same points as the reference"""
    z = np.load(os.path.join(golden_dir, "ref_encoder_structure.npz"))
    pname = "sd" if case.startswith("sd_") else case
    f32 = lambda a: torch.from_numpy(a.astype(np.float32) if a.dtype == np.float16 else a)  # (fp16-representable values, stored as fp16)
    P = {k[len(pname) + 3:]: f32(z[k]).clone() for k in z.files if k.startswith(pname + "/P/")}
    for k in R.trainable_keys(P):
        P[k].requires_grad_(True)
    common = dict(vocab=16, dropout=0, dropout_att=0, dropout_pre_encoder=0)
    if case == "feat_out":
        cfg = R.ConformerCfg(d_model=32, n_heads=4, n_layers=2, conv_kernel=9, feat_out=24, **common)
    elif case == "bypass":
        cfg = R.ConformerCfg(feat_in=10, d_model=16, n_heads=4, n_layers=3, conv_kernel=3, feat_out=8, conv_norm_type="layer_norm", **common)
    elif case == "causal_striding":   # CausalConv2D in the 'striding' x4 stack: 80 -> 41 -> 21 frequency bins, T -> T // 2 + 1 twice
        cfg = R.ConformerCfg(d_model=32, n_heads=4, n_layers=2, conv_kernel=9, causal_downsampling=True, **common)
    elif case == "local_attn":   # sliding-window attention [6, 6]: T' = 42 / 23 / 5 valid frames (shorter than the window, not a multiple of 2w)
        cfg = R.ConformerCfg(d_model=32, n_heads=4, n_layers=2, conv_kernel=9, self_attention_model="rel_pos_local_attn",
                             att_context_size=(6, 6), **common)
    elif case == "streaming_fastconformer":  # the cache-aware streaming recipe's encoder section, scaled down
        cfg = R.ConformerCfg(d_model=32, n_heads=4, n_layers=2, conv_kernel=9, causal_downsampling=True, att_context_size=(8, 3),
                             att_context_style="chunked_limited", conv_context_size=(8, 0), conv_norm_type="layer_norm", **common)
    else:
        cfg = R.ConformerCfg(d_model=32, n_heads=4, n_layers=4, conv_kernel=9, stochastic_depth_drop_prob=0.6,
                             stochastic_depth_mode=case[3:], **common)
        assert np.allclose(R.layer_drop_probs(cfg), z[f"{case}/probs"])
        torch.manual_seed(int(z[f"{case}/seed"]))
    dropped = []
    if case == "streaming_fastconformer":
        from oracle import fastconformer_ref as FC
        y, ylen = FC.encoder_forward(P, cfg, f32(z[f"{case}/x"]), torch.from_numpy(z[f"{case}/len"]), bn_training=True)
    else:
        y, ylen = R.encoder_forward(P, cfg, f32(z[f"{case}/x"]), torch.from_numpy(z[f"{case}/len"]), train=True,
                                    bypass_pre_encode=(case == "bypass"), dropped=dropped)
    assert np.array_equal(ylen.numpy(), z[f"{case}/ylen"])
    assert np.allclose(y.detach().numpy(), z[f"{case}/y"], atol=3e-5), np.abs(y.detach().numpy() - z[f"{case}/y"]).max()
    if case.startswith("sd_"):
        assert len(dropped) == 3 and any(dropped) and not all(dropped), dropped  # (the seeds were chosen to exercise both branches)
    (y * torch.from_numpy(z[f"{case}/probe"])).sum().backward()
    n = 0
    for k in z.files:
        if k.startswith(case + "/grad/"):
            name, ref = k[len(case) + 6:], z[k]
            scale = max(np.abs(ref).max(), 1e-4)
            # (analytically-zero gradients -- the key bias, the depthwise bias under BatchNorm -- are rounding noise of a few 1e-5
            #  under this O(1) probe on both sides)
            if np.abs(ref).max() < 2e-4:   # noise on both sides: it only has to stay noise
                assert np.abs(P[name].grad.numpy()).max() < 5e-4, name
            else:
                assert np.abs(P[name].grad.numpy() - ref).max() <= 1e-3 * scale + 5e-5, name
            n += 1
    assert n >= 10
    if case == "bypass":  # conformer_encoder.py:569-578: a [B, feat_in, T] tensor where pre-encoded frames are expected is a ValueError
        with pytest.raises(ValueError):
            R.encoder_forward(P, cfg, torch.rand(2, 10, 17), torch.tensor([17, 11]), bypass_pre_encode=True)


def test_stochastic_depth_drop_probabilities_follow_the_reference_rules():
    """regularization_utils.py:18-64 (the reference checks the same numbers in test_conformer_encoder.py:24-83)"""
    mk = lambda **kw: R.layer_drop_probs(R.ConformerCfg(n_layers=kw.pop("n"), **kw))
    assert mk(n=8, stochastic_depth_drop_prob=0.0) == [0.0] * 8
    assert mk(n=8, stochastic_depth_drop_prob=0.5, stochastic_depth_mode="uniform") == [0.0] + [0.5] * 7
    assert mk(n=8, stochastic_depth_drop_prob=0.5, stochastic_depth_mode="uniform", stochastic_depth_start_layer=3) == [0.0] * 3 + [0.5] * 5
    assert np.allclose(mk(n=5, stochastic_depth_drop_prob=0.8, stochastic_depth_mode="linear"), [0.0, 0.2, 0.4, 0.6, 0.8])
    assert np.allclose(mk(n=5, stochastic_depth_drop_prob=0.9, stochastic_depth_mode="linear", stochastic_depth_start_layer=2),
                       [0.0, 0.0, 0.3, 0.6, 0.9])
    for bad in (dict(stochastic_depth_drop_prob=1.0), dict(stochastic_depth_drop_prob=-0.1),
                dict(stochastic_depth_drop_prob=0.5, stochastic_depth_start_layer=0),
                dict(stochastic_depth_drop_prob=0.5, stochastic_depth_start_layer=9),
                dict(stochastic_depth_drop_prob=0.5, stochastic_depth_mode="weird")):
        with pytest.raises(ValueError):
            mk(n=8, **bad)


@pytest.mark.reference
@pytest.mark.parametrize("normalize,pad_to,pad_value", [("per_feature", 16, 0.0), ("all_features", 0, 0.0), ("NA", 16, -1.5),
                                                         ("per_feature", 0, 0.0)])
def test_front_end_options_match_the_live_reference(normalize, pad_to, pad_value):
    """FilterbankFeatures with the normalisation / padding options of the neighbouring recipes (the streaming recipes: normalize "NA";
    `pad_to: 16` is the class default): oracle/conformer_ref.py log_mel_features against the reference class run through the shim"""
    from oracle import ref_shim
    if not ref_shim.reference_available():
        pytest.skip("reference tree not present (GPU box): the committed fixtures cover the default options")
    FilterbankFeatures, _ = ref_shim.load_reference()
    f = FilterbankFeatures(sample_rate=16000, n_window_size=400, n_window_stride=160, nfilt=80, n_fft=512, dither=0.0, pad_to=pad_to,
                           normalize=normalize, pad_value=pad_value)
    f.eval()
    audio, _, _, _ = R.synthetic_batch(3, 1.3, vocab=16, seed=77)
    alen = torch.tensor([20800, 16000 + 77, 9999])
    want, want_len = f(audio.clone(), alen)
    got, got_len = R.log_mel_features(audio, alen, normalize=normalize, pad_to=pad_to, pad_value=pad_value)
    assert torch.equal(want_len, got_len) and want.shape == got.shape
    assert (want - got).abs().max().item() <= 1e-3 * max(1.0, want.abs().max().item())


@pytest.mark.reference
def test_restatement_matches_live_reference():
    from oracle import ref_shim

    if not ref_shim.reference_available():
        pytest.skip("/root/reference not present (GPU box)")
    torch.manual_seed(3)
    m = ref_shim.ReferenceCTCModel(d_model=48, n_heads=4, n_layers=1, vocab=12)
    m.eval()
    cfg = R.ConformerCfg(d_model=48, n_heads=4, n_layers=1, vocab=12, dropout=0, dropout_att=0, dropout_pre_encoder=0)
    P = {"encoder." + k: v for k, v in m.encoder.state_dict().items()}
    P.update({"decoder.decoder_layers." + k: v for k, v in m.decoder_layers.state_dict().items()})
    audio, alen, tok, tl = R.synthetic_batch(2, 0.8, vocab=12, seed=7)
    alen = torch.tensor([12800, 9000])
    loss, logp, enc, enc_len, mel, mel_len = m(audio, alen, tok, tl)
    out = R.model_forward(P, cfg, audio, alen, tok, tl)
    assert torch.allclose(out["mel"], mel, atol=1e-4)
    assert torch.allclose(out["logp"], logp, atol=2e-5)
    assert abs(out["loss"].item() - loss.item()) < 1e-4


# ---------------------------------------------------------------------------------------------- SpecAugment
def _specaug_cases():
    """(name, kwargs, seeding) exactly as oracle/make_golden.py:make_specaug_fixture ran the reference classes"""
    import random
    return [
        ("vec_adaptive", dict(freq_masks=2, time_masks=10, freq_width=27, time_width=0.05, mask_value=0.0), ("torch", 2024)),
        ("vec_int", dict(freq_masks=3, time_masks=4, freq_width=15, time_width=40, mask_value=-1.5), ("torch", 2025)),
        ("legacy", dict(freq_masks=2, time_masks=5, freq_width=27, time_width=0.05, mask_value=0.0), ("py", random.Random(7))),
    ]


def _unpack(z, name):
    x = z["x"]
    return np.unpackbits(z[name + "_mask"])[: x.size].reshape(x.shape).astype(bool)


@pytest.mark.parametrize("case", [0, 1, 2])
def test_specaug_restatement_matches_reference_fixture(golden_dir, case):
    from oracle import specaug_ref as SR
    z = np.load(os.path.join(golden_dir, "ref_specaug.npz"))
    x, length = torch.from_numpy(z["x"]), torch.from_numpy(z["length"])
    B, F, T = x.shape
    name, kw, (kind, seed) = _specaug_cases()[case]
    value = kw.pop("mask_value")
    if kind == "torch":
        torch.manual_seed(seed)
        rects = SR.vectorized_rects(B, F, T, length, **kw)
    else:
        rects = SR.legacy_rects(seed, B, F, T, length, **kw)
    out = SR.apply_rects(x, rects, value)
    mask = _unpack(z, name)
    assert np.array_equal((out != x).numpy(), mask)
    assert torch.all(out[torch.from_numpy(mask)] == value)


def test_specaug_cutout_restatement_matches_reference_fixture(golden_dir):
    import random
    from oracle import specaug_ref as SR
    z = np.load(os.path.join(golden_dir, "ref_specaug.npz"))
    x, length = torch.from_numpy(z["x"]), torch.from_numpy(z["length"])
    B, F, T = x.shape
    rng = random.Random(11)
    y = SR.apply_rects(x, SR.cutout_rects(rng, B, F, T, 5, 60, 20), 0.0)
    y = SR.apply_rects(y, SR.legacy_rects(rng, B, F, T, length, 1, 2, 10, 25), 0.0)
    assert np.array_equal((y != x).numpy(), _unpack(z, "cutout_then_legacy"))


# ---------------------------------------------------------------------------------------------- greedy decode / WER (8f-5)
def test_wer_restatement_matches_reference_test_vectors():
    """tests/collections/asr/test_asr_metrics.py:119-124 (the reference's own word_error_rate assertions)"""
    from oracle import decode_ref as D
    from nemo_amd.modules.ctc_decoding import word_error_rate
    for fn in (D.word_error_rate, word_error_rate):
        assert fn(hypotheses=['cat'], references=['cot']) == 1.0
        assert fn(hypotheses=['GPU'], references=['G P U']) == 1.0
        assert fn(hypotheses=['G P U'], references=['GPU']) == 3.0
        assert fn(hypotheses=['ducati motorcycle'], references=['motorcycle']) == 1.0
        assert fn(hypotheses=['ducati motorcycle'], references=['ducuti motorcycle']) == 0.5
        assert fn(hypotheses=['a B c'], references=['a b c']) == 1.0 / 3.0
        assert fn(hypotheses=['cat'], references=['cot'], use_cer=True) == 1.0 / 3.0
        with pytest.raises(ValueError):
            fn(hypotheses=['a'], references=['a', 'b'])


def test_greedy_decode_restatement_folds_like_the_reference_loop():
    from oracle import decode_ref as D
    V = 5  # blank = 5
    labels = [5, 1, 1, 5, 1, 2, 2, 2, 5, 5, 3, 3, 4, 5]
    logp = torch.full((1, len(labels), V + 1), -5.0)
    for t, c in enumerate(labels):
        logp[0, t, c] = -0.1 * (t + 1)
    (toks, score), = D.greedy_decode(logp, torch.tensor([len(labels)]), blank=V)
    assert toks == [1, 1, 2, 3, 4]
    assert abs(score - sum(-0.1 * (t + 1) for t, c in enumerate(labels) if c != V)) < 1e-5
    (toks, _), = D.greedy_decode(logp, torch.tensor([6]), blank=V)  # out_len truncates before decoding
    assert toks == [1, 1, 2]
    assert D.tokens_to_text([0, 1, 2], ["a", "b", "c"]) == "abc"



# ------------------------------------------------------------------ RNN-T loss oracle
def _rnnt_cases():
    import json
    with open(os.path.join(os.path.dirname(__file__), "golden", "rnnt_known_answers.json")) as f:
        return json.load(f)["cases"]


def test_rnnt_oracle_reproduces_the_reference_known_answers():
    """costs and gradients written into the reference's own tests (test_rnnt_pytorch.py:82-128, 190-310, 358-402)"""
    from oracle import rnnt_ref as RR
    cases = _rnnt_cases()
    c = cases["test_case_small"]
    acts = torch.tensor(c["acts"], dtype=torch.float32)
    labels = torch.tensor(c["labels"])
    lens, ll = torch.tensor([acts.shape[1]]), torch.tensor([labels.shape[1]])
    cost, grads = RR.rnnt_loss_and_grad(acts, labels, lens, ll, blank=0, reduction="sum")
    assert abs(cost.item() - c["expected_cost"]) < 1e-6
    assert np.allclose(grads.numpy(), np.array(c["expected_grads"]), atol=1e-7, rtol=1e-5)
    # FastEmit scales the cost by (1 + lambda) (test_case_small_fastemit_clamp, :436-439)
    for lam in (1.0, 0.01, 0.00001):
        cf, _ = RR.rnnt_loss_and_grad(acts, labels, lens, ll, blank=0, fastemit_lambda=lam, reduction="sum")
        assert abs(cf.item() - c["expected_cost"] * (1 + lam)) < 1e-6 * (1 + lam)
    c = cases["test_case_small_clamp"]
    cost, grads = RR.rnnt_loss_and_grad(acts, labels, lens, ll, blank=0, clamp=c["GRAD_CLAMP"], reduction="sum")
    assert abs(cost.item() - c["expected_cost"]) < 1e-6
    assert np.allclose(grads.numpy(), np.array(c["expected_grads"]), atol=1e-7, rtol=1e-5)
    c = cases["test_case_big_tensor"]
    acts = torch.tensor(c["activations"], dtype=torch.float32)
    labels = torch.tensor(c["labels"])
    B, T = acts.shape[:2]
    lens, ll = torch.full((B,), T), torch.full((B,), labels.shape[1])
    costs, grads = RR.rnnt_loss_and_grad(acts, labels, lens, ll, blank=0, reduction="none")
    assert np.allclose(costs.numpy(), np.array(c["expected_costs"]), atol=1e-6)
    assert np.allclose(grads.numpy(), np.array(c["expected_grads"]), atol=1e-6, rtol=1e-3)


def test_rnnt_oracle_closed_form_gradient_is_the_derivative():
    """the fused closed form (gpu_rnnt_kernel.py:355-396) against autograd through log-softmax + the alpha recursion,
    on a ragged random batch, any blank position"""
    from oracle import rnnt_ref as RR
    g = torch.Generator().manual_seed(3)
    B, T, U1, V1 = 3, 7, 5, 6
    acts = torch.randn(B, T, U1, V1, generator=g)
    lens, ll = torch.tensor([7, 4, 6]), torch.tensor([4, 2, 0])
    for blank in (0, V1 - 1, 2):
        labels = torch.randint(0, V1 - 1, (B, U1 - 1), generator=g)
        labels = labels + (labels >= blank).long()  # never the blank
        cost, grads = RR.rnnt_loss_and_grad(acts, labels, lens, ll, blank=blank, reduction="sum")
        nll, ag = RR.rnnt_nll_autograd(acts, labels, lens, ll, blank=blank)
        assert abs(cost.item() - nll.item()) < 1e-9 * abs(nll.item())
        assert (grads - ag).abs().max() < 1e-10
        assert grads[1, 4:].abs().max() == 0 and grads[2, :, 1:].abs().max() == 0  # padded cells carry no gradient


# ------------------------------------------------------------------ Squeezeformer oracle (the HIP path is next)
def test_squeezeformer_oracle_matches_the_reference_encoder():
    """oracle/squeezeformer_ref.py against the reference SqueezeformerEncoder run in the build container
    (tests/golden/ref_squeezeformer_tiny.npz): output, lengths, input gradient and every parameter gradient of a fixed
    linear functional -- dw_striding sub-sampling, scale/bias layers, Swish conv module on 2d channels with batch-statistics
    BatchNorm, time reduction + recovery, ragged lengths"""
    from oracle import squeezeformer_ref as SQ
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "ref_squeezeformer_tiny.npz"))
    P = {k[2:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("P.")}
    train_keys = [k[2:] for k in z.files if k.startswith("G.")]
    for k in train_keys:
        P[k] = P[k].clone().requires_grad_(True)
    cfg = SQ.SqueezeformerCfg(feat_in=40, d_model=32, n_heads=4, n_layers=4, conv_kernel=9, time_reduce_idx=1,
                              time_recovery_idx=3)
    x = torch.from_numpy(z["x"]).requires_grad_(True)
    y, yl = SQ.encoder_forward(P, cfg, x, torch.from_numpy(z["length"]), bn_training=True)
    assert yl.tolist() == z["y_len"].tolist() and tuple(y.shape) == z["y"].shape
    ref = torch.from_numpy(z["y"])
    assert (y - ref).abs().max() <= 2e-5 * ref.abs().max()
    valid = (torch.arange(y.shape[2]).unsqueeze(0) < yl.unsqueeze(1)).unsqueeze(1)
    (y * torch.from_numpy(z["w"]) * valid).sum().backward()
    gref = torch.from_numpy(z["dx"])
    assert (x.grad - gref).abs().max() <= 1e-4 * gref.abs().max()
    gmax = max(1.0, max(float(np.abs(z["G." + k]).max()) for k in train_keys))
    for k in train_keys:
        g, r = P[k].grad, torch.from_numpy(z["G." + k])
        g = torch.zeros_like(r) if g is None else g
        # floor relative to the largest gradient of the model: the depthwise-conv bias (in front of batch-statistics
        # BatchNorm) and the key bias (softmax shift) have analytically zero gradients -- both sides hold rounding noise there
        assert (g - r).abs().max().item() <= 2e-3 * r.abs().max().item() + 2e-4 * gmax, k


# ------------------------------------------------------------------ transducer head oracle
def test_transducer_head_oracle_matches_the_reference_modules():
    """oracle/transducer_ref.py (prediction network with a hand-written LSTM, joint) + oracle/rnnt_ref.py (loss, closed-form
    gradient w.r.t. the logits) against the reference's RNNTDecoder / RNNTJoint / RNNTLossPytorch run in the build container
    (tests/golden/ref_transducer_tiny.npz): decoder output, logits, loss, and the gradients w.r.t. the encoder output and
    every parameter obtained by pushing the oracle's logit gradient back through the restated head"""
    from oracle import rnnt_ref as RR
    from oracle import transducer_ref as TR
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "ref_transducer_tiny.npz"))
    PD = {k[4:]: torch.from_numpy(z[k]).clone().requires_grad_(True) for k in z.files if k.startswith("P.D.")}
    PJ = {k[4:]: torch.from_numpy(z[k]).clone().requires_grad_(True) for k in z.files if k.startswith("P.J.")}
    enc = torch.from_numpy(z["enc"]).requires_grad_(True)
    tgt, enc_len, tgt_len = torch.from_numpy(z["targets"]), torch.from_numpy(z["enc_len"]), torch.from_numpy(z["tgt_len"])
    V = PD["prediction.embed.weight"].shape[0] - 1
    g = TR.prediction_network(PD, tgt)
    assert (g - torch.from_numpy(z["dec_out"])).abs().max() < 1e-6
    logits = TR.joint_network(PJ, enc, g)
    # on CPU the reference joint returns log_softmax(logits) (rnnt.py: `log_softmax=None` -> applied unless the tensor is on
    # the GPU, where the loss fuses it); the restatement always returns the logits
    ref_logp = torch.from_numpy(z["logits"])
    assert (torch.log_softmax(logits, -1) - ref_logp).abs().max() <= 1e-5 * ref_logp.abs().max()
    cost, dlogits = RR.rnnt_loss_and_grad(logits, tgt.clamp(max=V - 1), enc_len, tgt_len, blank=V, reduction="sum")
    assert abs(cost.item() - float(z["loss"])) <= 1e-5 * abs(float(z["loss"]))
    logits.backward(dlogits.to(logits.dtype))  # the loss kernels hand out d cost / d logits; the head's backward does the rest
    ref = torch.from_numpy(z["d_enc"])
    assert (enc.grad - ref).abs().max() <= 1e-4 * ref.abs().max()
    for pre, P in (("D.", PD), ("J.", PJ)):
        for k, v in P.items():
            if ("G." + pre + k) not in z.files:
                continue
            r = torch.from_numpy(z["G." + pre + k])
            got = v.grad if v.grad is not None else torch.zeros_like(r)
            assert (got - r).abs().max().item() <= 2e-4 * r.abs().max().item() + 1e-6, k
    # the padding row of the embedding (the blank id) stays zero and receives no gradient
    assert PD["prediction.embed.weight"][V].abs().max() == 0


def test_fastconformer_encoder_oracle_matches_the_reference_encoder():
    """Conformer layers behind 'dw_striding' x8 sub-sampling, depthwise kernel 9 (cfg 4's encoder) against the reference
    ConformerEncoder run in the build container (tests/golden/ref_fastconformer_tiny.npz)"""
    from oracle import conformer_ref as CR
    from oracle import fastconformer_ref as FC
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "ref_fastconformer_tiny.npz"))
    P = {k[2:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("P.")}
    train_keys = [k[2:] for k in z.files if k.startswith("G.")]
    for k in train_keys:
        P[k] = P[k].clone().requires_grad_(True)
    cfg = CR.ConformerCfg(feat_in=40, d_model=32, n_heads=4, n_layers=2, conv_kernel=9, conv_channels=16, dropout=0.0,
                          dropout_pre_encoder=0.0, dropout_att=0.0)
    x = torch.from_numpy(z["x"]).requires_grad_(True)
    y, yl = FC.encoder_forward(P, cfg, x, torch.from_numpy(z["length"]), bn_training=True)
    assert yl.tolist() == z["y_len"].tolist() and tuple(y.shape) == z["y"].shape
    ref = torch.from_numpy(z["y"])
    assert (y - ref).abs().max() <= 2e-5 * ref.abs().max()
    valid = (torch.arange(y.shape[2]).unsqueeze(0) < yl.unsqueeze(1)).unsqueeze(1)
    (y * torch.from_numpy(z["w"]) * valid).sum().backward()
    gref = torch.from_numpy(z["dx"])
    assert (x.grad - gref).abs().max() <= 1e-4 * gref.abs().max()
    gmax = max(1.0, max(float(np.abs(z["G." + k]).max()) for k in train_keys))
    for k in train_keys:
        g, r = P[k].grad, torch.from_numpy(z["G." + k])
        g = torch.zeros_like(r) if g is None else g
        assert (g - r).abs().max().item() <= 2e-3 * r.abs().max().item() + 2e-4 * gmax, k


# ---------------------------------------------------------------------------------------------- BASELINE.json configs[0]
def cfg1_case():
    """SURVEY.md section 8(d) cfg 1 = BASELINE.json configs[0]: Conformer-CTC-Small (d=176, H=4, d_k=44, L=16, k=31),
    B = 2 x 10 s, vocab 128, fp32, dropout / dither off, batch-statistics BatchNorm; weights = init_params(seed 0)"""
    cfg = R.ConformerCfg.small(vocab=128, dropout=0.0, dropout_att=0.0, dropout_pre_encoder=0.0)
    P = R.init_params(cfg, seed=0)
    batch = R.synthetic_batch(2, 10.0, vocab=128, seed=1234)
    return cfg, P, batch


def test_cfg1_small_oracle_matches_reference_fixture(golden_dir):
    """pins the restatement at the geometry BASELINE.json names (16 layers deep, d_k = 44, T' = 251) against the reference's
    own files run through the shim (oracle/make_golden.py:make_cfg1_fixture): mel, log-probs, loss, EVERY gradient tensor
    (norm / max / random projection) and a dozen gradients element by element"""
    z = np.load(os.path.join(golden_dir, "ref_cfg1_small.npz"))
    cfg, P, (audio, alen, tok, tl) = cfg1_case()
    chk = np.array([sum(float(v.double().sum()) for k, v in sorted(P.items()) if v.is_floating_point()),
                    sum(float(v.double().abs().sum()) for k, v in sorted(P.items()) if v.is_floating_point())])
    assert np.allclose(chk, z["param_checksum"], rtol=1e-12), "init_params(seed=0) no longer reproduces the fixture's weights"
    for k in R.trainable_keys(P):
        P[k].requires_grad_(True)
    out = R.model_forward(P, cfg, audio, alen, tok, tl, train=False, bn_training=True)
    mel = out["mel"].numpy()
    assert mel.shape == (2, 80, 1001) and np.array_equal(out["mel_len"].numpy(), z["mel_len"])
    assert np.abs(mel[:, :, ::7] - z["mel_every7"]).max() < 2e-4
    assert np.allclose(mel.astype(np.float64).sum(2), z["mel_rowsum"], rtol=1e-3, atol=2e-2)
    assert np.allclose((mel.astype(np.float64) ** 2).sum(2), z["mel_rowsumsq"], rtol=1e-3)
    assert np.array_equal(out["enc_len"].numpy(), z["enc_len"])
    assert np.abs(out["enc"].detach().numpy()[:, :, ::5] - z["enc_every5"]).max() < 5e-4
    assert np.abs(out["logp"].detach().numpy() - z["logp"]).max() < 5e-4
    assert abs(out["loss"].item() - float(z["loss"])) <= 1e-5 * float(z["loss"])
    assert np.allclose(out["per_utt"].detach().numpy(), z["per_utt"], rtol=1e-5)
    out["loss"].backward()
    names = [str(n) for n in z["grad_names"]]
    assert sorted(names) == sorted(R.trainable_keys(P))
    gmax = float(z["grad_digest"][:, 1].max())
    for n, ref in zip(names, z["grad_digest"]):
        got = R.grad_digest(n, P[n].grad.numpy())
        numel = P[n].numel()
        # norm and max to 1e-3 of themselves; the projection to 1e-3 of the norm x sqrt(numel) bound it lives under.
        # analytically-zero gradients (depthwise bias under batch-statistics BN, key bias) are summation noise on both sides
        floor = 1e-4 * gmax if n.endswith(("depthwise_conv.bias", "linear_k.bias")) else 1e-7 * gmax
        assert abs(got[0] - ref[0]) <= 1e-3 * ref[0] + floor * np.sqrt(numel), (n, got, ref)
        assert abs(got[1] - ref[1]) <= 2e-3 * ref[1] + floor, (n, got, ref)
        assert abs(got[2] - ref[2]) <= 1e-3 * ref[0] + floor * np.sqrt(numel), (n, got, ref)
    full = [k[5:] for k in z.files if k.startswith("grad/")]
    assert len(full) >= 10
    for n in full:
        ref = z["grad/" + n]
        # conv.0.weight = sum over 80 k positions of (gradient x mel): the first tensor of the network and the last of the
        # backward chain, it carries the 2e-4 mel difference (rfft framing here, torch.stft there) plus 16 layers of fp32
        # reordering noise (the oracle itself moves by 2e-4 relative between fp32 and fp64 arithmetic at this depth)
        tol = 3e-3 if n.endswith("pre_encode.conv.0.weight") else 1e-3
        assert np.abs(P[n].grad.numpy() - ref).max() <= tol * np.abs(ref).max(), n
