"""TEST INFRASTRUCTURE ONLY -- generates the committed fixtures under tests/golden/.

Runs ONLY in the build container (needs /root/reference).  Two kinds of fixture:
  1. ctc_known_answers.json  -- the warp-ctc known-answer vectors held by the reference's own test
     tests/collections/asr/k2/test_ctc.py (test_case_small :85-120, test_case_small_blank_last :124-187,
     test_case_big_tensor :209-284), extracted by parsing that file's literals (nothing is executed).
  2. ref_*.npz -- inputs/outputs of the reference's own FilterbankFeatures / ConformerEncoder source files
     executed on CPU fp32 through oracle/ref_shim.py at fixed seeds (value parity for mel features and the
     encoder is *defined* as equality with these).

Usage:  python oracle/make_golden.py
"""
import ast
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, "tests", "golden")
REF = os.environ.get("NEMO_REFERENCE_ROOT", "/root/reference")


def _literal(node):
    """Evaluate list / number / np.array(list) literals."""
    if isinstance(node, ast.Call) and getattr(node.func, "attr", "") == "array":
        return _literal(node.args[0])
    return ast.literal_eval(node)


def extract_ctc_known_answers():
    src = open(os.path.join(REF, "tests/collections/asr/k2/test_ctc.py")).read()
    tree = ast.parse(src)
    wanted = {"test_case_small": 0, "test_case_small_blank_last": "last", "test_case_big_tensor": 0}
    out = {}
    for node in ast.walk(tree):
        if isinstance(node, ast.FunctionDef) and node.name in wanted:
            vals = {}
            for st in node.body:
                if isinstance(st, ast.Assign) and isinstance(st.targets[0], ast.Name):
                    name = st.targets[0].id
                    if name in ("acts", "labels", "expected_cost", "expected_costs", "expected_grads"):
                        try:
                            vals.setdefault(name, _literal(st.value))
                        except Exception:
                            pass
            acts = np.array(vals["acts"], dtype=np.float64)
            blank = acts.shape[-1] - 1 if wanted[node.name] == "last" else 0
            cost = vals.get("expected_costs", vals.get("expected_cost"))
            out[node.name] = dict(acts=acts.tolist(), labels=vals["labels"], blank=blank,
                                  expected_costs=np.atleast_1d(np.array(cost, dtype=np.float64)).tolist(),
                                  expected_grads=np.array(vals["expected_grads"], dtype=np.float64).tolist(),
                                  source="tests/collections/asr/k2/test_ctc.py::" + node.name)
    assert len(out) == 3, out.keys()
    with open(os.path.join(GOLD, "ctc_known_answers.json"), "w") as f:
        json.dump(out, f)
    for k, v in out.items():
        print(k, np.array(v["acts"]).shape, v["expected_costs"])


def make_reference_fixtures():
    from oracle.ref_shim import ReferenceCTCModel
    from oracle import conformer_ref as R

    # --- (1) mel front-end alone: B=3 x 1.5 s, ragged lengths ------------------------------------
    torch.manual_seed(0)
    m = ReferenceCTCModel(d_model=32, n_heads=4, n_layers=2, vocab=16)
    m.eval()
    audio, _, _, _ = R.synthetic_batch(3, 1.5, vocab=16, seed=1234)
    alen = torch.tensor([24000, 16000 + 77, 9999])
    mel, mel_len = m.features(audio.clone(), alen)
    np.savez_compressed(os.path.join(GOLD, "ref_mel_b3.npz"), audio=audio.numpy(), audio_len=alen.numpy(),
                        mel=mel.numpy(), mel_len=mel_len.numpy(), fb=m.featurizer.fb.numpy(),
                        window=m.featurizer.window.numpy())

    # --- (2) tiny model end to end (eval + train-mode BN), loss + selected grads --------------------
    with torch.no_grad():
        for n, p in m.named_parameters():
            if "pos_bias" in n:
                p.uniform_(-0.1, 0.1)
        for n, b in m.named_buffers():
            if "running_mean" in n:
                b.uniform_(-0.2, 0.2)
            if "running_var" in n:
                b.uniform_(0.5, 1.5)
    audio, _, tok, _ = R.synthetic_batch(3, 1.0, vocab=16, seed=4321)
    alen = torch.tensor([16000, 12000, 8123])
    tl = torch.tensor([3, 2, 3])
    fix = dict(audio=audio.numpy(), audio_len=alen.numpy(), tokens=tok.numpy(), token_len=tl.numpy())
    for k, v in m.encoder.state_dict().items():
        fix["P/encoder." + k] = v.numpy()
    for k, v in m.decoder_layers.state_dict().items():
        fix["P/decoder.decoder_layers." + k] = v.numpy()
    for mode in ("eval", "train"):
        m.train(mode == "train")
        m.featurizer.eval()
        # keep BN running stats untouched between the two passes
        saved = {n: b.clone() for n, b in m.named_buffers()}
        loss, logp, enc, enc_len, mel, mel_len = m(audio, alen, tok, tl)
        m.zero_grad()
        loss.backward()
        for n, b in m.named_buffers():
            b.copy_(saved[n])
        fix[f"{mode}/loss"] = loss.detach().numpy()
        fix[f"{mode}/logp"] = logp.detach().numpy()
        fix[f"{mode}/enc"] = enc.detach().numpy()
        fix[f"{mode}/enc_len"] = enc_len.numpy()
        for n, p in m.encoder.named_parameters():
            fix[f"{mode}/grad/encoder.{n}"] = p.grad.numpy().copy()
        for n, p in m.decoder_layers.named_parameters():
            fix[f"{mode}/grad/decoder.decoder_layers.{n}"] = p.grad.numpy().copy()
    np.savez_compressed(os.path.join(GOLD, "ref_tiny_model.npz"), **fix)
    print("tiny model fixture:", {k: float(fix[k]) for k in ("eval/loss", "train/loss")})


def make_specaug_fixture():
    """the reference's SpecAugment / SpecCutout classes (spectr_augment.py, loaded through the shim) on a seeded CPU input:
    vectorised mode (torch generator), legacy mode and cut-out (python random.Random) -> tests/golden/ref_specaug.npz"""
    import importlib
    import random
    from oracle import ref_shim
    ref_shim.install()
    sa = importlib.import_module("nemo.collections.asr.parts.submodules.spectr_augment")
    g = torch.Generator().manual_seed(99)
    x = torch.randn(4, 80, 523, generator=g)
    length = torch.tensor([523, 400, 77, 250])
    fix = {"x": x.numpy(), "length": length.numpy()}
    # (1) vectorised, adaptive time width (conformer_ctc_bpe.yaml:123-128: freq 2 x 27, time 10 x 0.05)
    torch.manual_seed(2024)
    fix["vec_adaptive"] = sa.SpecAugment(freq_masks=2, time_masks=10, freq_width=27, time_width=0.05)(
        input_spec=x.clone(), length=length).numpy()
    # (2) vectorised, integer time width, non-zero mask value
    torch.manual_seed(2025)
    fix["vec_int"] = sa.SpecAugment(freq_masks=3, time_masks=4, freq_width=15, time_width=40, mask_value=-1.5)(
        input_spec=x.clone(), length=length).numpy()
    # (3) legacy (python rng), adaptive width
    fix["legacy"] = sa.SpecAugment(freq_masks=2, time_masks=5, freq_width=27, time_width=0.05, rng=random.Random(7),
                                   use_vectorized_code=False)(input_spec=x.clone(), length=length).numpy()
    # (4) cut-out followed by SpecAugment, one shared rng (SpectrogramAugmentation.__init__ passes the same rng to both)
    rng = random.Random(11)
    y = sa.SpecCutout(rect_masks=5, rect_time=60, rect_freq=20, rng=rng)(input_spec=x.clone())
    fix["cutout_then_legacy"] = sa.SpecAugment(freq_masks=1, time_masks=2, freq_width=10, time_width=25, rng=rng,
                                               use_vectorized_code=False)(input_spec=y, length=length).numpy()
    # stored compactly: the input, and per case the bit-packed set of cells that changed (they hold the mask value)
    out = {"x": fix["x"], "length": fix["length"]}
    for k in ("vec_adaptive", "vec_int", "legacy", "cutout_then_legacy"):
        out[k + "_mask"] = np.packbits(fix[k] != fix["x"])
    np.savez_compressed(os.path.join(GOLD, "ref_specaug.npz"), **out)
    print("specaug fixture: masked cells", {k: int((fix[k] != fix["x"]).sum()) for k in fix if k not in ("x", "length")})


def _reference_function(rel_path, name, namespace):
    """compile ONE top-level function of a reference source file (read in place, executed, never copied) in `namespace`:
    for modules whose import needs packages this image does not have (webdataset, lhotse, soundfile ...)"""
    src = open(os.path.join(REF, rel_path)).read()
    tree = ast.parse(src)
    node = next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == name)
    mod = ast.Module(body=[node], type_ignores=[])
    exec(compile(mod, os.path.join(REF, rel_path), "exec"), namespace)
    return namespace[name]


def make_data_fixture():
    """the reference's input-pipeline pieces run here on seeded inputs -> tests/golden/ref_data_pipeline.json:
      * SemiSortBatchSampler (asr_batching.py, imported through the shim with its two heavy imports stubbed): the batches
        of every rank for several world sizes / drop_last / shuffle settings, numpy's global generator seeded per call;
      * _speech_collate_fn (audio_to_text.py:52-110, the function alone): ragged 4- and 5-field batches;
      * CharParser (parsers.py) on transcripts with OOV symbols, special labels and blanks;
      * manifest.item_iter on a manifest with every text / path spelling the parser accepts."""
    import importlib
    import sys
    import tempfile
    import types
    import json as js
    from oracle import ref_shim
    ref_shim.install()
    fake = types.ModuleType("nemo.collections.asr.data.audio_to_text")
    fake.AudioToBPEDataset = fake.AudioToCharDataset = type("D", (), {})
    sys.modules.setdefault("nemo.collections.asr.data", types.ModuleType("nemo.collections.asr.data"))
    sys.modules["nemo.collections.asr.data.audio_to_text"] = fake
    fm = types.ModuleType("nemo.collections.asr.models.asr_model")
    fm.ASRModel = type("ASRModel", (), {})
    sys.modules["nemo.collections.asr.models.asr_model"] = fm
    ab = importlib.import_module("nemo.collections.asr.parts.utils.asr_batching")
    out = {"sampler": [], "collate": [], "parser": [], "manifest": []}
    rs = np.random.RandomState(5)
    durations = np.round(rs.uniform(1.0, 30.0, size=157), 2).tolist()
    for world, bs, drop_last, shuffle, rf, seed in [(1, 8, False, True, None, 42), (2, 8, False, True, 0.2, 42),
                                                    (4, 6, True, True, 0.1, 7), (8, 4, False, False, 0.0, 3),
                                                    (4, 64, False, True, 0.1, 1)]:
        case = dict(world=world, batch_size=bs, drop_last=drop_last, shuffle=shuffle, randomization_factor=rf, seed=seed,
                    np_seed=1000 + world, epochs=[])
        for epoch in (0, 1):
            ranks = []
            for rank in range(world):
                sm = ab.SemiSortBatchSampler(rank, world, durations, bs, shuffle, drop_last, rf, seed)
                sm.set_epoch(epoch)
                np.random.seed(case["np_seed"] + epoch)  # what seed_everything gives every rank
                ranks.append([[int(i) for i in b] for b in sm])
            case["epochs"].append(ranks)
        out["sampler"].append(case)
    out["durations"] = durations

    collate = _reference_function("nemo/collections/asr/data/audio_to_text.py", "_speech_collate_fn", {"torch": torch})
    g = torch.Generator().manual_seed(3)
    for with_ids, pad_id in ((False, 0), (True, 7)):
        lens, tls = [160, 91, 160, 37], [5, 1, 9, 0]
        batch = []
        for i, (n, tl) in enumerate(zip(lens, tls)):
            item = (torch.randn(n, generator=g), torch.tensor(n).long(), torch.randint(1, 30, (tl,), generator=g).long(),
                    torch.tensor(tl).long())
            batch.append(item + (100 + i,) if with_ids else item)
        res = collate(batch, pad_id)
        out["collate"].append(dict(pad_id=pad_id, with_ids=with_ids, signals=[b[0].tolist() for b in batch],
                                   tokens=[b[2].tolist() for b in batch],
                                   out=[r.tolist() for r in res], out_dtypes=[str(r.dtype) for r in res]))

    sys.modules.setdefault("text_unidecode", types.SimpleNamespace(unidecode=lambda x: x))
    sys.modules.setdefault("inflect", types.SimpleNamespace(engine=lambda: None))
    parsers = importlib.import_module("nemo.collections.common.parts.preprocessing.parsers")
    labels = [" ", "a", "b", "c", "d", "e", "<unk>", "'", "zh"]
    for kw in (dict(), dict(unk_id=6), dict(unk_id=9, blank_id=9), dict(do_lowercase=False), dict(do_normalize=False)):
        p = parsers.CharParser(labels, **kw)
        texts = ["  A bad CAB ", "abc xyz de", "zh ab zh", "<unk> a'b", "", "a  b"]
        out["parser"].append(dict(labels=labels, kwargs=kw, texts=texts, ids=[p(t) for t in texts]))

    manifest = importlib.import_module("nemo.collections.common.parts.preprocessing.manifest")
    with tempfile.TemporaryDirectory() as d:
        os.makedirs(os.path.join(d, "wavs"))
        open(os.path.join(d, "wavs", "a.wav"), "wb").close()
        open(os.path.join(d, "t.txt"), "w").write("from a\nfile\n")
        lines = [dict(audio_filepath="wavs/a.wav", duration=1.5, text="hello"),
                 dict(audio_filename="wavs/missing.wav", duration=2.0, normalized_text="norm", offset=0.25),
                 dict(audio_filepath="/abs/x.wav", duration=3.0, text_filepath=os.path.join(d, "t.txt"), speaker=4,
                      orig_sample_rate=8000, lang="en"),
                 dict(audio_filepath="wavs/a.wav", duration=0.5, token_labels=[3, 4, 5])]
        mpath = os.path.join(d, "m.json")
        with open(mpath, "w") as f:
            for ln in lines:
                f.write(js.dumps(ln) + "\n")
            f.write("\n")
        keys = ("audio_file", "duration", "text", "offset", "speaker", "orig_sr", "token_labels", "lang", "id")
        items = [{k: it[k] for k in keys} for it in manifest.item_iter(mpath)]
        for it in items:
            it["audio_file"] = it["audio_file"].replace(d, "<DIR>")
        out["manifest"] = dict(lines=[js.dumps(ln).replace(d, "<DIR>") for ln in lines], items=items)
    with open(os.path.join(GOLD, "ref_data_pipeline.json"), "w") as f:
        js.dump(out, f)
    print("data fixture:", {k: len(v) for k, v in out.items()})


def extract_rnnt_known_answers():
    """literal inputs and expected costs / gradients of the reference's own RNN-T tests
    (tests/collections/asr/numba/rnnt_loss/test_rnnt_pytorch.py: test_case_small, test_case_big_tensor,
    test_case_small_clamp) -> tests/golden/rnnt_known_answers.json"""
    src = open(os.path.join(REF, "tests/collections/asr/numba/rnnt_loss/test_rnnt_pytorch.py")).read()
    tree = ast.parse(src)
    want = {"test_case_small": {}, "test_case_big_tensor": {}, "test_case_small_clamp": {}}
    consts = {}
    for node in tree.body:  # module-level constants such as GRAD_CLAMP
        if isinstance(node, ast.Assign) and isinstance(node.value, ast.Constant) and isinstance(node.targets[0], ast.Name):
            consts[node.targets[0].id] = node.value.value
    for cls in [n for n in tree.body if isinstance(n, ast.ClassDef)]:
        for fn in [n for n in cls.body if isinstance(n, ast.FunctionDef) and n.name in want and not want[n.name]]:
            vals = {}
            for st in ast.walk(fn):
                if isinstance(st, ast.Assign) and isinstance(st.targets[0], ast.Name):
                    name = st.targets[0].id
                    if name in ("acts", "activations", "labels", "expected_cost", "expected_costs", "expected_grads",
                                "GRAD_CLAMP"):
                        node = st.value
                        while isinstance(node, ast.Call) and getattr(node.func, "attr", "") in ("astype", "array"):
                            node = node.func.value if node.func.attr == "astype" else node.args[0]
                        try:
                            vals.setdefault(name, _literal(node))
                        except Exception:
                            pass
            want[fn.name] = vals
    out = {"cases": want}
    with open(os.path.join(GOLD, "rnnt_known_answers.json"), "w") as f:
        json.dump(out, f)
    print("rnnt known answers:", {k: sorted(v) for k, v in want.items()})


def make_squeezeformer_fixture():
    """the reference's SqueezeformerEncoder (squeezeformer_encoder.py, through the shim) on a tiny configuration that takes
    every branch -- 'dw_striding' sub-sampling, adaptive scale/bias, Swish conv module on 2d channels, batch-statistics
    BatchNorm, time reduction at layer 1 and recovery at layer 3, ragged lengths -> tests/golden/ref_squeezeformer_tiny.npz
    (inputs, every parameter, output, and the gradient of a fixed linear functional w.r.t. input and parameters)"""
    import importlib
    from oracle import ref_shim
    ref_shim.install()
    m = importlib.import_module("nemo.collections.asr.modules.squeezeformer_encoder")
    torch.manual_seed(11)
    enc = m.SqueezeformerEncoder(feat_in=40, n_layers=4, d_model=32, subsampling="dw_striding", subsampling_factor=4,
                                 subsampling_conv_channels=-1, ff_expansion_factor=4, n_heads=4, conv_kernel_size=9,
                                 dropout=0.0, dropout_emb=0.0, dropout_att=0.0, adaptive_scale=True, time_reduce_idx=1,
                                 time_recovery_idx=3)
    with torch.no_grad():  # move the trivially-initialised parameters off their defaults
        for n, p in enc.named_parameters():
            if n.endswith("_scale.scale"):
                p.add_(0.2 * torch.randn_like(p))
            elif n.endswith("_scale.bias") or "pos_bias" in n or n.endswith("norm_conv.bias"):
                p.add_(0.1 * torch.randn_like(p))
    enc.train()  # dropout is 0 everywhere: train mode only selects batch statistics in BatchNorm
    x = torch.randn(3, 40, 75, requires_grad=True)
    length = torch.tensor([75, 52, 31])
    y, yl = enc(audio_signal=x, length=length)
    w = torch.randn_like(y)
    valid = (torch.arange(y.shape[2]).unsqueeze(0) < yl.unsqueeze(1)).unsqueeze(1)
    (y * w * valid).sum().backward()
    out = {"x": x.detach().numpy(), "length": length.numpy(), "y": y.detach().numpy(), "y_len": yl.numpy(), "w": w.numpy(),
           "dx": x.grad.numpy()}
    for n, p in enc.state_dict().items():
        out["P." + n] = p.detach().numpy()
    for n, p in enc.named_parameters():
        out["G." + n] = (p.grad if p.grad is not None else torch.zeros_like(p)).numpy()
    np.savez_compressed(os.path.join(GOLD, "ref_squeezeformer_tiny.npz"), **out)
    print("squeezeformer fixture:", y.shape, yl.tolist(), len(out), "arrays")


def make_transducer_fixture():
    """the reference's RNNTDecoder + RNNTJoint (modules/rnnt.py through the shim) and its pure-torch loss
    (losses/rnnt_pytorch.py) on a tiny ragged batch -> tests/golden/ref_transducer_tiny.npz: parameters, decoder output,
    joint logits, per-batch loss, gradients w.r.t. the encoder output and every parameter"""
    import importlib
    from oracle import ref_shim
    ref_shim.install()
    m = importlib.import_module("nemo.collections.asr.modules.rnnt")
    lp = importlib.import_module("nemo.collections.asr.losses.rnnt_pytorch")
    torch.manual_seed(21)
    V, H, D, J = 12, 16, 24, 20
    dec = m.RNNTDecoder(prednet={"pred_hidden": H, "pred_rnn_layers": 2, "dropout": 0.0}, vocab_size=V,
                        normalization_mode=None, random_state_sampling=False, blank_as_pad=True)
    joint = m.RNNTJoint(jointnet={"encoder_hidden": D, "pred_hidden": H, "joint_hidden": J, "activation": "relu",
                                  "dropout": 0.0}, num_classes=V)
    B, T, U = 3, 9, 5
    enc = torch.randn(B, D, T, requires_grad=True)
    enc_len = torch.tensor([9, 7, 4])
    tgt_len = torch.tensor([5, 3, 0])
    tgt = torch.randint(0, V, (B, U))
    for b in range(B):
        tgt[b, tgt_len[b]:] = V  # padded with the blank id, as the collate function of the transducer models does
    g, _, _ = dec(targets=tgt, target_length=tgt_len)
    logits = joint(encoder_outputs=enc, decoder_outputs=g)
    loss = lp.RNNTLossPytorch(blank=V, reduction="sum")(acts=logits, labels=tgt.clamp(max=V - 1), act_lens=enc_len,
                                                         label_lens=tgt_len)
    loss.backward()
    out = {"enc": enc.detach().numpy(), "enc_len": enc_len.numpy(), "targets": tgt.numpy(), "tgt_len": tgt_len.numpy(),
           "dec_out": g.detach().numpy(), "logits": logits.detach().numpy(), "loss": np.array(loss.item()),
           "d_enc": enc.grad.numpy()}
    for pre, mod in (("D.", dec), ("J.", joint)):
        for n, p in mod.state_dict().items():
            out["P." + pre + n] = p.detach().numpy()
        for n, p in mod.named_parameters():
            out["G." + pre + n] = (p.grad if p.grad is not None else torch.zeros_like(p)).numpy()
    np.savez_compressed(os.path.join(GOLD, "ref_transducer_tiny.npz"), **out)
    print("transducer fixture: logits", tuple(logits.shape), "loss", loss.item())


def make_rnnt_greedy_fixture():
    """the reference's GreedyBatchedRNNTInfer (parts/submodules/rnnt_greedy_decoding.py:529, frame-looping algorithm, through the
    shim) on its own RNNTDecoder + RNNTJoint with random weights -> tests/golden/ref_rnnt_greedy.npz: parameters, encoder output,
    lengths, and per utterance the decoded token ids and their frame indices.  The blank bias is raised so that the search
    produces a mix of blank frames, single emissions and frames that hit `max_symbols`."""
    import importlib
    from oracle import ref_shim
    ref_shim.install()
    m = importlib.import_module("nemo.collections.asr.modules.rnnt")
    gd = importlib.import_module("nemo.collections.asr.parts.submodules.rnnt_greedy_decoding")
    torch.manual_seed(33)
    V, H, D, J, MAXS = 24, 32, 40, 36, 3
    dec = m.RNNTDecoder(prednet={"pred_hidden": H, "pred_rnn_layers": 1, "dropout": 0.0}, vocab_size=V,
                        normalization_mode=None, random_state_sampling=False, blank_as_pad=True)
    joint = m.RNNTJoint(jointnet={"encoder_hidden": D, "pred_hidden": H, "joint_hidden": J, "activation": "relu",
                                  "dropout": 0.0}, num_classes=V)
    with torch.no_grad():
        for p in list(dec.parameters()) + list(joint.parameters()):
            p.mul_(3.0)                                   # (default initialisation gives nearly flat logits)
        joint.joint_net[-1].bias[V] += 1.0
    dec.eval(); joint.eval()
    B, T = 5, 23
    enc = torch.randn(B, D, T) * 2.0
    enc_len = torch.tensor([23, 17, 23, 1, 9])
    infer = gd.GreedyBatchedRNNTInfer(decoder_model=dec, joint_model=joint, blank_index=V, max_symbols_per_step=MAXS,
                                      loop_labels=False, use_cuda_graph_decoder=False)
    hyps = infer(encoder_output=enc, encoded_lengths=enc_len)[0]
    out = {"enc": enc.numpy(), "enc_len": enc_len.numpy(), "blank": np.array(V), "max_symbols": np.array(MAXS)}
    for b, hy in enumerate(hyps):
        out[f"tokens{b}"] = np.array([int(t) for t in hy.y_sequence], dtype=np.int64)
        out[f"times{b}"] = np.array([int(t) for t in hy.timestamp], dtype=np.int64)
    for pre, mod in (("D.", dec), ("J.", joint)):
        for n, p in mod.state_dict().items():
            out["P." + pre + n] = p.detach().numpy()
    np.savez_compressed(os.path.join(GOLD, "ref_rnnt_greedy.npz"), **out)
    print("rnnt greedy fixture:", [len(h.y_sequence) for h in hyps], "tokens for lengths", enc_len.tolist())


def make_fastconformer_fixture():
    """the reference ConformerEncoder with 'dw_striding' x8 sub-sampling and depthwise kernel 9 (the FastConformer geometry)
    -> tests/golden/ref_fastconformer_tiny.npz (inputs, parameters, output, gradients of a fixed linear functional)"""
    from oracle import ref_shim
    ref_shim.install()
    import importlib
    m = importlib.import_module("nemo.collections.asr.modules.conformer_encoder")
    torch.manual_seed(31)
    enc = m.ConformerEncoder(feat_in=40, n_layers=2, d_model=32, feat_out=-1, subsampling="dw_striding", subsampling_factor=8,
                             subsampling_conv_channels=16, ff_expansion_factor=4, self_attention_model="rel_pos", n_heads=4,
                             conv_kernel_size=9, dropout=0.0, dropout_pre_encoder=0.0, dropout_emb=0.0, dropout_att=0.0)
    with torch.no_grad():
        for n, p in enc.named_parameters():
            if "pos_bias" in n:
                p.add_(0.1 * torch.randn_like(p))
    enc.train()
    x = torch.randn(3, 40, 131, requires_grad=True)
    length = torch.tensor([131, 90, 57])
    y, yl = enc(audio_signal=x, length=length)
    w = torch.randn_like(y)
    valid = (torch.arange(y.shape[2]).unsqueeze(0) < yl.unsqueeze(1)).unsqueeze(1)
    (y * w * valid).sum().backward()
    out = {"x": x.detach().numpy(), "length": length.numpy(), "y": y.detach().numpy(), "y_len": yl.numpy(), "w": w.numpy(),
           "dx": x.grad.numpy()}
    for n, p in enc.state_dict().items():
        out["P." + n] = p.detach().numpy()
    for n, p in enc.named_parameters():
        out["G." + n] = (p.grad if p.grad is not None else torch.zeros_like(p)).numpy()
    np.savez_compressed(os.path.join(GOLD, "ref_fastconformer_tiny.npz"), **out)
    print("fastconformer fixture:", tuple(y.shape), yl.tolist())


from oracle.conformer_ref import grad_digest  # noqa: E402  (3-number fingerprint of a gradient tensor)


CFG1_FULL_GRADS = ("decoder.decoder_layers.0.weight", "decoder.decoder_layers.0.bias", "encoder.pre_encode.conv.0.weight",
                   "encoder.layers.0.self_attn.pos_bias_u", "encoder.layers.0.self_attn.pos_bias_v",
                   "encoder.layers.0.conv.depthwise_conv.weight", "encoder.layers.0.norm_out.weight",
                   "encoder.layers.7.self_attn.linear_pos.weight", "encoder.layers.15.conv.depthwise_conv.weight",
                   "encoder.layers.15.self_attn.pos_bias_u", "encoder.layers.15.conv.batch_norm.weight",
                   "encoder.layers.15.feed_forward2.linear2.bias")


def make_cfg1_fixture():
    """BASELINE.json configs[0] itself (cfg 1): Conformer-CTC-Small (d=176, H=4 -> d_k=44, L=16, k=31),
    B = 2 x 10 s synthetic clips (seed 1234), vocab 128, fp32, dropout 0 / dither 0, batch-statistics BatchNorm -- run through
    the reference's own FilterbankFeatures + ConformerEncoder files on CPU -> tests/golden/ref_cfg1_small.npz.
    Weights are NOT stored (13 M values): they are oracle.conformer_ref.init_params(ConformerCfg.small(), seed=0), a pure
    function of torch's CPU generator; a checksum of them is stored so a generator change cannot pass silently.
    Stored: loss, per-utterance nll, lengths, log-probs [2,251,129], every 7th mel frame + per-row mel sums, a 3-number
    digest of EVERY gradient tensor and a dozen gradient tensors in full."""
    from oracle.ref_shim import ReferenceCTCModel
    from oracle import conformer_ref as R
    cfg = R.ConformerCfg.small(vocab=128, dropout=0.0, dropout_att=0.0, dropout_pre_encoder=0.0)
    P = R.init_params(cfg, seed=0)
    m = ReferenceCTCModel(d_model=176, n_heads=4, n_layers=16, vocab=128)
    sd = {k[len("encoder."):]: v for k, v in P.items() if k.startswith("encoder.")}
    missing, unexpected = m.encoder.load_state_dict(sd, strict=False)
    assert not unexpected and all("pos_enc" in k for k in missing), (missing, unexpected)
    m.decoder_layers.load_state_dict({k[len("decoder.decoder_layers."):]: v for k, v in P.items() if k.startswith("decoder.")})
    m.train()          # dropout is 0 everywhere: train mode selects batch statistics in BatchNorm
    m.featurizer.eval()  # no dither
    audio, alen, tok, tl = R.synthetic_batch(2, 10.0, vocab=128, seed=1234)
    loss, logp, enc, enc_len, mel, mel_len = m(audio, alen, tok, tl)
    per = m.ctc(logp.transpose(1, 0), tok.long(), enc_len.long(), tl.long())
    m.zero_grad()
    loss.backward()
    fix = {"loss": np.array(loss.item(), dtype=np.float64), "per_utt": per.detach().numpy().astype(np.float64),
           "enc_len": enc_len.numpy(), "mel_len": mel_len.numpy(), "logp": logp.detach().numpy(),
           "mel_every7": mel.numpy()[:, :, ::7].copy(), "mel_rowsum": mel.numpy().astype(np.float64).sum(2),
           "mel_rowsumsq": (mel.numpy().astype(np.float64) ** 2).sum(2),
           "enc_every5": enc.detach().numpy()[:, :, ::5].copy(),
           "param_checksum": np.array([sum(float(v.double().sum()) for k, v in sorted(P.items()) if v.is_floating_point()),
                                       sum(float(v.double().abs().sum()) for k, v in sorted(P.items()) if v.is_floating_point())])}
    grads = {"encoder." + n: p.grad for n, p in m.encoder.named_parameters()}
    grads.update({"decoder.decoder_layers." + n: p.grad for n, p in m.decoder_layers.named_parameters()})
    names = sorted(grads)
    assert set(names) == set(R.trainable_keys(P)), set(names) ^ set(R.trainable_keys(P))
    fix["grad_names"] = np.array(names)
    fix["grad_digest"] = np.stack([grad_digest(n, grads[n].numpy()) for n in names])
    for n in CFG1_FULL_GRADS:
        fix["grad/" + n] = grads[n].numpy().copy()
    np.savez_compressed(os.path.join(GOLD, "ref_cfg1_small.npz"), **fix)
    print("cfg1 (Small, B=2x10s) fixture: loss", loss.item(), "mel", tuple(mel.shape), "logp", tuple(logp.shape),
          "grad tensors", len(names))


# the encoder options the oracle restates ahead of the kernels (oracle/conformer_ref.py: att_context_size / att_context_style,
# conv_norm_type, conv_context_size, InterCTC): name -> (ReferenceCTCModel / ConformerEncoder kwargs, InterCTC (layers, weights) or None)
ENCODER_OPTION_CASES = {
    "att_regular_8_4": (dict(att_context_size=[8, 4], att_context_style="regular"), None),
    "att_regular_left_6": (dict(att_context_size=[6, -1], att_context_style="regular"), None),
    "att_chunked_8_3": (dict(att_context_size=[8, 3], att_context_style="chunked_limited"), None),
    "conv_layer_norm": (dict(conv_norm_type="layer_norm"), None),
    "conv_causal": (dict(conv_context_size="causal"), None),
    "conv_context_6_2": (dict(conv_context_size=[6, 2]), None),
    "interctc_l0_l1": (dict(), ([0, 1], [0.3, 0.1])),
    "streaming_recipe": (dict(att_context_size=[8, 3], att_context_style="chunked_limited", conv_context_size="causal",
                              conv_norm_type="layer_norm"), ([1], [0.25])),
}
OPTION_GRADS = ["encoder.pre_encode.out.weight", "encoder.layers.0.self_attn.linear_q.weight", "encoder.layers.0.self_attn.pos_bias_u",
                "encoder.layers.0.conv.depthwise_conv.weight", "encoder.layers.0.conv.batch_norm.weight",
                "encoder.layers.1.feed_forward2.linear2.weight", "encoder.layers.1.norm_out.weight",
                "decoder.decoder_layers.0.weight", "decoder.decoder_layers.0.bias"]


def make_encoder_options_fixture():
    """tests/golden/ref_encoder_options.npz: the reference's own ConformerEncoder run with the options of the streaming / long-form /
    InterCTC recipes on the tiny model of ref_tiny_model.npz (same parameters, same batch, conv kernel 9): encoder output, loss,
    InterCTC parts and a handful of gradients per case, train-mode BatchNorm statistics, dropout 0."""
    from oracle.ref_shim import ReferenceCTCModel
    from oracle import conformer_ref as R
    z = np.load(os.path.join(GOLD, "ref_tiny_model.npz"))
    audio, alen = torch.from_numpy(z["audio"]), torch.from_numpy(z["audio_len"])
    tok, tl = torch.from_numpy(z["tokens"]), torch.from_numpy(z["token_len"])
    fix = {}
    for name, (kw, inter) in ENCODER_OPTION_CASES.items():
        torch.manual_seed(0)
        m = ReferenceCTCModel(d_model=32, n_heads=4, n_layers=2, vocab=16, conv_kernel_size=9, **kw)
        sd = m.encoder.state_dict()
        with torch.no_grad():
            for k in sd:  # the tiny fixture's parameters wherever the shapes agree (the depthwise kernel is 9 taps here, 31 there)
                src = torch.from_numpy(z["P/encoder." + k]) if ("P/encoder." + k) in z.files else None
                if src is not None and src.shape == sd[k].shape:
                    sd[k].copy_(src)
            m.decoder_layers[0].weight.copy_(torch.from_numpy(z["P/decoder.decoder_layers.0.weight"]))
            m.decoder_layers[0].bias.copy_(torch.from_numpy(z["P/decoder.decoder_layers.0.bias"]))
        m.encoder.load_state_dict(sd)
        m.train(); m.featurizer.eval()
        if inter is None:
            loss, logp, enc, enc_len, mel, mel_len = m(audio, alen, tok, tl)
            out = {"loss": loss}
        else:
            out, enc, enc_len = m.forward_interctc(audio, alen, tok, tl, *inter)
        m.zero_grad()
        out["loss"].backward()
        for k, v in m.encoder.state_dict().items():
            if k.endswith("depthwise_conv.weight") or (kw.get("conv_norm_type") == "layer_norm" and "batch_norm" in k):
                fix[f"{name}/P/encoder.{k}"] = v.numpy().copy()  # the parameters that differ in shape / meaning from the tiny fixture
        fix[f"{name}/enc"] = enc.detach().numpy()
        fix[f"{name}/enc_len"] = enc_len.numpy()
        for k, v in out.items():
            fix[f"{name}/{k}"] = v.detach().numpy()
        grads = {"encoder." + n: p.grad for n, p in m.encoder.named_parameters()}
        grads.update({"decoder.decoder_layers." + n: p.grad for n, p in m.decoder_layers.named_parameters()})
        for n in OPTION_GRADS:
            fix[f"{name}/grad/{n}"] = grads[n].numpy().copy()
        print("encoder option", name, "loss", float(out["loss"]))
    np.savez_compressed(os.path.join(GOLD, "ref_encoder_options.npz"), **fix)


def make_encoder_structure_fixture():
    """tests/golden/ref_encoder_structure.npz: what the reference's own encoder tests exercise (tests/collections/asr/
    test_conformer_encoder.py: stochastic depth :24-124, bypass_pre_encode with feat_out and a LayerNorm conv module :129-199) as
    VALUES from the reference's ConformerEncoder, for the oracle restatement of the same options: (1) feat_out projection, (2)
    bypass_pre_encode on pre-encoded frames with conv_norm_type=layer_norm / kernel 3 / feat_out (the reference test's own
    geometry), (3) stochastic depth in training mode (uniform and linear), decisions drawn from torch's global generator."""
    from oracle.ref_shim import load_reference
    _, ConformerEncoder = load_reference()
    fix = {}

    def run(name, enc, x, n, seed=None, pname=None, **fw):
        # inputs and parameters are rounded to fp16-representable values BEFORE the reference runs and stored as float16 (exact):
        # half the bytes in tests/golden; outputs, probes and gradients stay float32
        x = x.half().float()
        with torch.no_grad():
            for t in list(enc.parameters()) + [b for k_, b in enc.named_buffers() if "pos_enc" not in k_ and b.is_floating_point()]:
                t.copy_(t.half().float())
        g = torch.Generator().manual_seed(99)
        probe = None
        if seed is not None:
            torch.manual_seed(seed)
        y, yl = enc(audio_signal=x, length=n, **fw)
        probe = torch.randn(y.shape, generator=g)
        enc.zero_grad()
        (y * probe).sum().backward()
        fix[f"{name}/x"], fix[f"{name}/len"] = x.numpy().astype(np.float16), n.numpy()
        fix[f"{name}/y"], fix[f"{name}/ylen"], fix[f"{name}/probe"] = y.detach().numpy(), yl.numpy(), probe.numpy()
        for k, v in enc.state_dict().items():
            if "pos_enc" not in k:  # (the 5000-position sinusoid table is a buffer, not a parameter)
                fix[f"{pname or name}/P/{k}"] = v.numpy().astype(np.float16) if v.is_floating_point() else v.numpy().copy()
        for k, p_ in enc.named_parameters():
            if k.startswith(("layers.0.", "out_proj", "pre_encode.conv.0", "layers.3.feed_forward2")) and p_.grad is not None:
                fix[f"{name}/grad/{k}"] = p_.grad.numpy().copy()  # (bypass_pre_encode leaves the sub-sampling without gradients)
        print("encoder structure", name, tuple(y.shape), float(y.detach().abs().mean()))

    # (1) feat_out projection on the ordinary path
    torch.manual_seed(3)
    enc = ConformerEncoder(feat_in=80, n_layers=2, d_model=32, n_heads=4, feat_out=24, conv_kernel_size=9, dropout=0.0,
                           dropout_pre_encoder=0.0, dropout_emb=0.0, dropout_att=0.0)
    enc.train()
    run("feat_out", enc, torch.randn(3, 80, 101, generator=torch.Generator().manual_seed(5)), torch.tensor([101, 77, 40]))
    # (2) bypass_pre_encode, the reference test's geometry (d_model 16, feat_out 8, layer_norm conv, kernel 3)
    torch.manual_seed(4)
    enc = ConformerEncoder(feat_in=10, n_layers=3, d_model=16, feat_out=8, stochastic_depth_drop_prob=0.0, dropout=0.0,
                           dropout_pre_encoder=0.0, dropout_emb=0.0, dropout_att=0.0, conv_norm_type="layer_norm", conv_kernel_size=3)
    enc.train()
    run("bypass", enc, torch.rand(2, 17, 16, generator=torch.Generator().manual_seed(6)), torch.tensor([17, 11]),
        bypass_pre_encode=True)
    # (3) stochastic depth, training mode
    for mode, seed in (("uniform", 11), ("linear", 12)):
        torch.manual_seed(7)
        enc = ConformerEncoder(feat_in=80, n_layers=4, d_model=32, n_heads=4, conv_kernel_size=9, dropout=0.0, dropout_pre_encoder=0.0,
                               dropout_emb=0.0, dropout_att=0.0, stochastic_depth_drop_prob=0.6, stochastic_depth_mode=mode,
                               stochastic_depth_start_layer=1)
        enc.train()
        fix[f"sd_{mode}/probs"] = np.array(enc.layer_drop_probs)
        fix[f"sd_{mode}/seed"] = np.array(seed)
        run(f"sd_{mode}", enc, torch.randn(2, 80, 65, generator=torch.Generator().manual_seed(8)), torch.tensor([65, 33]), seed=seed,
            pname="sd")  # (both modes start from the same initialisation: one copy of the parameters)
    # (4) causal down-sampling (CausalConv2D), 'striding' x4 and the cache-aware streaming recipe's whole combination:
    # 'dw_striding' x8 + causal_downsampling + chunked_limited attention [70, 13] + causal LayerNorm conv module, kernel 9
    # (examples/asr/conf/fastconformer/cache_aware_streaming/fastconformer_ctc_bpe_streaming.yaml: encoder section)
    torch.manual_seed(9)
    enc = ConformerEncoder(feat_in=80, n_layers=2, d_model=32, n_heads=4, conv_kernel_size=9, causal_downsampling=True, dropout=0.0,
                           dropout_pre_encoder=0.0, dropout_emb=0.0, dropout_att=0.0)
    enc.train()
    run("causal_striding", enc, torch.randn(3, 80, 101, generator=torch.Generator().manual_seed(10)), torch.tensor([101, 77, 40]))
    torch.manual_seed(13)
    enc = ConformerEncoder(feat_in=80, n_layers=2, d_model=32, n_heads=4, subsampling="dw_striding", subsampling_factor=8,
                           subsampling_conv_channels=16, causal_downsampling=True, att_context_size=[8, 3],
                           att_context_style="chunked_limited", conv_kernel_size=9, conv_context_size="causal",
                           conv_norm_type="layer_norm", dropout=0.0, dropout_pre_encoder=0.0, dropout_emb=0.0, dropout_att=0.0)
    enc.train()
    run("streaming_fastconformer", enc, torch.randn(3, 80, 301, generator=torch.Generator().manual_seed(14)), torch.tensor([301, 215, 96]))
    # (5) Longformer-style local attention with the relative-position term inside the window (conf/fastconformer/long_fastconformer):
    # window [6, 6] on ragged lengths -- a sequence shorter than the window, one that is not a multiple of 2w, padded queries
    torch.manual_seed(15)
    enc = ConformerEncoder(feat_in=80, n_layers=2, d_model=32, n_heads=4, conv_kernel_size=9, self_attention_model="rel_pos_local_attn",
                           att_context_size=[6, 6], dropout=0.0, dropout_pre_encoder=0.0, dropout_emb=0.0, dropout_att=0.0)
    enc.train()
    run("local_attn", enc, torch.randn(3, 80, 165, generator=torch.Generator().manual_seed(16)), torch.tensor([165, 90, 17]))
    np.savez_compressed(os.path.join(GOLD, "ref_encoder_structure.npz"), **fix)


if __name__ == "__main__":
    os.makedirs(GOLD, exist_ok=True)
    if len(sys.argv) > 1 and sys.argv[1] == "structure":
        make_encoder_structure_fixture()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "cfg1":
        make_cfg1_fixture()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "options":
        make_encoder_options_fixture()
        sys.exit(0)
    extract_ctc_known_answers()
    make_reference_fixtures()
    make_specaug_fixture()
    make_data_fixture()
    extract_rnnt_known_answers()
    make_squeezeformer_fixture()
    make_transducer_fixture()
    make_rnnt_greedy_fixture()
    make_fastconformer_fixture()
    make_cfg1_fixture()
    make_encoder_options_fixture()
    make_encoder_structure_fixture()
