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


if __name__ == "__main__":
    os.makedirs(GOLD, exist_ok=True)
    extract_ctc_known_answers()
    make_reference_fixtures()
    make_specaug_fixture()
