"""Host issue time of one fit_step next to the GPU step time: is a model's step bound by the host or by the GPU?

    python tools/host_phases.py [ctc|transducer|squeezeformer] [--json out.json]

Three measurements per launch mode (eager Python sequencer / recorded hipGraph segments):
  step_ms          wall time per step with the real kernels (the benchmark's number)
  issue_busy_ms    host time from the first launch of a step to the return of the optimizer call, GPU busy (includes queue
                   back-pressure: the host blocks when it runs too far ahead of the GPU)
  issue_null_ms    the same with mi355x_set_null_launch(1): every launch site issues an empty kernel, the GPU has nothing to do,
                   so this is the PURE issue time of the launch sequence (Python + ctypes + hipLaunchKernel / hipGraphLaunch)
A step is host-bound when issue_null_ms approaches step_ms.
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from nemo_amd import _lib
from nemo_amd.models import (EncDecCTCModel, EncDecRNNTModel, conformer_ctc_config, fastconformer_transducer_config,
                             squeezeformer_ctc_config)
from oracle import conformer_ref as R  # synthetic batch generator only

kind = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("-") else "ctc"
out_json = sys.argv[sys.argv.index("--json") + 1] if "--json" in sys.argv else None
dev = torch.device("cuda:0")
cdt = torch.bfloat16


def build():
    torch.manual_seed(0)
    if kind == "transducer":
        m = EncDecRNNTModel(fastconformer_transducer_config("large", vocab_size=1024, compute_dtype=cdt, spec_augment=True))
        m.decoder.compute_dtype = m.joint.compute_dtype = cdt
        vocab = 1024
    elif kind == "squeezeformer":
        m = EncDecCTCModel(squeezeformer_ctc_config("medium", vocab_size=128, compute_dtype=cdt, spec_augment=True))
        m.decoder.compute_dtype = cdt
        vocab = 128
    else:
        m = EncDecCTCModel(conformer_ctc_config("large", vocab_size=128, compute_dtype=cdt, spec_augment=True))
        m.decoder.compute_dtype = cdt
        vocab = 128
    m = m.to(dev).train()
    m.setup_optimization(dict(name="adamw", lr=1e-4, betas=[0.9, 0.98], weight_decay=1e-3))
    return m, vocab


def measure(m, batch, n, null):
    torch.cuda.synchronize()
    _lib.lib.mi355x_set_null_launch(1 if null else 0)
    try:
        host = 0.0
        t_all = time.perf_counter()
        for _ in range(n):
            t0 = time.perf_counter()
            m.fit_step(batch)
            host += time.perf_counter() - t0
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t_all) / n
    finally:
        _lib.lib.mi355x_set_null_launch(0)
    return wall * 1e3, host / n * 1e3


res = {"model": kind, "batch": "32 x 20 s", "dtype": "bf16"}
audio, alen, tok, tl = None, None, None, None
for mode in ("tape", "graphs", "eager"):
    m, vocab = build()
    m.encoder.use_graphs = mode != "eager"
    m.encoder.graph_tape = mode == "tape"
    m.encoder.graph_auto = False
    if audio is None:
        audio, alen, tok, tl = R.synthetic_batch(32, 20.0, vocab=vocab, seed=0)
    batch = [t.to(dev) for t in (audio, alen, tok, tl)]
    for _ in range(5):
        m.fit_step(batch)
    step_ms, issue_busy = measure(m, batch, 10, null=False)
    # (a recorded sequence keeps the real kernels it captured; the null switch only empties what is still launched live)
    null_wall, issue_null = measure(m, batch, 10, null=True)
    res[mode] = {"step_ms": round(step_ms, 2), "issue_busy_ms": round(issue_busy, 2), "issue_null_ms": round(issue_null, 2),
                 "null_wall_ms": round(null_wall, 2), "graph_info": m.encoder.graph_info()}
    if mode != "eager":
        # PURE issue time of the recorded sequence: a second instance RECORDS under the null switch, so what it replays are
        # empty kernels too and the GPU never pushes back
        del m
        torch.cuda.empty_cache()
        m, _ = build()
        m.encoder.use_graphs, m.encoder.graph_tape, m.encoder.graph_auto = True, mode == "tape", False
        _lib.lib.mi355x_set_null_launch(1)
        try:
            for _ in range(5):
                m.fit_step(batch)
        finally:
            _lib.lib.mi355x_set_null_launch(0)
        nw, ni = measure(m, batch, 10, null=True)
        res[mode]["recorded_null_issue_ms"] = round(ni, 2)
        res[mode]["recorded_null_wall_ms"] = round(nw, 2)
    print(kind, mode, res[mode], flush=True)
    del m
    torch.cuda.empty_cache()
if out_json:
    with open(out_json, "w") as f:
        json.dump(res, f, indent=1)
