#!/usr/bin/env python
"""CPU-leg calibration (build container only: needs /root/reference): how fast is the oracle restatement ("port",
oracle/conformer_ref.py) relative to the reference's OWN modules (FilterbankFeatures + ConformerEncoder loaded verbatim through
oracle/ref_shim.py) on the same host, threads, batch and train step (fwd + CTC + bwd + AdamW, fp32, dropout / dither on)?

bench.py's `cpu_baseline` on the GPU box can only run the port (kind "port": the reference tree does not travel); the ratio measured
here says what the same leg would read with the reference's modules, and bench.py prints it next to the value
(`cpu_baseline.port_over_reference`).  Writes profiles/r4_cpu_calibration.json.

    python tools/cpu_calibration.py [--threads 8] [--steps 5]
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--threads", type=int, default=min(8, os.cpu_count() or 1))
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r4_cpu_calibration.json"))
    a = ap.parse_args()
    from oracle import conformer_ref as R
    from oracle import ref_shim
    if not ref_shim.reference_available():
        raise SystemExit("cpu_calibration: /root/reference is not here -- this tool runs in the build container only")
    torch.set_num_threads(a.threads)
    out = {"host_cpus": os.cpu_count(), "threads": a.threads, "steps_per_leg": a.steps,
           "method": "alternating legs (reference, port, reference, port, ...) on the same synthetic batch; median step time per leg; "
                     "train mode, fp32, fwd + CTC loss + bwd + AdamW", "cases": []}
    saved = os.dup(1)
    os.dup2(2, 1)  # (the reference's logger prints to stdout)
    try:
        for size, batch, secs in (("small", 2, 10.0), ("large", 2, 20.0)):
            cfg = getattr(R.ConformerCfg, size)(vocab=128)
            data = R.synthetic_batch(batch, secs, vocab=128, seed=1234)
            torch.manual_seed(0)
            ref = ref_shim.ReferenceCTCModel(cfg.d_model, cfg.n_heads, cfg.n_layers, vocab=128, dropout=0.1, dropout_att=0.1, dither=1e-5).train()
            ropt = torch.optim.AdamW(ref.parameters(), lr=1e-4, betas=(0.9, 0.98), weight_decay=1e-3)
            P = R.init_params(cfg, seed=0, nonzero_pos_bias=False)
            keys = R.trainable_keys(P)
            for k in keys:
                P[k].requires_grad_(True)
            popt = torch.optim.AdamW([P[k] for k in keys], lr=1e-4, betas=(0.9, 0.98), weight_decay=1e-3)
            audio, alen, tok, tl = data

            def ref_step():
                t0 = time.perf_counter()
                ropt.zero_grad(set_to_none=True)
                ref(audio, alen, tok, tl)[0].backward()
                ropt.step()
                return time.perf_counter() - t0

            def port_step():
                t0 = time.perf_counter()
                popt.zero_grad(set_to_none=True)
                noise = torch.randn_like(audio)
                R.model_forward(P, cfg, audio, alen, tok, tl, train=True, noise=noise, dither=1e-5)["loss"].backward()
                popt.step()
                return time.perf_counter() - t0

            ref_step(); port_step()  # warm-up (allocator, thread pool)
            tr, tp = [], []
            for _ in range(a.steps):
                tr.append(ref_step())
                tp.append(port_step())
            tr.sort(); tp.sort()
            mr, mp = tr[len(tr) // 2], tp[len(tp) // 2]
            case = {"model": f"Conformer-CTC-{size}", "batch": batch, "clip_seconds": secs,
                    "reference_audio_sec_per_s": round(batch * secs / mr, 2), "port_audio_sec_per_s": round(batch * secs / mp, 2),
                    "port_over_reference": round(mr / mp, 4),
                    "reference_step_s": [round(x, 3) for x in tr], "port_step_s": [round(x, 3) for x in tp]}
            out["cases"].append(case)
            print(case, file=sys.stderr, flush=True)
    finally:
        sys.stdout.flush()
        os.dup2(saved, 1)
        os.close(saved)
    out["port_over_reference"] = {c["model"]: c["port_over_reference"] for c in out["cases"]}
    with open(a.out, "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out["port_over_reference"]))


if __name__ == "__main__":
    main()
