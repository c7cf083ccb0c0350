"""TEST INFRASTRUCTURE -- one-off generator of tests/golden/oracle_large_b32_loss.json: the CTC loss of BASELINE.json configs[1]
(Conformer-CTC-Large, B = 32 x 20 s, the benchmarked batch: R.synthetic_batch(32, 20.0, vocab=128, seed=1234), weights
R.init_params(ConformerCfg.large, seed=0)) by the CPU oracle (oracle/conformer_ref.py, which tests/test_oracle_pinning.py pins to
the reference's own files), dropout / dither / SpecAugment off, batch-statistics BatchNorm -- in fp32 and with bf16 rounding
emulated at the HIP path's storage points.  The oracle needs minutes for this batch, so it is run once here and the numbers are
committed; tests/test_baseline_configs_gpu.py compares the HIP path at the FULL benchmarked batch with them.

    python -m oracle.make_large_b32_loss
"""
import dataclasses
import json
import os
import time

import torch

from . import conformer_ref as R

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    torch.set_num_threads(os.cpu_count() or 8)
    cfg = R.ConformerCfg.large(vocab=128, dropout=0.0, dropout_att=0.0, dropout_pre_encoder=0.0)
    P = R.init_params(cfg, seed=0)
    batch = R.synthetic_batch(32, 20.0, vocab=128, seed=1234)
    out = {"config": "Conformer-CTC-Large, B=32x20s, R.synthetic_batch(32, 20.0, vocab=128, seed=1234), R.init_params(large, seed=0), "
                     "train-mode BatchNorm statistics, no dropout / dither / SpecAugment", "torch": torch.__version__}
    for name, emu in (("fp32", False), ("bf16_emulated", True)):
        t0 = time.time()
        with torch.no_grad():
            r = R.model_forward(P, dataclasses.replace(cfg, emulate_bf16=emu), *batch, train=False, bn_training=True)
        out[name] = {"loss": float(r["loss"]), "per_utt": [float(v) for v in r["per_utt"]], "seconds": round(time.time() - t0, 1)}
        print(name, out[name]["loss"], out[name]["seconds"], "s", flush=True)
    with open(os.path.join(ROOT, "tests", "golden", "oracle_large_b32_loss.json"), "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
