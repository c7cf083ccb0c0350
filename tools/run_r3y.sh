#!/bin/bash
# pack kernel fast path: parity + same-box A/B against the previous library
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r3y; mkdir -p $O
timeout 300 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -m gpu -x -q -k "pack or subsampling or model_matches or bf16 or tiny or conv" 2>&1 | tail -2
OLD=$PWD/nemo_amd/lib_ab/libmi355x_asr_prev.so
for rep in 1 2 3; do
  MI355X_ASR_LIB=$OLD timeout 120 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 > $O/bench_old_$rep.json
  timeout 120 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 > $O/bench_new_$rep.json
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r3y/bench_*.json")):
    try: print(f, json.loads(open(f).read())["ms_per_step"])
    except Exception as e: print(f, "ERR", e)
PY
