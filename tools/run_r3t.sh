#!/bin/bash
# round 3, attention instruction diet: parity of the new mask function / base-2 softmax, then old-vs-new on one box
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r3t; mkdir -p $O
timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -m gpu -x -q -k "flash or attention or attn or dropout or model" > $O/tests.log 2>&1; tail -3 $O/tests.log
OLD=$PWD/nemo_amd/lib_ab/libmi355x_asr_attnold.so
for rep in 1 2; do
  MI355X_ASR_LIB=$OLD timeout 120 python tools/attn_bench.py > $O/attn_old_$rep.txt 2>&1
  timeout 120 python tools/attn_bench.py > $O/attn_new_$rep.txt 2>&1
done
for rep in 1 2 3; do
  MI355X_ASR_LIB=$OLD timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 > $O/bench_old_$rep.json
  timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 > $O/bench_new_$rep.json
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r3t/bench_*.json")):
    try: print(f, json.loads(open(f).read())["ms_per_step"])
    except Exception as e: print(f, "ERR", e)
for f in sorted(glob.glob("gpurun_out/r3t/attn_*.txt")): print(f); print(open(f).read())
PY
