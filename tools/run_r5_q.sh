#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r5q; mkdir -p $O
timeout 900 python -m pytest tests/test_squeezeformer_gpu.py tests/test_baseline_configs_gpu.py tests/test_model_gpu.py -x -q 2>&1 | tail -5 | tee $O/tests_sq.txt
for rep in 1 2; do
for v in 0 1; do
  MI355X_FLASH_PAD_HEADS=$v timeout 200 python bench.py --model squeezeformer --size medium --steps 12 --warmup 6 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 \
   | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('squeezeformer-medium bf16 B=32x20s pad_heads=$v ms_per_step', d['ms_per_step'])" | tee -a $O/sq_pad_heads.txt
done; done
