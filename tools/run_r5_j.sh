#!/bin/bash
# where the grouped weight-gradient launch enters the side stream (MI355X_WGRAD_DEFER), live launches and launch tapes, one box
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r5j; mkdir -p $O
for dfr in 1 2; do
MI355X_WGRAD_DEFER=$dfr timeout 600 python -m pytest tests/test_model_gpu.py tests/test_graphs_gpu.py tests/test_baseline_configs_gpu.py -x -q -k "bf16 or hook or optimizer or large" 2>&1 | tail -2 | tee -a $O/tests.txt
done
run() { env "$@" timeout 200 python bench.py --steps 16 --warmup 10 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$*', d['ms_per_step'])" | tee -a $O/sweep.txt; }
for rep in 1 2; do
for g in 0 1; do
for dfr in 0 1 2; do
run MI355X_GRAPHS=$g MI355X_WGRAD_DEFER=$dfr
done; done; done
