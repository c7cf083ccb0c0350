#!/bin/bash
# weight-gradient lane: finer split-K (shorter workgroups) x low stream priority, in-step A/B on one box
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r5g; mkdir -p $O
timeout 300 python -m pytest tests/test_packed_gpu.py -x -q -k relu_mask 2>&1 | tail -3 | tee $O/tests.txt
run() { env "$@" MI355X_GRAPHS=0 timeout 200 python bench.py --steps 16 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$*', d['ms_per_step'])" | tee -a $O/sweep.txt; }
run A=0
run MI355X_WGRAD_SK=8
run MI355X_WGRAD_SK=16
run MI355X_WGRAD_PRIO=1 MI355X_WGRAD_SK=8
run MI355X_WGRAD_PRIO=1 MI355X_WGRAD_SK=16
run MI355X_WGRAD_PRIO=1
run A=0
run MI355X_WGRAD_SK=2
run MI355X_WGRAD_SK=3
run A=0
