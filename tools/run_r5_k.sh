#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r5k; mkdir -p $O
MI355X_GRAPHS_BWD_LIVE=1 timeout 600 python -m pytest tests/test_graphs_gpu.py -x -q 2>&1 | tail -2 | tee -a $O/tests.txt
run() { env "$@" timeout 200 python bench.py --steps 16 --warmup 10 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$*', d['ms_per_step'], d['launch'].get('host_issue_ms_per_step'))" | tee -a $O/sweep.txt; }
for rep in 1 2 3; do
run MI355X_GRAPHS=0
run MI355X_GRAPHS=1
run MI355X_GRAPHS=1 MI355X_GRAPHS_BWD_LIVE=1
done
