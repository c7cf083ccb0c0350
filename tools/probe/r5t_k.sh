#!/bin/bash
# round-5 session-3 call K: SpecAugment mask parameters in one launch -- parity vs the tensor-op path, model tests, step A/B, launches per step
cd "$(dirname "$0")/../.." || exit 1
R=$PWD; O=$R/gpurun_out/r5t_k; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py tests/test_graphs_gpu.py -x -q -m gpu -k "specaug or spec_augment or model or graphs" > $O/tests.txt 2>&1; echo "tests rc=$?"; tail -3 $O/tests.txt
for rep in 1 2 3; do for arm in 1 0; do
  MI355X_SPECAUG_FUSED=$arm MI355X_GRAPHS=0 timeout 200 python bench.py --steps 20 --warmup 6 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('specaug fused=$arm ms/step', d['ms_per_step'], 'host issue', d['launch']['host_issue_ms_per_step'])"
done; done | tee $O/ab.txt
for n in 4 12; do
  (cd /tmp && MI355X_GRAPHS=0 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/st$n -o out -- python $R/bench.py --steps $n --warmup 2 --no-cpu-baseline --no-roofline > /dev/null 2>&1)
  db=$(find /tmp/st$n -name "*.db" | head -1); [ -n "$db" ] && python tools/rocpd_stats.py $db $O/stats_s$n > /dev/null
done
python tools/per_step_stats.py $O/stats_s4.csv 4 $O/stats_s12.csv 12 $O/per_step.md | head -3
grep -c "_ZN2at" $O/per_step.md
