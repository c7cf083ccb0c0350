#!/bin/bash
# round 4 closing run on the final code (ON the GPU box): whole GPU suite, smoke(), default bench line (roofline with PMC traffic,
# cpu_baseline), the 2-rank rehearsal over the statistics mailbox and over the process group, BatchNorm-backward variants.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
O=gpurun_out/r4finalb; mkdir -p $O
timeout 800 python -m pytest tests -x -q -m gpu --durations=12 > $O/gpu_suite.log 2>&1; echo "gpu tests rc=$?"; tail -18 $O/gpu_suite.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.log
timeout 300 python bench.py --no-cpu-baseline > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
python - <<P
import json
d=json.loads(open('$O/bench_default.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step')}, d['roofline'].get('frac'), d['roofline'].get('frac_in_step'), d['roofline'].get('traffic'), d['roofline'].get('mfma_busy_frac_pmc'))
P
for mb in 1 0; do
  MI355X_SYNCBN_MAILBOX=$mb BENCH_DEVICE=0 BENCH_DIST_BACKEND=gloo timeout 200 python bench.py --gpus 2 --steps 6 --warmup 3 --no-cpu-baseline --no-roofline > $O/bench_gloo2_mailbox$mb.json 2> $O/bench_gloo2_mailbox$mb.err; echo "gloo2 mailbox=$mb rc=$?"
  python - <<P
import json
try:
    d=json.loads(open('$O/bench_gloo2_mailbox$mb.json').read().strip().splitlines()[-1]); x=d['distributed']
    print(d['ms_per_step'], x.get('syncbn_exchange'), x.get('syncbn_allreduces_per_step'), x.get('syncbn_exposed_ms_this_rank'))
except Exception as e: print('ERR', e)
P
done
bash tools/run_r4_sweep.sh "MI355X_BNR_ROWS=16" "MI355X_BNA_ROWS=8 MI355X_BNA_UNR=4" "MI355X_BNR_ROWS=16 MI355X_BNA_ROWS=8 MI355X_BNA_UNR=4" 2>&1 | tail -6
