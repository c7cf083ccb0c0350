#!/bin/bash
# Closing run of a round on the final code (ON the GPU box): whole GPU suite, smoke(), the default bench line (roofline with the
# PMC traffic table of this source hash, cpu_baseline), the other recipes' bench lines and the 2-rank rehearsals (two processes on
# the one GPU, gloo) over the statistics mailbox and over the process group.   usage: tools/run_final.sh <tag> [parts]
#   parts (default "suite smoke bench models gloo2")          (folds the former run_final_r2 ... r4b.sh)
cd "$(dirname "$0")/.." || exit 1
tag=${1:-rX}; PARTS=${2:-"suite smoke bench models gloo2"}
O=gpurun_out/${tag}final; mkdir -p $O
line() { python -c "
import json,sys
d=json.loads(open('$1').read().strip().splitlines()[-1]); r=d.get('roofline') or {}
print('$1', {k:d[k] for k in ('value','ms_per_step')}, r.get('frac'), r.get('frac_in_step'), r.get('traffic'), r.get('mfma_busy_frac_pmc'))" 2>&1 | tail -1; }
for part in $PARTS; do
case $part in
suite) timeout 1200 python -m pytest tests -x -q -m gpu --durations=8 > $O/gpu_suite.log 2>&1; echo "gpu tests rc=$?"; tail -14 $O/gpu_suite.log ;;
smoke) timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.log ;;
bench) timeout 400 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"; line $O/bench_default.json ;;
models)
  timeout 300 python bench.py --model squeezeformer --size medium --no-cpu-baseline --no-roofline > $O/bench_squeezeformer_medium.json 2>/dev/null; line $O/bench_squeezeformer_medium.json
  timeout 300 python bench.py --model transducer --no-cpu-baseline --no-roofline > $O/bench_transducer.json 2>/dev/null; line $O/bench_transducer.json
  timeout 300 python bench.py --size small --no-cpu-baseline --no-roofline > $O/bench_small_bf16.json 2>/dev/null; line $O/bench_small_bf16.json
  for s in semisort random; do
    timeout 300 python bench.py --var-len 5:30 --sampler $s --steps 16 --warmup 4 > $O/bench_varlen_$s.json 2>/dev/null; line $O/bench_varlen_$s.json
  done
  timeout 300 python bench.py --model squeezeformer --size medium --var-len 5:30 --sampler semisort --steps 16 --warmup 4 > $O/bench_squeezeformer_varlen_semisort.json 2>/dev/null; line $O/bench_squeezeformer_varlen_semisort.json ;;
gloo2)
  for mb in 1 0; do
    MI355X_SYNCBN_MAILBOX=$mb BENCH_DEVICE=0 BENCH_DIST_BACKEND=gloo timeout 300 python bench.py --gpus 2 --steps 6 --warmup 3 --no-cpu-baseline --no-roofline > $O/bench_gloo2_mailbox$mb.json 2> $O/bench_gloo2_mailbox$mb.err; echo "gloo2 mailbox=$mb rc=$?"
    python -c "
import json
try:
    d=json.loads(open('$O/bench_gloo2_mailbox$mb.json').read().strip().splitlines()[-1]); x=d['distributed']
    print(d['ms_per_step'], x.get('syncbn_exchange'), x.get('syncbn_allreduces_per_step'), x.get('syncbn_exposed_ms_this_rank'))
except Exception as e: print('ERR', e)"
  done ;;
esac
done
