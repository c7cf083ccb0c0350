#!/bin/bash
# round 4 closing evidence (ON the GPU box): default bench line, per-step kernel stats of the same command, the other two models.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
O=gpurun_out/r4final; mkdir -p $O
timeout 420 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
python - <<P
import json
d=json.loads(open('$O/bench_default.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step')}, d['roofline'].get('frac'), d['roofline'].get('frac_in_step'), str(d['launch'].get('mode'))[:60], d['launch'].get('host_issue_ms_per_step'))
P
MI355X_GRAPHS= bash tools/run_stats_r4.sh final > $O/stats.log 2>&1; echo "stats rc=$?"; cp gpurun_out/r4/prof_final/per_step.md $O/per_step.md 2>/dev/null; head -20 $O/per_step.md | cut -c1-120
timeout 200 python bench.py --model squeezeformer --size medium --var-len 5:30 --no-cpu-baseline --no-roofline > $O/bench_sqf_varlen.json 2> $O/bench_sqf_varlen.err; echo "sqf varlen rc=$?"; tail -1 $O/bench_sqf_varlen.json | cut -c1-300
timeout 200 python bench.py --model squeezeformer --size medium --no-cpu-baseline --no-roofline > $O/bench_sqf.json 2> $O/bench_sqf.err; echo "sqf rc=$?"; tail -1 $O/bench_sqf.json | cut -c1-200
timeout 200 python bench.py --model transducer --no-cpu-baseline --no-roofline > $O/bench_transducer.json 2> $O/bench_transducer.err; echo "transducer rc=$?"; tail -1 $O/bench_transducer.json | cut -c1-200
