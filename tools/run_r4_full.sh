#!/bin/bash
# round 4: the whole GPU suite + smoke + the default bench line
mkdir -p gpurun_out/r4full
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r4full/tests.log 2>&1; echo "gpu tests rc=$?"; tail -4 gpurun_out/r4full/tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r4full/smoke.log 2>&1; echo "smoke (driver style) rc=$?"; tail -2 gpurun_out/r4full/smoke.log
timeout 600 python __graft_entry__.py smoke > gpurun_out/r4full/smoke_main.log 2>&1; echo "smoke (build first, same process) rc=$?"; tail -2 gpurun_out/r4full/smoke_main.log
timeout 600 python bench.py > gpurun_out/r4full/bench_default.json 2>gpurun_out/r4full/bench_default.err; echo "bench rc=$?"
python - <<P
import json
d=json.loads(open('gpurun_out/r4full/bench_default.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step')}, d['roofline'].get('frac'), d['roofline'].get('frac_in_step'), d['launch']['mode'][:60], d['launch']['host_issue_ms_per_step'])
P
