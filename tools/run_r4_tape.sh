#!/bin/bash
# round 4: launch tapes -- parity tests, then the default step three ways (live launches / launch tape / hipGraph replay)
mkdir -p gpurun_out/r4tape
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
timeout 600 python -m pytest tests/test_graphs_gpu.py -x -q > gpurun_out/r4tape/tests.log 2>&1; echo "tests rc=$?" 
tail -5 gpurun_out/r4tape/tests.log
for mode in "0 1" "1 1" "1 0" "0 1" "1 1"; do
  set -- $mode
  MI355X_GRAPHS=$1 MI355X_TAPE=$2 timeout 300 python bench.py --steps 20 --warmup 8 --no-cpu-baseline --no-roofline > gpurun_out/r4tape/bench_g$1_t$2.json 2>gpurun_out/r4tape/bench_g$1_t$2.err
  echo "GRAPHS=$1 TAPE=$2: $(python - <<P
import json
try:
    d=json.loads(open('gpurun_out/r4tape/bench_g$1_t$2.json').read().strip().splitlines()[-1])
    print(d['ms_per_step'], d.get('launch'))
except Exception as e:
    print('ERR', e)
P
)"
done
timeout 600 python tools/host_phases.py ctc --json gpurun_out/r4tape/host_phases_ctc.json > gpurun_out/r4tape/host_phases_ctc.log 2>&1; echo "host_phases rc=$?"
grep "^ctc" gpurun_out/r4tape/host_phases_ctc.log | cut -c1-400
