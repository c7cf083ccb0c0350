#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/r5l; mkdir -p $O
MI355X_GRAPHS_BWD_LIVE=1 timeout 600 python -m pytest tests/test_graphs_gpu.py -x -q 2>&1 | grep -E "Error|assert|FAILED|passed|failed" | head -8 | tee $O/tests.txt
for mode in "0 eager" "1 tape"; do
  set -- $mode
  (cd /tmp && MI355X_GRAPHS=$1 timeout 300 rocprofv3 --kernel-trace -d $O/tr_$2 -o out -- python $R/bench.py --steps 6 --warmup 10 --no-cpu-baseline --no-roofline > $O/tr_$2.json 2> $O/tr_$2.err)
  db=$(find $O/tr_$2 -name "*.db" | head -1)
  echo "== $2" | tee -a $O/gap_sites.txt
  python tools/gap_sites.py $db 200 30 2>&1 | cut -c1-200 | tee -a $O/gap_sites.txt
  python -c "
import json
d=json.loads(open('$O/tr_$2.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], json.dumps(d['launch'].get('recorded'))[:600])" | tee -a $O/gap_sites.txt
  rm -rf $O/tr_$2
done
