#!/bin/bash
# why does the 128 x 256 half tile (key 8 mode 5) win alone and lose in the step?  kernel traces of both arms on one box
cd "$(dirname "$0")/../.." || exit 1
O=$PWD/gpurun_out/r6ah1; mkdir -p $O
export TMPDIR=/tmp MI355X_GRAPHS=0
for mode in 1 5 1 5; do
  n=m${mode}_$(date +%s)
  (cd /tmp && MI355X_GEMM_V8=$mode timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/rp_$n -o out -- python /root/repo/bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-roofline > $O/$n.json 2>/dev/null)
  db=$(find /tmp/rp_$n -name "*.db" | head -1); [ -n "$db" ] && python tools/rocpd_stats.py $db $O/$n > /dev/null; rm -rf /tmp/rp_$n
  echo "mode $mode: $(python -c "import json;print(json.loads(open('$O/$n.json').read().strip().splitlines()[-1])['ms_per_step'])")" | tee -a $O/summary.txt
done
