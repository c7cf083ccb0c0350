#!/bin/bash
# per-kernel time of the Squeezeformer-Medium step (where does the un-fused attention at d_k = 81 stand?)
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp MI355X_GRAPHS=0
R=$PWD; O=$R/gpurun_out/r5h; mkdir -p $O
finddb() { find $1 -name "*.db" | head -1; }
stats() {
  n=$1; shift
  (cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d $O/$n -o out -- python $R/bench.py "$@" > $O/$n.json 2> $O/$n.err)
  db=$(finddb $O/$n)
  [ -n "$db" ] && python tools/rocpd_stats.py $db $O/$n > /dev/null
  rm -rf $O/$n
  tail -1 $O/$n.json | cut -c1-200
}
stats sq_s4 --model squeezeformer --size medium --steps 4 --warmup 2 --no-cpu-baseline --no-roofline
stats sq_s12 --model squeezeformer --size medium --steps 12 --warmup 2 --no-cpu-baseline --no-roofline
python tools/per_step_stats.py $O/sq_s4.csv 4 $O/sq_s12.csv 12 $O/sq_per_step.md
head -40 $O/sq_per_step.md | cut -c1-200
