# Round-4 kernel-trace stats of the headline bench (ON the GPU box): two step counts -> per-step table (tools/per_step_stats.py).
# usage: tools/run_stats_r4.sh <tag> [env assignments...]   e.g. tools/run_stats_r4.sh unfused MI355X_FFN_FUSED=0
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
export MI355X_GRAPHS=0
tag=$1; shift
for kv in "$@"; do export "$kv"; done
O=$GRAFT_REPO_ROOT/gpurun_out/r4/prof_$tag
mkdir -p $O
finddb() { find $1 -name "*.db" | head -1; }
stats() {
  n=$1; shift
  (cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d $O/$n -o out -- python $GRAFT_REPO_ROOT/bench.py "$@" > $O/$n.json 2> $O/$n.err)
  db=$(finddb $O/$n)
  [ -n "$db" ] && python tools/rocpd_stats.py $db $O/$n > /dev/null
  rm -rf $O/$n
  tail -1 $O/$n.json | cut -c1-160
}
stats stats_s4 --steps 4 --warmup 2 --no-cpu-baseline --no-roofline
stats stats_s12 --steps 12 --warmup 2 --no-cpu-baseline --no-roofline
python tools/per_step_stats.py $O/stats_s4.csv 4 $O/stats_s12.csv 12 $O/per_step.md
head -45 $O/per_step.md
