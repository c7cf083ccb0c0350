# round-3 mid-round evidence run (ON the GPU box)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r3c
mkdir -p $O
finddb() { find $1 -name "*.db" | head -1; }
stats() {  # name, bench args...
  n=$1; shift
  (cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d $O/$n -o out -- python $GRAFT_REPO_ROOT/bench.py "$@" > $O/$n.json 2> $O/$n.err)
  db=$(finddb $O/$n)
  [ -n "$db" ] && python tools/rocpd_stats.py $db $O/$n > /dev/null
  rm -rf $O/$n
  tail -1 $O/$n.json | cut -c1-200
}
timeout 300 python bench.py > $O/bench_default.json 2> $O/bench_default.err
timeout 300 python tools/host_phases.py squeezeformer --json $O/host_squeezeformer.json > $O/host_squeezeformer.log 2>&1
timeout 200 python bench.py --model squeezeformer --size medium --no-roofline > $O/bench_sq_fixed.json 2> $O/bench_sq_fixed.err
timeout 200 python bench.py --model squeezeformer --size medium --var-len 5:30 --sampler semisort > $O/bench_sq_var_semisort.json 2> $O/bench_sq_var_semisort.err
timeout 200 python bench.py --model squeezeformer --size medium --var-len 5:30 --sampler bucket > $O/bench_sq_var_bucket.json 2> $O/bench_sq_var_bucket.err
timeout 200 python bench.py --model squeezeformer --size medium --var-len 5:30 --sampler random > $O/bench_sq_var_random.json 2> $O/bench_sq_var_random.err
timeout 200 python bench.py --var-len 5:30 --sampler semisort > $O/bench_ctc_var_semisort.json 2> $O/bench_ctc_var_semisort.err
timeout 200 python bench.py --model transducer --no-roofline > $O/bench_transducer.json 2> $O/bench_transducer.err
stats stats_s4 --steps 4 --warmup 3 --no-cpu-baseline --no-roofline
stats stats_s12 --steps 12 --warmup 3 --no-cpu-baseline --no-roofline
ls -la $O
