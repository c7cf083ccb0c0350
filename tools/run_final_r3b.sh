# Round-3 closing evidence run, part 2: the other recipes' bench lines, the variable-length workload, the 2-rank rehearsal
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r3final
mkdir -p $O
timeout 200 python bench.py --model transducer --no-roofline --no-cpu-baseline > $O/bench_transducer.json 2> $O/bench_transducer.err
timeout 200 python bench.py --model squeezeformer --size medium --no-roofline --no-cpu-baseline > $O/bench_sq_fixed.json 2> $O/bench_sq_fixed.err
timeout 200 python bench.py --model squeezeformer --size medium --var-len 5:30 --sampler semisort --no-cpu-baseline > $O/bench_sq_var_semisort.json 2> $O/bench_sq_var_semisort.err
timeout 200 python bench.py --var-len 5:30 --sampler semisort --no-cpu-baseline > $O/bench_ctc_var_semisort.json 2> $O/bench_ctc_var_semisort.err
BENCH_DEVICE=0 BENCH_DIST_BACKEND=gloo timeout 300 python bench.py --gpus 2 --steps 6 --warmup 3 --no-cpu-baseline --no-roofline > $O/bench_gloo2.json 2> $O/bench_gloo2.err
for f in $O/bench_*.json; do echo $f; tail -1 $f | cut -c1-220; done
