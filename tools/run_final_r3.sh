# Round-3 closing evidence run, part 1 (ON the GPU box): full GPU suite, smoke(), kernel-trace stats of the headline bench at two
# step counts (per-step launch counts), then the default bench line (roofline + cpu_baseline).  Everything lands in gpurun_out/r3final/.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r3final
mkdir -p $O
timeout 400 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 > $O/gpu_suite.log; cat $O/gpu_suite.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 > $O/smoke.log; cat $O/smoke.log
R3_PARTS=stats bash tools/run_profiles_r3.sh > $O/profiles.log 2>&1; tail -3 $O/profiles.log
timeout 300 python bench.py > $O/bench_default.json 2> $O/bench_default.err; cut -c1-300 $O/bench_default.json
