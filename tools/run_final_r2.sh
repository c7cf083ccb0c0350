# Round-2 closing evidence run (ON the GPU box, one gpurun call): full GPU suite, smoke(), kernel-trace stats of the headline
# bench, the three PMC passes (FETCH_SIZE | WRITE_SIZE | MFMA busy + GUI active; never mixed with API traces), the traffic
# table assembled from them (stamped with the GEMM source hash) and then the default bench line that reads it.
# Everything lands in gpurun_out/r2final/.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r2final
rm -rf $O; mkdir -p $O
timeout 300 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 > $O/gpu_suite.log; cat $O/gpu_suite.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 > $O/smoke.log; cat $O/smoke.log
python -c "import bench; print(bench._source_hash())" > $O/source_hash.txt 2>/dev/null
finddb() { find $1 -name "*.db" | head -1; }
stats() {
  n=$1; shift
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $O/$n -o out -- python $GRAFT_REPO_ROOT/bench.py "$@" > $O/$n.json 2> $O/$n.err)
  db=$(finddb $O/$n)
  [ -n "$db" ] && python tools/rocpd_stats.py $db $O/$n > /dev/null
  rm -rf $O/$n
  tail -1 $O/$n.json | cut -c1-160
}
pmc() {
  n=$1; shift
  (cd /tmp && MI355X_WGRAD_STREAM=0 timeout 300 rocprofv3 --kernel-trace --pmc "$@" -d $O/$n -o out -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline > $O/$n.json 2> $O/$n.err)
  db=$(finddb $O/$n)
  [ -n "$db" ] && python tools/pmc_dump.py $db $O/$n.pmc.json
  rm -rf $O/$n
}
stats stats_s12 --steps 12 --warmup 2 --no-cpu-baseline --no-roofline
pmc pmc_fetch FETCH_SIZE
pmc pmc_write WRITE_SIZE
pmc pmc_mfma SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE
mkdir -p $O/assembled
python tools/pmc_assemble.py $O $O/assembled r2 > $O/assemble.log 2>&1 && cp $O/assembled/r2_gemm_traffic.json profiles/r2_gemm_traffic.json
timeout 400 python bench.py > $O/bench_default.json 2> $O/bench_default.err
tail -1 $O/bench_default.json | cut -c1-400
ls $O
timeout 200 python bench.py --model transducer --steps 6 --warmup 2 --no-cpu-baseline --no-roofline > $O/bench_transducer.json 2> $O/bench_transducer.err; tail -1 $O/bench_transducer.json | cut -c1-200
timeout 200 python bench.py --model squeezeformer --size medium --steps 6 --warmup 2 --no-cpu-baseline --no-roofline > $O/bench_squeezeformer.json 2> $O/bench_squeezeformer.err; tail -1 $O/bench_squeezeformer.json | cut -c1-200
