cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/tr
mkdir -p $O
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d $O/p -o out -- python $GRAFT_REPO_ROOT/bench.py --model transducer --steps 3 --warmup 2 --no-roofline > $O/b.json 2> $O/b.err)
python tools/rocpd_stats.py $(find $O/p -name "*.db" | head -1) $O/stats > /dev/null
rm -rf $O/p
head -34 $O/stats.md | cut -c1-140
BENCH_GEMM_TABLE=$O/gemm_table.txt python bench.py --model transducer --steps 3 --warmup 2 > /dev/null 2>&1; head -14 $O/gemm_table.txt
