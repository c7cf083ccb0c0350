#!/bin/bash
# Evidence run of a round (ON the GPU box, inside one gpurun call).  Everything the round's profiles/ tables are made from:
#   tools/run_evidence.sh <tag> [parts]         parts (default "stats pmc models"):
#     stats   rocprofv3 --kernel-trace --stats of the headline bench at two step counts -> per-step launch counts / kernel time
#     pmc     three counter passes (FETCH_SIZE | WRITE_SIZE | SQ_VALU_MFMA_BUSY_CYCLES + GRBM_GUI_ACTIVE; never mixed with API traces,
#             weight-gradient side stream off so that every row is one kernel on its own) -> per-kernel HBM bytes and MFMA busy
#     models  kernel stats of the configs[3] / configs[4] recipes (FastConformer-Transducer, Squeezeformer-Medium)
#     kernel:<tool.py>  three SQ counter passes around one micro-benchmark of tools/ (attn_bench.py, gemm_bench.py, dw_bench.py ...)
# Output: gpurun_out/<tag>prof/ ; afterwards, HERE:  python tools/pmc_assemble.py gpurun_out/<tag>prof profiles <tag>
#                                                    python tools/roofline_table.py profiles <tag> 3
# (folds the former run_profiles_r2/r3.sh, run_stats_r4.sh, run_pmc_{attn,dw,ffn,gemm}.sh)
cd "$(dirname "$0")/.." || exit 1
R=$PWD
export TMPDIR=/tmp MI355X_GRAPHS=0   # live launches: no recording / trial steps inside the traces
tag=${1:-rX}; PARTS=${2:-"stats pmc models"}
O=$R/gpurun_out/${tag}prof; mkdir -p $O
python -c "import bench; print(bench._source_hash())" > $O/source_hash.txt 2>/dev/null
finddb() { find $1 -name "*.db" | head -1; }
stats() {  # name, bench args...
  n=$1; shift
  (cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d $O/$n -o out -- python $R/bench.py "$@" > $O/$n.json 2> $O/$n.err)
  db=$(finddb $O/$n); [ -n "$db" ] && python tools/rocpd_stats.py $db $O/$n > /dev/null
  rm -rf $O/$n; tail -1 $O/$n.json | cut -c1-200
}
pmc() {  # name, counters...
  n=$1; shift
  (cd /tmp && MI355X_WGRAD_STREAM=0 timeout 500 rocprofv3 --kernel-trace --pmc "$@" -d $O/$n -o out -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline > $O/$n.json 2> $O/$n.err)
  db=$(finddb $O/$n); [ -n "$db" ] && python tools/pmc_dump.py $db $O/$n.pmc.json
  rm -rf $O/$n
}
for part in $PARTS; do
case $part in
stats)
  stats stats_s4 --steps 4 --warmup 2 --no-cpu-baseline --no-roofline
  stats stats_s12 --steps 12 --warmup 2 --no-cpu-baseline --no-roofline
  python tools/per_step_stats.py $O/stats_s4.csv 4 $O/stats_s12.csv 12 $O/per_step.md && head -30 $O/per_step.md | cut -c1-160 ;;
pmc)
  pmc pmc_fetch FETCH_SIZE
  pmc pmc_write WRITE_SIZE
  pmc pmc_mfma SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE ;;
models)
  stats stats_transducer --model transducer --steps 3 --warmup 2 --no-roofline --no-cpu-baseline
  stats stats_squeezeformer --model squeezeformer --size medium --steps 3 --warmup 2 --no-roofline --no-cpu-baseline ;;
kernel:*)
  tool=${part#kernel:}; K=$O/kernel_${tool%.py}; mkdir -p $K; : > $K/summary.md
  P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT"
  P2="SQ_LDS_IDX_ACTIVE SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU"
  P3="GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_RD SQ_INSTS_VMEM_RD SQ_WAVES"
  i=0
  for P in "$P1" "$P2" "$P3"; do
    i=$((i+1))
    (cd /tmp && ITERS=4 timeout 400 rocprofv3 --kernel-trace --pmc $P -d $K/p$i -o out -- python $R/tools/$tool > /dev/null 2>&1)
    db=$(finddb $K/p$i); echo "## pass $i" >> $K/summary.md
    [ -n "$db" ] && python tools/pmc_summary.py $db | cut -c1-300 >> $K/summary.md
    rm -rf $K/p$i
  done
  head -60 $K/summary.md ;;
esac
done
ls $O
