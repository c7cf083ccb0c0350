# PMC diagnosis of the fused feed-forward kernels (run ON the GPU box): which pipe is busy for how much of the kernel?
# Counters in their own runs with --kernel-trace only (no API tracing).  Output: gpurun_out/r4/pmc_ffn/summary.md
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r4/pmc_ffn
mkdir -p $O; : > $O/summary.md
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT"
P2="SQ_LDS_IDX_ACTIVE SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS"
P3="GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM_RD SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_VALU SQ_INSTS_SALU"
P4="FETCH_SIZE"
P5="WRITE_SIZE"
i=0
for P in "$P1" "$P2" "$P3" "$P4" "$P5"; do
  i=$((i+1))
  (cd /tmp && ITERS=3 ROUNDS=1 ROTATE=4 timeout 150 rocprofv3 --kernel-trace --pmc $P -d $GRAFT_REPO_ROOT/$O/p$i -o out -- python $GRAFT_REPO_ROOT/tools/ffn_bench.py > /dev/null 2>&1)
  db=$(find $O/p$i -name "*.db" | head -1)
  echo "## pass $i: $P" >> $O/summary.md
  python tools/pmc_summary.py $db | grep -E "ffn_fused|gemm_bf16_v[245]" >> $O/summary.md
  rm -rf $O/p$i
done
cat $O/summary.md
