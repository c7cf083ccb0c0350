# PMC diagnosis of the GEMM structures (run ON the GPU box): where do the cycles of the K loop go?
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/pmcg
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT"
P2="SQ_LDS_IDX_ACTIVE SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS"
P3="GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM_RD SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_VALU SQ_INSTS_SALU"
i=0
for P in "$P1" "$P2" "$P3"; do
  i=$((i+1))
  for cfg in "v4:MI355X_GEMM_V5=0" "v5:MI355X_GEMM_V5=1" "v2:MI355X_GEMM_V5=0 MI355X_GEMM_V4=0"; do
    name=${cfg%%:*}; envs=${cfg#*:}
    (cd /tmp && env $envs ITERS=4 PMC_SHAPES=1 timeout 120 rocprofv3 --kernel-trace --pmc $P -d $GRAFT_REPO_ROOT/gpurun_out/pmcg/${name}_p$i -o out -- python $GRAFT_REPO_ROOT/tools/gemm_bench.py > /dev/null 2>&1)
    db=$(ls gpurun_out/pmcg/${name}_p$i/*/*.db 2>/dev/null | head -1)
    [ -z "$db" ] && db=$(find gpurun_out/pmcg/${name}_p$i -name "*.db" | head -1)
    echo "## $name pass $i" >> gpurun_out/pmcg/summary.md
    python tools/pmc_summary.py $db | grep gemm >> gpurun_out/pmcg/summary.md
    rm -rf gpurun_out/pmcg/${name}_p$i
  done
done
cat gpurun_out/pmcg/summary.md
