# PMC diagnosis of the fused attention kernels (run ON the GPU box): where do the cycles of a key / query step go?
# NOTE (round 3): the one attempt with `timeout 150` per pass ended in three time-outs and no output (a fresh box needs 1-2 min for
# the first `import torch` alone, more under rocprofv3) -- give it 400 s per pass and a gpurun budget of >= 10 minutes.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/pmca; mkdir -p $O
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT"
P2="SQ_LDS_IDX_ACTIVE SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL"
P3="GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_VALU SQ_INSTS_SALU SQ_INST_CYCLES_SALU SQ_WAVES"
i=0
: > $O/summary.md
for P in "$P1" "$P2" "$P3"; do
  i=$((i+1))
  (cd /tmp && timeout 400 rocprofv3 --kernel-trace --pmc $P -d $GRAFT_REPO_ROOT/$O/p$i -o out -- python $GRAFT_REPO_ROOT/tools/attn_bench.py > /dev/null 2>&1)
  db=$(find $O/p$i -name "*.db" | head -1)
  echo "## pass $i" >> $O/summary.md
  python tools/pmc_summary.py $db | grep -i "relpos\|kernel\|---" >> $O/summary.md
  rm -rf $O/p$i
done
cat $O/summary.md
