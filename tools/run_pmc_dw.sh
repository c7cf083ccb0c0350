# PMC diagnosis of the depthwise-conv kernels (run ON the GPU box): issue-bound, memory-bound or waiting?
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/pmcdw; mkdir -p $O
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES"
P2="GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_MISC"
P3="TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"
i=0
: > $O/summary.md
for P in "$P1" "$P2" "$P3"; do
  i=$((i+1))
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $P -d $GRAFT_REPO_ROOT/$O/p$i -o out -- python $GRAFT_REPO_ROOT/tools/dw_bench.py > $GRAFT_REPO_ROOT/$O/p$i.log 2>&1)
  db=$(find $O/p$i -name "*.db" | head -1)
  echo "## pass $i" >> $O/summary.md
  python tools/pmc_summary.py $db | grep -i "dwconv\|kernel\|---" >> $O/summary.md
  rm -rf $O/p$i
done
cat $O/summary.md
