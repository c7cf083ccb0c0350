#!/bin/bash
# round 4: BatchNorm-Swish backward kernels -- rows per workgroup / per thread and loop unroll, alone (tools/bn_bench.py) and in the step
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
O=gpurun_out/r4bn; mkdir -p $O; : > $O/bn.txt
for cfg in "A=1" "MI355X_BNR_ROWS=16" "MI355X_BNA_ROWS=4" "MI355X_BNA_ROWS=8" "MI355X_BNA_ROWS=8 MI355X_BNA_UNR=4" "MI355X_BNA_ROWS=4 MI355X_BNA_UNR=4" "MI355X_BNA_ROWS=16 MI355X_BNA_UNR=4"; do
  echo "## $cfg" | tee -a $O/bn.txt
  env $cfg timeout 60 python tools/bn_bench.py 2>/dev/null | grep bwd | tee -a $O/bn.txt
done
bash tools/run_r4_sweep.sh "MI355X_BNR_ROWS=16" "MI355X_BNA_ROWS=8 MI355X_BNA_UNR=4" "MI355X_BNR_ROWS=16 MI355X_BNA_ROWS=8 MI355X_BNA_UNR=4" "MI355X_BNR_ROWS=16 MI355X_BNA_ROWS=4 MI355X_BNA_UNR=4"
