#!/bin/bash
# round-5 session-3 call F: which stream pattern sends hip::Stream::EndCapture into its recursion at the Small geometry
cd "$(dirname "$0")/../.." || exit 1
O=$PWD/gpurun_out/r5t_f; mkdir -p $O
for v in "MI355X_WGRAD_STREAM=0" "MI355X_SUB_WGRAD_STREAM=0" "MI355X_DPOS_STREAM=0" "MI355X_POSPROJ_SIDE=0" "MI355X_TAP_REDUCE_SIDE=0" "MI355X_WGRAD_DEFER=0" "MI355X_WGRAD_LAYERS=1" "MI355X_FLASH_PAD_HEADS=0" "MI355X_WGRAD_GROUPED=0" "MI355X_OPT_IN_BACKWARD=0"; do
  env MI355X_GRAPHS=1 $v timeout 200 python bench.py --size small --steps 4 --warmup 2 --no-cpu-baseline --no-roofline > $O/v.json 2> $O/v.err; echo "$v rc=$? $(tail -c 200 $O/v.json | cut -c1-60)"
done 2>&1 | tee $O/summary.txt
