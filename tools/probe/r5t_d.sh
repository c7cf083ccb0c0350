#!/bin/bash
# round-5 session-3 call D: bench.py --size small died with SIGSEGV in the closing run -- traceback and bisection by switch
cd "$(dirname "$0")/../.." || exit 1
O=$PWD/gpurun_out/r5t_d; mkdir -p $O
run() { lab=$1; shift; env "$@" timeout 200 python -X faulthandler bench.py --size small --steps 4 --warmup 2 --no-cpu-baseline --no-roofline > $O/$lab.json 2> $O/$lab.err; echo "$lab rc=$? $(tail -c 300 $O/$lab.json | cut -c1-120)"; grep -n "File \|Fatal\|Segmentation" $O/$lab.err | head -12; }
run default A=1
run graphs0 MI355X_GRAPHS=0
run padskip0 MI355X_PAD_SKIP=0 MI355X_GRAPHS=0
run glufuse0 MI355X_GLU_DW_FUSE_FWD=0 MI355X_GLU_DW_FUSE=0 MI355X_BN_DW_FUSE=0 MI355X_GRAPHS=0
run ctcwave0 MI355X_CTC_WAVE=0 MI355X_GRAPHS=0
run padheads0 MI355X_FLASH_PAD_HEADS=0 MI355X_GRAPHS=0
