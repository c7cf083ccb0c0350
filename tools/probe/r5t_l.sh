#!/bin/bash
# round-5 session-3 call L: the GEMM structure knobs re-judged by the step on the closing code (in-process A/B)
cd "$(dirname "$0")/../.." || exit 1
O=$PWD/gpurun_out/r5t_l; mkdir -p $O
timeout 700 python tools/step_ab.py "gemm6=0,1;gemm7=1,0;gemm5=1,2;gemm4=1,2" 3 8 2>/dev/null | tee $O/step_ab.txt
