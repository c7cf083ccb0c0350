#!/bin/bash
# round-5 session-3 call I: what the side streams are worth on the current code (in-process A/B)
cd "$(dirname "$0")/../.." || exit 1
O=$PWD/gpurun_out/r5t_i; mkdir -p $O
timeout 800 python tools/step_ab.py "enc.wgrad_side_stream=1,0;enc.sub_wgrad_side_stream=1,0;enc.dpos_side_stream=1,0;enc.wgrad_layers=2,3,6" 4 8 2>/dev/null | tee $O/step_ab.txt
