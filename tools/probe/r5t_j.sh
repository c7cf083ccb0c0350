#!/bin/bash
# round-5 session-3 call J: staggered start of the lockstep GEMM workgroups -- per launch (cold operands), then in the step
cd "$(dirname "$0")/../.." || exit 1
O=$PWD/gpurun_out/r5t_j; mkdir -p $O
timeout 400 python tools/stagger_bench.py 2>&1 | grep -v amdgpu.ids | tee $O/per_launch.txt
timeout 600 python tools/step_ab.py "gemm8=0,300,600" 4 8 2>/dev/null | tee $O/step_ab.txt
