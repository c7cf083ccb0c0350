#!/bin/bash
# round-5 session-2 call D: forward GLU fusion + deferred tap reduction -- parity and in-step A/B
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r5s_d; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py tests/test_graphs_gpu.py tests/test_baseline_configs_gpu.py tests/test_packed_gpu.py -q -m gpu -k "dwconv or model or graphs or baseline or packed or large or cfg1" > $O/tests.txt 2>&1; echo "tests rc=$?"; tail -6 $O/tests.txt
timeout 500 python tools/step_ab.py "enc.fuse_glu_dwconv_fwd=0,1;enc.tap_reduce_side=0,1" 6 8 2>/dev/null | tee $O/step_ab.txt
