#!/bin/bash
mkdir -p gpurun_out/r4dw
timeout 300 python -m pytest tests/test_kernels_gpu.py -x -q -k "dwconv_bn_swish" > gpurun_out/r4dw/tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r4dw/tests.log
for m in 0 1 0 1; do echo "STREAM=$m"; MI355X_DWCONV_STREAM=$m timeout 120 python tools/dw_bench.py 2>&1 | tail -4; done
