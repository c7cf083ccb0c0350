#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r5c; mkdir -p $O
timeout 900 python -m pytest tests/test_packed_gpu.py -x -q 2>&1 | tail -40 | tee $O/tests_packed.txt
