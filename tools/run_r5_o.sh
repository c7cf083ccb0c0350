#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r5o; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 | tee $O/gpu_suite.txt
