#!/bin/bash
# builds an ablation variant of the library into a scratch dir and times the v2 GEMM with parts switched off
set -e
export MI355X_GEMM_V3=0
for dbg in 0 1 2 3 4 8 12 13; do
  echo "== MI355X_GEMM_DBG=$dbg (1 skip epilogue, 2 skip K loop, 4 skip MFMA, 8 skip in-loop loads)"
  MI355X_GEMM_DBG=$dbg ONLY=${ONLY:-ffn} ITERS=20 python tools/gemm_bench.py 2>&1 | grep -v amdgpu.ids
done
