#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r5n; mkdir -p $O
echo "# upper bounds: what the dropout masks cost inside the step (attention probabilities / GEMM epilogues + LN casts)" | tee -a $O/ab.txt
timeout 400 python tools/step_ab.py enc.dropout_att=0.1,0.0 5 10 2>/dev/null | tail -2 | tee -a $O/ab.txt
timeout 400 python tools/step_ab.py enc.dropout=0.1,0.0 5 10 2>/dev/null | tail -2 | tee -a $O/ab.txt
