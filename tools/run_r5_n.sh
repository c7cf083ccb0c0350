#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r5n; mkdir -p $O
timeout 400 python tools/step_ab.py enc.wgrad_layers=1,2,3,4 6 10 2>/dev/null | tail -4 | tee -a $O/ab.txt
