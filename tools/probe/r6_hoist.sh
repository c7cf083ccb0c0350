#!/bin/bash
# generic-epilogue LDS-read hoisting: micro (layer shapes, rotating operands) and step bench, two libraries interleaved on one box
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r6hoist; mkdir -p $O
for rep in 1 2; do for v in base hoist; do
  echo "== $v rep $rep" >> $O/micro.txt
  MI355X_ASR_LIB=$PWD/nemo_amd/lib_ab/libmi355x_asr_$v.so ROTATE=4 timeout 200 python tools/gemm_bench.py 2>&1 | grep -v "^$" >> $O/micro.txt
done; done
bash tools/run_ab.sh libs 16 > $O/libs.txt 2>&1
