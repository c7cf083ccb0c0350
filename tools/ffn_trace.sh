#!/bin/bash
out=${1:-gpurun_out/r4/ffn_trace.log}
mkdir -p $(dirname $out); : > $out
touch nemo_amd/csrc/ffn.hip
MI355X_EXTRA_HIPCC_FLAGS="-DFFN_TRACE ${EXTRA}" python -m nemo_amd.build >> $out 2>&1
timeout 120 python tools/ffn_trace.py 2>&1 | grep -v amdgpu.ids >> $out
touch nemo_amd/csrc/ffn.hip
python -m nemo_amd.build >> $out 2>&1
grep -v "^\[build\]" $out
