#!/bin/bash
# ablation ladder of the fused feed-forward kernel: rebuilds csrc/ffn.hip with -DFFN_ABLATE (run-time MI355X_FFN_DBG bits:
# 1 no LDS-DMA, 2 no phase-1 MFMAs, 4 no phase-2 MFMAs, 8 no transform, 16 no fragment reads) and times each combination.
out=${1:-gpurun_out/r4/ffn_ablate.log}
mkdir -p $(dirname $out); : > $out
touch nemo_amd/csrc/ffn.hip
MI355X_EXTRA_HIPCC_FLAGS=-DFFN_ABLATE python -m nemo_amd.build >> $out 2>&1
for dbg in ${DBGS:-0 1 2 4 6 8 16 22 30 31 9 25}; do
  echo "== MI355X_FFN_DBG=$dbg" >> $out
  MI355X_FFN_DBG=$dbg ONLY=fused ROUNDS=2 ITERS=20 timeout 120 python tools/ffn_bench.py 2>&1 | grep -v amdgpu.ids >> $out
done
touch nemo_amd/csrc/ffn.hip
python -m nemo_amd.build >> $out 2>&1
cat $out
