#!/bin/bash
# round-5 session-3 call C: batched staging in the depthwise kernels (loads of all passes in flight together): parity of the conv
# module and the models, kernels alone (tools/dw_bench.py if it covers the fused forms), then library A/B interleaved on one box
cd "$(dirname "$0")/../.." || exit 1
R=$PWD
O=$R/gpurun_out/r5t_c; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py tests/test_packed_gpu.py tests/test_squeezeformer_gpu.py tests/test_baseline_configs_gpu.py -x -q -m gpu -k "dwconv or conv or model or packed or squeezeformer or large or cfg1" > $O/tests.txt 2>&1; echo "tests rc=$?"; tail -4 $O/tests.txt
for rep in 1 2 3; do
  for arm in new old; do
    lib=$R/nemo_amd/lib/libmi355x_asr.so; [ $arm = old ] && lib=$R/nemo_amd/lib_ab/before_dwbatch.so
    MI355X_ASR_LIB=$lib MI355X_GRAPHS=0 timeout 200 python bench.py --steps 20 --warmup 6 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ctc-large $arm ms/step', d['ms_per_step'])"
  done
done | tee $O/ab.txt
for arm in new old; do
  lib=$R/nemo_amd/lib/libmi355x_asr.so; [ $arm = old ] && lib=$R/nemo_amd/lib_ab/before_dwbatch.so
  (cd /tmp && MI355X_ASR_LIB=$lib MI355X_GRAPHS=0 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/st_$arm -o out -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-roofline > /dev/null 2>&1)
  db=$(find /tmp/st_$arm -name "*.db" | head -1); [ -n "$db" ] && python tools/rocpd_stats.py $db $O/stats_$arm > /dev/null
  grep -i "dwconv\|bn_swish_bwd_reduce\|grouped" $O/stats_$arm.md | cut -c1-150
done
