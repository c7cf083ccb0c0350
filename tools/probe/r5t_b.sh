#!/bin/bash
# round-5 session-3 call B: padded-tile skip in the sub-sampling backward (conv2 dgrad row tiles, conv2 wgrad K-tiles, out-Linear
# dgrad row tiles): parity, then variable-length benches with / without the hints on one box
cd "$(dirname "$0")/../.." || exit 1
R=$PWD
O=$R/gpurun_out/r5t_b; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py tests/test_packed_gpu.py tests/test_graphs_gpu.py tests/test_baseline_configs_gpu.py -x -q -m gpu -k "conv2 or gemm or model or packed or graphs or large or cfg1" > $O/tests.txt 2>&1; echo "tests rc=$?"; tail -5 $O/tests.txt
for pass in 1 2; do
for s in random semisort; do
for arm in 1 0; do
  MI355X_PAD_SKIP=$arm timeout 300 python bench.py --var-len 5:30 --sampler $s --packed 1 --steps 16 --warmup 4 --no-cpu-baseline --no-roofline > $O/varlen_${s}_skip$arm.json 2>/dev/null
  python -c "
import json; d=json.loads(open('$O/varlen_${s}_skip$arm.json').read().strip().splitlines()[-1]); print('$s pad_skip=$arm valid audio-s/s', d['value'], 'ms/step', d['ms_per_step'], d['config']['variable_length'].get('padded_sample_fraction'))"
done; done; done | tee $O/summary.txt
for arm in 1 0 1 0; do
  MI355X_PAD_SKIP=$arm timeout 300 python bench.py --steps 16 --warmup 6 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fixed 32x20s pad_skip=$arm ms/step', d['ms_per_step'])"
done | tee -a $O/summary.txt
