#!/bin/bash
# round 4, second batch: Swish-derivative-in-forward GEMM pair + streaming depthwise conv, tests and in-step A/B
mkdir -p gpurun_out/r4b
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "swish or gemm_epilogues or dwconv" > gpurun_out/r4b/tests_kernels.log 2>&1; echo "kernel tests rc=$?"; tail -3 gpurun_out/r4b/tests_kernels.log
timeout 900 python -m pytest tests/test_model_gpu.py tests/test_baseline_configs_gpu.py -x -q > gpurun_out/r4b/tests_model.log 2>&1; echo "model tests rc=$?"; tail -3 gpurun_out/r4b/tests_model.log
timeout 300 python tools/step_ab.py enc.swish_g=0,1 5 8 2>&1 | tail -2
for m in 0 1 0 1; do
  MI355X_DWCONV_STREAM=$m timeout 300 python bench.py --steps 20 --warmup 8 --no-cpu-baseline --no-roofline > gpurun_out/r4b/bench_dw$m.json 2>gpurun_out/r4b/bench_dw$m.err
  echo "DWCONV_STREAM=$m: $(python -c "import json;d=json.loads(open('gpurun_out/r4b/bench_dw$m.json').read().strip().splitlines()[-1]);print(d['ms_per_step'])")"
done
