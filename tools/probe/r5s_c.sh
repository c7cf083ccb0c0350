#!/bin/bash
# round-5 session-2 call C: pre-pack on the side stream -- parity (model / graph / baseline suites) and in-step A/B
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r5s_c; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_model_gpu.py tests/test_graphs_gpu.py tests/test_baseline_configs_gpu.py tests/test_squeezeformer_gpu.py tests/test_packed_gpu.py -q -m gpu > $O/tests.txt 2>&1; echo "tests rc=$?"; tail -6 $O/tests.txt
timeout 500 python tools/step_ab.py "enc.prepack_side=0,1;enc.fuse_glu_dwconv_bwd=1,0" 6 8 2>/dev/null | tee $O/step_ab.txt
