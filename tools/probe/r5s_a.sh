#!/bin/bash
# round-5 session-2 call A: parity of the new kernels / launch-sequence logic, in-step A/B of the fused conv-module backward and the
# weight-gradient entry points, variable-length Squeezeformer with recorded sequences per padded length
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r5s_a; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_graphs_gpu.py tests/test_model_gpu.py tests/test_squeezeformer_gpu.py tests/test_packed_gpu.py -x -q -m gpu -k "ctc or dwconv_bn or graphs or model or squeezeformer or packed" --durations=6 > $O/tests.txt 2>&1; echo "tests rc=$?"; tail -12 $O/tests.txt
timeout 120 python tools/probe/ctc_time.py 2>&1 | tee $O/ctc_time.txt
timeout 500 python tools/step_ab.py "enc.fuse_bn_dwconv_bwd=0,1;enc.wgrad_defer=1,5,6;cfg:ctc=0,1" 5 8 2>/dev/null | tee $O/step_ab.txt
for mode in tape live; do
  timeout 400 python bench.py --model squeezeformer --size medium --var-len 5:30 --sampler semisort --pad-to 64 --launch $mode --steps 40 --warmup 4 > $O/sq_varlen_$mode.json 2> $O/sq_varlen_$mode.err
  echo "sq varlen $mode rc=$?"
  python - <<PY
import json
try:
    d=json.loads(open("$O/sq_varlen_$mode.json").read().strip().splitlines()[-1])
    v=d["config"]["variable_length"]; l=d["launch"]
    print("$mode", d["value"], d["ms_per_step"], v.get("ms_per_step_min_median_max"), v.get("distinct_padded_lengths"), l.get("host_issue_ms_per_step"), l.get("variable_length_previsit"))
except Exception as e:
    print("ERR", e); print(open("$O/sq_varlen_$mode.err").read()[-1500:])
PY
done
