#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r5f; mkdir -p $O
timeout 900 python -m pytest tests/test_packed_gpu.py tests/test_kernels_gpu.py -x -q 2>&1 | tail -6 | tee $O/tests.txt
for sampler in semisort random; do
for pk in 0 1; do
  timeout 300 python bench.py --var-len 5:30 --sampler $sampler --packed $pk --steps 16 --warmup 4 2>/dev/null | tail -1 > $O/varlen_${sampler}_pk$pk.json
  python - <<PY
import json
d=json.load(open("$O/varlen_${sampler}_pk$pk.json"))
v=d["config"].get("variable_length") or {}
print("$sampler packed=$pk", "valid audio-s/s", d["value"], "ms/step", d["ms_per_step"], "host_issue", d.get("launch",{}).get("host_issue_ms_per_step"), {k:v.get(k) for k in ("padded_sample_fraction","last_step_rows_packed_vs_padded","ms_per_step_min_median_max")})
PY
done; done 2>&1 | tee $O/summary.txt
