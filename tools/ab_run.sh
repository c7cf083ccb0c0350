#!/bin/bash
# Same-box A/B of the library variants built by tools/ab_build.py.  Run ON the GPU box (inside one gpurun call):
#   bash tools/ab_run.sh [bench-steps] [gemm-bench filter]
# For every nemo_amd/lib_ab/libmi355x_asr_<name>.so: the GEMM / model parity subset, the GEMM micro-benchmark, and the
# step benchmark twice (interleaved over the variants, so drift hits all of them alike).  Results: gpurun_out/ab_<name>.*
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
STEPS=${1:-12}
FILTER=${2:-}
LIBS=(nemo_amd/lib_ab/libmi355x_asr_*.so)
for lib in "${LIBS[@]}"; do
  name=$(basename "$lib" .so); name=${name#libmi355x_asr_}
  export MI355X_ASR_LIB=$PWD/$lib
  timeout 120 python -m pytest tests -m gpu -x -q -k "gemm or wgrad or conv2 or model_matches or bf16" 2>&1 | tail -1 | sed "s/^/[$name] parity: /"
  ONLY=$FILTER ITERS=30 timeout 90 python tools/gemm_bench.py 2>&1 | grep TFLOP > gpurun_out/ab_${name}.gemm.txt
done
for rep in 1 2; do
  for lib in "${LIBS[@]}"; do
    name=$(basename "$lib" .so); name=${name#libmi355x_asr_}
    export MI355X_ASR_LIB=$PWD/$lib
    timeout 100 python bench.py --steps "$STEPS" --warmup 3 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 \
      | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('[$name] rep $rep ms_per_step', d['ms_per_step'])" | tee -a gpurun_out/ab_steps.txt
  done
done
unset MI355X_ASR_LIB
python - <<'PY'
import glob, os
files = sorted(glob.glob("gpurun_out/ab_*.gemm.txt"))
names = [os.path.basename(f)[3:-9] for f in files]
rows = {}
for n, f in zip(names, files):
    for line in open(f):
        p = line.split()
        rows.setdefault(p[0], {})[n] = p[-4]
print("shape".ljust(28), *[n.rjust(10) for n in names])
for k, v in rows.items():
    print(k.ljust(28), *[v.get(n, "-").rjust(10) for n in names])
PY
