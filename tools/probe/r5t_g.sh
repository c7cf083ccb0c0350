#!/bin/bash
# round-5 session-3 call G: the nested side-stream scope fixed -- Small geometry under every launch mode, graph tests, model tests
cd "$(dirname "$0")/../.." || exit 1
O=$PWD/gpurun_out/r5t_g; mkdir -p $O
for v in "MI355X_GRAPHS=auto" "MI355X_GRAPHS=1" "MI355X_GRAPHS=1 MI355X_TAPE=0" "MI355X_GRAPHS=0"; do
  env $v timeout 200 python bench.py --size small --steps 8 --warmup 3 --no-cpu-baseline --no-roofline > $O/v.json 2> $O/v.err; echo "$v rc=$? $(python -c "
import json; d=json.loads(open('$O/v.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['launch']['mode'][:60])" 2>&1 | tail -1)"
done 2>&1 | tee $O/small.txt
timeout 900 python -m pytest tests/test_graphs_gpu.py tests/test_model_gpu.py tests/test_squeezeformer_gpu.py -x -q -m gpu > $O/tests.txt 2>&1; echo "tests rc=$?"; tail -4 $O/tests.txt
