#!/bin/bash
# same-box A/B of the round-3 HEAD, the round-4 HEAD and this tree: headline, Squeezeformer-Medium,
# FastConformer-Transducer; arms interleaved, two passes
cd "$(dirname "$0")/.." || exit 1
R=$PWD; O=$R/gpurun_out/r5p; mkdir -p $O
run() {  # tree tag args...
  tree=$1; tag=$2; shift 2
  (cd $tree && timeout 300 python bench.py "$@" --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', '$*', d['ms_per_step'])") | tee -a $O/ab.txt
}
for rep in 1 2; do
for t in "ab_head/r3 r3" "ab_head/r4 r4" ". r5"; do
  set -- $t
  run $1 $2 --steps 16 --warmup 8
  run $1 $2 --model squeezeformer --size medium --steps 12 --warmup 6
  run $1 $2 --model transducer --steps 12 --warmup 6
done; done
