#!/bin/bash
# round-5 session-3 call H: robustness sweep -- every model size / recipe / launch mode bench.py offers runs to its JSON line
cd "$(dirname "$0")/../.." || exit 1
O=$PWD/gpurun_out/r5t_h; mkdir -p $O
run() { lab=$1; shift; timeout 250 python bench.py "$@" --no-cpu-baseline --no-roofline > $O/$lab.json 2> $O/$lab.err; rc=$?; echo "$lab rc=$rc $(python -c "
import json; d=json.loads(open('$O/$lab.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], round(d['value']), d['launch']['mode'][:40])" 2>&1 | tail -1)"; [ $rc != 0 ] && tail -3 $O/$lab.err; }
for sz in xs sm medium ml; do run ctc_$sz --size $sz --steps 6 --warmup 3; done
for sz in xs sm ml large; do run sq_$sz --model squeezeformer --size $sz --steps 6 --warmup 3; done
run tr_large --model transducer --steps 6 --warmup 3
run ctc_fp32_small --size small --dtype fp32 --batch 4 --secs 5 --steps 4 --warmup 2
run ctc_large_b8_5s --batch 8 --secs 5 --steps 6 --warmup 3
run varlen_tape --var-len 5:30 --sampler semisort --launch tape --pad-to 64 --steps 12 --warmup 3
run varlen_auto --var-len 5:30 --sampler bucket --launch auto --steps 12 --warmup 3
run sq_varlen_random_packed --model squeezeformer --size medium --var-len 5:30 --sampler random --packed 1 --steps 8 --warmup 3
run tr_varlen --model transducer --var-len 5:30 --sampler semisort --steps 8 --warmup 3
