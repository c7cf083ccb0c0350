#!/bin/bash
# round-5 session-2 call B: whole GPU suite on the fused conv-module backward / wave CTC / per-length recordings, in-step A/B of the
# GLU write-out and the weight-gradient entry point, step benches of the three recipes
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r5s_b; mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -q -m gpu --durations=8 > $O/gpu_suite.txt 2>&1; echo "suite rc=$?"; tail -15 $O/gpu_suite.txt
timeout 500 python tools/step_ab.py "enc.fuse_glu_dwconv_bwd=0,1;enc.wgrad_defer=1,5,6" 5 8 2>/dev/null | tee $O/step_ab.txt
ms() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], d['launch']['mode'][:40])" 2>/dev/null || echo ERR; }
echo "ctc-large $(timeout 200 python bench.py --steps 20 --warmup 6 --no-cpu-baseline --no-roofline 2>/dev/null | ms)" | tee -a $O/bench.txt
echo "squeezeformer-medium $(timeout 300 python bench.py --model squeezeformer --size medium --steps 12 --warmup 6 --no-cpu-baseline --no-roofline 2>/dev/null | ms)" | tee -a $O/bench.txt
echo "transducer $(timeout 300 python bench.py --model transducer --steps 12 --warmup 6 --no-cpu-baseline --no-roofline 2>/dev/null | ms)" | tee -a $O/bench.txt
