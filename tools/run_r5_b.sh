#!/bin/bash
# round 5, GPU call B: the ADVICE fixes' tests + heads padded to the fused attention width (Conformer-Small bf16)
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r5b; mkdir -p $O
timeout 600 python -m pytest tests/test_graphs_gpu.py tests/test_mailbox_gpu.py -x -q 2>&1 | tail -5 | tee $O/tests_advice.txt
timeout 300 python -m pytest tests/test_model_gpu.py -x -q -k "bf16_mfma or optimizer_behind" 2>&1 | tail -5 | tee $O/tests_pad.txt
timeout 300 python -m pytest tests/test_rnnt_decoding.py tests/test_squeezeformer_gpu.py -x -q 2>&1 | tail -3 | tee $O/tests_rnnt_sq.txt
for rep in 1 2; do
for v in 0 1; do
  MI355X_FLASH_PAD_HEADS=$v timeout 200 python bench.py --size small --steps 12 --warmup 4 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 \
   | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('small bf16 B=32x20s pad_heads=$v ms_per_step', d['ms_per_step'])" | tee -a $O/small_pad_heads.txt
done; done
