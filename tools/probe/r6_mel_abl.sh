#!/bin/bash
# where the register-FFT front-end kernel's time goes: compile-time ablations (tools/ab_build.py ...=@mel:-DR16_ABL=n)
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r6mel; mkdir -p $O
for v in abl1 abl2 abl3; do
  echo "== $v" >> $O/abl.txt
  MI355X_ASR_LIB=$PWD/nemo_amd/lib_ab/libmi355x_asr_$v.so python tools/mel_bench.py 2>&1 | grep "variant=2" >> $O/abl.txt
done
