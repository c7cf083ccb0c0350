#!/bin/bash
# v8 K-loop ablations: every tools/ab_build.py variant in nemo_amd/lib_ab on the square problems and conv2 forward (one box, one after the other)
out=gpurun_out/${1:-r6c}
mkdir -p $out
for lib in nemo_amd/lib_ab/libmi355x_asr_*.so; do
  name=$(basename $lib .so); name=${name#libmi355x_asr_}
  echo "== $name" | tee -a $out/ablate.txt
  MI355X_ASR_LIB=$PWD/$lib CHECK=0 ONLY=${ONLY:-sq} REPS=2 timeout 300 python tools/v8_probe.py 2>&1 | grep -v amdgpu.ids | tee -a $out/ablate.txt
done
