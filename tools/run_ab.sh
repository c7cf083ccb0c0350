#!/bin/bash
# Same-box A/B harness (ON the GPU box).  Box-to-box spread is +-1 ms per step: a change is only ever claimed from arms that ran
# interleaved on ONE box.    usage: tools/run_ab.sh <mode> ...        (folds the former ab_run.sh, run_r4_sweep.sh, run_r3c.sh ...)
#   knobs  "K=V ..." "K2=V2" ...   every argument = one arm of environment assignments on top of the default; the default arm is
#                                  repeated every five arms (bench.py --steps 20 --warmup 6, live launches)
#   step   KEY=V0,V1[,V2] [rounds] [steps]   in-process, interleaved (tools/step_ab.py): gemmN, env:NAME, enc.attr, model.attr
#   libs   [bench-steps]           every nemo_amd/lib_ab/libmi355x_asr_<name>.so built by tools/ab_build.py: parity subset + step bench, twice
#   trees  DIR1 DIR2 ...           whole source trees (e.g. `git archive <round head>` exports with their own built library):
#                                  headline, Squeezeformer-Medium, Transducer step benches, two interleaved passes
cd "$(dirname "$0")/.." || exit 1
mode=$1; shift
O=gpurun_out/ab; mkdir -p $O
ms() { python -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])" 2>/dev/null || echo ERR; }
case $mode in
knobs)
  run() { lab=$1; shift; echo "$lab $(env MI355X_GRAPHS=0 "$@" timeout 150 python bench.py --steps 20 --warmup 6 --no-cpu-baseline --no-roofline 2>/dev/null | ms)" | tee -a $O/knobs.txt; }
  i=0; run default A=1
  for arm in "$@"; do run "$arm" $arm; i=$((i+1)); [ $((i % 5)) -eq 0 ] && run default A=1; done
  run default A=1 ;;
step) timeout 600 python tools/step_ab.py "$@" 2>/dev/null | tail -4 | tee -a $O/step.txt ;;
libs)
  STEPS=${1:-12}
  for rep in 1 2; do for lib in nemo_amd/lib_ab/libmi355x_asr_*.so; do
    name=$(basename "$lib" .so); name=${name#libmi355x_asr_}
    [ $rep = 1 ] && MI355X_ASR_LIB=$PWD/$lib timeout 200 python -m pytest tests -m gpu -x -q -k "gemm or wgrad or conv2 or model_matches or bf16" 2>&1 | tail -1 | sed "s/^/[$name] parity: /"
    echo "[$name] rep $rep $(MI355X_ASR_LIB=$PWD/$lib timeout 150 python bench.py --steps $STEPS --warmup 4 --no-cpu-baseline --no-roofline 2>/dev/null | ms)" | tee -a $O/libs.txt
  done; done ;;
trees)
  for rep in 1 2; do for t in "$@"; do
    for args in "" "--model squeezeformer --size medium" "--model transducer"; do
      echo "$t [$args] $(cd $t && timeout 300 python bench.py $args --steps 12 --warmup 6 --no-cpu-baseline --no-roofline 2>/dev/null | ms)" | tee -a $O/trees.txt
    done
  done; done ;;
*) sed -n 2,12p "$0" ;;
esac
