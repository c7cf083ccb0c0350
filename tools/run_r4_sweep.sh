#!/bin/bash
# round 4: in-step sweep of every run-time knob (ON the GPU box): each arm is the default configuration with ONE knob moved, the
# default arm repeated between groups (one box, live launches); lesson of r4_dwconv_in_step.md -- the step, not a warm
# micro-benchmark, decides defaults.  usage: tools/run_r4_sweep.sh "<K=V ...>" ...   (no arguments: the standard list)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
O=gpurun_out/r4sweep; mkdir -p $O
: > $O/sweep.txt
run() {  # label, env assignments...
  lab=$1; shift
  env MI355X_GRAPHS=0 "$@" timeout 120 python bench.py --steps 20 --warmup 6 --no-cpu-baseline --no-roofline > $O/run.json 2> $O/run.err
  ms=$(python -c "import json;print(json.loads(open('$O/run.json').read().strip().splitlines()[-1])['ms_per_step'])" 2>/dev/null || echo ERR)
  echo "$lab $ms" | tee -a $O/sweep.txt
}
if [ $# -gt 0 ]; then
  ARMS=("$@")
else
  ARMS=("MI355X_GEMM_V4=0" "MI355X_GEMM_V4=2" "MI355X_GEMM_V5=0" "MI355X_GEMM_V6=1" "MI355X_GEMM_V7=0" "MI355X_LN2=0" "MI355X_LN2_BWD=0"
        "MI355X_LN_CAST_FUSE=0" "MI355X_SWISH_G=0" "MI355X_WGRAD_GROUPED=0" "MI355X_WGRAD_STREAM=0" "MI355X_DPOS_STREAM=0"
        "MI355X_SUB_WGRAD_STREAM=0" "MI355X_WGRAD_PRIO=1" "MI355X_WGRAD_PRIO=-1" "MI355X_OPT_IN_BACKWARD=0" "MI355X_ARENA=0"
        "MI355X_GEMM_FEW_TILES=0" "MI355X_GEMM_FEW_TILES=300" "MI355X_CONV2_IMPLICIT=0" "MI355X_FLASH_DELTA_LO=0")
fi
i=0
run default A=1
for arm in "${ARMS[@]}"; do
  run "$arm" $arm
  i=$((i+1))
  if [ $((i % 5)) -eq 0 ]; then run default A=1; fi
done
run default A=1
