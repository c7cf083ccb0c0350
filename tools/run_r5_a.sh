#!/bin/bash
# round 5, GPU call A: box baseline, vendor GEMM anatomy (kernel names + counters), -fno-slp-vectorize A/B
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
O=gpurun_out/r5a; mkdir -p $O
R=$PWD
timeout 150 python bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 > $O/bench_base.json
python -c "import json;d=json.load(open('$O/bench_base.json'));print('baseline ms_per_step',d['ms_per_step'])"
# vendor anatomy: names + times
(cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats -d $R/$O/va_trace -o out -- python $R/tools/vendor_anatomy.py > $R/$O/vendor_anatomy.txt 2>$R/$O/va_err.txt)
cat $O/vendor_anatomy.txt
f=$(find $O/va_trace -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -30 "$f" > $O/vendor_kernel_stats.csv && cat $O/vendor_kernel_stats.csv | cut -c1-400
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT"
P2="GRBM_GUI_ACTIVE SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU"
i=0
for P in "$P1" "$P2"; do
  i=$((i+1))
  (cd /tmp && ITERS=4 ROTATE=2 timeout 150 rocprofv3 --kernel-trace --pmc $P -d $R/$O/va_pmc$i -o out -- python $R/tools/vendor_anatomy.py > /dev/null 2>&1)
  db=$(find $O/va_pmc$i -name "*.db" | head -1)
  echo "## pass $i" >> $O/vendor_pmc.md
  [ -n "$db" ] && python tools/pmc_summary.py $db | grep -i "cijk\|gemm_bf16\|kernel |" | cut -c1-300 >> $O/vendor_pmc.md
  rm -rf $O/va_pmc$i
done
cat $O/vendor_pmc.md
find $O/va_trace -name "*.db" -delete; find $O/va_trace -name "*trace.csv" -delete
# SLP-vectoriser A/B
for rep in 1 2; do
  for lib in nemo_amd/lib_ab/libmi355x_asr_*.so; do
    name=$(basename "$lib" .so); name=${name#libmi355x_asr_}
    export MI355X_ASR_LIB=$PWD/$lib
    if [ $rep = 1 ]; then timeout 60 python tools/attn_bench.py 2>&1 | sed "s/^/[$name] /" | tee -a $O/ab_attn.txt; fi
    timeout 120 python bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 \
      | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('[$name] rep $rep ms_per_step', d['ms_per_step'])" | tee -a $O/ab_steps.txt
  done
done
unset MI355X_ASR_LIB
