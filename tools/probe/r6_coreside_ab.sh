#!/bin/bash
# same-box A/B of two library builds (old / new in nemo_amd/lib_ab/), interleaved three times, plus a kernel trace of the new one
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r6co; mkdir -p $O
ms() { python -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])" 2>/dev/null || echo ERR; }
for rep in 1 2 3; do for v in old new; do
  echo "[$v] rep $rep $(MI355X_GRAPHS=0 MI355X_ASR_LIB=$PWD/nemo_amd/lib_ab/libmi355x_asr_$v.so timeout 150 python bench.py --steps 16 --warmup 4 --no-cpu-baseline --no-roofline 2>/dev/null | ms)" | tee -a $O/ab_${1:-x}.txt
done; done
export TMPDIR=/tmp
(cd /tmp && MI355X_GRAPHS=0 MI355X_ASR_LIB=/root/repo/nemo_amd/lib_ab/libmi355x_asr_new.so timeout 300 rocprofv3 --kernel-trace --stats -d $PWD/rp_co -o out -- python /root/repo/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-roofline > /dev/null 2>&1; db=$(find /tmp/rp_co -name "*.db" | head -1); cd /root/repo; [ -n "$db" ] && python tools/rocpd_stats.py $db $O/trace_${1:-x} > /dev/null; rm -rf /tmp/rp_co)
head -30 $O/trace_${1:-x}.md | cut -c1-150
