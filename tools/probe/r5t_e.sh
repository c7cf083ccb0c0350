#!/bin/bash
# round-5 session-3 call E: native backtrace of the SIGSEGV of `bench.py --size small` under MI355X_GRAPHS=auto
cd "$(dirname "$0")/../.." || exit 1
O=$PWD/gpurun_out/r5t_e; mkdir -p $O
export TMPDIR=/tmp
timeout 400 /opt/rocm/bin/rocgdb -batch -ex "set pagination off" -ex "handle SIGUSR1 nostop noprint" -ex run -ex "bt 40" -ex "info threads" -ex "py-bt" --args python bench.py --size small --steps 4 --warmup 2 --no-cpu-baseline --no-roofline > $O/gdb.txt 2>&1
echo "gdb rc=$?"; grep -n "SIGSEGV\|^#" $O/gdb.txt | head -60
for v in "MI355X_GRAPHS=1" "MI355X_GRAPHS=1 MI355X_TAPE=0" "MI355X_GRAPHS=auto MI355X_TAPE=0" "MI355X_GRAPHS=auto MI355X_ARENA=0"; do
  env $v timeout 200 python -X faulthandler bench.py --size small --steps 4 --warmup 2 --no-cpu-baseline --no-roofline > $O/v.json 2> $O/v.err; echo "$v rc=$? $(tail -c 200 $O/v.json | cut -c1-100)"; grep -n "Fatal\|File " $O/v.err | head -8
done
