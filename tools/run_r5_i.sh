#!/bin/bash
# why does a launch tape lose to live launches on the device?  kernel traces of both, per-stream busy time and gaps
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/r5i; mkdir -p $O
timeout 300 python -m pytest tests/test_rnnt_decoding.py -x -q 2>&1 | tail -2 | tee $O/tests_rnnt.txt
for mode in "0 1 eager" "1 1 tape" "0 1 eager2" "1 1 tape2"; do
  set -- $mode
  MI355X_GRAPHS=$1 MI355X_TAPE=$2 timeout 200 python bench.py --steps 16 --warmup 10 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$3', d['ms_per_step'], d['launch'].get('host_issue_ms_per_step'), str(d['launch'].get('mode'))[:60])" | tee -a $O/ab.txt
done
for mode in "0 1 eager" "1 1 tape"; do
  set -- $mode
  (cd /tmp && MI355X_GRAPHS=$1 MI355X_TAPE=$2 timeout 300 rocprofv3 --kernel-trace -d $O/tr_$3 -o out -- python $R/bench.py --steps 6 --warmup 10 --no-cpu-baseline --no-roofline > $O/tr_$3.json 2> $O/tr_$3.err)
  db=$(find $O/tr_$3 -name "*.db" | head -1)
  echo "== $3" | tee -a $O/gaps.txt
  python tools/stream_gaps.py $db 200 2>&1 | cut -c1-400 | tee -a $O/gaps.txt
  python - "$db" <<'PY' | tee -a $O/gaps.txt
import sqlite3, sys, re
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
cols = [r[1] for r in cur.execute(f"pragma table_info({kd})")]
qcol = "stream_id" if "stream_id" in cols else "queue_id"
scol = [r[1] for r in cur.execute(f"pragma table_info({ks})")]
ncol = "kernel_name" if "kernel_name" in scol else "display_name"
rows = list(cur.execute(f"select d.{qcol}, d.start, d.end, s.{ncol} from {kd} d join {ks} s on d.kernel_id = s.id order by d.start"))
t_end = max(r[2] for r in rows); rows = [r for r in rows if r[1] >= t_end - 200e6]
# per-kernel-name average duration (top 14 by total), for the eager / tape comparison
agg = {}
for q, s, e, n in rows:
    n = re.sub(r"\(.*", "", n)[:60]
    a = agg.setdefault(n, [0, 0]); a[0] += 1; a[1] += e - s
for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:14]:
    print(f"   {n:60s} n={c:5d} avg {t/c/1e3:8.1f} us total {t/1e6:7.2f} ms")
PY
  rm -rf $O/tr_$3
done
