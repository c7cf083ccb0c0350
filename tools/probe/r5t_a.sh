#!/bin/bash
# round-5 session-3 call A: whole GPU suite + smoke on HEAD (Squeezeformer Swish fusion included), Squeezeformer-Medium with /
# without the fused conv-module activation, and where the ~80 copyBuffer dispatches per step come from (HIP API + copy trace)
cd "$(dirname "$0")/../.." || exit 1
R=$PWD
O=$R/gpurun_out/r5t_a; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests -x -q -m gpu --durations=6 > $O/gpu_suite.log 2>&1; echo "gpu tests rc=$?"; tail -12 $O/gpu_suite.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.log
for arm in 1 0 1 0; do
  MI355X_GLU_DW_FUSE_FWD=$arm MI355X_GLU_DW_FUSE=$arm timeout 300 python bench.py --model squeezeformer --size medium --no-cpu-baseline --no-roofline > $O/sq_fuse$arm.json 2>/dev/null
  python -c "
import json; d=json.loads(open('$O/sq_fuse$arm.json').read().strip().splitlines()[-1]); print('squeezeformer fuse=$arm', d['ms_per_step'], d.get('launch'))"
done
# copies: API trace + memory-copy trace of three live steps
(cd /tmp && MI355X_GRAPHS=0 timeout 300 rocprofv3 --hip-trace --kernel-trace --memory-copy-trace --output-format csv -d /tmp/cptrace -o out -- python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-roofline > $O/cptrace.json 2> $O/cptrace.err)
python - <<'EOF' > gpurun_out/r5t_a/copies.txt 2>&1
import csv, glob, collections
for pat in ("*memory_copy_trace.csv", "*hip_api_trace.csv", "*kernel_trace.csv"):
    for f in glob.glob("/tmp/cptrace/**/" + pat, recursive=True):
        rows = list(csv.DictReader(open(f)))
        print("==", pat, len(rows), list(rows[0].keys()) if rows else None)
        if "memory_copy" in pat:
            c = collections.Counter((r.get("Direction"), r.get("Bytes") or r.get("Size")) for r in rows)
            for k, v in c.most_common(40): print(v, k)
        elif "hip_api" in pat:
            c = collections.Counter(r.get("Function") for r in rows)
            for k, v in c.most_common(40): print(v, k)
        else:
            c = collections.Counter(r.get("Kernel_Name", "")[:60] for r in rows if "copyBuffer" in r.get("Kernel_Name", "") or "fillBuffer" in r.get("Kernel_Name", ""))
            for k, v in c.most_common(10): print(v, k)
            # what runs right before / after a copyBuffer on the same queue
            rows.sort(key=lambda r: int(r["Start_Timestamp"]))
            ctx = collections.Counter()
            for i, r in enumerate(rows):
                if "copyBuffer" in r["Kernel_Name"] and 0 < i < len(rows) - 1:
                    ctx[(rows[i - 1]["Kernel_Name"][:50], rows[i + 1]["Kernel_Name"][:50], r.get("Queue_Id"))] += 1
            for k, v in ctx.most_common(25): print(v, k)
EOF
head -120 $O/copies.txt
