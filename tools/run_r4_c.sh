#!/bin/bash
# round 4, session 4 (ON the GPU box): statistics-mailbox tests (two processes on the one GPU), in-step A/B of the depthwise-conv
# forward kernel choice, then the three PMC passes of the headline step on the final GEMM sources (-> profiles/r4_gemm_traffic.json).
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
O=gpurun_out/r4c; mkdir -p $O
timeout 300 python -m pytest tests/test_mailbox_gpu.py "tests/test_model_gpu.py::test_data_parallel_two_ranks_equal_one_process_on_the_joint_batch" -x -q -rs > $O/mailbox_tests.log 2>&1; echo "mailbox tests rc=$?"; tail -8 $O/mailbox_tests.log
for lvl in 0 1 0 1; do
  MI355X_GRAPHS=0 MI355X_DWCONV_STREAM=$lvl timeout 200 python bench.py --steps 20 --warmup 8 --no-cpu-baseline --no-roofline > $O/dw_$lvl.json 2>$O/dw_$lvl.err
  echo "DWCONV_STREAM=$lvl: $(python -c "import json;print(json.loads(open('$O/dw_$lvl.json').read().strip().splitlines()[-1])['ms_per_step'])")"
done
if [ "${SKIP_PMC:-0}" != "1" ]; then
  R3_PARTS=pmc bash tools/run_profiles_r3.sh > $O/pmc.log 2>&1; echo "pmc rc=$?"; tail -3 $O/pmc.log
  mkdir -p $O/prof; cp gpurun_out/r3prof/*.pmc.json gpurun_out/r3prof/source_hash.txt $O/prof/ 2>/dev/null
  python tools/pmc_assemble.py $O/prof $O r4 2>&1 | tail -2
fi
