cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_kernels_gpu.py -q -x -k "persistent" 2>&1 | tail -3
V5_AB=1 REPS=3 ITERS=30 timeout 300 python tools/gemm_bench.py 2>&1 | grep TFLOP > gpurun_out/v5_ab.txt
cat gpurun_out/v5_ab.txt
