cd $GRAFT_REPO_ROOT
( time python bench.py ) > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
tail -3 gpurun_out/bench_default.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/bench_default.json").read().strip().split("\n")[-1])
print("ms", d["ms_per_step"], "roof", d["roofline"]["frac"], d["roofline"]["traffic_note"])
print("hbm", d["roofline_hbm"]["achieved"], {k:(v["GBps"],v["us"]) for k,v in d["roofline_hbm"]["per_kernel"].items()})
print("cpu", d["cpu_baseline"])
print("dist", d["distributed"])
PY
BENCH_DEVICE=0 BENCH_DIST_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 3 --warmup 1 --batch 8 --no-cpu-baseline > gpurun_out/bench_gloo2.json 2> gpurun_out/bench_gloo2.err
tail -2 gpurun_out/bench_gloo2.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/bench_gloo2.json").read().strip().split("\n")[-1])
print("gloo2 ms", d["ms_per_step"], d["value"], d["distributed"])
PY
