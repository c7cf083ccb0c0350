"""BatchNorm + Swish kernels of the conv module alone at the Large shape (M = 16032, d = 512, bf16), cold operands (ROTATE sets)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nemo_amd import ops
dev = "cuda"; M, d = 16032, 512; ROT = 6
g = torch.Generator(device=dev).manual_seed(0)
xs = [torch.randn(M, d, device=dev, generator=g).to(torch.bfloat16) for _ in range(ROT)]
dys = [torch.randn(M, d, device=dev, generator=g).to(torch.bfloat16) for _ in range(ROT)]
ys = [torch.empty_like(xs[0]) for _ in range(ROT)]
gamma = torch.randn(d, device=dev); beta = torch.randn(d, device=dev)
stats = torch.zeros(2 * d + 8, device=dev, dtype=torch.float64)
stats[:d] = torch.randn(d, device=dev, dtype=torch.float64) * M * 0.1; stats[d:2 * d] = (torch.rand(d, device=dev, dtype=torch.float64) + 1) * M
mean = torch.empty(d, device=dev); rstd = torch.empty(d, device=dev); rm = torch.zeros(d, device=dev); rv = torch.ones(d, device=dev)
sums = torch.zeros(2, d, device=dev, dtype=torch.float64); dg = torch.zeros(d, device=dev); db = torch.zeros(d, device=dev)
cnt = [0]
def nxt():
    cnt[0] += 1; return cnt[0] % ROT
def t(name, fn):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30): fn()
    e1.record(); torch.cuda.synchronize(); print(f"{name:34s} {e0.elapsed_time(e1) / 30 * 1e3:8.1f} us", flush=True)
def two():
    i = nxt(); ops.bn_finalize(stats, float(M), mean, rstd, rm, rv, 0.1, 1e-5, d); ops.bn_swish_fwd(xs[i], mean, rstd, gamma, beta, ys[i], M, d)
def one():
    i = nxt(); ops.bn_stats_swish_fwd(xs[i], stats, float(M), gamma, beta, ys[i], mean, rstd, rm, rv, 0.1, 1e-5, M, d)
def swish_only():
    i = nxt(); ops.bn_swish_fwd(xs[i], mean, rstd, gamma, beta, ys[i], M, d)
def red():
    i = nxt(); ops.bn_swish_bwd_reduce(dys[i], xs[i], mean, rstd, gamma, beta, sums, M, d, dgamma=dg, dbeta=db)
def app():
    i = nxt(); ops.bn_swish_bwd_apply(dys[i], xs[i], mean, rstd, gamma, beta, sums, float(M), True, ys[i], M, d)
ops.bn_finalize(stats, float(M), mean, rstd, rm, rv, 0.1, 1e-5, d)
t("bn_swish_fwd alone", swish_only)
t("bn_finalize + bn_swish_fwd", two)
t("bn_stats_swish_fwd (one launch)", one)
t("bn_swish_bwd_reduce (+ 2nd stage)", red)
t("bn_swish_bwd_apply", app)
