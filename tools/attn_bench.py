"""per-kernel timing of the fused rel-pos attention kernels at the Conformer-CTC-Large shape (B=32, H=8, T=501, dk=64)"""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nemo_amd import ops
dev = "cuda"
B, H, T, dk = 32, 8, 501, 64
d = H * dk; Tp = (T + 7) // 8 * 8; scale = 1 / math.sqrt(dk)
g = torch.Generator(device=dev).manual_seed(0)
bf = torch.bfloat16
qkv = (torch.randn(B * T, 3 * d, device=dev, generator=g) * 0.5).to(bf)
pos = (torch.randn(2 * T - 1, d, device=dev, generator=g) * 0.5).to(bf)
u = torch.randn(d, device=dev, generator=g) * 0.1; v = torch.randn(d, device=dev, generator=g) * 0.1
lens = torch.full((B,), 500, device=dev, dtype=torch.int64)
dO = torch.randn(B * T, d, device=dev, generator=g).to(bf)
ctx = torch.empty(B * T, d, device=dev, dtype=bf); lse = torch.zeros(B, H, T, device=dev)
qu = torch.empty(B * T, d, device=dev, dtype=bf); qv = torch.empty_like(qu)
dlt = torch.zeros(B, H, T, device=dev)
dqu = torch.empty_like(qu); dqv = torch.empty_like(qu)
dqkv = torch.empty(B * T, 3 * d, device=dev, dtype=bf)
dp = torch.zeros(2 * T - 1, d, device=dev)
drop = ops.Dropout(0.1, 1, 2)
def t(name, fn, flops):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): fn()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 10 * 1e3
    print(f"{name:12s} {us:9.1f} us   {flops / us / 1e6:7.1f} TFLOP/s (useful)", flush=True)
pair = 2.0 * B * H * T * T * dk  # one T x T x dk product
t("fwd", lambda: ops.relpos_flash_fwd(qkv, 3 * d, pos, d, u, v, lens, ctx, d, lse, B, H, T, dk, Tp, scale, drop), 3 * pair)
ops.qbias(qkv, 3 * d, u, v, qu, qv, B * T, d); ops.attn_delta(dO, ctx, dlt, B, H, T, d)
dS = ops.relpos_ds_buffer(B, H, T, dev, fill=0.0)
t("bwd_dq", lambda: ops.relpos_flash_bwd_dq(qu, qv, qkv, 3 * d, pos, d, lens, dO, lse, dlt, dqu, dqv, B, H, T, dk, scale, drop, ds_out=dS), 5 * pair)
t("bwd_dkv", lambda: ops.relpos_flash_bwd_dkv(qu, qv, qkv, 3 * d, pos, d, lens, dO, lse, dlt, dqkv, 3 * d, B, H, T, dk, Tp, scale, drop), 5 * pair)
t("bwd_dpos", lambda: ops.relpos_flash_bwd_dpos(qv, dS, lens, dp, B, H, T, dk), 1 * pair)
