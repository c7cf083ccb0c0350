"""where the cycles of a step of the fused feed-forward kernel go: needs csrc/ffn.hip built with -DFFN_TRACE (tools/ffn_trace.sh).
Prints, for three workgroups (first / middle / last) and one wave of each kind, the average shader cycles per step and segment."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nemo_amd import ops
from nemo_amd._lib import lib
from nemo_amd.packing import PackPlan

dev = "cuda"
bf = torch.bfloat16
M, d, dff = int(os.environ.get("M", "16032")), 512, 2048
g = torch.Generator(device=dev).manual_seed(0)
W1 = torch.randn(dff, d, device=dev, generator=g) * d ** -0.5
W2 = torch.randn(d, dff, device=dev, generator=g) * dff ** -0.5
b1 = torch.randn(dff, device=dev, generator=g); b2 = torch.randn(d, device=dev, generator=g)
p = PackPlan(bf, dev)
p.add_ffn("f", W1, W2)
p.finalize(); p.run()
x = torch.randn(M, d, device=dev, generator=g); y = torch.randn(M, d, device=dev, generator=g).to(bf)
h = torch.empty(M, dff, device=dev, dtype=bf); out = torch.empty(M, d, device=dev)
a = torch.empty(M, dff, device=dev, dtype=bf); dh = torch.empty(M, dff, device=dev, dtype=bf); dy = torch.empty(M, d, device=dev, dtype=bf)
df = torch.randn(M, d, device=dev, generator=g).to(bf)
d_in, d_res = ops.Dropout(0.1, 1, 1), ops.Dropout(0.1, 1, 2)
trace = torch.zeros(48, dtype=torch.int32, device=dev)
fn = lib.mi355x_ffn_debug_trace
fn.argtypes = [C.c_void_p]
fn(trace.data_ptr())
nsteps = 4 * (dff // 64 + 1) + 3
names = {0: ["(stamp)", "mfma+rd+dma", "handoff/xform", "(stamp)", "(stamp)", "vmcnt-wait", "lgkm+barrier"],
         1: ["(stamp)", "mfma+rd+dma", "transform", "(stamp)", "(stamp)", "vmcnt-wait", "lgkm+barrier"]}
for which in ("fwd", "bwd"):
    for _ in range(3):
        trace.zero_()
        if which == "fwd":
            ops.ffn_fwd(y, p["f.w1p"], b1, p["f.w2p"], b2, x, h, out, M, d, dff, 0.5, d_in, d_res)
        else:
            ops.ffn_bwd_dgrad(df, p["f.w2tp"], p["f.w1tp"], h, dh, a, dy, M, d, dff, d_in)
        torch.cuda.synchronize()
    t = trace.cpu().view(3, 2, 8).long() & 0xFFFFFFFF
    print(f"== {which}: average shader cycles per step ({nsteps} steps), workgroups first / middle / last")
    for role in (0, 1):
        print("  phase-%d wave:" % (role + 1))
        for k in range(7):
            print(f"    {names[role][k]:16s} " + "  ".join(f"{t[b, role, k].item() / nsteps:8.1f}" for b in range(3)))
        print(f"    {'total':16s} " + "  ".join(f"{t[b, role, :7].sum().item() / nsteps:8.1f}" for b in range(3)))
