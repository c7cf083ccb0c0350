"""time mi355x_ctc_loss (lattice + gradient launches) at the headline shape with the wave-resident and the LDS / barrier lattice kernel"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from nemo_amd import ops
from nemo_amd._lib import lib
dev = "cuda"
B, T, C, U = 32, 501, 129, 60
g = torch.Generator().manual_seed(0)
logp = torch.log_softmax(torch.randn(B, T, C, generator=g), -1).to(dev)
tgt = torch.randint(0, C - 1, (B, U), generator=g).to(dev)
il = torch.full((B,), T, dtype=torch.int64, device=dev); tl = torch.full((B,), U, dtype=torch.int64, device=dev)
grad = torch.empty_like(logp)
out = {}
for mode in (0, 1, 0, 1):
    lib.mi355x_ctc_config(mode)
    for _ in range(3):
        nll = ops.ctc_loss(logp, tgt, il, tl, C - 1, grad=grad)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        nll = ops.ctc_loss(logp, tgt, il, tl, C - 1, grad=grad)
    e1.record(); torch.cuda.synchronize()
    out[mode] = (nll.clone(), grad.clone())
    print(f"ctc lattice mode {mode}: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us per loss+grad call, nll[0] {nll[0].item():.4f}")
d = (out[0][0] - out[1][0]).abs().max().item(); dg = (out[0][1] - out[1][1]).abs().max().item()
print(f"max |nll diff| {d:.3e} (nll ~ {out[0][0].abs().mean().item():.1f}), max |grad diff| {dg:.3e}")
