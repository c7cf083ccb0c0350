"""micro-benchmark of the fused feed-forward kernels (csrc/ffn.hip) against the GEMM pairs they replace, Conformer-CTC-Large
shape (M = 16032, d = 512, d_ff = 2048), HIP-event timing.  ROTATE > 1 cycles over independent operand / output sets so that a
launch finds its activations where the training step finds them (HBM, not an L2 the previous launch warmed); the weights of one
"layer" are shared by the rotated sets on purpose (in the step they are re-read by 251 workgroups of the same launch anyway).
Variants are interleaved in one process (cdna_hip_programming.md rule 24)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nemo_amd import ops
from nemo_amd.packing import PackPlan

dev = "cuda"
bf = torch.bfloat16
iters = int(os.environ.get("ITERS", "30"))
ROT = int(os.environ.get("ROTATE", "6"))
rounds = int(os.environ.get("ROUNDS", "3"))
M = int(os.environ.get("M", "16032"))
d, dff = 512, int(os.environ.get("DFF", "2048"))
pdrop = float(os.environ.get("PDROP", "0.1"))


def timeit(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3  # us


g = torch.Generator(device=dev).manual_seed(0)
b1 = torch.randn(dff, device=dev, generator=g)
b2 = torch.randn(d, device=dev, generator=g)
COLDW = os.environ.get("COLD_WEIGHTS", "1") != "0"   # every rotated set has its own weights, like the 36 blocks of a step
plans = []
for r in range(ROT if COLDW else 1):
    W1 = torch.randn(dff, d, device=dev, generator=g) * d ** -0.5
    W2 = torch.randn(d, dff, device=dev, generator=g) * dff ** -0.5
    p = PackPlan(bf, dev)
    p.add_ffn("f", W1, W2)
    p.add_matrix("w1", W1); p.add_matrix("w2", W2); p.add_matrix("w1t", W1, True); p.add_matrix("w2t", W2, True)
    p.finalize(); p.run()
    plans.append(p)
d_in, d_res = ops.Dropout(pdrop, 1, 1), ops.Dropout(pdrop, 1, 2)

sets = []
for r in range(ROT):
    s = dict(x=torch.randn(M, d, device=dev, generator=g), y=torch.randn(M, d, device=dev, generator=g).to(bf),
             h=torch.empty(M, dff, device=dev, dtype=bf), a=torch.empty(M, dff, device=dev, dtype=bf),
             out=torch.empty(M, d, device=dev), df=torch.randn(M, d, device=dev, generator=g).to(bf),
             dh=torch.empty(M, dff, device=dev, dtype=bf), dy=torch.empty(M, d, device=dev, dtype=bf))
    s["h"].copy_(torch.randn(M, dff, device=dev, generator=g))
    s["p"] = plans[r % len(plans)]
    sets.append(s)
cnt = [0]


def nxt():
    cnt[0] += 1
    return sets[cnt[0] % ROT]


def fwd_fused():
    s = nxt()
    p = s["p"]
    ops.ffn_fwd(s["y"], p["f.w1p"], b1, p["f.w2p"], b2, s["x"], s["h"], s["out"], M, d, dff, 0.5, d_in, d_res)


def fwd_pair():
    s = nxt()
    p = s["p"]
    ops.gemm(s["y"], p["w1"], s["a"], M, dff, d, d, p.pitch("w1"), dff, bias=b1, epi=ops.EPI_SWISH_DROP, aux_out=s["h"], drop=d_in)
    ops.gemm(s["a"], p["w2"], s["out"], M, d, dff, dff, p.pitch("w2"), d, bias=b2, alpha=0.5, epi=ops.EPI_RESID, aux_in=s["x"],
             drop=d_res)


def bwd_fused():
    s = nxt()
    p = s["p"]
    ops.ffn_bwd_dgrad(s["df"], p["f.w2tp"], p["f.w1tp"], s["h"], s["dh"], s["a"], s["dy"], M, d, dff, d_in)


def bwd_pair():
    s = nxt()
    p = s["p"]
    ops.gemm(s["df"], p["w2t"], s["dh"], M, dff, d, d, p.pitch("w2t"), dff, epi=ops.EPI_DSWISH, aux_in=s["h"], drop=d_in)
    ops.gemm(s["dh"], p["w1t"], s["dy"], M, d, dff, dff, p.pitch("w1t"), d)


flops = 2.0 * 2.0 * M * d * dff
res = {}
for rd in range(rounds):
    for name, fn in (("fwd_pair", fwd_pair), ("fwd_fused", fwd_fused), ("bwd_pair", bwd_pair), ("bwd_fused", bwd_fused)):
        if os.environ.get("ONLY") and os.environ["ONLY"] not in name:
            continue
        res.setdefault(name, []).append(timeit(fn))
for name, ts in res.items():
    best, med = min(ts), sorted(ts)[len(ts) // 2]
    print(f"{name:10s} M={M} dff={dff} p={pdrop} rotate={ROT}: median {med:8.1f} us  min {best:8.1f} us  "
          f"{flops / med / 1e6:7.1f} TFLOP/s (median)  all={['%.1f' % t for t in ts]}", flush=True)
