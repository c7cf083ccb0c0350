"""Fused feed-forward block (csrc/ffn.hip: mi355x_ffn_fwd / mi355x_ffn_bwd_dgrad).

CPU: the packed weight images (PackPlan.add_ffn_k512 / add_ffn_kchunk) are the order documented in include/mi355x_asr.h -- checked
by evaluating the pack entries' index maps in numpy.  GPU (-m gpu): the kernels against (a) an fp32 torch restatement of
ConformerFeedForward.forward + the macaron residual (conformer_modules.py:366-387, :174-181) on the bf16-rounded operands, with
the hidden pre-activation rounded to bf16 as the reference's autocast does, and (b) the unfused GEMM-epilogue path, which draws
the SAME dropout masks -- including a ragged last workgroup and the benchmark's row count."""
import numpy as np
import pytest
import torch

dev = "cuda"


def _emulate_pack(plan):
    """numpy evaluation of every pending block of a PackPlan (dst[r*pitch + c] = src[r1*sr1 + r2*sr2 + c1*sc1 + c2*sc2])"""
    out = {}
    for name, src, rows, cols, ro, co, nr2, nc2, sr1, sr2, sc1, sc2 in plan._pending:
        off, irows, pitch = plan._images[name]
        img = out.setdefault(name, np.zeros((irows, pitch), np.float32))
        r = np.arange(rows)[:, None]
        c = np.arange(cols)[None, :]
        idx = (r // nr2) * sr1 + (r % nr2) * sr2 + (c // nc2) * sc1 + (c % nc2) * sc2
        img[ro:ro + rows, co:co + cols] = src.reshape(-1).numpy()[idx]
    return out


def test_ffn_pack_maps_are_the_documented_orders():
    from nemo_amd.packing import PackPlan
    dff = 192
    g = torch.Generator().manual_seed(0)
    W1 = torch.randn(dff, 512, generator=g)
    W2 = torch.randn(512, dff, generator=g)
    p = PackPlan(torch.bfloat16, "cpu")
    p.add_ffn_k512("w1p", W1)
    p.add_ffn_kchunk("w2p", W2)
    p.add_ffn_k512("w2tp", W2, transpose=True)     # logical A = W2^T [dff, 512]
    p.add_ffn_kchunk("w1tp", W1, transpose=True)   # logical B = W1^T [512, dff]
    img = {k: v.reshape(-1) for k, v in _emulate_pack(p).items()}
    A = {"w1p": W1.numpy(), "w2tp": W2.t().numpy()}
    for name, a in A.items():
        c, k16, r, e = np.meshgrid(np.arange(dff // 64), np.arange(32), np.arange(64), np.arange(16), indexing="ij")
        want = a[c * 64 + r, k16 * 16 + e].reshape(-1)
        assert np.array_equal(img[name], want), name
    Bm = {"w2p": W2.numpy(), "w1tp": W1.t().numpy()}
    for name, b in Bm.items():
        t, o, e = np.meshgrid(np.arange(dff // 16), np.arange(512), np.arange(16), indexing="ij")
        want = b[o, t * 16 + e].reshape(-1)
        assert np.array_equal(img[name], want), name


# ------------------------------------------------------------------------------------------------------------------ GPU
def _ops():
    from nemo_amd import ops
    return ops


def _rel(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return ((a - b).abs().max() / (b.abs().max() + 1e-12)).item()


def _bfr(x):
    return x.to(torch.bfloat16).float()


def _setup(M, dff, seed):
    from nemo_amd.packing import PackPlan
    g = torch.Generator().manual_seed(seed)
    d = 512
    W1 = (torch.randn(dff, d, generator=g) * d ** -0.5).to(dev)
    W2 = (torch.randn(d, dff, generator=g) * dff ** -0.5).to(dev)   # asymmetric, non-square: transpose-detecting
    b1 = (torch.randn(dff, generator=g) * 0.5).to(dev)
    b2 = (torch.randn(d, generator=g) * 0.5).to(dev)
    x = torch.randn(M, d, generator=g).to(dev)
    y = torch.randn(M, d, generator=g).to(dev).to(torch.bfloat16)
    p = PackPlan(torch.bfloat16, dev)
    p.add_ffn_k512("w1p", W1); p.add_ffn_kchunk("w2p", W2)
    p.add_ffn_k512("w2tp", W2, transpose=True); p.add_ffn_kchunk("w1tp", W1, transpose=True)
    p.add_matrix("w1", W1); p.add_matrix("w2", W2); p.add_matrix("w1t", W1, True); p.add_matrix("w2t", W2, True)
    p.finalize(); p.run()
    return d, W1, W2, b1, b2, x, y, p


@pytest.mark.gpu
@pytest.mark.parametrize("M,dff", [(64, 128), (200, 256), (16032, 2048), (33, 2048)])
def test_ffn_fwd_matches_fp32_restatement(M, dff):
    o = _ops()
    d, W1, W2, b1, b2, x, y, p = _setup(M, dff, 3)
    h = torch.full((M, dff), float("nan"), device=dev, dtype=torch.bfloat16)
    out = torch.full((M, d), float("nan"), device=dev)
    o.ffn_fwd(y, p["w1p"], b1, p["w2p"], b2, x, h, out, M, d, dff, alpha=0.5)
    torch.cuda.synchronize()
    hr = y.float() @ _bfr(W1).t() + b1
    assert _rel(h, hr) < 1e-2
    hq = h.float()                                     # (the kernel's own rounding of h: isolates the second half)
    act = _bfr(hq * torch.sigmoid(hq))
    ref = x + 0.5 * (act @ _bfr(W2).t() + b2)
    assert _rel(out, ref) < 2e-4, _rel(out, ref)
    assert torch.isfinite(out).all() and torch.isfinite(h.float()).all()


@pytest.mark.gpu
@pytest.mark.parametrize("M,dff,pdrop", [(200, 256, 0.0), (1000, 2048, 0.1), (16032, 2048, 0.1)])
def test_ffn_fwd_matches_unfused_path_with_the_same_dropout_masks(M, dff, pdrop):
    o = _ops()
    d, W1, W2, b1, b2, x, y, p = _setup(M, dff, 4)
    d_in, d_res = o.Dropout(pdrop, seed=11, site=5), o.Dropout(pdrop, seed=11, site=6)
    h = torch.empty(M, dff, device=dev, dtype=torch.bfloat16)
    out = torch.empty(M, d, device=dev)
    o.ffn_fwd(y, p["w1p"], b1, p["w2p"], b2, x, h, out, M, d, dff, alpha=0.5, drop_in=d_in, drop_res=d_res)
    h0 = torch.empty_like(h); a0 = torch.empty_like(h); out0 = torch.empty_like(out)
    o.gemm(y, p["w1"], a0, M, dff, d, d, p.pitch("w1"), dff, bias=b1, epi=o.EPI_SWISH_DROP, aux_out=h0, drop=d_in)
    o.gemm(a0, p["w2"], out0, M, d, dff, dff, p.pitch("w2"), d, bias=b2, alpha=0.5, epi=o.EPI_RESID, aux_in=x, drop=d_res)
    torch.cuda.synchronize()
    assert _rel(h, h0) < 1e-2                           # (accumulation order only; both bf16)
    assert ((h.float() - h0.float()).abs() > 0).float().mean().item() < 0.02
    assert _rel(out, out0) < 3e-3, _rel(out, out0)      # act differs by one bf16 rounding of h inside the Swish
    if pdrop > 0:  # identical masks: the dropped residual-branch elements are exactly x
        z0, z1 = (out0 == x), (out == x)
        assert (z0 == z1).float().mean().item() > 0.9999
        assert abs(z1.float().mean().item() - pdrop) < 0.01


@pytest.mark.gpu
@pytest.mark.parametrize("M,dff,pdrop", [(64, 128, 0.0), (200, 256, 0.1), (16032, 2048, 0.1), (33, 2048, 0.0)])
def test_ffn_bwd_dgrad_matches_unfused_path_and_fp32(M, dff, pdrop):
    o = _ops()
    d, W1, W2, b1, b2, x, y, p = _setup(M, dff, 5)
    g = torch.Generator().manual_seed(9)
    d_in = o.Dropout(pdrop, seed=13, site=2)
    h = (torch.randn(M, dff, generator=g) * 1.5).to(dev).to(torch.bfloat16)
    df = torch.randn(M, d, generator=g).to(dev).to(torch.bfloat16)
    dh = torch.full((M, dff), float("nan"), device=dev, dtype=torch.bfloat16)
    act = torch.full((M, dff), float("nan"), device=dev, dtype=torch.bfloat16)
    dy = torch.full((M, d), float("nan"), device=dev, dtype=torch.bfloat16)
    o.ffn_bwd_dgrad(df, p["w2tp"], p["w1tp"], h, dh, act, dy, M, d, dff, drop_in=d_in)
    # unfused: dh = (df @ W2) * mask * swish'(h) ; dy = dh @ W1 ; act = the forward epilogue's output on the same h
    dh0 = torch.empty_like(dh); dy0 = torch.empty_like(dy)
    o.gemm(df, p["w2t"], dh0, M, dff, d, d, p.pitch("w2t"), dff, epi=o.EPI_DSWISH, aux_in=h, drop=d_in)
    o.gemm(dh0, p["w1t"], dy0, M, d, dff, dff, p.pitch("w1t"), d)
    mask = torch.empty(M, dff, device=dev)
    o.drop_scale_cast(torch.ones(M, dff, device=dev), mask, M * dff, 1.0, d_in)
    torch.cuda.synchronize()
    hq = h.float()
    s = torch.sigmoid(hq)
    act_ref = hq * s * mask
    assert _rel(act, act_ref) < 1e-2
    assert ((act.float() == 0) == (act_ref == 0)).float().mean().item() > 0.9999
    gref = (df.float() @ _bfr(W2)) * mask * (s * (1 + hq * (1 - s)))
    assert _rel(dh, gref) < 2e-2 and _rel(dh, dh0) < 2e-2
    assert _rel(dy, dh.float() @ _bfr(W1)) < 1e-2
    assert _rel(dy, dy0) < 2e-2
    assert torch.isfinite(dy.float()).all()
