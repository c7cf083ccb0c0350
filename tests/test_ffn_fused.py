"""Fused feed-forward block (csrc/ffn.hip: mi355x_ffn_fwd / mi355x_ffn_bwd_dgrad).

CPU: PackPlan.add_ffn book-keeping.  GPU (-m gpu): mi355x_ffn_pack against a numpy statement of the fragment orders documented in
include/mi355x_asr.h (bit-exact), and the kernels against (a) an fp32 torch restatement of
ConformerFeedForward.forward + the macaron residual (conformer_modules.py:366-387, :174-181) on the bf16-rounded operands, with
the hidden pre-activation rounded to bf16 as the reference's autocast does, and (b) the unfused GEMM-epilogue path, which draws
the SAME dropout masks -- including a ragged last workgroup and the benchmark's row count."""
import numpy as np
import pytest
import torch

dev = "cuda"


def k512_image(A):
    """include/mi355x_asr.h: fragment (c, k16, mt) of a logical A [dff, 512] at ((c*32 + k16)*2 + mt) KiB, stored [hh][lr][8]"""
    dff = A.shape[0]
    c, k16, mt, hh, lr, e = np.meshgrid(np.arange(dff // 64), np.arange(32), np.arange(2), np.arange(2), np.arange(32), np.arange(8),
                                        indexing="ij")
    return A[c * 64 + mt * 32 + lr, k16 * 16 + hh * 8 + e].reshape(-1)


def kchunk_image(B):
    """fragment (t, q, mt4) of a logical B [512, dff] at ((t*4 + q)*4 + mt4) KiB, stored [hh][lr][8]"""
    dff = B.shape[1]
    t, q, mt, hh, lr, e = np.meshgrid(np.arange(dff // 16), np.arange(4), np.arange(4), np.arange(2), np.arange(32), np.arange(8),
                                      indexing="ij")
    return B[q * 128 + mt * 32 + lr, t * 16 + hh * 8 + e].reshape(-1)


def test_ffn_plan_declares_four_aligned_images_per_block():
    from nemo_amd.packing import PackPlan
    dff = 192
    p = PackPlan(torch.bfloat16, "cpu")
    p.add_matrix("other", torch.zeros(10, 24))
    p.add_ffn("L0.ff1", torch.zeros(dff, 512), torch.zeros(512, dff))
    offs = [p._images[f"L0.ff1.{nm}"] for nm in ("w1p", "w1tp", "w2p", "w2tp")]
    assert all(o[0] % 8 == 0 and o[1] * o[2] == 512 * dff for o in offs)                 # 16-byte aligned, 512 * dff elements each
    spans = sorted((o[0], o[0] + o[1] * o[2]) for o in offs)
    assert all(a[1] <= b[0] for a, b in zip(spans, spans[1:]))                           # disjoint
    with pytest.raises(AssertionError):
        p.add_ffn("bad", torch.zeros(dff, 256), torch.zeros(256, dff))                   # d_model must be 512


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
    p.add_ffn("f", W1, W2)
    p.add_matrix("w1", W1); p.add_matrix("w2", W2); p.add_matrix("w1t", W1, True); p.add_matrix("w2t", W2, True)
    p.finalize(); p.run()
    return d, W1, W2, b1, b2, x, y, p


@pytest.mark.gpu
@pytest.mark.parametrize("dff", [128, 2048])
def test_ffn_pack_writes_the_documented_fragment_orders(dff):
    """bit-exact: the images are a permutation of the bf16-rounded weights"""
    d, W1, W2, b1, b2, x, y, p = _setup(8, dff, 1)
    torch.cuda.synchronize()
    w1q, w2q = W1.to(torch.bfloat16).float().cpu().numpy(), W2.to(torch.bfloat16).float().cpu().numpy()
    got = {nm: p[f"f.{nm}"].float().cpu().numpy().reshape(-1) for nm in ("w1p", "w1tp", "w2p", "w2tp")}
    assert np.array_equal(got["w1p"], k512_image(w1q))
    assert np.array_equal(got["w2tp"], k512_image(w2q.T.copy()))
    assert np.array_equal(got["w2p"], kchunk_image(w2q))
    assert np.array_equal(got["w1tp"], kchunk_image(w1q.T.copy()))


@pytest.mark.gpu
@pytest.mark.parametrize("M,dff", [(64, 128), (200, 256), (16032, 2048), (33, 2048)])
def test_ffn_fwd_matches_fp32_restatement(M, dff):
    o = _ops()
    d, W1, W2, b1, b2, x, y, p = _setup(M, dff, 3)
    h = torch.full((M, dff), float("nan"), device=dev, dtype=torch.bfloat16)
    out = torch.full((M, d), float("nan"), device=dev)
    o.ffn_fwd(y, p["f.w1p"], b1, p["f.w2p"], b2, x, h, out, M, d, dff, alpha=0.5)
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
    o.ffn_fwd(y, p["f.w1p"], b1, p["f.w2p"], b2, x, h, out, M, d, dff, alpha=0.5, drop_in=d_in, drop_res=d_res)
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
    o.ffn_bwd_dgrad(df, p["f.w2tp"], p["f.w1tp"], h, dh, act, dy, M, d, dff, drop_in=d_in)
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


@pytest.mark.gpu
def test_model_with_fused_feed_forward_matches_the_gemm_pair_path():
    """Conformer-CTC-Large geometry (4 layers), bf16, dropout off: loss and every gradient of the model whose feed-forward blocks
    run as one launch per direction (encoder.ffn_fused) against the same model on the two-GEMM path -- the two differ by the bf16
    rounding of the hidden pre-activation inside the Swish only."""
    from nemo_amd.models import EncDecCTCModel, conformer_ctc_config
    from oracle import conformer_ref as R

    def run(fused):
        torch.manual_seed(0)
        cfg = conformer_ctc_config("large", vocab_size=128, n_layers=4, dropout=0.0, dropout_pre_encoder=0.0, dropout_att=0.0,
                                   dropout_emb=0.0, compute_dtype=torch.bfloat16)
        cfg["preprocessor"]["dither"] = 0.0
        m = EncDecCTCModel(cfg)
        m.decoder.compute_dtype = torch.bfloat16
        m.encoder.ffn_fused = fused
        m = m.to(dev).train()
        audio, alen, tok, tl = R.synthetic_batch(3, 4.0, vocab=128, seed=5)
        out = m.training_step([audio.to(dev), alen.to(dev), tok.to(dev), tl.to(dev)])
        out["loss"].backward()
        torch.cuda.synchronize()
        return out["loss"].item(), {k: p.grad.detach().float().cpu() for k, p in m.named_parameters() if p.grad is not None}

    l1, g1 = run(True)
    l0, g0 = run(False)
    assert abs(l1 - l0) <= 2e-3 * abs(l0), (l1, l0)
    floor = 1e-3 * max(g.norm().item() for g in g0.values())
    worst = max(((g1[k] - g0[k]).norm().item() / max(g0[k].norm().item(), floor), k) for k in g0)
    assert worst[0] < 3e-2, worst
