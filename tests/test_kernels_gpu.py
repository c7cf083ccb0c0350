"""-m gpu: every HIP kernel, called through the C ABI, against the CPU oracle (plain torch fp32 / numpy restatements)
on the same seeded inputs.  Tolerances: fp32 kernels 1e-4..1e-3 relative (north_star: 1e-3 rel fp32); bf16 MFMA
kernels are compared against an fp32 computation on the bf16-rounded operands (so only accumulation order differs)."""
import json
import math
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from oracle import conformer_ref as R
from oracle import ctc_ref

dev = "cuda"


def ops():
    from nemo_amd import ops as _ops
    return _ops


def rel_err(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return ((a - b).abs().max() / (b.abs().max() + 1e-12)).item()


def bf(x):
    return x.to(torch.bfloat16)


# ---------------------------------------------------------------------------------------------- GEMM
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("layout", ["NT", "NN", "TN"])
@pytest.mark.parametrize("shape", [(300, 200, 136), (1000, 512, 512), (130, 129, 64)])
def test_gemm_layouts(dtype, layout, shape):
    o = ops()
    M, N, K = shape
    g = torch.Generator().manual_seed(1)
    A = torch.randn(M, K, generator=g)
    Bm = torch.randn(N, K, generator=g)  # asymmetric on purpose (transpose-detecting)
    Aq, Bq = A.to(dtype), Bm.to(dtype)
    ref = Aq.float() @ Bq.float().t()
    tA, tB = layout[0] == "T", layout[1] == "N"  # "NN": B stored [K][N]; "TN": A stored [K][M], B stored [K][N]
    pad8 = lambda n: (n + 7) // 8 * 8
    if tA:
        Ast = torch.zeros(K, pad8(M), dtype=dtype); Ast[:, :M] = Aq.t(); lda = pad8(M)
    else:
        Ast = Aq.contiguous(); lda = K
    if tB:
        Bst = torch.zeros(K, pad8(N), dtype=dtype); Bst[:, :N] = Bq.t(); ldb = pad8(N)
    else:
        Bst = Bq.contiguous(); ldb = K
    Cd = torch.full((M, N), float("nan"), device=dev)
    o.gemm(Ast.to(dev), Bst.to(dev), Cd, M, N, K, lda, ldb, N, transA=tA, transB=tB)
    torch.cuda.synchronize()
    assert rel_err(Cd, ref) < (2e-5 if dtype == torch.float32 else 1e-4), (layout, shape, rel_err(Cd, ref))


@pytest.mark.parametrize("layout", ["NT", "NN", "TN", "TT"])
@pytest.mark.parametrize("shape", [(300, 200, 136), (1000, 512, 512), (130, 129, 64), (33, 31, 7), (64, 64, 1030)])
def test_gemm_f32_on_the_matrix_cores_matches_the_vector_unit_kernel(layout, shape):
    """fp32 problems run on v_mfma_f32_32x32x2_f32 (gemm_f32_mfma_kernel); mi355x_gemm_config(3, 0) selects the vector-unit kernel
    it replaces.  Both against an f64 product of the same operands: the matrix-core kernel must be as exact as the VALU one
    (fp32 operands, fp32 accumulation: only the summation order differs), ragged tiles and every operand layout included."""
    o = ops()
    M, N, K = shape
    g = torch.Generator().manual_seed(5)
    A = torch.randn(M, K, generator=g)
    Bm = torch.randn(N, K, generator=g)
    ref = (A.double() @ Bm.double().t())
    tA, tB = layout[0] == "T", layout[1] == "N"
    Ast, lda = (A.t().contiguous(), M) if tA else (A, K)
    Bst, ldb = (Bm.t().contiguous(), N) if tB else (Bm, K)
    errs = {}
    prev = o.gemm_config(3, 1)
    try:
        for mode in (1, 0):
            o.gemm_config(3, mode)
            Cd = torch.full((M, N), float("nan"), device=dev)
            o.gemm(Ast.to(dev), Bst.to(dev), Cd, M, N, K, lda, ldb, N, transA=tA, transB=tB)
            torch.cuda.synchronize()
            errs[mode] = ((Cd.double().cpu() - ref).abs().max() / ref.abs().max()).item()
    finally:
        o.gemm_config(3, prev if prev >= 0 else 1)
    assert errs[1] < 2e-6 and errs[0] < 2e-6, errs
    assert errs[1] < 4 * errs[0] + 1e-7, errs


def test_gemm_f32_on_the_matrix_cores_epilogues_batch_and_split_k():
    o = ops()
    g = torch.Generator().manual_seed(6)
    # batched + strided, bias, residual epilogue
    nb, M, N, K = 3, 70, 90, 50
    A = torch.randn(nb, M, K, generator=g); Bm = torch.randn(nb, N, K, generator=g)
    bias = torch.randn(N, generator=g); res = torch.randn(nb, M, N, generator=g)
    want = res + 0.5 * (torch.einsum("bmk,bnk->bmn", A.double(), Bm.double()) + bias.double())
    got = {}
    prev = o.gemm_config(3, 1)
    try:
        for mode in (1, 0):
            o.gemm_config(3, mode)
            Cd = torch.full((nb, M, N), float("nan"), device=dev)
            o.gemm(A.to(dev), Bm.to(dev), Cd, M, N, K, K, K, N, batch=nb, sA=(M * K, 0), sB=(N * K, 0), sC=(M * N, 0),
                   bias=bias.to(dev), alpha=0.5, epi=o.EPI_RESID, aux_in=res.to(dev), ldaux=N)
            torch.cuda.synchronize()
            got[mode] = Cd.cpu()
            assert ((Cd.double().cpu() - want).abs().max() / want.abs().max()).item() < 2e-6, mode
        # split-K with atomic accumulation into a pre-filled C (weight-gradient form: both operands reduction-major)
        M2, N2, K2 = 96, 80, 1000
        X = torch.randn(K2, M2, generator=g); Y = torch.randn(K2, N2, generator=g)
        base = torch.randn(M2, N2, generator=g)
        want2 = base.double() + X.double().t() @ Y.double()
        for mode in (1, 0):
            o.gemm_config(3, mode)
            Cd = base.clone().to(dev)
            o.gemm(X.to(dev), Y.to(dev), Cd, M2, N2, K2, M2, N2, N2, transA=True, transB=True, atomic=True, splitk=4)
            torch.cuda.synchronize()
            assert ((Cd.double().cpu() - want2).abs().max() / want2.abs().max()).item() < 5e-6, mode
    finally:
        o.gemm_config(3, prev if prev >= 0 else 1)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_gemm_epilogues(dtype):
    o = ops()
    M, N, K = 260, 384, 192
    g = torch.Generator().manual_seed(2)
    A = torch.randn(M, K, generator=g).to(dtype)
    W = (torch.randn(N, K, generator=g) * 0.1).to(dtype)
    bias = torch.randn(N, generator=g)
    res = torch.randn(M, N, generator=g)
    acc = A.float() @ W.float().t() + bias
    Ad, Wd, bd, resd = A.to(dev), W.to(dev), bias.to(dev), res.to(dev)
    tol = 2e-5 if dtype == torch.float32 else 1e-2
    # STORE with alpha, bf16/f32 out
    out = torch.empty(M, N, device=dev, dtype=dtype)
    o.gemm(Ad, Wd, out, M, N, K, K, K, N, bias=bd, alpha=0.5)
    assert rel_err(out, 0.5 * acc) < tol
    # SWISH_DROP (no dropout): aux_out = pre-act, C = swish
    h = torch.empty(M, N, device=dev, dtype=dtype)
    a = torch.empty(M, N, device=dev, dtype=dtype)
    o.gemm(Ad, Wd, a, M, N, K, K, K, N, bias=bd, epi=o.EPI_SWISH_DROP, aux_out=h)
    assert rel_err(h, acc) < tol and rel_err(a, acc * torch.sigmoid(acc)) < tol
    # RESID: C(f32) = res + alpha*(acc)
    r2 = torch.empty(M, N, device=dev)
    o.gemm(Ad, Wd, r2, M, N, K, K, K, N, bias=bd, alpha=0.5, epi=o.EPI_RESID, aux_in=resd)
    assert rel_err(r2, res + 0.5 * acc) < tol
    # RESID in place
    r3 = resd.clone()
    o.gemm(Ad, Wd, r3, M, N, K, K, K, N, bias=bd, alpha=1.0, epi=o.EPI_RESID, aux_in=r3)
    assert rel_err(r3, res + acc) < tol
    # DSWISH: C = acc * swish'(aux)
    pre = torch.randn(M, N, generator=g)
    sg = torch.sigmoid(pre)
    o.gemm(Ad, Wd, out, M, N, K, K, K, N, epi=o.EPI_DSWISH, aux_in=pre.to(dev).to(dtype))
    pre_q = pre.to(dtype).float(); sg = torch.sigmoid(pre_q)
    assert rel_err(out, (acc - bias) * (sg * (1 + pre_q * (1 - sg)))) < tol
    # MUL_POS
    o.gemm(Ad, Wd, out, M, N, K, K, K, N, epi=o.EPI_MUL_POS, aux_in=pre.to(dev).to(dtype))
    assert rel_err(out, (acc - bias) * (pre_q > 0)) < tol
    # RELU_MASK: rows m = (b, t, f) with 4 inner rows; lens per b
    rows_per_b, inner = 52, 4  # 5 batches x 13 t x 4
    lens = torch.tensor([13, 7, 1, 0, 10], dtype=torch.int64)
    o.gemm(Ad, Wd, out, M, N, K, K, K, N, bias=bd, epi=o.EPI_RELU_MASK, row_len=lens.to(dev), rows_per_b=rows_per_b,
           rows_inner=inner)
    m = torch.arange(M)
    keep = ((m % rows_per_b) // inner) < lens[m // rows_per_b]
    assert rel_err(out, torch.relu(acc) * keep[:, None]) < tol
    # atomic split-K accumulate on top of existing content
    base = torch.ones(M, N, device=dev)
    o.gemm(Ad, Wd, base, M, N, K, K, K, N, atomic=True, splitk=3)
    assert rel_err(base, 1.0 + (acc - bias)) < tol
    torch.cuda.synchronize()


@pytest.mark.parametrize("shape", [(260, 384, 192), (1031, 2048, 512), (4100, 2048, 512)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_gemm_swish_derivative_taken_in_the_forward_epilogue(dtype, shape):
    """MI355X_EPI_SWISH_DROP_G / _DSWISH_G against the pre-activation pair (same dropout site): the activations are the same bits,
    the stored factor is swish'(h) * mask, and the backward product follows the pre-activation form within one more bf16 rounding.
    Shapes: the generic 256 x 128 structure, the 256 x 256 structure (N = 2048) and the persistent one (bf16, more tiles than CUs)."""
    o = ops()
    M, N, K = shape
    g = torch.Generator().manual_seed(12)
    A = torch.randn(M, K, generator=g).to(dtype)
    W = (torch.randn(N, K, generator=g) * 0.1).to(dtype)
    bias = torch.randn(N, generator=g)
    Ad, Wd, bd = A.to(dev), W.to(dev), bias.to(dev)
    drop = o.Dropout(0.1, 77, 5)
    h = torch.empty(M, N, device=dev, dtype=dtype); a = torch.empty_like(h)
    gf = torch.empty_like(h); a2 = torch.empty_like(h)
    o.gemm(Ad, Wd, a, M, N, K, K, K, N, bias=bd, epi=o.EPI_SWISH_DROP, aux_out=h, drop=drop)
    o.gemm(Ad, Wd, a2, M, N, K, K, K, N, bias=bd, epi=o.EPI_SWISH_DROP_G, aux_out=gf, drop=drop)
    torch.cuda.synchronize()
    assert torch.equal(a, a2)
    acc = A.float() @ W.float().t() + bias
    sg = torch.sigmoid(acc)
    mask = (a.float().cpu() != 0) | (acc * sg == 0)            # kept elements (the mask is regenerated, not stored)
    want_g = sg * (1 + acc * (1 - sg)) * mask / 0.9
    tol = 2e-5 if dtype == torch.float32 else 1e-2
    assert rel_err(gf, want_g) < tol
    # backward: dh = (dy @ W2^T) * swish'(h) * mask
    dy = torch.randn(M, K, generator=g).to(dtype).to(dev)
    dh1 = torch.empty_like(h); dh2 = torch.empty_like(h)
    o.gemm(dy, Wd, dh1, M, N, K, K, K, N, epi=o.EPI_DSWISH, aux_in=h, drop=drop)
    o.gemm(dy, Wd, dh2, M, N, K, K, K, N, epi=o.EPI_DSWISH_G, aux_in=gf)
    torch.cuda.synchronize()
    assert rel_err(dh2, dh1) < tol
    assert torch.equal(dh1 == 0, dh2 == 0) or dtype == torch.float32   # the same elements are dropped


def test_gemm_batched_strided_bf16():
    """attention-style: q [B*T, 3d] head slices, scores [H,B,T,Tp] -- two-level batch strides."""
    o = ops()
    Bn, H, T, dk = 3, 4, 77, 32
    d = H * dk
    Tp = 128
    g = torch.Generator().manual_seed(3)
    qkv = bf(torch.randn(Bn * T, 3 * d, generator=g))
    ref = torch.einsum("bihe,bjhe->hbij", qkv[:, :d].float().view(Bn, T, H, dk), qkv[:, d:2 * d].float().view(Bn, T, H, dk))
    out = torch.zeros(H, Bn, T, Tp, device=dev)
    # the P.V product below runs with K = Tp > T: for the last utterance the reduction walks Tp - T rows past its own V rows
    # (multiplied by the zero pad columns of P).  They must be finite memory of OUR allocation -- 0 x (whatever the
    # allocator left behind, possibly NaN bits) made this test flaky -- so the device copy carries zero rows behind it
    qd = torch.zeros(Bn * T + Tp, 3 * d, device=dev, dtype=torch.bfloat16)
    qd[: Bn * T] = qkv.to(dev)
    o.gemm(qd, qd, out, T, T, dk, 3 * d, 3 * d, Tp, batch=H * Bn, nb0=Bn, sA=(T * 3 * d, dk), sB=(T * 3 * d, dk),
           sC=(T * Tp, Bn * T * Tp), b_off=d)
    torch.cuda.synchronize()
    assert rel_err(out[..., :T], ref) < 1e-4
    # P @ V  (NN): probs [H,B,T,Tp] bf16 @ v -> ctx [B*T, d] written strided
    p = bf(torch.rand(H, Bn, T, Tp, generator=g)); p[..., T:] = 0
    ctx = torch.zeros(Bn * T, d, device=dev, dtype=torch.bfloat16)
    o.gemm(p.to(dev), qd, ctx, T, dk, Tp, Tp, 3 * d, d, transB=True, batch=H * Bn, nb0=Bn, sA=(T * Tp, Bn * T * Tp),
           sB=(T * 3 * d, dk), sC=(T * d, dk), b_off=2 * d)
    v = qkv[:, 2 * d:].float().view(Bn, T, H, dk)
    ref2 = torch.einsum("hbij,bjhe->bihe", p.float()[..., :T], v).reshape(Bn * T, d)
    torch.cuda.synchronize()
    assert rel_err(ctx, ref2) < 1e-2


def test_gemm_wgrad_fused_bias_grad():
    """TN wgrad with the bias-gradient column sums riding along (colsum_out), split-K atomics, M tail"""
    o = ops()
    rows, n_out, n_in = 1000, 200, 136
    g = torch.Generator().manual_seed(21)
    dY = bf(torch.randn(rows, 208, generator=g)); dY[:, 200:] = 0
    X = bf(torch.randn(rows, n_in, generator=g))
    dW = torch.ones(n_out, n_in, device=dev); db = torch.ones(n_out, device=dev)
    o.gemm(dY.to(dev), X.to(dev), dW, n_out, n_in, rows, 208, n_in, n_in, transA=True, transB=True, atomic=True, splitk=4,
           colsum_out=db)
    torch.cuda.synchronize()
    assert rel_err(dW, 1 + dY[:, :200].float().t() @ X.float()) < 1e-4
    assert rel_err(db, 1 + dY[:, :200].float().sum(0)) < 1e-4


def test_gemm_grouped_weight_gradients():
    """several weight-gradient problems (dW += dY^T X, bias gradient riding along) in ONE launch == one by one"""
    o = ops()
    g = torch.Generator().manual_seed(23)
    rows = 1500
    shapes = [(512, 256), (256, 512), (192, 128), (640, 96), (256, 256)]
    probs, refs = [], []
    dYall = bf(torch.randn(rows, 1024, generator=g)).to(dev)  # two problems slice columns of one matrix (q/k/v style)
    for i, (n_out, n_in) in enumerate(shapes):
        X = bf(torch.randn(rows, n_in, generator=g)).to(dev)
        if i < 2:
            dY, ldy, off = dYall, 1024, i * 512
            dYv = dYall[:, off:off + n_out]
        else:
            dY = bf(torch.randn(rows, n_out, generator=g)).to(dev); ldy, off = n_out, 0
            dYv = dY
        dW = torch.randn(n_out, n_in, generator=g).to(dev)          # accumulate onto existing content
        db = torch.randn(n_out, generator=g).to(dev) if i % 2 == 0 else None
        refs.append((dW.clone() + dYv.float().t() @ X.float(), None if db is None else db.clone() + dYv.float().sum(0)))
        probs.append((dY, ldy, off, X, n_in, 0, dW, n_out, n_in, db))
    o.wgrad_grouped(probs, rows, 3)
    torch.cuda.synchronize()
    for (dY, ldy, off, X, ldx, xo, dW, n_out, n_in, db), (rw, rb) in zip(probs, refs):
        assert rel_err(dW, rw) < 2e-5, (n_out, n_in)
        if db is not None:
            assert rel_err(db, rb) < 2e-5, (n_out, n_in)


def test_gemm_dropout_consistency():
    """EPI_SWISH_DROP forward mask == mask regenerated by drop_scale_cast (same key, idx = m*N+n)."""
    o = ops()
    M, N, K = 128, 256, 64
    A = torch.ones(M, K, device=dev); W = torch.ones(N, K, device=dev) / K
    drop = o.Dropout(0.25, seed=7, site=3)
    h = torch.empty(M, N, device=dev); a = torch.empty(M, N, device=dev)
    o.gemm(A, W, a, M, N, K, K, K, N, epi=o.EPI_SWISH_DROP, aux_out=h, drop=drop)
    ones = torch.ones(M, N, device=dev); m2 = torch.empty(M, N, device=dev)
    o.drop_scale_cast(ones, m2, M * N, 1.0, drop)
    torch.cuda.synchronize()
    sw = 1.0 * torch.sigmoid(torch.tensor(1.0)).item()
    assert torch.allclose(a, m2 * sw, rtol=1e-5)
    keep = (m2 > 0).float().mean().item()
    assert abs(keep - 0.75) < 0.02, keep
    assert torch.allclose(m2[m2 > 0], torch.tensor(1 / 0.75, device=dev))


@pytest.mark.parametrize("M", [100, 700])
def test_gemm_width_4_mod_8_vector_tail_and_unaligned_dropout_runs(M):
    """N = 324 (Squeezeformer-Medium's d_model): the last four columns go out as one 4-wide access, full-width tiles take the
    templated epilogue although every other row starts its dropout run in the middle of a hash group, f32 rows of pitch 324 count
    as vector-aligned (csrc/gemm.hip: epilogue4 / epilogue_chunk / drop_mask8u).  M = 100: the 128x128 structure, 700: 256x128.
    Masks must be the ones drop_scale_cast regenerates for idx = m * N + n; pad columns of a pitched C stay untouched."""
    o = ops()
    N, K, ldp = 324, 136, 328
    g = torch.Generator().manual_seed(M)
    A = torch.randn(M, K, generator=g).to(torch.bfloat16).to(dev); W = (torch.randn(N, K, generator=g) / 8).to(torch.bfloat16).to(dev)
    bias = torch.randn(N, generator=g).to(dev)
    ref = A.float() @ W.float().t() + bias
    drop = o.Dropout(0.25, seed=11, site=5)
    ones = torch.ones(M, N, device=dev); mask = torch.empty(M, N, device=dev)
    o.drop_scale_cast(ones, mask, M * N, 1.0, drop)
    # (a) bf16 C with pitch 328, bias + dropout
    C = torch.full((M, ldp), float("nan"), device=dev, dtype=torch.bfloat16)
    o.gemm(A, W, C, M, N, K, K, K, ldp, bias=bias, drop=drop)
    torch.cuda.synchronize()
    assert torch.isnan(C[:, N:].float()).all(), "pad columns written"
    assert rel_err(C[:, :N].float(), ref * mask) < 6e-3
    assert ((C[:, :N].float() == 0) == (mask == 0)).all()
    # (b) f32 residual epilogue, C and aux rows of pitch 324
    R = torch.randn(M, N, generator=g).to(dev); C2 = torch.empty(M, N, device=dev)
    o.gemm(A, W, C2, M, N, K, K, K, N, bias=bias, alpha=0.5, epi=o.EPI_RESID, aux_in=R, drop=drop)
    torch.cuda.synchronize()
    assert rel_err(C2, R + 0.5 * ref * mask) < 1e-3
    # (c) Swish + dropout with the second (bf16, pitched) output
    H = torch.full((M, ldp), float("nan"), device=dev, dtype=torch.bfloat16)
    C3 = torch.full((M, ldp), float("nan"), device=dev, dtype=torch.bfloat16)
    o.gemm(A, W, C3, M, N, K, K, K, ldp, bias=bias, epi=o.EPI_SWISH_DROP, aux_out=H, ldaux=ldp, drop=drop)
    torch.cuda.synchronize()
    assert torch.isnan(C3[:, N:].float()).all() and torch.isnan(H[:, N:].float()).all()
    assert rel_err(H[:, :N].float(), ref) < 6e-3
    assert rel_err(C3[:, :N].float(), ref * torch.sigmoid(ref) * mask) < 8e-3
    # (d) no dropout, plain f32 store of pitch 324 == the same product through an aligned width (N = 320 columns of it)
    C4 = torch.empty(M, N, device=dev); C5 = torch.empty(M, 320, device=dev)
    o.gemm(A, W, C4, M, N, K, K, K, N, bias=bias)
    o.gemm(A, W, C5, M, 320, K, K, K, 320, bias=bias[:320].contiguous())
    torch.cuda.synchronize()
    assert torch.equal(C4[:, :320], C5) and rel_err(C4, ref) < 1e-3


# ---------------------------------------------------------------------------------------------- LayerNorm etc.
@pytest.mark.parametrize("d", [176, 512])
@pytest.mark.parametrize("xdt,ydt", [(torch.float32, torch.float32), (torch.float32, torch.bfloat16)])
def test_layernorm(d, xdt, ydt):
    o = ops()
    M = 333
    g = torch.Generator().manual_seed(4)
    x = torch.randn(M, d, generator=g) * 2 + 0.5
    gamma = torch.randn(d, generator=g); beta = torch.randn(d, generator=g)
    xr = x.clone().requires_grad_(True); gr = gamma.clone().requires_grad_(True); br = beta.clone().requires_grad_(True)
    y_ref = F.layer_norm(xr, (d,), gr, br, 1e-5)
    dy = torch.randn(M, d, generator=g).to(ydt)
    y_ref.backward(dy.float())
    xd = x.to(dev); y = torch.empty(M, d, device=dev, dtype=ydt)
    mean = torch.empty(M, device=dev); rstd = torch.empty(M, device=dev)
    o.layernorm_fwd(xd, gamma.to(dev), beta.to(dev), y, mean, rstd, M, d)
    assert rel_err(y, y_ref) < (1e-5 if ydt == torch.float32 else 5e-3)
    dres = torch.ones(M, d, device=dev)
    dg = torch.zeros(d, device=dev); db = torch.zeros(d, device=dev)
    o.layernorm_bwd(dy.to(dev), xd, gamma.to(dev), mean, rstd, dres, True, dg, db, M, d)
    torch.cuda.synchronize()
    assert rel_err(dres - 1.0, xr.grad) < 1e-4
    assert rel_err(dg, gr.grad) < 1e-4 and rel_err(db, br.grad) < 1e-4


@pytest.mark.parametrize("ydt", [torch.float32, torch.bfloat16])
def test_layernorm2_is_two_layernorms(ydt):
    """mi355x_layernorm2_fwd = LN1 (f32 out) followed by LN2 (compute-dtype out), bit for bit, statistics included"""
    o = ops()
    M, d = 333, 512
    g = torch.Generator().manual_seed(41)
    x = (torch.randn(M, d, generator=g) * 2 + 0.5).to(dev)
    g1, b1, g2, b2 = (torch.randn(d, generator=g).to(dev) for _ in range(4))
    y1 = torch.empty(M, d, device=dev); y2 = torch.empty(M, d, device=dev, dtype=ydt)
    m1, r1, m2, r2 = (torch.empty(M, device=dev) for _ in range(4))
    o.layernorm_fwd(x, g1, b1, y1, m1, r1, M, d)
    o.layernorm_fwd(y1, g2, b2, y2, m2, r2, M, d)
    z1 = torch.empty_like(y1); z2 = torch.empty_like(y2)
    n1, s1, n2, s2 = (torch.empty(M, device=dev) for _ in range(4))
    o.layernorm2_fwd(x, g1, b1, z1, n1, s1, g2, b2, z2, n2, s2, M, d)
    torch.cuda.synchronize()
    assert torch.equal(z1, y1) and torch.equal(z2, y2)
    assert torch.equal(n1, m1) and torch.equal(s1, r1) and torch.equal(n2, m2) and torch.equal(s2, r2)


@pytest.mark.parametrize("N", [64, 136, 256, 512, 1024])
def test_colsum_bf16_all_lane_layouts(N):
    """bf16 column sums (bias gradients): 16-byte chunks per lane; rows narrower than a wave's 512 columns share the wave"""
    o = ops()
    M = 777
    g = torch.Generator().manual_seed(N)
    x = torch.randn(M, N, generator=g).to(torch.bfloat16)
    out = torch.ones(N, device=dev)
    o.colsum(x.to(dev), out, M, N, alpha=0.5)
    torch.cuda.synchronize()
    assert rel_err(out, 1 + 0.5 * x.float().sum(0)) < 1e-5


@pytest.mark.parametrize("dydt", [torch.float32, torch.bfloat16])
def test_layernorm2_bwd_is_two_layernorm_backwards(dydt):
    """mi355x_layernorm2_bwd (layer i+1's norm_feed_forward1 backward + layer i's norm_out backward, the gradient between them
    kept in registers) = the two single launches: residual gradient and its dropped bf16 copy to the last bit or two (the compiler
    contracts the same expressions into FMAs independently per kernel), the same dropout mask, the four parameter gradients up to
    the order of their atomics"""
    o = ops()
    M, d = 777, 512
    g = torch.Generator().manual_seed(43)
    r4 = (torch.randn(M, d, generator=g) * 2 + 0.3).to(dev)
    g1, b1, g2, b2 = (torch.randn(d, generator=g).to(dev) for _ in range(4))
    xo = torch.empty(M, d, device=dev); y = torch.empty(M, d, device=dev, dtype=dydt)
    m5, r5, m1, r1 = (torch.empty(M, device=dev) for _ in range(4))
    o.layernorm2_fwd(r4, g2, b2, xo, m5, r5, g1, b1, y, m1, r1, M, d) if dydt == torch.bfloat16 else (
        o.layernorm_fwd(r4, g2, b2, xo, m5, r5, M, d), o.layernorm_fwd(xo, g1, b1, y, m1, r1, M, d))
    dy1 = torch.randn(M, d, generator=g).to(dydt).to(dev)
    dres_in = torch.randn(M, d, generator=g).to(dev)
    drop = o.Dropout(0.1, 5, 7)
    # reference: two launches
    dxo = dres_in.clone()
    dg1, db1, dg2, db2 = (torch.zeros(d, device=dev) for _ in range(4))
    o.layernorm_bwd(dy1, xo, g1, m1, r1, dxo, True, dg1, db1, M, d)
    dr = torch.empty(M, d, device=dev); cast = torch.empty(M, d, device=dev, dtype=torch.bfloat16)
    o.layernorm_bwd(dxo, r4, g2, m5, r5, dr, False, dg2, db2, M, d, cast_out=cast, cast_scale=0.5, cast_drop=drop)
    # fused
    eg1, eb1, eg2, eb2 = (torch.zeros(d, device=dev) for _ in range(4))
    dr2 = torch.full((M, d), float("nan"), device=dev); cast2 = torch.full((M, d), float("nan"), device=dev, dtype=torch.bfloat16)
    o.layernorm2_bwd(dy1, xo, g1, m1, r1, eg1, eb1, dres_in, r4, g2, m5, r5, eg2, eb2, dr2, M, d, cast_out=cast2, cast_scale=0.5,
                     cast_drop=drop)
    torch.cuda.synchronize()
    assert rel_err(dr2, dr) < 1e-6 and rel_err(cast2, cast) < 1e-3
    assert torch.equal(cast2 == 0, cast == 0) and 0.05 < (cast == 0).float().mean().item() < 0.15
    for a, b in ((eg1, dg1), (eb1, db1), (eg2, dg2), (eb2, db2)):
        assert rel_err(a, b) < 1e-5


def test_colsum_logsoftmax():
    o = ops()
    M, C_ = 700, 129
    g = torch.Generator().manual_seed(5)
    x = torch.randn(M, 136, generator=g)
    out = torch.ones(C_, device=dev)
    o.colsum(x.to(dev), out, M, C_, ld=136, alpha=2.0)
    assert rel_err(out, 1 + 2 * x[:, :C_].sum(0)) < 1e-5
    logits = x[:, :C_].contiguous().requires_grad_(True)
    lp_ref = torch.log_softmax(logits, -1)
    dl = torch.randn(M, C_, generator=g)
    lp_ref.backward(dl)
    lp = torch.empty(M, C_, device=dev)
    o.log_softmax_fwd(x.to(dev), 136, lp, C_, M, C_)
    assert rel_err(lp, lp_ref) < 1e-5
    dx = torch.full((M, 136), float("nan"), device=dev)
    o.log_softmax_bwd(dl.to(dev), lp, C_, dx, 136, M, C_, 1.0)
    torch.cuda.synchronize()
    assert rel_err(dx[:, :C_], logits.grad) < 1e-5
    assert torch.all(dx[:, C_:] == 0)


def test_specaug_fill_rects_and_module(golden_dir):
    """mi355x_fill_rects against the oracle's slice assignment (incl. clipped / empty / overlapping rectangles), and the
    drop-in SpectrogramAugmentation on the GPU: same cells as its own parameter draw at the same device seed, input intact"""
    from nemo_amd.modules import SpectrogramAugmentation
    from oracle import specaug_ref as SR
    o = ops()
    z = np.load(os.path.join(golden_dir, "ref_specaug.npz"))
    x = torch.from_numpy(z["x"]); length = torch.from_numpy(z["length"])
    B, F, T = x.shape
    rects = torch.tensor([[0, 0, F, 10, 60], [1, 5, 32, 0, T], [1, 20, 40, 100, 90], [2, -3, 4, T - 5, T + 9],
                          [3, 79, 200, 0, 1], [3, 0, F, 0, T], [0, 7, 7, 0, T], [2, 10, 12, 300, 301]], dtype=torch.long)
    want = SR.apply_rects(x, rects, -2.0)
    got = o.fill_rects(x.clone().to(dev), rects.to(torch.int32).to(dev), -2.0)
    torch.cuda.synchronize()
    assert torch.equal(got.cpu(), want)
    # reference-generated rectangles (legacy stream) through the kernel reproduce the reference's output cells
    import random
    r2 = SR.legacy_rects(random.Random(7), B, F, T, length, 2, 5, 27, 0.05)
    got = o.fill_rects(x.clone().to(dev), r2.to(torch.int32).to(dev), 0.0).cpu()
    mask = np.unpackbits(z["legacy_mask"])[: x.numel()].reshape(tuple(x.shape)).astype(bool)
    assert np.array_equal((got != x).numpy(), mask) and torch.all(got[torch.from_numpy(mask)] == 0)
    # module on the device
    m = SpectrogramAugmentation(freq_masks=2, time_masks=10, freq_width=27, time_width=0.05)
    xd, ld = x.to(dev), length.to(dev)
    torch.manual_seed(5)
    groups = m.mask_rects(B, F, T, ld, xd.device)
    torch.manual_seed(5)
    y = m(input_spec=xd, length=ld)
    torch.cuda.synchronize()
    ref = x
    for r, v in groups:
        ref = SR.apply_rects(ref, r.cpu(), v)
    assert torch.equal(y.cpu(), ref) and torch.equal(xd.cpu(), x)
    masked_frac = (y.cpu() != x).float().mean().item()
    assert 0.05 < masked_frac < 0.9


@pytest.mark.parametrize("tw", [0.05, 1.0, 25])
def test_specaug_mask_parameters_in_one_launch_equal_the_tensor_ops(tw):
    """mi355x_specaug_rects (the reference's four uniform draws -> rectangle table in one launch) against the module's tensor-op
    path, which restates spectr_augment.py:155-195 op by op and is pinned to the reference fixture on CPU: same device seed, same
    integers -- fraction-of-utterance and frame-count time widths, utterances shorter than a mask"""
    from nemo_amd.modules import SpectrogramAugmentation
    B, F, T = 9, 80, 1201
    length = torch.tensor([1201, 1200, 777, 640, 333, 100, 17, 3, 1], dtype=torch.int64, device=dev)
    m = SpectrogramAugmentation(freq_masks=2, time_masks=10, freq_width=27, time_width=tw)
    got, want = [], []
    for seed in (1, 2, 3):
        m.fused_rects = True
        torch.manual_seed(seed)
        got.append(m.mask_rects(B, F, T, length, torch.device(dev))[0][0])
        m.fused_rects = False
        torch.manual_seed(seed)
        want.append(m.mask_rects(B, F, T, length, torch.device(dev))[0][0])
    torch.cuda.synchronize()
    for g_, w_ in zip(got, want):
        assert g_.dtype == torch.int32 and g_.shape == w_.shape == (B * 12, 5)
        assert torch.equal(g_.long().cpu(), w_.long().cpu())
    assert not torch.equal(got[0], got[1])   # (the seeds draw different masks)


@pytest.mark.parametrize("M,d", [(700, 64), (1003, 512), (130, 176)])
def test_add2_colsum(M, d):
    """dq = dqu + dqv fused with the pos_bias_u / pos_bias_v gradients (multi_head_attention.py:288-291)"""
    o = ops()
    g = torch.Generator().manual_seed(11)
    a = bf(torch.randn(M, d, generator=g)); b = bf(torch.randn(M, d, generator=g))
    out = torch.full((M, 3 * d), float("nan"), device=dev, dtype=torch.bfloat16)
    sums = torch.ones(2 * d, device=dev)
    o.add2_colsum(a.to(dev), b.to(dev), out, 3 * d, M, d, sums)
    torch.cuda.synchronize()
    assert torch.equal(out[:, :d].float().cpu(), bf(a.float() + b.float()).float())
    assert torch.isnan(out[:, d:].float()).all()
    assert rel_err(sums[:d], 1 + a.float().sum(0)) < 1e-5
    assert rel_err(sums[d:], 1 + b.float().sum(0)) < 1e-5


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_glu(dtype):
    o = ops()
    Bn, T, d = 3, 50, 64
    lens = torch.tensor([50, 31, 7])
    g = torch.Generator().manual_seed(6)
    x = torch.randn(Bn * T, 2 * d, generator=g).to(dtype)
    xr = x.float().requires_grad_(True)
    valid = (torch.arange(T)[None] < lens[:, None]).reshape(-1, 1).float()
    ref = xr[:, :d] * torch.sigmoid(xr[:, d:]) * valid
    dout = torch.randn(Bn * T, d, generator=g).to(dtype)
    ref.backward(dout.float())
    out = torch.empty(Bn * T, d, device=dev, dtype=dtype)
    o.glu_fwd(x.to(dev), out, lens.to(dev), T, Bn * T, d)
    din = torch.empty(Bn * T, 2 * d, device=dev, dtype=dtype)
    o.glu_bwd(x.to(dev), dout.to(dev), din, lens.to(dev), T, Bn * T, d)
    torch.cuda.synchronize()
    tol = 1e-5 if dtype == torch.float32 else 1e-2
    assert rel_err(out, ref) < tol and rel_err(din, xr.grad) < tol


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_relpos_softmax(dtype):
    o = ops()
    H, Bn, T = 2, 3, 45
    Tp, Pp = 64, 96
    lens = torch.tensor([45, 30, 1])
    g = torch.Generator().manual_seed(7)
    ac = torch.randn(H, Bn, T, Tp, generator=g).requires_grad_(True)
    bdf = torch.randn(H, Bn, T, Pp, generator=g).requires_grad_(True)
    scale = 0.3
    ii = torch.arange(T)[:, None]; jj = torch.arange(T)[None]
    bd = bdf[:, :, ii, T - 1 + jj - ii]
    scores = (ac[..., :T] + bd) * scale
    valid = torch.arange(T)[None] < lens[:, None]
    masked = ~(valid[:, :, None] & valid[:, None, :])
    sc = scores.masked_fill(masked[None], -10000.0)
    s_ref = torch.softmax(sc, -1).masked_fill(masked[None], 0.0)
    dpd = torch.randn(H, Bn, T, Tp, generator=g)
    (s_ref * dpd[..., :T]).sum().backward()
    s = torch.full((H, Bn, T, Tp), float("nan"), device=dev, dtype=dtype)
    o.relpos_softmax_fwd(ac.detach().to(dev), bdf.detach().to(dev), s, None, lens.to(dev), H, Bn, T, Tp, Pp, scale)
    tol = 1e-5 if dtype == torch.float32 else 1e-2
    assert rel_err(s[..., :T], s_ref) < tol
    assert torch.all(s[..., T:] == 0)
    dscore = torch.full((H, Bn, T, Tp), float("nan"), device=dev, dtype=dtype)
    dbdf = torch.full((H, Bn, T, Pp), float("nan"), device=dev, dtype=dtype)
    o.relpos_softmax_bwd(dpd.to(dev), s, dscore, dbdf, H, Bn, T, Tp, Pp, scale)
    torch.cuda.synchronize()
    assert rel_err(dscore[..., :T], ac.grad[..., :T]) < 3 * tol
    assert rel_err(dbdf, bdf.grad) < 3 * tol
    assert torch.all(dscore[..., T:] == 0)


def _attn_ref(qkv, pos, u, v, lens, B, H, T, dk):
    """fp32 reference of multi_head_attention.py:272-354 on the (bf16-rounded) inputs; returns ctx [B*T, d], lse [B,H,T]"""
    d = H * dk
    q = qkv[:, :d].float().view(B, T, H, dk); k = qkv[:, d:2 * d].float().view(B, T, H, dk).transpose(1, 2)
    vv = qkv[:, 2 * d:].float().view(B, T, H, dk).transpose(1, 2)
    p = pos.float().view(2 * T - 1, H, dk).transpose(0, 1)
    qu = bf(q + u.view(H, dk)).float().transpose(1, 2); qv = bf(q + v.view(H, dk)).float().transpose(1, 2)
    ac = qu @ k.transpose(-1, -2)
    bdf = qv @ p.transpose(-1, -2).unsqueeze(0)
    ii = torch.arange(T)[:, None]; jj = torch.arange(T)[None]
    sc = (ac + bdf[:, :, ii, T - 1 + jj - ii]) / math.sqrt(dk)
    valid = torch.arange(T)[None] < lens[:, None]
    masked = ~(valid[:, :, None] & valid[:, None, :])[:, None]
    sc = sc.masked_fill(masked, -10000.0)
    attn = torch.softmax(sc, -1).masked_fill(masked, 0.0)
    lse = torch.logsumexp(sc.masked_fill(masked, -float("inf")), -1)
    ctx = (attn @ vv).transpose(1, 2).reshape(B * T, d)
    return ctx, lse, attn


@pytest.mark.parametrize("dk", [64, 128])   # 128: round 5, the kernels' second head width (d_k 65..128 zero-padded)
@pytest.mark.parametrize("T", [45, 160, 501])
def test_relpos_flash_attention_fwd(T, dk):
    o = ops()
    B, H = 3, 2
    d = H * dk
    g = torch.Generator().manual_seed(31)
    qkv = bf(torch.randn(B * T, 3 * d, generator=g))
    pos = bf(torch.randn(2 * T - 1, d, generator=g))
    u = torch.randn(d, generator=g) * 0.5; v = torch.randn(d, generator=g) * 0.5
    lens = torch.tensor([T, max(1, T // 2 + 3), 1])
    ctx_ref, lse_ref, _ = _attn_ref(qkv, pos, u, v, lens, B, H, T, dk)
    ctx = torch.full((B * T, d), float("nan"), device=dev, dtype=torch.bfloat16)
    ctx_lo = torch.full((B * T, d), float("nan"), device=dev, dtype=torch.bfloat16)
    lse = torch.zeros(B, H, T, device=dev)
    Tp = (T + 7) // 8 * 8
    o.relpos_flash_fwd(qkv.to(dev), 3 * d, pos.to(dev), d, u.to(dev), v.to(dev), lens.to(dev), ctx, d, lse, B, H, T, dk, Tp,
                       1.0 / math.sqrt(dk), ctx_lo=ctx_lo)
    torch.cuda.synchronize()
    assert torch.isfinite(ctx.float()).all()
    assert rel_err(ctx, ctx_ref) < 5e-3, rel_err(ctx, ctx_ref)  # (measured 1.3-1.9e-3: bf16 probabilities)
    # the rounding residual: |lo| <= half an ulp of ctx, and it is what bf16 dropped of the kernel's own f32 accumulators --
    # checked through the one place it is used: delta = sum dO * (ctx + lo) against the same product with ctx alone has to
    # move by the rounding of ctx (~2^-9 relative per element), not more
    lo = ctx_lo.float()
    assert torch.isfinite(lo).all() and (lo.abs() <= ctx.float().abs() * 2.0 ** -8 + 1e-30).all()
    assert lo.abs().max() > 0
    for b in range(B):
        n = int(lens[b])
        assert (lse[b, :, :n].cpu() - lse_ref[b, :, :n]).abs().max() < 2e-2
        assert torch.all(ctx.view(B, T, d)[b, n:] == 0)


def _attn_ref_grads(qkv, pos, u, v, lens, dO, B, H, T, dk):
    """autograd of the fp32 reference wrt qu, qv, k, v, p (inputs bf16-rounded like the kernels see them)"""
    d = H * dk
    q = qkv[:, :d].float().view(B, T, H, dk)
    qu = bf(q + u.view(H, dk)).float().requires_grad_(True)
    qv = bf(q + v.view(H, dk)).float().requires_grad_(True)
    k = qkv[:, d:2 * d].float().view(B, T, H, dk).clone().requires_grad_(True)
    vv = qkv[:, 2 * d:].float().view(B, T, H, dk).clone().requires_grad_(True)
    p = pos.float().view(2 * T - 1, H, dk).clone().requires_grad_(True)
    ac = qu.transpose(1, 2) @ k.transpose(1, 2).transpose(-1, -2)
    bdf = qv.transpose(1, 2) @ p.transpose(0, 1).transpose(-1, -2).unsqueeze(0)
    ii = torch.arange(T)[:, None]; jj = torch.arange(T)[None]
    sc = (ac + bdf[:, :, ii, T - 1 + jj - ii]) / math.sqrt(dk)
    valid = torch.arange(T)[None] < lens[:, None]
    masked = ~(valid[:, :, None] & valid[:, None, :])[:, None]
    attn = torch.softmax(sc.masked_fill(masked, -10000.0), -1).masked_fill(masked, 0.0)
    ctx = (attn @ vv.transpose(1, 2)).transpose(1, 2).reshape(B * T, d)
    ctx.backward(dO.float())
    return dict(ctx=ctx.detach(), dqu=qu.grad.reshape(B * T, d), dqv=qv.grad.reshape(B * T, d), dk=k.grad.reshape(B * T, d),
                dv=vv.grad.reshape(B * T, d), dp=p.grad.reshape(2 * T - 1, d))


@pytest.mark.parametrize("dk", [64, 128])
@pytest.mark.parametrize("T", [45, 160, 501])
def test_relpos_flash_attention_bwd(T, dk):
    o = ops()
    B, H = 3, 2
    d = H * dk
    g = torch.Generator().manual_seed(32)
    qkv = bf(torch.randn(B * T, 3 * d, generator=g) * 0.7)
    pos = bf(torch.randn(2 * T - 1, d, generator=g) * 0.7)
    u = torch.randn(d, generator=g) * 0.3; v = torch.randn(d, generator=g) * 0.3
    lens = torch.tensor([T, max(1, T // 2 + 3), 1])
    dO = bf(torch.randn(B * T, d, generator=g))
    ref = _attn_ref_grads(qkv, pos, u, v, lens, dO, B, H, T, dk)
    Tp = (T + 7) // 8 * 8
    scale = 1.0 / math.sqrt(dk)
    qkv_d, pos_d, lens_d, dO_d = qkv.to(dev), pos.to(dev), lens.to(dev), dO.to(dev)
    ctx = torch.empty(B * T, d, device=dev, dtype=torch.bfloat16)
    lse = torch.zeros(B, H, T, device=dev)
    o.relpos_flash_fwd(qkv_d, 3 * d, pos_d, d, u.to(dev), v.to(dev), lens_d, ctx, d, lse, B, H, T, dk, Tp, scale)
    qu = torch.empty(B * T, d, device=dev, dtype=torch.bfloat16); qv = torch.empty_like(qu)
    o.qbias(qkv_d, 3 * d, u.to(dev), v.to(dev), qu, qv, B * T, d)
    delta = torch.zeros(B, H, T, device=dev)
    o.attn_delta(dO_d, ctx, delta, B, H, T, d)
    dqu = torch.full((B * T, d), float("nan"), device=dev, dtype=torch.bfloat16); dqv = torch.full_like(dqu, float("nan"))
    dS = o.relpos_ds_buffer(B, H, T, dev, fill=float("nan"))  # every block the gradient kernel reads must have been written
    o.relpos_flash_bwd_dq(qu, qv, qkv_d, 3 * d, pos_d, d, lens_d, dO_d, lse, delta, dqu, dqv, B, H, T, dk, scale, ds_out=dS)
    torch.cuda.synchronize()
    assert rel_err(dqu, ref["dqu"]) < 8e-3, rel_err(dqu, ref["dqu"])  # (measured 2.9-3.7e-3)
    assert rel_err(dqv, ref["dqv"]) < 8e-3, rel_err(dqv, ref["dqv"])  # (measured 2.9-3.7e-3)
    # the one-pass prologue (q + u, q + v, delta) = the two kernels above, bit for bit
    qu2 = torch.full_like(qu, float("nan")); qv2 = torch.full_like(qv, float("nan")); delta2 = torch.full_like(delta, float("nan"))
    o.attn_bwd_prep(dO_d, ctx, delta2, qkv_d, 3 * d, u.to(dev), v.to(dev), qu2, qv2, B, H, T, d)
    torch.cuda.synchronize()
    assert torch.equal(qu2, qu) and torch.equal(qv2, qv) and torch.equal(delta2, delta)
    # fused outputs of the dQ kernel: dq = dqu + dqv straight into the q third of the [M, 3d] gradient (the other two thirds
    # untouched), pos_bias_u / pos_bias_v gradients = column sums of dQu | dQv added to what the buffer held
    dq3 = torch.full((B * T, 3 * d), 7.0, device=dev, dtype=torch.bfloat16)
    bg = torch.ones(2 * d, device=dev)
    o.relpos_flash_bwd_dq(qu, qv, qkv_d, 3 * d, pos_d, d, lens_d, dO_d, lse, delta, None, None, B, H, T, dk, scale,
                          dq_out=dq3, ld_dq=3 * d, bias_grads=bg)
    torch.cuda.synchronize()
    assert torch.all(dq3[:, d:] == 7.0)
    assert rel_err(dq3[:, :d], dqu.float() + dqv.float()) < 6e-3  # (one bf16 rounding of the f32 sum instead of three)
    assert rel_err(dq3[:, :d], ref["dqu"] + ref["dqv"]) < 8e-3
    assert rel_err(bg[:d] - 1.0, ref["dqu"].sum(0)) < 8e-3 and rel_err(bg[d:] - 1.0, ref["dqv"].sum(0)) < 8e-3
    if hasattr(o, "relpos_flash_bwd_dkv"):
        dqkv = torch.full((B * T, 3 * d), float("nan"), device=dev, dtype=torch.bfloat16)
        o.relpos_flash_bwd_dkv(qu, qv, qkv_d, 3 * d, pos_d, d, lens_d, dO_d, lse, delta, dqkv, 3 * d, B, H, T, dk, Tp, scale)
        torch.cuda.synchronize()
        assert rel_err(dqkv[:, d:2 * d], ref["dk"]) < 8e-3, rel_err(dqkv[:, d:2 * d], ref["dk"])
        assert rel_err(dqkv[:, 2 * d:], ref["dv"]) < 8e-3, rel_err(dqkv[:, 2 * d:], ref["dv"])
    if hasattr(o, "relpos_flash_bwd_dpos"):
        dp = torch.zeros(2 * T - 1, d, device=dev)
        dp_c = torch.full((2 * T - 1, d), float("nan"), device=dev, dtype=torch.bfloat16)
        o.relpos_flash_bwd_dpos(qv, dS, lens_d, dp, B, H, T, dk, dpos_cast=dp_c)
        torch.cuda.synchronize()
        assert rel_err(dp, ref["dp"]) < 8e-3, rel_err(dp, ref["dp"])
        # the reduction stage also writes the GEMM-operand copy: every element, the bf16 rounding of the f32 result
        assert torch.equal(dp_c, dp.to(torch.bfloat16))


@pytest.mark.parametrize("dk", [64, 128])
def test_relpos_flash_attention_dropout_consistency(dk):
    """With one-hot V (T <= d_k) the context IS the dropped probability matrix, so the forward mask can be read out and
    every backward kernel checked against the same mask (masks are regenerated, never stored)."""
    o = ops()
    B, H, T = 2, 2, 45
    d = H * dk
    g = torch.Generator().manual_seed(33)
    qkv = torch.randn(B * T, 3 * d, generator=g) * 0.5
    eye = torch.zeros(T, dk); eye[torch.arange(T), torch.arange(T)] = 1.0
    qkv[:, 2 * d:] = eye.repeat(B, H)  # V[b, j, h, :] = e_j
    qkv = bf(qkv)
    pos = bf(torch.randn(2 * T - 1, d, generator=g) * 0.5)
    u = torch.randn(d, generator=g) * 0.3; v = torch.randn(d, generator=g) * 0.3
    lens = torch.tensor([T, 30])
    _, _, attn = _attn_ref(qkv, pos, u, v, lens, B, H, T, dk)          # [B,H,T,T] undropped probabilities
    pdrop = 0.25
    drop = o.Dropout(pdrop, seed=5, site=9)
    Tp = (T + 7) // 8 * 8
    scale = 1.0 / math.sqrt(dk)
    qkv_d, pos_d, lens_d = qkv.to(dev), pos.to(dev), lens.to(dev)
    ctx = torch.empty(B * T, d, device=dev, dtype=torch.bfloat16); lse = torch.zeros(B, H, T, device=dev)
    o.relpos_flash_fwd(qkv_d, 3 * d, pos_d, d, u.to(dev), v.to(dev), lens_d, ctx, d, lse, B, H, T, dk, Tp, scale, drop)
    torch.cuda.synchronize()
    Pd = ctx.float().cpu().view(B, T, H, dk)[..., :T].permute(0, 2, 1, 3)  # [B,H,T(i),T(j)]
    keep = Pd > 0.5 * attn / (1 - pdrop) * (attn > 1e-3)
    big = attn > 1e-3
    assert torch.all(((Pd - attn / (1 - pdrop)).abs() < 0.02 + 0.02 * attn)[keep & big])
    assert torch.all(Pd[~keep & big].abs() < 1e-3)
    rate = keep[big].float().mean().item()
    assert abs(rate - (1 - pdrop)) < 0.04, rate
    # backward with the SAME mask: reference gradients from the extracted mask
    mask = torch.where(keep | ~big, torch.tensor(1.0 / (1 - pdrop)), torch.tensor(0.0))
    dO = bf(torch.randn(B * T, d, generator=g))
    q = qkv[:, :d].float().view(B, T, H, dk)
    qu = bf(q + u.view(H, dk)).float().requires_grad_(True); qv = bf(q + v.view(H, dk)).float().requires_grad_(True)
    k = qkv[:, d:2 * d].float().view(B, T, H, dk).clone().requires_grad_(True)
    vv = qkv[:, 2 * d:].float().view(B, T, H, dk).clone().requires_grad_(True)
    pp = pos.float().view(2 * T - 1, H, dk).clone().requires_grad_(True)
    ac = qu.transpose(1, 2) @ k.transpose(1, 2).transpose(-1, -2)
    bdf = qv.transpose(1, 2) @ pp.transpose(0, 1).transpose(-1, -2).unsqueeze(0)
    ii = torch.arange(T)[:, None]; jj = torch.arange(T)[None]
    sc = (ac + bdf[:, :, ii, T - 1 + jj - ii]) * scale
    valid = torch.arange(T)[None] < lens[:, None]
    masked = ~(valid[:, :, None] & valid[:, None, :])[:, None]
    a2 = torch.softmax(sc.masked_fill(masked, -10000.0), -1).masked_fill(masked, 0.0) * mask
    c2 = (a2 @ vv.transpose(1, 2)).transpose(1, 2).reshape(B * T, d)
    c2.backward(dO.float())
    quq = torch.empty(B * T, d, device=dev, dtype=torch.bfloat16); qvq = torch.empty_like(quq)
    o.qbias(qkv_d, 3 * d, u.to(dev), v.to(dev), quq, qvq, B * T, d)
    delta = torch.zeros(B, H, T, device=dev)
    o.attn_delta(dO.to(dev), ctx, delta, B, H, T, d)
    dqu = torch.empty_like(quq); dqv = torch.empty_like(quq)
    dS = o.relpos_ds_buffer(B, H, T, dev, fill=float("nan"))
    o.relpos_flash_bwd_dq(quq, qvq, qkv_d, 3 * d, pos_d, d, lens_d, dO.to(dev), lse, delta, dqu, dqv, B, H, T, dk, scale, drop,
                          ds_out=dS)
    dqkv = torch.zeros(B * T, 3 * d, device=dev, dtype=torch.bfloat16)
    o.relpos_flash_bwd_dkv(quq, qvq, qkv_d, 3 * d, pos_d, d, lens_d, dO.to(dev), lse, delta, dqkv, 3 * d, B, H, T, dk, Tp, scale, drop)
    dp = torch.zeros(2 * T - 1, d, device=dev)
    o.relpos_flash_bwd_dpos(qvq, dS, lens_d, dp, B, H, T, dk)
    torch.cuda.synchronize()
    assert rel_err(dqu, qu.grad.reshape(B * T, d)) < 4e-2
    assert rel_err(dqv, qv.grad.reshape(B * T, d)) < 4e-2
    assert rel_err(dqkv[:, d:2 * d], k.grad.reshape(B * T, d)) < 4e-2
    assert rel_err(dqkv[:, 2 * d:], vv.grad.reshape(B * T, d)) < 4e-2
    assert rel_err(dp, pp.grad.reshape(2 * T - 1, d)) < 4e-2


@pytest.mark.parametrize("geom", [(3, 150, 80), (2, 501, 512), (2, 77, 136)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_dwconv_bn_swish(dtype, geom):
    """(bf16, even d: the streaming kernels of csrc/convmod.hip -- a partly filled channel group, the headline geometry, a
    ragged last group and time tile; f32: the LDS-tile kernels)"""
    o = ops()
    from nemo_amd._lib import lib
    prev = lib.mi355x_dwconv_config(2)  # streaming kernels in both directions (the default runs the tile kernels both ways)
    try:
        _dwconv_bn_swish(o, dtype, geom)
    finally:
        lib.mi355x_dwconv_config(prev)
    _dwconv_bn_swish(o, dtype, geom)


def _dwconv_bn_swish(o, dtype, geom):
    (Bn, T, d), k = geom, 31
    g = torch.Generator().manual_seed(8)
    x = torch.randn(Bn, T, d, generator=g).to(dtype)
    w = torch.randn(d, 1, k, generator=g) * 0.2
    bias = torch.randn(d, generator=g)
    gamma = torch.rand(d, generator=g) + 0.5; beta = torch.randn(d, generator=g) * 0.1
    xr = x.float().requires_grad_(True); wr = w.clone().requires_grad_(True); br = bias.clone().requires_grad_(True)
    gr = gamma.clone().requires_grad_(True); ber = beta.clone().requires_grad_(True)
    c_ref = F.conv1d(F.pad(xr.transpose(1, 2), (15, 15)), wr, br, groups=d)  # [B,d,T]
    c_q = c_ref.to(dtype).float() if dtype == torch.bfloat16 else c_ref
    mean = c_q.mean((0, 2)); var = c_q.var((0, 2), unbiased=False)
    z = (c_ref - mean[None, :, None]) * torch.rsqrt(var[None, :, None] + 1e-5) * gr[None, :, None] + ber[None, :, None]
    y_ref = (z * torch.sigmoid(z)).transpose(1, 2)
    dy = torch.randn(Bn, T, d, generator=g).to(dtype)
    y_ref.backward(dy.float())
    # device
    xd = x.to(dev); c = torch.empty(Bn, T, d, device=dev, dtype=dtype)
    stats = torch.zeros(2, d, device=dev, dtype=torch.float64)
    o.dwconv_fwd(xd, w.to(dev), bias.to(dev), c, stats, Bn, T, d, k)
    tol = 2e-5 if dtype == torch.float32 else 1e-2
    assert rel_err(c, c_ref.transpose(1, 2)) < tol
    mu = torch.empty(d, device=dev); rs = torch.empty(d, device=dev)
    rm = torch.zeros(d, device=dev); rv = torch.ones(d, device=dev)
    n = Bn * T
    o.bn_finalize(stats, n, mu, rs, rm, rv, 0.1, 1e-5, d)
    assert rel_err(mu, mean) < 1e-3 and rel_err(rs, torch.rsqrt(var + 1e-5)) < 1e-3
    assert rel_err(rm, 0.1 * mean) < 1e-3 and rel_err(rv, 0.9 + 0.1 * var * n / (n - 1)) < 1e-3
    y = torch.empty(Bn, T, d, device=dev, dtype=dtype)
    o.bn_swish_fwd(c, mu, rs, gamma.to(dev), beta.to(dev), y, n, d)
    assert rel_err(y, y_ref) < 2 * tol
    # the one-launch training forward (statistics finalised in the kernel) = the two launches above, bit for bit
    mu2 = torch.empty(d, device=dev); rs2 = torch.empty(d, device=dev)
    rm2 = torch.zeros(d, device=dev); rv2 = torch.ones(d, device=dev)
    y2 = torch.empty(Bn, T, d, device=dev, dtype=dtype)
    o.bn_stats_swish_fwd(c, stats, n, gamma.to(dev), beta.to(dev), y2, mu2, rs2, rm2, rv2, 0.1, 1e-5, n, d)
    cnt = torch.tensor([float(n)], device=dev, dtype=torch.float64)
    y3 = torch.empty_like(y2)
    o.bn_stats_swish_fwd(c, stats, cnt, gamma.to(dev), beta.to(dev), y3, mu2.clone(), rs2.clone(), None, None, 0.1, 1e-5, n, d)
    torch.cuda.synchronize()
    assert torch.equal(y2, y) and torch.equal(y3, y) and torch.equal(mu2, mu) and torch.equal(rs2, rs)
    assert torch.equal(rm2, rm) and torch.equal(rv2, rv)
    sums = torch.zeros(2, d, device=dev, dtype=torch.float64)
    dgam2 = torch.zeros(d, device=dev); dbet2 = torch.zeros(d, device=dev)
    o.bn_swish_bwd_reduce(dy.to(dev), c, mu, rs, gamma.to(dev), beta.to(dev), sums, n, d, dgamma=dgam2, dbeta=dbet2)
    dc = torch.empty(Bn, T, d, device=dev, dtype=dtype)
    o.bn_swish_bwd_apply(dy.to(dev), c, mu, rs, gamma.to(dev), beta.to(dev), sums, n, True, dc, n, d)
    dgam = torch.zeros(d, device=dev); dbet = torch.zeros(d, device=dev)
    o.bn_param_grad(sums, dgam, dbet, d)
    assert rel_err(dgam, gr.grad) < 5 * tol and rel_err(dbet, ber.grad) < 5 * tol
    # ... and the parameter gradients accumulated by the reduction's second stage
    assert rel_err(dgam2, dgam) < 1e-5 and rel_err(dbet2, dbet) < 1e-5 + 1e-5 / max(dbet.abs().max().item(), 1e-6)
    dx = torch.empty(Bn, T, d, device=dev, dtype=dtype)
    dw = torch.zeros(d, 1, k, device=dev); dbias = torch.zeros(d, device=dev)
    o.dwconv_bwd(dc, xd, w.to(dev), dx, dw, dbias, Bn, T, d, k)
    torch.cuda.synchronize()
    assert rel_err(dx, xr.grad) < 5 * tol, rel_err(dx, xr.grad)
    assert rel_err(dw, wr.grad) < 5 * tol, rel_err(dw, wr.grad)
    # dbias is analytically ~0 under batch-stat BN; compare absolutely against the scale of dw
    assert (dbias.cpu() - br.grad).abs().max() < 5 * tol * wr.grad.abs().max() + 1e-3
    # BatchNorm + Swish backward fused into the depthwise backward (mi355x_dwconv_bwd_bnswish) = the two launches above: the
    # intermediate is rounded the same way, only the weight-gradient summation order (f32 partial slabs) is shared too
    from nemo_amd._lib import lib
    if lib.mi355x_dwconv_config(-1) == 0 or dtype == torch.float32:   # (the fused form extends the LDS-tile kernel)
        for count in (n, torch.tensor([float(n)], device=dev, dtype=torch.float64)):
            dx2 = torch.empty(Bn, T, d, device=dev, dtype=dtype)
            dw2 = torch.zeros(d, 1, k, device=dev); dbias2 = torch.zeros(d, device=dev)
            o.dwconv_bwd_bnswish(dy.to(dev), c, mu, rs, gamma.to(dev), beta.to(dev), sums, count, True, xd, w.to(dev), dx2, dw2,
                                 dbias2, Bn, T, d, k)
            torch.cuda.synchronize()
            assert rel_err(dx2, dx) < 1e-5, rel_err(dx2, dx)
            assert rel_err(dw2, dw) < 1e-5 and (dbias2 - dbias).abs().max() < 1e-5 * max(1.0, dw.abs().max().item())
            assert rel_err(dx2, xr.grad) < 5 * tol and rel_err(dw2, wr.grad) < 5 * tol
        # ... and with the GLU backward in its write-out (padded grid with ragged lengths, then packed rows) = mi355x_glu_bwd of dx
        lens = torch.tensor([max(1, T - 7 * i) for i in range(Bn)], dtype=torch.int64)
        pw1 = torch.randn(n, 2 * d, generator=g).to(dtype).to(dev)
        want = torch.empty(n, 2 * d, device=dev, dtype=dtype)
        o.glu_bwd(pw1, dx, want, lens.to(dev), T, n, d)
        got = torch.full((n, 2 * d), float("nan"), device=dev, dtype=dtype)
        dw3 = torch.zeros(d, 1, k, device=dev); dbias3 = torch.zeros(d, device=dev)
        o.dwconv_bwd_bnswish(dy.to(dev), c, mu, rs, gamma.to(dev), beta.to(dev), sums, n, True, xd, w.to(dev), None, dw3, dbias3,
                             Bn, T, d, k, glu_in=pw1, glu_din=got, glu_len=lens.to(dev))
        torch.cuda.synchronize()
        assert torch.isfinite(got.float()).all() and rel_err(got, want) < 1e-5, rel_err(got, want)
        assert rel_err(dw3, dw) < 1e-5
        # forward twin: GLU + pad mask in the depthwise forward's tile staging = mi355x_glu_fwd then mi355x_dwconv_fwd
        g_ref = torch.empty(Bn, T, d, device=dev, dtype=dtype); c_two = torch.empty(Bn, T, d, device=dev, dtype=dtype)
        st_two = torch.zeros(2, d, device=dev, dtype=torch.float64)
        o.glu_fwd(pw1, g_ref.view(n, d), lens.to(dev), T, n, d)
        o.dwconv_fwd(g_ref, w.to(dev), bias.to(dev), c_two, st_two, Bn, T, d, k)
        g_one = torch.full((Bn, T, d), float("nan"), device=dev, dtype=dtype); c_one = torch.empty(Bn, T, d, device=dev, dtype=dtype)
        st_one = torch.zeros(2, d, device=dev, dtype=torch.float64)
        o.dwconv_fwd_glu(pw1, lens.to(dev), None, g_one, w.to(dev), bias.to(dev), c_one, st_one, Bn, T, d, k)
        torch.cuda.synchronize()
        assert torch.equal(g_one, g_ref) and rel_err(c_one, c_two) < 1e-6 and rel_err(st_one, st_two) < 1e-6
        cu = torch.zeros(Bn + 1, dtype=torch.int64); cu[1:] = torch.cumsum(lens, 0)
        Mp = int(cu[-1])
        rows = torch.cat([torch.arange(int(lens[b])) + b * T for b in range(Bn)]).to(dev)
        pw1p = pw1[rows].contiguous()
        gotp = torch.full((Mp, 2 * d), float("nan"), device=dev, dtype=dtype)
        o.dwconv_bwd_bnswish(dy.to(dev), c, mu, rs, gamma.to(dev), beta.to(dev), sums, n, True, xd, w.to(dev), None,
                             torch.zeros(d, 1, k, device=dev), torch.zeros(d, device=dev), Bn, T, d, k, glu_in=pw1p, glu_din=gotp,
                             glu_len=lens.to(dev), glu_cu=cu.to(dev))
        torch.cuda.synchronize()
        assert torch.isfinite(gotp.float()).all() and rel_err(gotp, want[rows]) < 1e-5, rel_err(gotp, want[rows])
        g_pk = torch.full((Bn, T, d), float("nan"), device=dev, dtype=dtype); c_pk = torch.empty(Bn, T, d, device=dev, dtype=dtype)
        o.dwconv_fwd_glu(pw1p, lens.to(dev), cu.to(dev), g_pk, w.to(dev), bias.to(dev), c_pk, None, Bn, T, d, k)
        torch.cuda.synchronize()
        assert torch.equal(g_pk, g_ref) and rel_err(c_pk, c_two) < 1e-6
        # Squeezeformer's conv module: Swish + pad mask instead of the GLU (act = 1) = mi355x_swish_mask_fwd / _bwd around the core
        sw_in = torch.randn(n, d, generator=g).to(dtype).to(dev)
        s_ref = torch.empty(Bn, T, d, device=dev, dtype=dtype); c_sw2 = torch.empty(Bn, T, d, device=dev, dtype=dtype)
        o.swish_mask_fwd(sw_in, s_ref.view(n, d), lens.to(dev), T, n, d)
        o.dwconv_fwd(s_ref, w.to(dev), bias.to(dev), c_sw2, None, Bn, T, d, k)
        s_one = torch.full((Bn, T, d), float("nan"), device=dev, dtype=dtype); c_sw1 = torch.empty(Bn, T, d, device=dev, dtype=dtype)
        o.dwconv_fwd_glu(sw_in, lens.to(dev), None, s_one, w.to(dev), bias.to(dev), c_sw1, None, Bn, T, d, k, act=1)
        torch.cuda.synchronize()
        assert torch.equal(s_one, s_ref) and rel_err(c_sw1, c_sw2) < 1e-6
        want_s = torch.empty(n, d, device=dev, dtype=dtype)
        o.swish_mask_bwd(sw_in, dx.view(n, d), want_s, lens.to(dev), T, n, d)
        got_s = torch.full((n, d), float("nan"), device=dev, dtype=dtype)
        o.dwconv_bwd_bnswish(dy.to(dev), c, mu, rs, gamma.to(dev), beta.to(dev), sums, n, True, xd, w.to(dev), None,
                             torch.zeros(d, 1, k, device=dev), torch.zeros(d, device=dev), Bn, T, d, k, glu_in=sw_in, glu_din=got_s,
                             glu_len=lens.to(dev), glu_act=1)
        torch.cuda.synchronize()
        assert torch.isfinite(got_s.float()).all() and rel_err(got_s, want_s) < 1e-5, rel_err(got_s, want_s)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_subsampling_pieces(dtype):
    o = ops()
    Bn, Fq, T, C_ = 2, 80, 61, 32
    g = torch.Generator().manual_seed(9)
    mel = torch.randn(Bn, Fq, T, generator=g)
    len0 = torch.tensor([61, 40]); len1 = R.conv_out_len(len0, 1)
    w1 = torch.randn(C_, 1, 3, 3, generator=g) * 0.3; b1 = torch.randn(C_, generator=g) * 0.1
    T1, F1 = (T - 1) // 2 + 1, (Fq - 1) // 2 + 1
    x = mel.transpose(1, 2).unsqueeze(1)
    tm0 = (torch.arange(T)[None] < len0[:, None]).float().view(Bn, 1, T, 1)
    tm1 = (torch.arange(T1)[None] < len1[:, None]).float().view(Bn, 1, T1, 1)
    w1r = w1.clone().requires_grad_(True); b1r = b1.clone().requires_grad_(True)
    o1_ref = torch.relu(F.conv2d(x * tm0, w1r, b1r, stride=2, padding=1)) * tm1  # [B,C,T1,F1]
    out1 = torch.empty(Bn, T1, F1, C_, device=dev, dtype=dtype)
    o.conv1_fwd(mel.to(dev), w1.to(dev), b1.to(dev), out1, len0.to(dev), len1.to(dev), C_)
    tol = 1e-5 if dtype == torch.float32 else 1e-2
    assert rel_err(out1, o1_ref.permute(0, 2, 3, 1)) < tol
    # conv1 backward (params): feed a gated upstream grad
    dout1 = (torch.randn(Bn, T1, F1, C_, generator=g) * (o1_ref.permute(0, 2, 3, 1) > 0)).to(dtype)
    o1_ref.backward(dout1.float().permute(0, 3, 1, 2))
    dw = torch.zeros(C_, 1, 3, 3, device=dev); db = torch.zeros(C_, device=dev)
    o.conv1_bwd(dout1.to(dev), mel.to(dev), len0.to(dev), dw, db, C_)
    assert rel_err(dw, w1r.grad) < 5 * tol and rel_err(db, b1r.grad) < 5 * tol
    # im2col + GEMM == conv2d
    T2, F2 = (T1 - 1) // 2 + 1, (F1 - 1) // 2 + 1
    col = torch.empty(Bn * T2 * F2, 9 * C_, device=dev, dtype=dtype)
    o.im2col(out1, col, Bn, T1, F1, C_)
    w2 = torch.randn(C_, C_, 3, 3, generator=g) * 0.1
    w2p = w2.permute(0, 2, 3, 1).reshape(C_, 9 * C_).to(dtype)
    out2 = torch.empty(Bn * T2 * F2, C_, device=dev, dtype=torch.float32)
    o.gemm(col, w2p.to(dev), out2, Bn * T2 * F2, C_, 9 * C_, 9 * C_, 9 * C_, C_)
    o1q = out1.float().cpu().permute(0, 3, 1, 2)
    ref2 = F.conv2d(o1q, w2.to(dtype).float(), stride=2, padding=1).permute(0, 2, 3, 1).reshape(-1, C_)
    assert rel_err(out2, ref2) < (1e-4 if dtype == torch.float32 else 2e-3)
    # col2im with relu gate == conv2d input-gradient * (act > 0)
    dcol = torch.randn(Bn * T2 * F2, 9 * C_, generator=g).to(dtype)
    din = torch.empty(Bn, T1, F1, C_, device=dev, dtype=dtype)
    o.col2im_relu(dcol.to(dev), out1, din, Bn, T1, F1, C_)
    cols = dcol.float().view(Bn, T2 * F2, 9, C_).permute(0, 3, 2, 1).reshape(Bn, C_ * 9, T2 * F2)  # fold wants (C*kh*kw)
    fold = F.fold(cols, (T1, F1), kernel_size=3, stride=2, padding=1)  # [B,C,T1,F1]
    ref_din = fold.permute(0, 2, 3, 1) * (out1.float().cpu() > 0)
    torch.cuda.synchronize()
    assert rel_err(din, ref_din) < 2 * tol


@pytest.mark.parametrize("T1,F1,pad", [(61, 40, 1), (38, 17, 1), (61, 40, 2), (38, 17, 2), (37, 18, 2)])
def test_conv2_implicit_gemm_forward_and_dgrad(T1, F1, pad):
    """Conv2d(C->C, 3x3, stride 2, pad 1) of ConvSubsampling as an implicit GEMM (gathered A operand, no im2col) and its
    input gradient as four parity-class implicit GEMMs with scattered output rows + ReLU gate (no col2im), against
    torch conv2d / autograd on the same bf16-rounded operands.  pad = 2: CausalConv2D (causal_downsampling: F.pad (2, 1) on both
    axes, no symmetric padding) -- other tap offsets, the parity classes swap, taps that reach past the last output."""
    o = ops()
    Bn, C_ = 3, 256  # (the gather lives in the LDS-DMA GEMM structures: M >= 192, N >= 96)
    g = torch.Generator().manual_seed(17)
    T2, F2 = (T1 + pad - 2) // 2 + 1, (F1 + pad - 2) // 2 + 1
    conv = lambda t, w, b: F.conv2d(F.pad(t, (pad, 1, pad, 1)), w, b, stride=2)
    x = bf(torch.relu(torch.randn(Bn, T1, F1, C_, generator=g)))               # post-ReLU activations: ~half are zero
    w2 = bf(torch.randn(C_, C_, 3, 3, generator=g) * 0.1)
    b2 = torch.randn(C_, generator=g) * 0.1
    xr = x.float().permute(0, 3, 1, 2).requires_grad_(True)
    ref = conv(xr, w2.float(), b2)                                              # [B,C,T2,F2]
    assert tuple(ref.shape[2:]) == (T2, F2)
    # forward: gather taps (kh-1, kw-1), K order (kh, kw, ci) = the packed weight image [co][(kh,kw,ci)]
    w2p = w2.permute(0, 2, 3, 1).reshape(C_, 9 * C_).contiguous()
    M2 = Bn * T2 * F2
    out2 = torch.empty(M2, C_, device=dev, dtype=torch.float32)
    taps = [(kh - pad, kw - pad) for kh in range(3) for kw in range(3)]
    o.gemm(x.to(dev), w2p.to(dev), out2, M2, C_, 9 * C_, C_, 9 * C_, C_, bias=b2.to(dev),
           gather=dict(nI=T2, nJ=F2, SI=T1, SJ=F1, C=C_, si=2, sj=2, taps=taps))
    assert rel_err(out2, ref.permute(0, 2, 3, 1).reshape(M2, C_)) < 2e-3
    # dgrad: dx[b,t1,f1,ci] = (x > 0) * sum_{taps of the parity class} dy[b, i+dt, j+df, :] . W[:, ci, kh, kw]
    dy = bf(torch.randn(Bn, T2, F2, C_, generator=g))
    ref.backward(dy.float().permute(0, 3, 1, 2))
    ref_dx = xr.grad.permute(0, 2, 3, 1) * (x.float() > 0)
    dx = torch.full((Bn, T1, F1, C_), float("nan"), device=dev, dtype=torch.bfloat16)
    xd, dyd = x.to(dev), dy.to(dev)
    for pt in (0, 1):
        for pf in (0, 1):
            nI, nJ = (T1 - pt + 1) // 2, (F1 - pf + 1) // 2
            khs = [1] if (pt + pad) % 2 else [0, 2]       # t1 = 2 t2 - pad + kh
            kws = [1] if (pf + pad) % 2 else [0, 2]
            slots = [(kh, kw) for kh in khs for kw in kws]
            taps_d = [((pt + pad - kh) // 2, (pf + pad - kw) // 2) for kh, kw in slots]
            wimg = torch.cat([w2[:, :, kh, kw].t() for kh, kw in slots], dim=1).contiguous()   # [ci][(slot, co)]
            K = len(slots) * C_
            o.gemm(dyd, wimg.to(dev), dx, Bn * nI * nJ, C_, K, C_, K, C_, epi=o.EPI_MUL_POS, aux_in=xd, ldaux=C_,
                   gather=dict(nI=nI, nJ=nJ, SI=T2, SJ=F2, C=C_, si=1, sj=1, taps=taps_d),
                   rowmap=dict(nI=nI, nJ=nJ, OI=T1, OJ=F1, si=2, sj=2, oi=pt, oj=pf))
    torch.cuda.synchronize()
    assert not torch.isnan(dx.float()).any()
    assert rel_err(dx, ref_dx) < 1e-2
    # wgrad: dW[co, ci, kh, kw] += sum_m dy[m, co] * x[b, 2*t2+kh-1, 2*f2+kw-1, ci]  (B operand gathered, batch = tap,
    # written straight into the reference's [co, ci, 3, 3] layout: column stride 9, batch offset 1)
    w2r = w2.float().clone().requires_grad_(True)
    conv(x.float().permute(0, 3, 1, 2), w2r, None).backward(dy.float().permute(0, 3, 1, 2))
    dW = torch.ones(C_, C_, 3, 3, device=dev)
    for sk in (1, 3):
        dW.fill_(1.0)
        o.gemm(dyd, xd, dW, C_, C_, M2, C_, C_, 9 * C_, transA=True, transB=True, atomic=True, splitk=sk, batch=9, nb0=9,
               sC=(1, 0), c_col_stride=9, c_dtype=o.F32,
               gather=dict(operand=1, nI=T2, nJ=F2, SI=T1, SJ=F1, C=C_, si=2, sj=2, taps=taps))
        torch.cuda.synchronize()
        assert rel_err(dW - 1.0, w2r.grad) < 2e-3, sk


def test_conv2_dgrad_skips_row_tiles_beyond_the_utterances():
    """The four parity-class input-gradient GEMMs with `row_len` (round 5): 256-row tiles whose rows all lie beyond their utterance
    are zero-filled through the row map without a K loop.  Same launches with and without the hint on a ragged batch that is large
    enough for the 256 x 256 structure (>= 224 workgroups): identical results, every row written."""
    o = ops()
    Bn, C_, T1, F1 = 5, 256, 1200, 40
    T2, F2 = (T1 - 1) // 2 + 1, (F1 - 1) // 2 + 1
    g = torch.Generator().manual_seed(23)
    len1 = torch.tensor([1200, 700, 301, 64, 1])
    len2 = (len1 + 1) // 2
    x = torch.relu(torch.randn(Bn, T1, F1, C_, generator=g)).to(torch.bfloat16)
    x = x * (torch.arange(T1)[None, :, None, None] < len1[:, None, None, None])          # conv1's output: zero beyond len1
    w2 = (torch.randn(C_, C_, 3, 3, generator=g) * 0.1).to(torch.bfloat16)
    dy = torch.randn(Bn, T2, F2, C_, generator=g).to(torch.bfloat16)
    xd, dyd, l2 = x.to(dev), dy.to(dev), len2.to(dev)
    outs = []
    for hint in (False, True):
        dx = torch.full((Bn, T1, F1, C_), float("nan"), device=dev, dtype=torch.bfloat16)
        for pt in (0, 1):
            for pf in (0, 1):
                nI, nJ = (T1 - pt + 1) // 2, (F1 - pf + 1) // 2
                slots = [(kh, kw) for kh in ([1] if pt == 0 else [0, 2]) for kw in ([1] if pf == 0 else [0, 2])]
                taps_d = [(1 if kh == 0 else 0, 1 if kw == 0 else 0) for kh, kw in slots]
                wimg = torch.cat([w2[:, :, kh, kw].t() for kh, kw in slots], dim=1).contiguous().to(dev)
                K = len(slots) * C_
                kw_ = dict(row_len=l2, rows_per_b=nI * nJ, rows_inner=nJ) if hint else {}
                o.gemm(dyd, wimg, dx, Bn * nI * nJ, C_, K, C_, K, C_, epi=o.EPI_MUL_POS, aux_in=xd, ldaux=C_,
                       gather=dict(nI=nI, nJ=nJ, SI=T2, SJ=F2, C=C_, si=1, sj=1, taps=taps_d),
                       rowmap=dict(nI=nI, nJ=nJ, OI=T1, OJ=F1, si=2, sj=2, oi=pt, oj=pf), **kw_)
        torch.cuda.synchronize()
        outs.append(dx.float())
    a, b = outs
    assert not torch.isnan(a).any() and not torch.isnan(b).any()
    assert (a == b).all()                                                                  # (-0 == +0)
    beyond = torch.arange(T1, device=dev)[None, :, None, None] >= len1.to(dev)[:, None, None, None]
    assert (b * beyond == 0).all() and b.abs().max() > 0
    # weight gradient: K runs over the output positions; K-tiles beyond an utterance (dy is masked there) are neither loaded nor
    # multiplied with the hint.  Same sums up to the order of the split-K atomics.
    dym = (dy * (torch.arange(T2)[None, :, None, None] < len2[:, None, None, None])).to(dev)
    taps = [(kh - 1, kw - 1) for kh in range(3) for kw in range(3)]
    M2 = Bn * T2 * F2
    res = []
    for hint in (False, True):
        kw_ = dict(row_len=l2, rows_per_b=T2 * F2, rows_inner=F2) if hint else {}
        dW = torch.zeros(C_, C_, 3, 3, device=dev)
        o.gemm(dym, xd, dW, C_, C_, M2, C_, C_, 9 * C_, transA=True, transB=True, atomic=True, splitk=6, batch=9, nb0=9,
               sC=(1, 0), c_col_stride=9, c_dtype=o.F32,
               gather=dict(operand=1, nI=T2, nJ=F2, SI=T1, SJ=F1, C=C_, si=2, sj=2, taps=taps), **kw_)
        torch.cuda.synchronize()
        res.append(dW)
    assert torch.isfinite(res[1]).all() and rel_err(res[1], res[0]) < 1e-5, rel_err(res[1], res[0])


# ---------------------------------------------------------------------------------------------- mel front-end
def _fb_sparse(fb):
    from nemo_amd.modules.audio_preprocessing import sparsify_filterbank
    return tuple(t.to(dev) for t in sparsify_filterbank(fb))


def test_logmel_vs_reference_fixture(golden_dir):
    o = ops()
    z = np.load(os.path.join(golden_dir, "ref_mel_b3.npz"))
    audio = torch.from_numpy(z["audio"]); alen = torch.from_numpy(z["audio_len"])
    fb = torch.from_numpy(z["fb"][0]); win = torch.from_numpy(z["window"])
    raw = o.logmel(audio.to(dev), alen.to(dev), win.to(dev), _fb_sparse(fb), 80)
    seq = torch.from_numpy(z["mel_len"]).to(dev)
    feat = o.feat_normalize(raw, seq)
    torch.cuda.synchronize()
    ref = torch.from_numpy(z["mel"])
    err = (feat.cpu() - ref).abs().max().item()
    # north_star: mel features within 1e-3 relative fp32 of the reference CPU path (features are O(1) after normalisation)
    assert err < 1e-3 * max(1.0, ref.abs().max().item()), err
    assert np.allclose(feat.cpu().numpy(), z["mel"], rtol=1e-3, atol=1e-3)


def test_logmel_ragged_and_padding_invariance():
    o = ops()
    g = torch.Generator().manual_seed(11)
    S = 16000 + 37
    audio = 0.1 * torch.randn(2, S, generator=g)
    alen = torch.tensor([S, 4000])
    fb = torch.from_numpy(R.mel_filterbank()); win = R.hann_window_sym(400)
    ref, ref_len = R.log_mel_features(audio, alen)
    raw = o.logmel(audio.to(dev), alen.to(dev), win.to(dev), _fb_sparse(fb), 80)
    feat = o.feat_normalize(raw, ref_len.to(dev))
    assert (feat.cpu() - ref).abs().max() < 1e-3
    # right-padding the batch must not change valid frames (reference test_padding_and_batch_size_invariance.py:22-45)
    audio2 = torch.cat([audio, torch.zeros(2, 3200)], 1)
    raw2 = o.logmel(audio2.to(dev), alen.to(dev), win.to(dev), _fb_sparse(fb), 80)
    feat2 = o.feat_normalize(raw2, ref_len.to(dev))
    torch.cuda.synchronize()
    T = feat.shape[-1]
    assert (feat2[..., : int(ref_len[1])][1] - feat[..., : int(ref_len[1])][1]).abs().max() < 1e-5


@pytest.mark.parametrize("S,dither", [(16000 + 37, 0.0), (48000, 1e-5), (331, 1e-5), (160 * 64, 0.0)])
def test_logmel_wave_synchronised_kernel_against_the_round_1_kernel(S, dither):
    """csrc/mel.hip: the default front-end kernel (round 6: FFT in registers, 16 lanes per frame, one LDS transpose, the dither
    evaluated once per sample) and the round-4 kernel (wave-local hand-overs instead of workgroup barriers, in-place radix-4 stages)
    against the round-1 kernel on the same input -- ragged lengths, a clip shorter than one window, dither on (counter-based: the
    same noise every way).  The radix-4 kernels share their arithmetic up to fused-multiply-add contraction; the register kernel is
    the same DFT in another summation order (fp32: ~1e-6 relative on the powers).  The wave-local hand-overs must be RACE-FREE:
    five runs of each, identical bits."""
    o = ops()
    from nemo_amd._lib import lib
    g = torch.Generator().manual_seed(S)
    audio = (0.1 * torch.randn(3, S, generator=g)).to(dev)
    alen = torch.tensor([S, max(1, S // 3), max(1, S - 161)]).to(dev)
    fb = _fb_sparse(torch.from_numpy(R.mel_filterbank())); win = R.hann_window_sym(400).to(dev)
    outs = []
    prev = lib.mi355x_logmel_config(-1)
    try:
        for variant in (0, 1, 1, 1, 1, 1, 2, 2, 2, 2, 2):
            lib.mi355x_logmel_config(variant)
            raw = torch.full((3, 80, 1 + S // 160), float("nan"), device=dev)
            o.logmel(audio, alen, win, fb, 80, dither=dither, seed=123, out=raw)
            torch.cuda.synchronize()
            outs.append(raw.cpu())
    finally:
        lib.mi355x_logmel_config(prev)
    assert prev == 2
    assert torch.isfinite(outs[1]).all() and torch.isfinite(outs[6]).all()
    for k in range(2, 6):
        assert torch.equal(outs[1], outs[k]), k
    for k in range(7, 11):
        assert torch.equal(outs[6], outs[k]), k
    # log(power + 2^-24): values in [-16.6, ~10]; a differently contracted / ordered FFT moves a power by ~1e-6 relative
    err = (outs[0] - outs[1]).abs().max().item()
    assert err < 1e-4, err
    err = (outs[0] - outs[6]).abs().max().item()
    assert err < 1e-4, err


# ---------------------------------------------------------------------------------------------- CTC
@pytest.mark.parametrize("name", ["test_case_small", "test_case_small_blank_last", "test_case_big_tensor"])
def test_ctc_known_answers(golden_dir, name):
    o = ops()
    case = json.load(open(os.path.join(golden_dir, "ctc_known_answers.json")))[name]
    acts = torch.tensor(case["acts"], dtype=torch.float32)
    Bn, T, C_ = acts.shape
    labels = torch.tensor(case["labels"], dtype=torch.int64)
    logp = torch.log_softmax(acts, -1).to(dev).contiguous()
    grad = torch.empty_like(logp)
    nll = o.ctc_loss(logp, labels.to(dev), torch.full((Bn,), T, dtype=torch.int64, device=dev),
                     torch.tensor([len(l) for l in case["labels"]], device=dev), case["blank"], grad=grad)
    torch.cuda.synchronize()
    assert np.allclose(nll.cpu().numpy(), case["expected_costs"], rtol=1e-5)
    g = grad.cpu()
    gx = g - torch.exp(logp.cpu()) * g.sum(-1, keepdim=True)  # log-softmax backward of the reference harness
    assert np.allclose(gx.numpy(), np.array(case["expected_grads"]), atol=2e-6, rtol=1e-3)


def test_ctc_random_ragged_vs_oracle():
    o = ops()
    rng = np.random.RandomState(0)
    Bn, T, C_, U = 5, 60, 17, 12
    acts = rng.randn(Bn, T, C_).astype(np.float32)
    tgt = rng.randint(0, C_ - 1, size=(Bn, U))
    tgt[0, 3] = tgt[0, 2]
    in_len = np.array([60, 41, 25, 13, 8]); tl = np.array([12, 9, 12, 6, 7])  # last one infeasible (7 labels w/ repeats?)
    tgt[4, :7] = [1, 1, 1, 1, 1, 1, 1]  # needs 13 frames > 8 -> infinite -> zero_infinity
    logp = torch.log_softmax(torch.from_numpy(acts), -1)
    nll_ref, g_ref = ctc_ref.ctc_loss_and_grad(logp.numpy().astype(np.float64), tgt, in_len, tl, blank=C_ - 1)
    grad = torch.empty(Bn, T, C_, device=dev)
    nll = o.ctc_loss(logp.to(dev).contiguous(), torch.from_numpy(tgt).to(dev), torch.from_numpy(in_len).to(dev),
                     torch.from_numpy(tl).to(dev), C_ - 1, grad=grad, grad_scale=0.5)
    torch.cuda.synchronize()
    assert nll_ref[4] == 0.0
    assert np.allclose(nll.cpu().numpy(), nll_ref, rtol=1e-4, atol=1e-5)
    assert np.allclose(grad.cpu().numpy(), 0.5 * g_ref, rtol=1e-3, atol=1e-5)


# ---------------------------------------------------------------------------------------------- optimizer / packing
def test_ctc_greedy_decode_vs_oracle():
    """mi355x_ctc_greedy_decode against the restated reference loop on random ragged log-probs (with ties and long
    repeats), and the text / WER plumbing of the drop-in decoder"""
    from nemo_amd.modules import GreedyCTCDecoder, WER
    from oracle import decode_ref as D
    o = ops()
    g = torch.Generator().manual_seed(41)
    B, T, V = 5, 301, 28
    x = torch.randn(B, T, V + 1, generator=g)
    x[:, :, V] += 1.5                                   # plenty of blanks
    x[1, 40:90] = x[1, 40:41]                           # a long repeat
    x[2, :, :] = torch.round(x[2] * 2) / 2              # exact ties: the first maximum wins (torch.max)
    logp = torch.log_softmax(x, -1)
    lens = torch.tensor([T, 250, 301, 0, 17])
    ref = D.greedy_decode(logp, lens, blank=V)
    tok, olen, score = o.ctc_greedy_decode(logp.to(dev), lens.to(dev), V)
    torch.cuda.synchronize()
    tok, olen, score = tok.cpu(), olen.cpu(), score.cpu()
    for b, (rt, rs) in enumerate(ref):
        assert int(olen[b]) == len(rt), b
        assert tok[b, : len(rt)].tolist() == rt, b
        assert torch.all(tok[b, len(rt):] == -1)
        assert abs(score[b].item() - rs) <= 1e-4 * max(1.0, abs(rs)), b
    vocab = [chr(ord("a") + i) for i in range(26)] + [" ", "'"]
    dec = GreedyCTCDecoder(vocab)
    texts = dec(logp.to(dev), lens.to(dev))
    assert texts == [D.tokens_to_text(rt, vocab) for rt, _ in ref]
    wer = WER(dec)
    tgt = torch.randint(0, V, (B, 12), generator=g); tl = torch.tensor([12, 10, 12, 3, 5])
    wer.update(logp.to(dev), lens.to(dev), tgt.to(dev), tl.to(dev))
    refs = [D.tokens_to_text(tgt[b, : int(tl[b])].tolist(), vocab) for b in range(B)]
    assert abs(wer.compute()[0] - D.word_error_rate(texts, refs)) < 1e-12


def test_adamw_matches_torch():
    o = ops()
    n = 4096 + 64
    g = torch.Generator().manual_seed(12)
    p0 = torch.randn(n, generator=g)
    pr = p0.clone().requires_grad_(True)
    opt = torch.optim.AdamW([pr], lr=1e-2, betas=(0.9, 0.98), eps=1e-8, weight_decay=1e-3)
    p = p0.to(dev); m = torch.zeros(n, device=dev); v = torch.zeros(n, device=dev)
    for step in range(1, 4):
        gr = torch.randn(n, generator=g)
        pr.grad = gr.clone(); opt.step()
        o.adamw_step(p, (gr * 2).to(dev), m, v, 1e-2, 0.9, 0.98, 1e-8, 1e-3, step, grad_scale=0.5)
    torch.cuda.synchronize()
    assert rel_err(p, pr) < 1e-5


def test_adamw_in_pieces_is_bit_identical_to_one_step():
    """begin_step / step_range / finish_step (the optimizer running behind backward, slice by slice, in any order)
    against step() on the same gradients: same kernel per element, so the weights, moments and EMA are bit-identical"""
    from nemo_amd.flat import FlatParams
    from nemo_amd.optim import FusedAdamW
    torch.manual_seed(5)
    def make():
        torch.manual_seed(5)
        ms = [torch.nn.Sequential(torch.nn.Linear(64, 96), torch.nn.Linear(96, 160), torch.nn.Linear(160, 32)).to(dev),
              torch.nn.Linear(32, 8).to(dev)]
        fl = [FlatParams(m) for m in ms]
        for fp in fl:
            fp.build(dev)
        return ms, fl, FusedAdamW(fl, lr=3e-3, betas=(0.9, 0.98), weight_decay=1e-2, ema_decay=0.99)
    ma, fa, oa = make()
    mb, fb, ob = make()
    g = torch.Generator(device=dev).manual_seed(3)
    side = torch.cuda.Stream()
    for step in range(4):
        oa.zero_grad(); ob.zero_grad()
        for x, y in zip(fa, fb):
            gr = torch.randn(x.grad.shape, device=dev, generator=g)
            x.grad.copy_(gr); y.grad.copy_(gr)
        oa.step(lr=1e-3 * (step + 1), grad_scale=0.5)
        assert ob.begin_step(lr=1e-3 * (step + 1), grad_scale=0.5)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):  # reverse order, a gap left for finish_step, second buffer untouched until then
            hi = fb[0].range_of("2.")
            ob.step_range(fb[0], *hi)
            ob.step_range(fb[0], *fb[0].range_of("0."))
        torch.cuda.current_stream().wait_stream(side)
        ob.finish_step()
    torch.cuda.synchronize()
    assert oa.step_count == ob.step_count == 4
    for x, y in zip(fa, fb):
        assert torch.equal(x.flat, y.flat)
        assert torch.equal(oa._moments(x)[0], ob._moments(y)[0]) and torch.equal(oa._moments(x)[1], ob._moments(y)[1])
        assert torch.equal(oa._ema_of(x), ob._ema_of(y))
    # with clipping the global norm needs every gradient first: the piecewise mode declines
    oc = FusedAdamW(fb, lr=1e-3, max_grad_norm=1.0)
    assert not oc.begin_step()


def test_adamw_clip_and_ema_match_torch():
    """gradient_clip_val + EMA inside the fused launch: against torch.nn.utils.clip_grad_norm_ + torch.optim.AdamW +
    the reference's ema_update (_foreach_mul_/_foreach_add_, ema.py:150-157) over several steps and two flat buffers"""
    from nemo_amd.flat import FlatParams
    from nemo_amd.optim import FusedAdamW
    torch.manual_seed(4)
    mods = [torch.nn.Sequential(torch.nn.Linear(64, 96), torch.nn.Linear(96, 32)).to(dev), torch.nn.Linear(32, 8).to(dev)]
    refs = [torch.nn.Sequential(torch.nn.Linear(64, 96), torch.nn.Linear(96, 32)).to(dev), torch.nn.Linear(32, 8).to(dev)]
    for a, b in zip(mods, refs):
        b.load_state_dict(a.state_dict())
    flats = [FlatParams(m) for m in mods]
    for fp in flats:
        fp.build(dev)
    opt = FusedAdamW(flats, lr=3e-3, betas=(0.9, 0.98), weight_decay=1e-2, max_grad_norm=0.7, ema_decay=0.9)
    rparams = [p for r in refs for p in r.parameters()]
    ropt = torch.optim.AdamW(rparams, lr=3e-3, betas=(0.9, 0.98), weight_decay=1e-2)
    ema_ref = [p.detach().clone() for p in rparams]
    g = torch.Generator(device=dev).manual_seed(9)
    for step in range(5):
        opt.zero_grad(); ropt.zero_grad()
        grads = [torch.randn(p.shape, device=dev, generator=g) * (3.0 if step % 2 == 0 else 0.01) for p in rparams]
        for p, q_, gr in zip([p for m in mods for p in m.parameters()], rparams, grads):
            p.grad.copy_(gr); q_.grad = gr.clone()
        total = torch.nn.utils.clip_grad_norm_(rparams, 0.7)
        ropt.step()
        torch._foreach_mul_(ema_ref, 0.9); torch._foreach_add_(ema_ref, [p.detach() for p in rparams], alpha=0.1)
        opt.step()
        torch.cuda.synchronize()
        assert abs(opt.last_grad_norm.item() - total.item()) <= 1e-5 * total.item()
    for p, q_ in zip([p for m in mods for p in m.parameters()], rparams):
        assert rel_err(p, q_) < 2e-6
    with opt.swap_ema_weights():
        for p, e in zip([p for m in mods for p in m.parameters()], ema_ref):
            assert rel_err(p, e) < 2e-6
    for p, q_ in zip([p for m in mods for p in m.parameters()], rparams):
        assert rel_err(p, q_) < 2e-6  # swapped back


def test_pack_weights():
    from nemo_amd.packing import PackPlan
    g = torch.Generator().manual_seed(13)
    W = torch.randn(70, 45, generator=g).to(dev)
    W2 = torch.randn(16, 8, 3, 3, generator=g).to(dev)
    plan = PackPlan(torch.bfloat16, dev)
    a = plan.add_matrix("w", W)                       # [70, 48] (K padded to 8)
    at = plan.add_matrix("wt", W, transpose=True)     # [45, 72]
    c2 = plan.add_conv3x3("c2", W2)                   # [16, 72]: k = (kh*3+kw)*8 + ci
    # 16-byte-aligned rows: whole 64 x 64 tiles take the vectorised path, the ragged last tile of each axis the scalar one
    W3 = torch.randn(200, 136, generator=g).to(dev)
    plan.add_matrix("w3", W3); plan.add_matrix("w3t", W3, transpose=True)
    plan.finalize()
    plan.run()
    torch.cuda.synchronize()
    assert torch.equal(plan["w3"][:, :136].float().cpu(), W3.to(torch.bfloat16).float().cpu())
    assert torch.equal(plan["w3t"][:, :200].float().cpu(), W3.t().to(torch.bfloat16).float().cpu())
    assert torch.equal(plan["w"][:, :45].float().cpu(), W.to(torch.bfloat16).float().cpu())
    assert torch.all(plan["w"][:, 45:] == 0)
    assert torch.equal(plan["wt"][:, :70].float().cpu(), W.t().to(torch.bfloat16).float().cpu())
    assert torch.equal(plan["c2"].float().cpu(), W2.permute(0, 2, 3, 1).reshape(16, 72).to(torch.bfloat16).float().cpu())


# ------------------------------------------------------------------ RNN-T loss
def _rnnt_known():
    import json
    with open(os.path.join(os.path.dirname(__file__), "golden", "rnnt_known_answers.json")) as f:
        return json.load(f)["cases"]


def test_rnnt_loss_known_answers_of_the_reference_tests():
    from nemo_amd.modules import RNNTLoss
    cases = _rnnt_known()

    def run(acts, labels, **kw):
        a = torch.tensor(acts, dtype=torch.float32, device=dev).requires_grad_(True)
        lab = torch.tensor(labels, device=dev)
        B, T = a.shape[:2]
        lens = torch.full((B,), T, device=dev, dtype=torch.int64)
        ll = torch.full((B,), lab.shape[1], device=dev, dtype=torch.int64)
        cost = RNNTLoss(blank=0, **kw)(a, lab, lens, ll)
        cost.sum().backward()
        return cost.detach().cpu().numpy(), a.grad.cpu().numpy()
    c = cases["test_case_small"]
    cost, grads = run(c["acts"], c["labels"], reduction="sum")
    assert np.allclose(cost, c["expected_cost"], atol=1e-5) and np.allclose(grads, np.array(c["expected_grads"]), atol=1e-6, rtol=1e-4)
    for lam in (1.0, 0.01, 0.00001):  # test_case_small_fastemit_clamp: the cost scales by (1 + lambda)
        cf, _ = run(c["acts"], c["labels"], reduction="sum", fastemit_lambda=lam, clamp=0.1)
        assert np.allclose(cf, c["expected_cost"] * (1 + lam), rtol=1e-5)
    c = cases["test_case_small_clamp"]
    cost, grads = run(c["acts"], c["labels"], reduction="sum", clamp=c["GRAD_CLAMP"])
    assert np.allclose(cost, c["expected_cost"], atol=1e-5) and np.allclose(grads, np.array(c["expected_grads"]), atol=1e-6, rtol=1e-4)
    c = cases["test_case_big_tensor"]
    cost, grads = run(c["activations"], c["labels"], reduction="sum")
    assert np.allclose(cost, sum(c["expected_costs"]), atol=1e-5)
    assert np.allclose(grads, np.array(c["expected_grads"]), atol=1e-6, rtol=1e-3)
    cost, _ = run(c["activations"], c["labels"], reduction="none")
    assert np.allclose(cost, np.array(c["expected_costs"]), atol=1e-5)


@pytest.mark.parametrize("B,T,U1,V1,blank", [(3, 9, 6, 7, 0), (4, 33, 17, 29, 28), (2, 70, 70, 1025, 1024), (5, 12, 1, 8, 3)])
def test_rnnt_loss_matches_the_oracle_on_ragged_batches(B, T, U1, V1, blank):
    """fused log-softmax + lattice + gradient kernels against oracle/rnnt_ref.py (float64): ragged T and U, U1 beyond one
    wave, vector (V1 % 4 == 0) and scalar row paths, an empty-label utterance, FastEmit and clamp, all reductions"""
    from oracle import rnnt_ref as RR
    from nemo_amd.modules import RNNTLoss
    g = torch.Generator().manual_seed(B * 1000 + T)
    acts = torch.randn(B, T, U1, V1, generator=g) * 2.0
    lens = torch.randint(max(1, T // 2), T + 1, (B,), generator=g); lens[0] = T
    ll = torch.randint(0, U1, (B,), generator=g) if U1 > 1 else torch.zeros(B, dtype=torch.int64)
    ll[-1] = U1 - 1
    if B > 2 and U1 > 1:
        ll[1] = 0
    labels = torch.randint(0, V1 - 1, (B, max(U1 - 1, 0)), generator=g)
    labels = labels + (labels >= blank).long()
    for kw in (dict(reduction="sum"), dict(reduction="mean", fastemit_lambda=0.3), dict(reduction="none", clamp=0.05),
               dict(reduction="mean", fastemit_lambda=0.01, clamp=0.2)):
        a = acts.to(dev).requires_grad_(True)
        cost = RNNTLoss(blank=blank, **kw)(a, labels.to(dev), lens.to(dev), ll.to(dev))
        up = torch.linspace(0.5, 1.5, cost.numel(), device=dev)  # a non-trivial upstream gradient
        (cost * up).sum().backward()
        rc, rg = RR.rnnt_loss_and_grad(acts, labels, lens, ll, blank=blank, fastemit_lambda=kw.get("fastemit_lambda", 0.0),
                                       clamp=kw.get("clamp", 0.0), reduction=kw["reduction"])
        upc = up.cpu().double()
        rg = rg * (upc.view(-1, 1, 1, 1) if kw["reduction"] == "none" else upc)
        assert torch.allclose(cost.detach().cpu().double(), rc, rtol=2e-5, atol=1e-4), (kw, cost, rc)
        err = (a.grad.cpu().double() - rg).abs().max().item()
        # fp32 lattice values reach |alpha| ~ (T+U) * log V (~ 1e3 for the large case: ulp 6e-5), so exp(alpha+beta-ll) carries
        # ~1e-4 relative error there -- the reference's own big-tensor test allows rtol 1e-3 (test_rnnt_pytorch.py:306-309)
        tol = 2e-5 if T + U1 < 64 else 1e-3
        assert err <= tol * max(1.0, rg.abs().max().item()), (kw, err)


def test_rnnt_loss_argument_checks():
    from nemo_amd.modules import RNNTLoss
    a = torch.randn(2, 5, 3, 4, device=dev)
    lab = torch.ones(2, 2, dtype=torch.int64, device=dev)
    lens = torch.tensor([5, 4], device=dev); ll = torch.tensor([2, 1], device=dev)
    RNNTLoss()(a, lab, lens, ll)
    with pytest.raises(ValueError):  # T must equal max(lengths)  (rnnt_numpy.py:92-95)
        RNNTLoss()(a, lab, torch.tensor([4, 4], device=dev), ll)
    with pytest.raises(ValueError):  # U must equal max(label_lengths) + 1
        RNNTLoss()(a, lab, lens, torch.tensor([1, 1], device=dev))
    with pytest.raises(TypeError):
        RNNTLoss()(a, lab.int(), lens, ll)
    with pytest.raises(RuntimeError):
        RNNTLoss()(a.cpu(), lab.cpu(), lens.cpu(), ll.cpu())


@pytest.mark.parametrize("M,N,K", [(8000, 1024, 512), (16032, 2048, 512), (8200, 1024, 584), (16032, 1024, 2048)])
def test_gemm_persistent_structure_matches_the_tiled_one(M, N, K):
    """gemm_bf16_v5_kernel (persistent workgroups, epilogue of tile t inside the K loop of tile t+1, wave-private LDS windows,
    counted vmcnt across tiles, inline-asm fragment / aux loads) against gemm_bf16_v2/v4 on the same inputs INCLUDING the regenerated dropout masks, for the
    four epilogues it carries; ragged M (last tile partly / rounds wholly past M), K tail (584 = 9 K-tiles + 8), K = 2048
    (rounds only in the first 8 of 32 iterations).  Five launches per case must agree bit for bit (no atomics in this
    structure: any run-to-run difference would be a race between DMA, window and barrier)."""
    o = ops()
    g = torch.Generator().manual_seed(M + K)
    A = bf(torch.randn(M, K, generator=g)).to(dev)
    W = bf(torch.randn(N, K, generator=g) * 0.05).to(dev)
    bias = torch.randn(N, generator=g).to(dev)
    res = torch.randn(M, N, generator=g).to(dev)
    pre = bf(torch.randn(M, N, generator=g)).to(dev)
    drop = o.Dropout(0.1, 7, 3)

    def run(kind):
        if kind == "store":
            c = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
            o.gemm(A, W, c, M, N, K, K, K, N, bias=bias, alpha=0.5, drop=drop)
            return (c,)
        if kind == "store_f32":
            c = torch.empty(M, N, device=dev)
            o.gemm(A, W, c, M, N, K, K, K, N, bias=bias)
            return (c,)
        if kind == "swish":
            h = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
            a = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
            o.gemm(A, W, a, M, N, K, K, K, N, bias=bias, epi=o.EPI_SWISH_DROP, aux_out=h, drop=drop)
            return (h, a)
        if kind == "resid":
            c = torch.empty(M, N, device=dev)
            o.gemm(A, W, c, M, N, K, K, K, N, bias=bias, alpha=0.5, epi=o.EPI_RESID, aux_in=res, drop=drop)
            return (c,)
        if kind == "resid_inplace":
            c = res.clone()
            o.gemm(A, W, c, M, N, K, K, K, N, bias=bias, alpha=0.5, epi=o.EPI_RESID, aux_in=c, drop=drop)
            return (c,)
        c = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        o.gemm(A, W, c, M, N, K, K, K, N, epi=o.EPI_DSWISH, aux_in=pre, drop=drop)
        return (c,)

    old = o.gemm_config(5, 1)
    try:
        for kind in ("store", "store_f32", "swish", "resid", "resid_inplace", "dswish"):
            o.gemm_config(5, 0)
            want = [t.float() for t in run(kind)]
            o.gemm_config(5, 2)  # 2 = wherever the persistent structure can run (1 = only where it measured faster)
            first = run(kind)
            torch.cuda.synchronize()
            for t, w_ in zip(first, want):
                assert torch.isfinite(t).all(), kind
                # same products, same k order; the bias enters the sum first instead of last: fp32 re-association only
                assert rel_err(t, w_) < (5e-6 if t.dtype == torch.float32 else 5e-3), (kind, rel_err(t, w_))
                assert (t.float() - w_).abs().max() <= 0.02 * w_.abs().max() , kind
            for _ in range(4):
                again = run(kind)
                torch.cuda.synchronize()
                for t, f in zip(again, first):
                    assert torch.equal(t, f), (kind, "run-to-run difference")
    finally:
        o.gemm_config(5, old if old >= 0 else 1)


@pytest.mark.parametrize("V1,ldp", [(1025, 1032), (29, 32), (16, 16)])
def test_rnnt_loss_pitched_rows_and_bf16_operand_gradient(V1, ldp):
    """mi355x_rnnt_loss_ex (the fused joint + loss path): logits with a row pitch, gradient written as the pitched GEMM operand.
    f32 pitched output equals the dense entry (to the ulp); bf16 output is its rounding; pad columns are zero."""
    o = ops()
    B, T, U1 = 3, 19, 7
    g = torch.Generator().manual_seed(5)
    acts = torch.randn(B, T, U1, V1, generator=g).to(dev)
    labels = torch.randint(0, V1 - 1, (B, U1 - 1), generator=g).to(dev)
    xl = torch.tensor([19, 13, 7]).to(dev)
    yl = torch.tensor([6, 3, 0]).to(dev)
    gd = torch.empty_like(acts)
    cd = o.rnnt_loss(acts, labels, xl, yl, V1 - 1, grads=gd, fastemit_lambda=0.01, grad_scale=0.5)
    n = B * T * U1
    ap = torch.full((n, ldp), float("nan"), device=dev)
    ap[:, :V1] = acts.view(n, V1)
    for dtype in (torch.float32, torch.bfloat16):
        gp = torch.full((n, ldp), 7.0, device=dev, dtype=dtype)
        cp = o.rnnt_loss_pitched(ap, ldp, B, T, U1, V1, labels, xl, yl, V1 - 1, gp, ldp, fastemit_lambda=0.01, grad_scale=0.5)
        torch.cuda.synchronize()
        assert torch.equal(cp, cd)
        assert (gp[:, V1:] == 0).all()
        if dtype == torch.float32:
            # (an element may go through the scalar head / tail site in one walk and the float4 site in the other: the
            #  compiler contracts the two copies of the same expression differently -- 1 ulp)
            assert torch.allclose(gp[:, :V1], gd.view(n, V1), rtol=3e-5, atol=1e-7)  # (blank column: difference of exponentials)
            g32 = gp.clone()
        else:
            assert torch.equal(gp[:, :V1], g32[:, :V1].to(torch.bfloat16))  # same walk, rounded once


@pytest.mark.parametrize("tile", [256, 128])
@pytest.mark.parametrize("M,N,K", [(16032, 2048, 512), (3000, 512, 256), (5000, 1280, 2048), (777, 384, 1024)])
def test_gemm_register_prefetch_structure_matches_the_lds_dma_one(M, N, K, tile):
    """gemm_bf16_v6_kernel (256x256x64 tile, operands prefetched through registers two K-tiles deep, ds_write into the same LDS
    images) against gemm_bf16_v4_kernel (LDS-DMA) on the same inputs: same products in the same order, so the results must be
    BIT-identical, dropout masks included; ragged M / N (clamped rows), K = 4 ... 32 K-tiles, and run-to-run determinism (a
    difference between two launches would be a race between the register copies, the stages and the barriers)."""
    o = ops()
    g = torch.Generator().manual_seed(M + N + K)
    A = bf(torch.randn(M, K, generator=g)).to(dev)
    W = bf(torch.randn(N, K, generator=g) * 0.05).to(dev)
    bias = torch.randn(N, generator=g).to(dev)
    res = torch.randn(M, N, generator=g).to(dev)
    drop = o.Dropout(0.1, 11, 5)
    ref = (A.float() @ W.float().t() + bias).cpu()

    def run(kind):
        if kind == "store":
            c = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
            o.gemm(A, W, c, M, N, K, K, K, N, bias=bias)
            return (c,)
        if kind == "swish":
            h = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
            a = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
            o.gemm(A, W, a, M, N, K, K, K, N, bias=bias, epi=o.EPI_SWISH_DROP, aux_out=h, drop=drop)
            return (h, a)
        c = torch.empty(M, N, device=dev)
        o.gemm(A, W, c, M, N, K, K, K, N, bias=bias, alpha=0.5, epi=o.EPI_RESID, aux_in=res, drop=drop)
        return (c,)

    # tile 256: the 256x256 structure wherever N > 128; tile 128: the 256x128 structure everywhere; never the persistent one
    key = 6 if tile == 256 else 7
    old4, old5, old6 = o.gemm_config(4, 2 if tile == 256 else 0), o.gemm_config(5, 0), o.gemm_config(key, 0)
    try:
        for kind in ("store", "swish", "resid"):
            o.gemm_config(key, 0)
            want = [t.float() for t in run(kind)]
            o.gemm_config(key, 1)
            for rep in range(3):
                got = [t.float() for t in run(kind)]
                torch.cuda.synchronize()
                for w_, g_ in zip(want, got):
                    assert torch.equal(w_, g_), (kind, rep, (w_ - g_).abs().max().item())
            if kind == "store":
                assert rel_err(got[0], ref) < 1e-2
    finally:
        o.gemm_config(4, old4 if old4 >= 0 else 1); o.gemm_config(5, old5 if old5 >= 0 else 1); o.gemm_config(key, old6 if old6 >= 0 else 1)
