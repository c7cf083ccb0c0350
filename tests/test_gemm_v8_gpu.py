"""-m gpu: the phase-staggered 256x256 GEMM structure (gemm_bf16_v8_kernel, `mi355x_gemm_config(8, .)`; 16x16x32 MFMAs, two wave
rows one barrier apart, counted vmcnt, LDS-DMA seven half-tiles ahead) against the lock-step structures and against fp32 products
of the same bf16-rounded operands.  The K-contiguous layouts accumulate the same products in the same k order as the 32x32x16
structures, so their results must be BIT-identical (dropout masks, ReLU gates, row maps included); the reduction-major layouts
(weight gradients) end in f32 atomics whose order is free, so they are held to the fp32 product instead.  Every case is launched
several times: a run-to-run difference would be a race between the DMA, the fragment reads and the staggered barriers."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

dev = "cuda"
bf16 = torch.bfloat16


def ops():
    from nemo_amd import ops as _ops
    return _ops


def rel_l2(a, b):
    a, b = a.detach().float(), b.detach().float()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


class _modes:
    """mi355x_gemm_config keys set for the duration of a `with` block, previous values restored"""

    def __init__(self, **kv):
        self.kv = {int(k[1:]): v for k, v in kv.items()}

    def __enter__(self):
        o = ops()
        self.old = {k: o.gemm_config(k, v) for k, v in self.kv.items()}

    def __exit__(self, *a):
        o = ops()
        for k, v in self.old.items():
            o.gemm_config(k, v if v >= 0 else (0 if k == 6 else 1))


@pytest.mark.parametrize("M,N,K", [(16032, 2048, 512), (3000, 520, 256), (5000, 1280, 2048), (777, 384, 1024), (300, 1536, 128),
                                   (8200, 1024, 576)])
def test_phase_staggered_structure_is_bit_identical_to_the_lock_step_one(M, N, K):
    """dense NT: ragged M / N (clamped rows, partial column tiles: 520 = 2 tiles + 8 columns), K = 2 ... 32 K-tiles (odd counts
    included), five epilogues with their dropout masks; three launches each."""
    o = ops()
    g = torch.Generator().manual_seed(M + N + K)
    A = (torch.rand(M, K, generator=g) * 2 - 1).to(bf16).to(dev)
    W = ((torch.rand(N, K, generator=g) * 2 - 1) * 0.05).to(bf16).to(dev)
    bias = torch.randn(N, generator=g).to(dev)
    res = torch.randn(M, N, generator=g).to(dev)
    pre = torch.randn(M, N, generator=g).to(bf16).to(dev)
    drop = o.Dropout(0.1, 11, 5)
    ref = A.float() @ W.float().t() + bias

    def run(kind):
        if kind == "store_f32":
            c = torch.full((M, N), float("nan"), device=dev)
            o.gemm(A, W, c, M, N, K, K, K, N, bias=bias)
            return (c,)
        if kind == "store":
            c = torch.empty(M, N, device=dev, dtype=bf16)
            o.gemm(A, W, c, M, N, K, K, K, N, bias=bias, alpha=0.5, drop=drop)
            return (c,)
        if kind == "swish":
            h = torch.empty(M, N, device=dev, dtype=bf16)
            a = torch.empty(M, N, device=dev, dtype=bf16)
            o.gemm(A, W, a, M, N, K, K, K, N, bias=bias, epi=6, aux_out=h, drop=drop)  # Swish + dropout, g = swish'(h) * mask stored
            return (h, a)
        if kind == "resid":
            c = torch.empty(M, N, device=dev)
            o.gemm(A, W, c, M, N, K, K, K, N, bias=bias, alpha=0.5, epi=o.EPI_RESID, aux_in=res, drop=drop)
            return (c,)
        c = torch.empty(M, N, device=dev, dtype=bf16)
        o.gemm(A, W, c, M, N, K, K, K, N, epi=o.EPI_DSWISH, aux_in=pre, drop=drop)
        return (c,)

    for kind in ("store_f32", "store", "swish", "resid", "dswish"):
        with _modes(k8=0, k4=2, k5=0, k6=0):   # the third structure (256x256, lock step) wherever N > 128
            want = run(kind)
        with _modes(k8=2, k5=0):
            for rep in range(3):
                got = run(kind)
                torch.cuda.synchronize()
                for w_, g_ in zip(want, got):
                    assert torch.equal(w_, g_), (kind, rep, (w_.float() - g_.float()).abs().max().item())
        if kind == "store_f32":
            assert rel_l2(got[0], ref) < 2e-6


@pytest.mark.parametrize("M,N,K", [(16032, 512, 2048), (16032, 512, 512), (9000, 768, 1024), (8100, 520, 192)])
def test_one_a_half_tile_of_the_phase_staggered_structure(M, N, K):
    """the 128x256 tile (one A half, three K-tile buffers, two phases per K-tile; key 8 mode 2 forces it): problems whose 256x256 tiles leave
    the chip half empty.  Bit-identical to the 256x128 lock-step structure (same products, same k order), four epilogues, odd K-tile
    counts, ragged M / N, three launches each."""
    o = ops()
    g = torch.Generator().manual_seed(M + N + K)
    A = (torch.rand(M, K, generator=g) * 2 - 1).to(bf16).to(dev)
    W = ((torch.rand(N, K, generator=g) * 2 - 1) * 0.05).to(bf16).to(dev)
    bias = torch.randn(N, generator=g).to(dev)
    res = torch.randn(M, N, generator=g).to(dev)
    drop = o.Dropout(0.1, 3, 9)

    def run(kind):
        if kind == "store":
            c = torch.empty(M, N, device=dev, dtype=bf16)
            o.gemm(A, W, c, M, N, K, K, K, N, bias=bias)
            return c
        if kind == "store_f32":
            c = torch.full((M, N), float("nan"), device=dev)
            o.gemm(A, W, c, M, N, K, K, K, N, bias=bias, alpha=0.5, drop=drop)
            return c
        c = torch.empty(M, N, device=dev)
        o.gemm(A, W, c, M, N, K, K, K, N, bias=bias, alpha=0.5, epi=o.EPI_RESID, aux_in=res, drop=drop)
        return c

    for kind in ("store", "store_f32", "resid"):
        with _modes(k8=0, k4=0, k5=0, k7=0):   # the 256x128 lock-step structure (LDS-DMA K loop)
            want = run(kind)
        with _modes(k8=2, k5=0):
            for rep in range(3):
                got = run(kind)
                torch.cuda.synchronize()
                assert torch.equal(want, got), (kind, rep, (want.float() - got.float()).abs().max().item())
    assert rel_l2(run("store").float(), (A.float() @ W.float().t() + bias)) < 5e-3


def test_phase_staggered_structure_split_k_slices_and_batches():
    """NT with atomic split-K (every slice >= 2 K-tiles, uneven last slice) and a strided batch: against the fp32 product"""
    o = ops()
    g = torch.Generator().manual_seed(5)
    M, N, K, nb = 1500, 640, 1344, 3   # 21 K-tiles: split 4 -> 6, 6, 6, 3
    A = (torch.rand(nb, M, K, generator=g) * 2 - 1).to(bf16).to(dev)
    W = ((torch.rand(nb, N, K, generator=g) * 2 - 1) * 0.1).to(bf16).to(dev)
    ref = torch.einsum("bmk,bnk->bmn", A.float(), W.float())
    with _modes(k8=2):
        for sk in (1, 4):
            c = torch.zeros(nb, M, N, device=dev)
            o.gemm(A, W, c, M, N, K, K, K, N, batch=nb, sA=(M * K, 0), sB=(N * K, 0), sC=(M * N, 0), atomic=sk > 1, splitk=sk,
                   c_dtype=o.F32)
            torch.cuda.synchronize()
            assert rel_l2(c, ref) < 2e-6, sk


def test_phase_staggered_structure_gathered_convolution():
    """conv2 of the 'striding' sub-sampling as the encoder issues it: forward (gathered A rows, ReLU + time mask, row tiles beyond an
    utterance zero-filled) and the four input-gradient GEMMs (gathered dY, row map, ReLU gate): bit-identical to the third structure;
    weight gradient (gathered reduction-major B, batch = taps, column-strided C, K-tiles beyond an utterance skipped; key 8 mode 3)
    against the third structure and torch's conv2d gradient."""
    o = ops()
    Bn, C_, T1, F1 = 5, 256, 1200, 40
    T2, F2 = (T1 - 1) // 2 + 1, (F1 - 1) // 2 + 1
    M2 = Bn * T2 * F2
    g = torch.Generator().manual_seed(23)
    len1 = torch.tensor([1200, 700, 301, 64, 1])
    len2 = ((len1 + 1) // 2).to(dev)
    x = torch.relu(torch.randn(Bn, T1, F1, C_, generator=g)).to(bf16)
    x = (x * (torch.arange(T1)[None, :, None, None] < len1[:, None, None, None])).to(dev)
    w2 = (torch.randn(C_, C_, 3, 3, generator=g) * 0.1).to(bf16)
    b2 = (torch.randn(C_, generator=g) * 0.1).to(dev)
    w2p = w2.permute(0, 2, 3, 1).reshape(C_, 9 * C_).contiguous().to(dev)
    taps = [(kh - 1, kw - 1) for kh in range(3) for kw in range(3)]
    dy = torch.randn(Bn, T2, F2, C_, generator=g).to(bf16)
    dy = (dy * (torch.arange(T2)[None, :, None, None] < ((len1 + 1) // 2)[:, None, None, None])).to(dev)

    def forward():
        out2 = torch.full((M2, C_), 7.0, device=dev, dtype=bf16)
        o.gemm(x, w2p, out2, M2, C_, 9 * C_, C_, 9 * C_, C_, bias=b2, epi=o.EPI_RELU_MASK, row_len=len2, rows_per_b=T2 * F2,
               rows_inner=F2, gather=dict(nI=T2, nJ=F2, SI=T1, SJ=F1, C=C_, si=2, sj=2, taps=taps))
        return out2

    def dgrad():
        dx = torch.full((Bn, T1, F1, C_), float("nan"), device=dev, dtype=bf16)
        for pt in (0, 1):
            for pf in (0, 1):
                nI, nJ = (T1 - pt + 1) // 2, (F1 - pf + 1) // 2
                slots = [(kh, kw) for kh in ([1] if pt == 0 else [0, 2]) for kw in ([1] if pf == 0 else [0, 2])]
                taps_d = [(1 if kh == 0 else 0, 1 if kw == 0 else 0) for kh, kw in slots]
                wimg = torch.cat([w2[:, :, kh, kw].t() for kh, kw in slots], dim=1).contiguous().to(dev)
                K = len(slots) * C_
                o.gemm(dy, wimg, dx, Bn * nI * nJ, C_, K, C_, K, C_, epi=o.EPI_MUL_POS, aux_in=x, ldaux=C_, row_len=len2,
                       rows_per_b=nI * nJ, rows_inner=nJ, gather=dict(nI=nI, nJ=nJ, SI=T2, SJ=F2, C=C_, si=1, sj=1, taps=taps_d),
                       rowmap=dict(nI=nI, nJ=nJ, OI=T1, OJ=F1, si=2, sj=2, oi=pt, oj=pf))
        return dx

    def wgrad(hint):
        kw_ = dict(row_len=len2, rows_per_b=T2 * F2, rows_inner=F2) if hint else {}
        dW = torch.zeros(C_, C_, 3, 3, device=dev)
        o.gemm(dy, x, dW, C_, C_, M2, C_, C_, 9 * C_, transA=True, transB=True, atomic=True, splitk=6, batch=9, nb0=9, sC=(1, 0),
               c_col_stride=9, c_dtype=o.F32, gather=dict(operand=1, nI=T2, nJ=F2, SI=T1, SJ=F1, C=C_, si=2, sj=2, taps=taps), **kw_)
        return dW

    with _modes(k8=0, k4=2):
        want_f, want_d, want_w = forward(), dgrad(), wgrad(True)
    with _modes(k8=2):
        for rep in range(3):
            got_f, got_d = forward(), dgrad()
            torch.cuda.synchronize()
            assert torch.equal(want_f, got_f), rep
            assert not torch.isnan(got_d.float()).any()
            assert torch.equal(want_d.float(), got_d.float()), rep      # (float: -0 == +0)
        for hint in (False, True):
            got_w = wgrad(hint)
            torch.cuda.synchronize()
            assert torch.isfinite(got_w).all() and rel_l2(got_w, want_w) < 2e-6, (hint, rel_l2(got_w, want_w))
    w2r = w2.float().clone().requires_grad_(True)
    F.conv2d(x.float().cpu().permute(0, 3, 1, 2), w2r, None, stride=2, padding=1).backward(dy.float().cpu().permute(0, 3, 1, 2))
    assert rel_l2(got_w.cpu(), w2r.grad) < 1e-5


@pytest.mark.parametrize("M,N,K,sk", [(512, 2048, 16032, 4), (300, 520, 1000, 2), (1024, 512, 777, 1), (264, 136, 200, 1)])
def test_phase_staggered_structure_weight_gradient_layouts(M, N, K, sk):
    """both operands reduction-major (ds_read_b64_tr_b16 fragments, second swizzle term), K tails (16032 = 250.5 K-tiles), ragged
    M / N (columns past the matrix read the zero page), fused bias-gradient column sums, atomic split-K: against fp32"""
    o = ops()
    g = torch.Generator().manual_seed(M + K)
    lda, ldb = (M + 7) // 8 * 8, (N + 7) // 8 * 8
    dY = (torch.rand(K, lda, generator=g) * 2 - 1).to(bf16).to(dev)
    X = (torch.rand(K, ldb, generator=g) * 2 - 1).to(bf16).to(dev)
    ref = dY[:, :M].float().t() @ X[:, :N].float()
    refb = dY[:, :M].float().sum(0)
    with _modes(k8=2):
        for rep in range(2):
            dW = torch.zeros(M, N, device=dev)
            db = torch.zeros(M, device=dev)
            o.gemm(dY, X, dW, M, N, K, lda, ldb, N, transA=True, transB=True, atomic=True, splitk=sk, c_dtype=o.F32, colsum_out=db)
            torch.cuda.synchronize()
            assert rel_l2(dW, ref) < 3e-6 and rel_l2(db, refb) < 1e-6, (rep, rel_l2(dW, ref), rel_l2(db, refb))


def test_phase_staggered_structure_grouped_weight_gradients():
    """a Conformer layer's weight gradients as ONE grouped launch on the eighth structure (key 8 mode 3): every dW and bias gradient
    against fp32, accumulated on top of what the buffers held"""
    o = ops()
    g = torch.Generator().manual_seed(9)
    rows, d, dff = 4000, 256, 1024
    mk = lambda n: (torch.rand(rows, n, generator=g) * 2 - 1).to(bf16).to(dev)
    shapes = [(d, dff), (dff, d), (d, d), (2 * d, d), (3 * d, d)]
    probs, refs = [], []
    for no, ni in shapes:
        dY, X = mk(no), mk(ni)
        dW = torch.full((no, ni), 0.5, device=dev)
        db = torch.full((no,), -1.0, device=dev)
        probs.append((dY, no, 0, X, ni, 0, dW, no, ni, db))
        refs.append((0.5 + dY.float().t() @ X.float(), -1.0 + dY.float().sum(0)))
    with _modes(k8=3):
        o.wgrad_grouped(probs, rows, 4)
        torch.cuda.synchronize()
    for q, (rw, rb) in zip(probs, refs):
        assert rel_l2(q[6], rw) < 3e-6 and rel_l2(q[9], rb) < 1e-6
