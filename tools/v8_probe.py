"""The phase-staggered 256x256 GEMM structure (gemm_bf16_v8_kernel, key 8) next to the older structures: correctness against an
fp32 product, run-to-run determinism (race screen), and per-launch time on the Conformer-CTC-Large shapes and on square problems.

    python tools/v8_probe.py            # check + bench
    CHECK=0 python tools/v8_probe.py    # bench only
    ROTATE=8                            # cold operands (8 independent operand / output sets in turn)

All timings: HIP events around ITERS launches, random uniform [-1, 1) operands (never zeros: DVFS), interleaved A/B in one process.
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nemo_amd import ops

dev = "cuda"
bf = torch.bfloat16
ITERS = int(os.environ.get("ITERS", "20"))
ROT = int(os.environ.get("ROTATE", "1"))
REPS = int(os.environ.get("REPS", "3"))


def rel_err(a, b):
    return ((a.float() - b.float()).norm() / b.float().norm().clamp_min(1e-30)).item()


def set_modes(v8, v4=None, v5=None, v6=None, v7=None):
    old = {8: ops.gemm_config(8, v8)}
    for k, v in ((4, v4), (5, v5), (6, v6), (7, v7)):
        if v is not None:
            old[k] = ops.gemm_config(k, v)
    return old


def restore(old):
    for k, v in old.items():
        ops.gemm_config(k, v if v >= 0 else (0 if k == 6 else 1))


def check():
    g = torch.Generator(device=dev).manual_seed(1)
    bad = 0
    shapes = [(256, 256, 128), (512, 512, 256), (777, 384, 1024), (3000, 520, 256), (16032, 2048, 512), (16032, 512, 2048),
              (5000, 1280, 2048), (4096, 4096, 4096), (300, 1536, 512), (8200, 1024, 576)]
    for (M, N, K) in shapes:
        A = (torch.rand(M, K, device=dev, generator=g) * 2 - 1).to(bf)
        W = ((torch.rand(N, K, device=dev, generator=g) * 2 - 1) * 0.1).to(bf)
        bias = torch.randn(N, device=dev, generator=g)
        res = torch.randn(M, N, device=dev, generator=g)
        ref = A.float() @ W.float().t() + bias
        drop = ops.Dropout(0.1, 11, 5)
        for kind in ("store", "store_f32", "swish", "resid"):
            outs = {}
            for mode in (0, 2):
                old = set_modes(mode, v5=0)
                try:
                    if kind == "store":
                        c = torch.empty(M, N, device=dev, dtype=bf)
                        ops.gemm(A, W, c, M, N, K, K, K, N, bias=bias)
                        o = (c,)
                    elif kind == "store_f32":
                        c = torch.empty(M, N, device=dev)
                        ops.gemm(A, W, c, M, N, K, K, K, N, bias=bias)
                        o = (c,)
                    elif kind == "swish":
                        h = torch.empty(M, N, device=dev, dtype=bf)
                        a = torch.empty(M, N, device=dev, dtype=bf)
                        ops.gemm(A, W, a, M, N, K, K, K, N, bias=bias, epi=ops.EPI_SWISH_DROP, aux_out=h, drop=drop)
                        o = (h, a)
                    else:
                        c = torch.empty(M, N, device=dev)
                        ops.gemm(A, W, c, M, N, K, K, K, N, bias=bias, alpha=0.5, epi=ops.EPI_RESID, aux_in=res, drop=drop)
                        o = (c,)
                    torch.cuda.synchronize()
                    outs[mode] = o
                finally:
                    restore(old)
            errs = [rel_err(x, y) for x, y in zip(outs[2], outs[0])]
            line = f"check M={M} N={N} K={K} {kind:9s} v8-vs-old rel {max(errs):.2e}"
            if kind in ("store", "store_f32"):
                e_ref = rel_err(outs[2][0], ref)
                e_old = rel_err(outs[0][0], ref)
                line += f"  vs fp32 product: v8 {e_ref:.2e} old {e_old:.2e}"
                if e_ref > max(2.0 * e_old, 1e-6):
                    bad += 1
                    line += "  <-- BAD"
            tol = 1e-5 if outs[2][0].dtype == torch.float32 and kind != "swish" else 6e-3
            if max(errs) > tol or not all(torch.isfinite(x.float()).all() for x in outs[2]):
                bad += 1
                line += "  <-- BAD"
            # race screen: the same launch again, several times, bit for bit
            if kind == "store_f32":
                old = set_modes(2, v5=0)
                try:
                    for rep in range(6):
                        c2 = torch.empty(M, N, device=dev)
                        ops.gemm(A, W, c2, M, N, K, K, K, N, bias=bias)
                        torch.cuda.synchronize()
                        if not torch.equal(c2, outs[2][0]):
                            bad += 1
                            line += f"  <-- RUN-TO-RUN DIFFERENCE (rep {rep}, {(c2 - outs[2][0]).abs().max().item():.3e})"
                            break
                finally:
                    restore(old)
            print(line, flush=True)
    print("CHECK", "FAILED" if bad else "ok", bad, flush=True)
    return bad


def timeit(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(ITERS):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / ITERS * 1e-3


def make(M, N, K, epi):
    g = torch.Generator(device=dev).manual_seed(0)
    fs = []
    for r in range(ROT):
        A = (torch.rand(M, K, device=dev, generator=g) * 2 - 1).to(bf)
        B = (torch.rand(N, K, device=dev, generator=g) * 2 - 1).to(bf)
        bias = torch.randn(N, device=dev, generator=g)
        if epi == "store":
            C = torch.empty(M, N, device=dev, dtype=bf)
            f = lambda A=A, B=B, C=C, bias=bias: ops.gemm(A, B, C, M, N, K, K, K, N, bias=bias)
        elif epi == "plain":
            C = torch.empty(M, N, device=dev, dtype=bf)
            f = lambda A=A, B=B, C=C: ops.gemm(A, B, C, M, N, K, K, K, N)
        elif epi == "swish":
            C = torch.empty(M, N, device=dev, dtype=bf)
            H = torch.empty(M, N, device=dev, dtype=bf)
            d = ops.Dropout(0.1, 1, 1)
            f = lambda A=A, B=B, C=C, bias=bias, H=H, d=d: ops.gemm(A, B, C, M, N, K, K, K, N, bias=bias, epi=6, aux_out=H, drop=d)
        elif epi == "resid":
            R = torch.randn(M, N, device=dev)
            C = torch.empty(M, N, device=dev)
            d = ops.Dropout(0.1, 1, 2)
            f = lambda A=A, B=B, C=C, bias=bias, R=R, d=d: ops.gemm(A, B, C, M, N, K, K, K, N, bias=bias, alpha=0.5, epi=ops.EPI_RESID, aux_in=R, drop=d)
        elif epi == "dswish":
            C = torch.empty(M, N, device=dev, dtype=bf)
            H = torch.randn(M, N, device=dev, generator=g).to(bf)
            f = lambda A=A, B=B, C=C, H=H: ops.gemm(A, B, C, M, N, K, K, K, N, epi=7, aux_in=H)
        fs.append(f)
    cnt = [0]

    def call():
        fs[cnt[0] % ROT]()
        cnt[0] += 1
    return call


def bench():
    M = 16032
    cases = [("sq8192", 8192, 8192, 8192, "plain"), ("sq4096", 4096, 4096, 4096, "plain"),
             ("ffn1_fwd_swish", M, 2048, 512, "swish"), ("ffn1_plain", M, 2048, 512, "plain"),
             ("ffn2_dgrad_dswish", M, 2048, 512, "dswish"),
             ("ffn2_fwd_resid", M, 512, 2048, "resid"), ("ffn2_plain", M, 512, 2048, "plain"),
             ("ffn1_dgrad_store", M, 512, 2048, "store"),
             ("qkv_fwd_store", M, 1536, 512, "store"), ("qkv_dgrad_store", M, 512, 1536, "store"),
             ("pw1_fwd_store", M, 1024, 512, "store"), ("pw1_dgrad", M, 512, 1024, "store"),
             ("proj_fwd_resid", M, 512, 512, "resid")]
    only = os.environ.get("ONLY")
    # arms: the shipped dispatch with key 8 off, and the new structure forced wherever it can run
    arms = [("old", dict(v8=0)), ("v8", dict(v8=2))]
    if os.environ.get("ARMS") == "half":   # the 128x256 tile where 256x256 tiles do not fill the chip
        arms = [("old", dict(v8=0)), ("v8+half", dict(v8=5))]
    if os.environ.get("ARMS") == "all":
        arms = [("old", dict(v8=0)), ("v4", dict(v8=0, v4=2, v5=0, v6=0)), ("v6", dict(v8=0, v4=2, v5=0, v6=1)), ("v8", dict(v8=2))]
    for name, M_, N_, K_, epi in cases:
        if only and only not in name:
            continue
        call = make(M_, N_, K_, epi)
        res = {a: [] for a, _ in arms}
        for rep in range(REPS):
            for a, kw in arms:
                old = set_modes(**kw)
                try:
                    res[a].append(timeit(call))
                finally:
                    restore(old)
        fl = 2.0 * M_ * N_ * K_
        txt = "  ".join(f"{a}: {min(v)*1e6:7.1f} us {fl/min(v)/1e12:7.1f} TF (med {sorted(v)[len(v)//2]*1e6:7.1f})" for a, v in res.items())
        print(f"bench {name:20s} M={M_:6d} N={N_:5d} K={K_:5d} {epi:6s} rot={ROT}  {txt}", flush=True)


def conv2():
    """conv2 of the 'striding' sub-sampling as the encoder issues it (implicit GEMM: gathered A rows, ReLU + time-mask epilogue) and
    one of its four input-gradient GEMMs (gathered dout2, row map, ReLU gate): v8 against the third structure, bit for bit, + time"""
    B_, T1, F1, C_ = 32, 1001, 40, 512
    if os.environ.get("CONV2_SMALL"):
        B_, T1, F1, C_ = 4, 301, 40, 256
    T2, F2 = (T1 - 1) // 2 + 1, (F1 - 1) // 2 + 1
    M2 = B_ * T2 * F2
    g = torch.Generator(device=dev).manual_seed(0)
    out1 = (torch.rand(B_, T1, F1, C_, device=dev, generator=g) * 2 - 1).to(bf)
    W = ((torch.rand(C_, 9 * C_, device=dev, generator=g) * 2 - 1) * 0.05).to(bf)
    bias = torch.randn(C_, device=dev, generator=g)
    len2 = torch.full((B_,), T2, device=dev, dtype=torch.int64)
    len2[1] = T2 // 2
    len2[2] = 5
    taps = [(kh - 1, kw - 1) for kh in range(3) for kw in range(3)]

    def fwd(out2):
        ops.gemm(out1, W, out2, M2, C_, 9 * C_, C_, 9 * C_, C_, bias=bias, epi=ops.EPI_RELU_MASK, row_len=len2,
                 rows_per_b=T2 * F2, rows_inner=F2, gather=dict(nI=T2, nJ=F2, SI=T1, SJ=F1, C=C_, si=2, sj=2, taps=taps))

    res, outs = {}, {}
    for rep in range(REPS):
        for arm, mode in (("old", 0), ("v8", 2)):
            old = set_modes(mode)
            try:
                out2 = torch.full((M2, C_), 7.0, device=dev, dtype=bf)
                fwd(out2)
                torch.cuda.synchronize()
                outs[arm] = out2
                res.setdefault(arm, []).append(timeit(lambda: fwd(out2)))
            finally:
                restore(old)
    same = torch.equal(outs["old"], outs["v8"])
    fl = 2.0 * M2 * C_ * 9 * C_
    print(f"conv2_fwd gathered M={M2} N={C_} K={9*C_}: bit-identical={same} (max diff {(outs['old'].float()-outs['v8'].float()).abs().max().item():.3e})  " +
          "  ".join(f"{a}: {min(v)*1e6:8.1f} us {fl/min(v)/1e12:7.1f} TF" for a, v in res.items()), flush=True)
    # input gradient, parity class (1, 1): taps k = 0, 2 on both axes -> 4 slots; rows scattered through the row map, gate = out1 > 0
    pt = pf = 1
    nI, nJ = (T1 - pt + 1) // 2, (F1 - pf + 1) // 2
    dout2 = (torch.rand(M2, C_, device=dev, generator=g) * 2 - 1).to(bf)
    slots = [(0, 0), (0, 1), (1, 0), (1, 1)]  # (di, dj) into the [T2, F2] grid of dout2 for the class's positions
    Wd = ((torch.rand(C_, len(slots) * C_, device=dev, generator=g) * 2 - 1) * 0.05).to(bf)

    def dgrad(dout1):
        ops.gemm(dout2, Wd, dout1, B_ * nI * nJ, C_, len(slots) * C_, C_, len(slots) * C_, C_, epi=ops.EPI_MUL_POS, aux_in=out1, ldaux=C_,
                 row_len=len2, rows_per_b=nI * nJ, rows_inner=nJ,
                 gather=dict(nI=nI, nJ=nJ, SI=T2, SJ=F2, C=C_, si=1, sj=1, taps=slots),
                 rowmap=dict(nI=nI, nJ=nJ, OI=T1, OJ=F1, si=2, sj=2, oi=pt, oj=pf))

    res, outs = {}, {}
    for rep in range(REPS):
        for arm, mode in (("old", 0), ("v8", 2)):
            old = set_modes(mode)
            try:
                d1 = torch.full((B_ * T1 * F1, C_), 3.0, device=dev, dtype=bf)
                dgrad(d1)
                torch.cuda.synchronize()
                outs[arm] = d1
                res.setdefault(arm, []).append(timeit(lambda: dgrad(d1)))
            finally:
                restore(old)
    same = torch.equal(outs["old"], outs["v8"])
    fl = 2.0 * B_ * nI * nJ * C_ * len(slots) * C_
    print(f"conv2_dgrad11 gathered+rowmap M={B_*nI*nJ} N={C_} K={len(slots)*C_}: bit-identical={same}  " +
          "  ".join(f"{a}: {min(v)*1e6:8.1f} us {fl/min(v)/1e12:7.1f} TF" for a, v in res.items()), flush=True)
    return 0 if same else 1


def tn():
    """weight-gradient layouts (both operands reduction-major, f32 atomic split-K, fused bias-gradient column sums): dense, grouped
    and the gathered conv2 weight gradient -- v8 against the older structures and against an fp32 product, + time"""
    bad = 0
    g = torch.Generator(device=dev).manual_seed(3)
    for (M, N, K, sk) in [(512, 2048, 16032, 4), (2048, 512, 16032, 4), (300, 520, 1000, 2), (512, 512, 16032, 8), (1024, 512, 777, 1)]:
        lda, ldb = (M + 7) // 8 * 8, (N + 7) // 8 * 8  # (pitches are multiples of 8 elements by contract; pad columns hold junk)
        dY = (torch.rand(K, lda, device=dev, generator=g) * 2 - 1).to(bf)
        X = (torch.rand(K, ldb, device=dev, generator=g) * 2 - 1).to(bf)
        ref = dY[:, :M].float().t() @ X[:, :N].float()
        refb = dY[:, :M].float().sum(0)
        outs, res = {}, {}
        for rep in range(REPS):
            for arm, mode in (("old", 0), ("v8", 2)):
                old = set_modes(mode)
                try:
                    dW = torch.zeros(M, N, device=dev)
                    db = torch.zeros(M, device=dev)
                    f = lambda: ops.gemm(dY, X, dW, M, N, K, lda, ldb, N, transA=True, transB=True, atomic=True, splitk=sk, c_dtype=ops.F32,
                                         colsum_out=db)
                    f()
                    torch.cuda.synchronize()
                    outs[arm] = (dW.clone(), db.clone())
                    res.setdefault(arm, []).append(timeit(f))
                finally:
                    restore(old)
        e8, eo = rel_err(outs["v8"][0], ref), rel_err(outs["old"][0], ref)
        b8, bo = rel_err(outs["v8"][1], refb), rel_err(outs["old"][1], refb)
        ok = e8 <= max(2 * eo, 2e-6) and b8 <= max(2 * bo, 2e-6)
        bad += 0 if ok else 1
        fl = 2.0 * M * N * K
        print(f"tn dense M={M} N={N} K={K} sk={sk}: dW rel err v8 {e8:.2e} old {eo:.2e}; colsum v8 {b8:.2e} old {bo:.2e} {'ok' if ok else '<-- BAD'}  " +
              "  ".join(f"{a}: {min(v)*1e6:8.1f} us {fl/min(v)/1e12:7.1f} TF" for a, v in res.items()), flush=True)
    # one Conformer layer's weight gradients as the grouped launch
    Mr, d, dff = 16032, 512, 2048
    mk = lambda n: (torch.rand(Mr, n, device=dev, generator=g) * 2 - 1).to(bf)
    shapes = [(d, dff), (dff, d), (d, dff), (dff, d), (d, d), (d, d), (d, d), (d, d), (2 * d, d), (d, d)]
    ops_ = [(mk(no), mk(ni)) for no, ni in shapes]
    outs, res = {}, {}
    for rep in range(REPS):
        for arm, mode in (("old", 0), ("v8", 3)):
            old = set_modes(mode)
            try:
                probs = []
                for (dY, X), (no, ni) in zip(ops_, shapes):
                    probs.append((dY, no, 0, X, ni, 0, torch.zeros(no, ni, device=dev), no, ni, torch.zeros(no, device=dev)))
                f = lambda: ops.wgrad_grouped(probs, Mr, 4)
                f()
                torch.cuda.synchronize()
                outs[arm] = [(q[6].clone(), q[9].clone()) for q in probs]
                res.setdefault(arm, []).append(timeit(f))
            finally:
                restore(old)
    worst = 0.0
    for (dY, X), (w8, b8), (wo, bo_) in zip(ops_, outs["v8"], outs["old"]):
        ref = dY.float().t() @ X.float()
        worst = max(worst, rel_err(w8, ref), rel_err(b8, dY.float().sum(0)))
    ok = worst < 5e-6
    bad += 0 if ok else 1
    fl = sum(2.0 * Mr * a * b for a, b in shapes)
    print(f"tn grouped layer x{len(shapes)}: worst rel err vs fp32 {worst:.2e} {'ok' if ok else '<-- BAD'}  " +
          "  ".join(f"{a}: {min(v)*1e6:8.1f} us {fl/min(v)/1e12:7.1f} TF" for a, v in res.items()), flush=True)
    # conv2 weight gradient (gathered reduction-major B, batch = taps, column-strided C)
    B_, T1, F1, C_ = 32, 1001, 40, 512
    T2, F2 = (T1 - 1) // 2 + 1, (F1 - 1) // 2 + 1
    M2 = B_ * T2 * F2
    dout2 = (torch.rand(M2, C_, device=dev, generator=g) * 2 - 1).to(bf)
    out1 = (torch.rand(B_, T1, F1, C_, device=dev, generator=g) * 2 - 1).to(bf)
    len2 = torch.full((B_,), T2, device=dev, dtype=torch.int64)
    len2[3] = T2 // 3
    dv = dout2.view(B_, T2, F2, C_)
    dv[3, T2 // 3:] = 0  # dY is masked beyond an utterance (what makes the K-tile skip legal)
    taps = [(kh - 1, kw - 1) for kh in range(3) for kw in range(3)]
    from nemo_amd.modules.conformer_encoder import ConformerEncoder as CE
    sk = CE._splitk(CE._tiles(C_, C_, True) * 9, M2)
    outs, res = {}, {}
    for rep in range(REPS):
        for arm, mode in (("old", 0), ("v8", 3)):
            old = set_modes(mode)
            try:
                dW = torch.zeros(C_, C_, 3, 3, device=dev)
                f = lambda: ops.gemm(dout2, out1, dW, C_, C_, M2, C_, C_, 9 * C_, transA=True, transB=True, atomic=True, splitk=sk, batch=9,
                                     nb0=9, sC=(1, 0), c_col_stride=9, c_dtype=ops.F32, row_len=len2, rows_per_b=T2 * F2, rows_inner=F2,
                                     gather=dict(operand=1, nI=T2, nJ=F2, SI=T1, SJ=F1, C=C_, si=2, sj=2, taps=taps))
                f()
                torch.cuda.synchronize()
                outs[arm] = dW.clone()
                res.setdefault(arm, []).append(timeit(f))
            finally:
                restore(old)
    e = rel_err(outs["v8"], outs["old"])
    ok = e < 5e-6
    bad += 0 if ok else 1
    fl = 2.0 * M2 * C_ * 9 * C_
    print(f"tn conv2_wgrad gathered splitk={sk}: v8 vs old rel {e:.2e} {'ok' if ok else '<-- BAD'}  " +
          "  ".join(f"{a}: {min(v)*1e6:8.1f} us {fl/min(v)/1e12:7.1f} TF" for a, v in res.items()), flush=True)
    return bad


if __name__ == "__main__":
    rc = 0
    if os.environ.get("CHECK", "1") != "0":
        rc = check()
    if os.environ.get("CONV2", "1") != "0":
        rc += conv2()
    if os.environ.get("TN", "1") != "0":
        rc += tn()
    if os.environ.get("BENCH", "1") != "0":
        bench()
    sys.exit(1 if rc else 0)
