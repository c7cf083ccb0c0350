"""micro-benchmark of mi355x_gemm on the Conformer-CTC-Large shapes (per-launch HIP-event timing)"""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nemo_amd import ops

dev = "cuda"
bf = torch.bfloat16
iters = int(os.environ.get("ITERS", "20"))
only = os.environ.get("ONLY")

def timeit(fn):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3

ROT = int(os.environ.get("ROTATE", "1"))  # > 1: cycle over that many independent operand / output sets, so that a launch finds
# its operands where the training step finds them (HBM / MALL, not an L2 that the previous launch of the same buffers warmed)


def run(name, M, N, K, epi="store", out=bf, layout="NT", splitk=1):
    if only and only not in name: return
    g = torch.Generator(device=dev).manual_seed(0)
    tA, tB = layout[0] == "T", layout[1] == "N"
    kw = dict(transA=tA, transB=tB)
    lda = M if tA else K; ldb = N if tB else K
    fs = []
    for r in range(ROT):
        A = torch.randn((K, M) if tA else (M, K), device=dev, generator=g).to(bf)
        B = torch.randn((K, N) if tB else (N, K), device=dev, generator=g).to(bf)
        C = torch.zeros(M, N, device=dev, dtype=out)
        bias = torch.randn(N, device=dev, generator=g)
        if epi == "store": f = lambda A=A, B=B, C=C, bias=bias: ops.gemm(A, B, C, M, N, K, lda, ldb, N, bias=bias, **kw)
        elif epi == "nobias": f = lambda A=A, B=B, C=C: ops.gemm(A, B, C, M, N, K, lda, ldb, N, **kw)
        elif epi == "swish":
            H = torch.empty(M, N, device=dev, dtype=bf); d = ops.Dropout(0.1, 1, 1)
            f = lambda A=A, B=B, C=C, bias=bias, H=H, d=d: ops.gemm(A, B, C, M, N, K, lda, ldb, N, bias=bias, epi=ops.EPI_SWISH_DROP, aux_out=H, drop=d, **kw)
        elif epi == "resid":
            R = torch.randn(M, N, device=dev); C = torch.empty(M, N, device=dev); d = ops.Dropout(0.1, 1, 2)
            f = lambda A=A, B=B, C=C, bias=bias, R=R, d=d: ops.gemm(A, B, C, M, N, K, lda, ldb, N, bias=bias, alpha=0.5, epi=ops.EPI_RESID, aux_in=R, drop=d, **kw)
        elif epi == "dswish":
            H = torch.randn(M, N, device=dev, generator=g).to(bf); d = ops.Dropout(0.1, 1, 3)
            f = lambda A=A, B=B, C=C, H=H, d=d: ops.gemm(A, B, C, M, N, K, lda, ldb, N, epi=ops.EPI_DSWISH, aux_in=H, drop=d, **kw)
        elif epi == "mulpos":
            H = torch.randn(M, N, device=dev, generator=g).to(bf)
            f = lambda A=A, B=B, C=C, H=H: ops.gemm(A, B, C, M, N, K, lda, ldb, N, epi=ops.EPI_MUL_POS, aux_in=H, **kw)
        elif epi == "atomic":
            C = torch.zeros(M, N, device=dev)
            f = lambda A=A, B=B, C=C: ops.gemm(A, B, C, M, N, K, lda, ldb, N, atomic=True, splitk=splitk, **kw)
        fs.append(f)
    cnt = [0]

    def call():
        fs[cnt[0] % ROT]()
        cnt[0] += 1
    t = timeit(call)
    print(f"{name:34s} {layout} M={M:7d} N={N:5d} K={K:6d} epi={epi:7s} {t*1e6:9.1f} us  {2.0*M*N*K/t/1e12:8.1f} TFLOP/s", flush=True)

def run_grouped(name, M=16032, d=512, dff=2048, splitk=4):
    """one Conformer layer's ten weight gradients as ONE grouped TN launch (what the backward sequencer issues per layer)"""
    if only and only not in name: return
    g = torch.Generator(device=dev).manual_seed(0)
    mk = lambda n: torch.randn(M, n, device=dev, generator=g).to(bf)
    shapes = [(d, dff), (dff, d), (d, dff), (dff, d), (d, d), (d, d), (d, d), (d, d), (2 * d, d), (d, d)]  # (n_out, n_in)
    probs = []
    for n_out, n_in in shapes:
        dY, X = mk(n_out), mk(n_in)
        dW = torch.zeros(n_out, n_in, device=dev)
        db = torch.zeros(n_out, device=dev)
        probs.append((dY, n_out, 0, X, n_in, 0, dW, n_out, n_in, db))
    t = timeit(lambda: ops.wgrad_grouped(probs, M, splitk))
    fl = sum(2.0 * M * a * b for a, b in shapes)
    print(f"{name:34s} TN grouped x{len(shapes)} M={M} splitk={splitk} {t*1e6:9.1f} us  {fl/t/1e12:8.1f} TFLOP/s", flush=True)


M = 16032
if os.environ.get("WGRAD_AB"):
    for rep in range(int(os.environ.get("REPS", "3"))):
        run_grouped("layer_wgrad_grouped_sk4", splitk=4)
        run("conv2_wgrad_like", 512, 4608, 320640 // 4, "atomic", torch.float32, "TN", splitk=8)
        run("ffn1_fwd_swish", M, 2048, 512, "swish")
        run("ffn2_fwd_resid", M, 512, 2048, "resid")
        run("proj_fwd_resid", M, 512, 512, "resid")
        run("ffn1_dgrad_store", M, 512, 2048, "store")
    sys.exit(0)
if os.environ.get("CONV2_WGRAD"):  # the sub-sampling conv2 weight gradient exactly as the encoder issues it (gathered operand)
    from nemo_amd.modules.conformer_encoder import ConformerEncoder as CE
    B_, T1, F1, C_ = 32, 1001, 40, 512
    T2, F2 = (T1 - 1) // 2 + 1, (F1 - 1) // 2 + 1
    M2 = B_ * T2 * F2
    g = torch.Generator(device=dev).manual_seed(0)
    dout2 = torch.randn(M2, C_, device=dev, generator=g).to(bf)
    out1 = torch.randn(B_, T1, F1, C_, device=dev, generator=g).to(bf)
    dW = torch.zeros(C_, C_, 3, 3, device=dev)
    sk = CE._splitk(CE._tiles(C_, C_, True) * 9, M2)
    f = lambda: ops.gemm(dout2, out1, dW, C_, C_, M2, C_, C_, 9 * C_, transA=True, transB=True, atomic=True, splitk=sk, batch=9,
                         nb0=9, sC=(1, 0), c_col_stride=9, c_dtype=ops.F32,
                         gather=dict(operand=1, nI=T2, nJ=F2, SI=T1, SJ=F1, C=C_, si=2, sj=2,
                                     taps=[(kh - 1, kw - 1) for kh in range(3) for kw in range(3)]))
    for rep in range(int(os.environ.get("REPS", "3"))):
        t = timeit(f)
        print(f"conv2_wgrad_gathered splitk={sk} {t*1e6:9.1f} us  {2.0*M2*C_*9*C_/t/1e12:8.1f} TFLOP/s", flush=True)
    sys.exit(0)
if os.environ.get("JOINT"):  # the transducer joint's output layer at fused_batch_size 4: n = 4 * 251 * 61 rows, V+1 = 1025, J = 640
    n, V1, J = 4 * 251 * 61, 1025, 640
    V1p = 1032
    g = torch.Generator(device=dev).manual_seed(0)
    h = torch.randn(n, J, device=dev, generator=g).to(bf)
    w = torch.randn(V1, J, device=dev, generator=g).to(bf)
    wt = torch.randn(J, V1p, device=dev, generator=g).to(bf)
    logits = torch.empty(n, V1, device=dev)
    dlog = torch.randn(n, V1p, device=dev, generator=g).to(bf); dlog[:, V1:] = 0
    dh = torch.empty(n, J, device=dev, dtype=bf)
    gW = torch.zeros(V1, J, device=dev)
    for rep in range(2):
        t = timeit(lambda: ops.gemm(h, w, logits, n, V1, J, J, J, V1))
        print(f"joint fwd  NT {n}x{V1}x{J} f32 out ldc=1025  {t*1e6:8.1f} us {2.0*n*V1*J/t/1e12:7.1f} TF", flush=True)
        t = timeit(lambda: ops.gemm(dlog, wt, dh, n, J, V1p, V1p, V1p, J))
        print(f"joint dgrad NT {n}x{J}x{V1p}                  {t*1e6:8.1f} us {2.0*n*V1*J/t/1e12:7.1f} TF", flush=True)
        for sk in (4, 10, 16):
            t = timeit(lambda: ops.gemm(dlog, h, gW, V1, J, n, V1p, J, J, transA=True, transB=True, atomic=True, splitk=sk,
                                        c_dtype=ops.F32))
            print(f"joint wgrad TN {V1}x{J}x{n} splitk={sk:2d}        {t*1e6:8.1f} us {2.0*n*V1*J/t/1e12:7.1f} TF", flush=True)
        t = timeit(lambda: ops.gemm(dlog, h, gW, 1024, J, n, V1p, J, J, transA=True, transB=True, atomic=True, splitk=10,
                                    c_dtype=ops.F32))
        print(f"   (M = 1024 instead of 1025, splitk=10)        {t*1e6:8.1f} us", flush=True)
    sys.exit(0)
if os.environ.get("HALF_M"):  # FastConformer x8 (M = 8032 rows): do the 256-row tiles still fill the chip?
    Mh = 8032
    for rep in range(2):
        run("ffn2_fwd_resid_M8032", Mh, 512, 2048, "resid")
        run("ffn1_fwd_swish_M8032", Mh, 2048, 512, "swish")
        run("proj_resid_M8032", Mh, 512, 512, "resid")
        run("qkv_store_M8032", Mh, 1536, 512, "store")
        run("dgrad_store_M8032_k1536", Mh, 512, 1536, "store")
    sys.exit(0)
if os.environ.get("PMC_SHAPES"):  # two shapes for counter collection (structure chosen by the MI355X_GEMM_* environment)
    run("big_square_bias", 8192, 8192, 8192, "store")
    run("ffn1_fwd_swish", M, 2048, 512, "swish")
    sys.exit(0)
if os.environ.get("V5_AB"):  # persistent overlapped-epilogue structure on / off, same process, interleaved
    for rep in range(int(os.environ.get("REPS", "3"))):
        for mode in (0, 2):
            ops.gemm_config(5, mode)
            tag = f"v5={mode} "
            run(tag + "ffn1_fwd_swish", M, 2048, 512, "swish")
            run(tag + "ffn2_dgrad_dswish", M, 2048, 512, "dswish")
            run(tag + "qkv_fwd_store", M, 1536, 512, "store")
            run(tag + "pw1_fwd_store", M, 1024, 512, "store")
            run(tag + "resid_n1024_k2048", M, 1024, 2048, "resid")
            run(tag + "store_n2048_k2048", M, 2048, 2048, "store")
            run(tag + "pw1_dgrad_store_k1024", M, 512, 1024, "store")
            run(tag + "resid_n1024_k512", M, 1024, 512, "resid")
    sys.exit(0)
if os.environ.get("V6_AB"):  # register-prefetch 256x256 structure on / off (LDS-DMA), same process, interleaved; v5 off so that
    # every shape with enough 256x256 tiles takes the structure under test
    for rep in range(int(os.environ.get("REPS", "3"))):
        for v5, v6 in ((0, 0), (0, 2), (0, 1)):
            ops.gemm_config(5, v5); ops.gemm_config(6, v6)
            tag = f"v5={v5} v6={v6} "
            run(tag + "ffn1_fwd_swish", M, 2048, 512, "swish")
            run(tag + "ffn1_fwd_store", M, 2048, 512, "store")
            run(tag + "ffn2_dgrad_dswish", M, 2048, 512, "dswish")
            run(tag + "qkv_fwd_store", M, 1536, 512, "store")
            run(tag + "pw1_fwd_store", M, 1024, 512, "store")
            run(tag + "store_n2048_k2048", M, 2048, 2048, "store")
            run(tag + "big_square", 8192, 8192, 8192, "nobias")
            run(tag + "sq4096", 4096, 4096, 4096, "nobias")
    sys.exit(0)
if os.environ.get("COLD_AB"):  # the Conformer layer's NT shapes with rotating operand sets, every structure knob interleaved
    for rep in range(int(os.environ.get("REPS", "2"))):
        for v5, v6, v7 in ((1, 0, 0), (1, 1, 1), (0, 0, 0), (0, 1, 1)):
            ops.gemm_config(5, v5); ops.gemm_config(6, v6); ops.gemm_config(7, v7)
            tag = f"v5={v5} v6={v6} v7={v7} "
            run(tag + "ffn1_fwd_swish", M, 2048, 512, "swish")
            run(tag + "ffn2_dgrad_dswish", M, 2048, 512, "dswish")
            run(tag + "ffn2_fwd_resid", M, 512, 2048, "resid")
            run(tag + "ffn1_dgrad_store", M, 512, 2048, "store")
            run(tag + "proj_fwd_resid", M, 512, 512, "resid")
            run(tag + "proj_dgrad_store", M, 512, 512, "store")
            run(tag + "qkv_fwd_store", M, 1536, 512, "store")
            run(tag + "qkv_dgrad_store", M, 512, 1536, "store")
            run(tag + "pw1_fwd_store", M, 1024, 512, "store")
    sys.exit(0)
if os.environ.get("EPI_COST"):  # what the Swish-gradient epilogue's arithmetic costs: the same launch with a 2-op epilogue
    for rep in range(3):
        for v5 in (1, 0):
            ops.gemm_config(5, v5)
            run(f"v5={v5} dswish", M, 2048, 512, "dswish")
            run(f"v5={v5} mulpos", M, 2048, 512, "mulpos")
            run(f"v5={v5} store_nobias", M, 2048, 512, "nobias")
    sys.exit(0)
if os.environ.get("V6N_AB"):  # the N = 512 shapes (256x128 tiles): LDS-DMA / register prefetch / persistent structure
    for rep in range(int(os.environ.get("REPS", "2"))):
        for v5, v6 in ((0, 0), (0, 1), (1, 0), (1, 1)):
            ops.gemm_config(5, v5); ops.gemm_config(7, v6)
            tag = f"v5={v5} v6={v6} "
            run(tag + "ffn2_fwd_resid", M, 512, 2048, "resid")
            run(tag + "ffn1_dgrad_store", M, 512, 2048, "store")
            run(tag + "proj_fwd_resid", M, 512, 512, "resid")
            run(tag + "proj_dgrad_store", M, 512, 512, "store")
            run(tag + "qkv_dgrad_store", M, 512, 1536, "store")
            run(tag + "pw1_dgrad_store", M, 512, 1024, "store")
            run(tag + "ffn2_dgrad_dswish", M, 2048, 512, "dswish")
            run(tag + "ffn1_fwd_swish", M, 2048, 512, "swish")
    sys.exit(0)
run("ffn1_fwd", M, 2048, 512, "swish")
run("ffn1_fwd_store", M, 2048, 512, "store")
run("ffn1_fwd_nobias", M, 2048, 512, "nobias")
run("ffn2_fwd", M, 512, 2048, "resid")
run("ffn2_fwd_store", M, 512, 2048, "store")
run("qkv_fwd", M, 1536, 512, "store")
run("proj_fwd", M, 512, 512, "store")
run("conv2_fwd", 320640, 512, 4608, "store")
run("big_square", 8192, 8192, 8192, "nobias")
run("ffn1_wgrad", 2048, 512, M, "atomic", torch.float32, "TN", splitk=8)
run("ffn2_wgrad", 512, 2048, M, "atomic", torch.float32, "TN", splitk=8)
run("proj_wgrad", 512, 512, M, "atomic", torch.float32, "TN", splitk=32)
run("pv_like_NN", 4096, 4096, 4096, "nobias", bf, "NN")
run("tn_big_store", 4096, 4096, 4096, "nobias", bf, "TN")
run("ffn1_wgrad_sk4", 2048, 512, M, "atomic", torch.float32, "TN", splitk=4)
run("ffn1_wgrad_sk16", 2048, 512, M, "atomic", torch.float32, "TN", splitk=16)
run("ffn1_wgrad_sk1_store", 2048, 512, M, "nobias", torch.float32, "TN")
run("proj_wgrad_sk16", 512, 512, M, "atomic", torch.float32, "TN", splitk=16)
run("proj_wgrad_sk64", 512, 512, M, "atomic", torch.float32, "TN", splitk=64)
run("ffn1_dgrad_dswish", M, 2048, 512, "store")
run("conv2_dgrad", 320640, 4608, 512, "nobias")
run("qkv_wgrad_like", 1536, 512, M, "atomic", torch.float32, "TN", splitk=8)
run("conv2_wgrad_like", 512, 4608, 320640 // 4, "atomic", torch.float32, "TN", splitk=8)
