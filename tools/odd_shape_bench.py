"""Which dimension makes the Squeezeformer-Medium GEMMs slow (d = 324)?  mi355x_gemm on N / K / pitch variations of its two
feed-forward shapes, plain bf16 store and the f32 residual epilogue, HIP events, rotating operands.
    python tools/odd_shape_bench.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from nemo_amd import ops

dev, bf = "cuda", torch.bfloat16
ROT, ITERS = 4, 20


def timeit(fs):
    for f in fs: f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(ITERS): fs[i % len(fs)]()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / ITERS * 1e-3


def case(tag, M, N, K, lda, ldb, ldc, cdt=bf, epi="store"):
    fs = []
    for _ in range(ROT):
        A = torch.randn(M, lda, device=dev).to(bf); B = torch.randn(N, ldb, device=dev).to(bf)
        C = torch.zeros(M, ldc, device=dev, dtype=cdt); bias = torch.randn(N + 8, device=dev)[:N]
        if epi == "store":
            fs.append(lambda A=A, B=B, C=C, bias=bias: ops.gemm(A, B, C, M, N, K, lda, ldb, ldc, bias=bias))
        else:
            R = torch.randn(M, ldc, device=dev)
            fs.append(lambda A=A, B=B, C=C, bias=bias, R=R: ops.gemm(A, B, C, M, N, K, lda, ldb, ldc, bias=bias, alpha=0.5,
                                                                      epi=ops.EPI_RESID, aux_in=R, ldaux=ldc))
    t = timeit(fs)
    print(f"{tag:44s} M={M} N={N:5d} K={K:5d} lda={lda:5d} ldb={ldb:5d} ldc={ldc:5d} {str(cdt)[6:]:9s} {epi:6s} {t*1e6:8.1f} us {2.0*M*N*K/t/1e12:7.1f} TFLOP/s", flush=True)


M = 16032
for rep in range(2):
    case("FFN2 as the encoder issues it?  N=324 ldc=328", M, 324, 1296, 1296, 1296, 328)
    case("FFN2 N=324 ldc=324 (unaligned rows)", M, 324, 1296, 1296, 1296, 324)
    case("FFN2 N=328 ldc=328", M, 328, 1296, 1296, 1296, 328)
    case("FFN2 N=320 ldc=328", M, 320, 1296, 1296, 1296, 328)
    case("FFN2 N=384 ldc=384", M, 384, 1296, 1296, 1296, 384)
    case("FFN2 N=384 K=1344", M, 384, 1344, 1344, 1344, 384)
    case("FFN2 f32 resid N=324 ldc=324", M, 324, 1296, 1296, 1296, 324, torch.float32, "resid")
    case("FFN2 f32 resid N=328 ldc=328", M, 328, 1296, 1296, 1296, 328, torch.float32, "resid")
    case("FFN2 f32 resid N=384 ldc=384", M, 384, 1296, 1296, 1296, 384, torch.float32, "resid")
    case("FFN1 N=1296 K=324 lda=328", M, 1296, 324, 328, 328, 1296)
    case("FFN1 N=1296 K=328 lda=328", M, 1296, 328, 328, 328, 1296)
    case("FFN1 N=1296 K=320 lda=328", M, 1296, 320, 328, 328, 1296)
    case("FFN1 N=1296 K=384 lda=384", M, 1296, 384, 384, 384, 1296)
    case("FFN1 N=1280 K=320 lda=320", M, 1280, 320, 320, 320, 1280)
    case("FFN1 N=1536 K=384 lda=384", M, 1536, 384, 384, 384, 1536)
