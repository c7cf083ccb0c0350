"""Calibration only (never on the product path): what the vendor library (hipBLASLt through torch.matmul) reaches on the same
box for the shapes of tools/gemm_bench.py -- the practical MFMA ceiling under the part's power limit, next to mi355x_gemm."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from nemo_amd import ops

dev, bf = "cuda", torch.bfloat16


def timeit(f, iters=20):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


for name, M, N, K in [("square_8192", 8192, 8192, 8192), ("square_4096", 4096, 4096, 4096), ("ffn1_fwd", 16032, 2048, 512),
                      ("ffn2_fwd", 16032, 512, 2048), ("qkv_fwd", 16032, 1536, 512), ("proj", 16032, 512, 512),
                      ("conv2_like", 320640, 512, 4608)]:
    A = torch.randn(M, K, device=dev).to(bf)
    B = torch.randn(N, K, device=dev).to(bf)
    C = torch.empty(M, N, device=dev, dtype=bf)
    t_lib = timeit(lambda: torch.matmul(A, B.t(), out=C))
    t_own = timeit(lambda: ops.gemm(A, B, C, M, N, K, K, K, N))
    fl = 2.0 * M * N * K
    print(f"{name:12s} M={M:7d} N={N:5d} K={K:5d}  hipBLASLt {t_lib*1e6:9.1f} us {fl/t_lib/1e12:7.1f} TF | mi355x_gemm {t_own*1e6:9.1f} us "
          f"{fl/t_own/1e12:7.1f} TF", flush=True)


# the FUSED launches against the library doing the same work in its own kernels (GEMM with bias epilogue + elementwise)
import torch.nn.functional as F  # noqa: E402

M, d, dff = 16032, 512, 2048
x = torch.randn(M, d, device=dev).to(bf)
w1 = torch.randn(dff, d, device=dev).to(bf); b1 = torch.randn(dff, device=dev).to(bf)
t_lib = timeit(lambda: F.dropout(F.silu(F.linear(x, w1, b1)), 0.1, True))
h = torch.empty(M, dff, device=dev, dtype=bf); a = torch.empty(M, dff, device=dev, dtype=bf)
b1f = b1.float()
dr = ops.Dropout(0.1, 1, 1)
t_own = timeit(lambda: ops.gemm(x, w1, a, M, dff, d, d, d, dff, bias=b1f, epi=ops.EPI_SWISH_DROP, aux_out=h, drop=dr))
print(f"FFN1 forward (Linear + bias + Swish + dropout; ours also stores the pre-activation): library {t_lib*1e6:.1f} us | fused "
      f"mi355x_gemm {t_own*1e6:.1f} us")
w2 = torch.randn(d, dff, device=dev).to(bf); b2 = torch.randn(d, device=dev).to(bf)
res = torch.randn(M, d, device=dev)
t_lib = timeit(lambda: res + 0.5 * F.dropout(F.linear(a, w2, b2), 0.1, True).float())
r = torch.empty(M, d, device=dev)
b2f = b2.float()
t_own = timeit(lambda: ops.gemm(a, w2, r, M, d, dff, dff, dff, d, bias=b2f, alpha=0.5, epi=ops.EPI_RESID, aux_in=res, drop=dr))
print(f"FFN2 forward (Linear + bias + dropout + 0.5 x + fp32 residual): library {t_lib*1e6:.1f} us | fused mi355x_gemm {t_own*1e6:.1f} us")

# weight-gradient (TN) shapes: dW[n_out, n_in] = dY[rows, n_out]^T @ X[rows, n_in], f32 output; ours with atomic split-K
print("TN (weight gradient) shapes, f32 result:")
for name, n_out, n_in, rows, sk in [("ffn_w1", 2048, 512, 16032, 8), ("ffn_w2", 512, 2048, 16032, 8), ("proj", 512, 512, 16032, 16),
                                    ("conv2_like", 512, 4608, 80160, 8)]:
    dY = torch.randn(rows, n_out, device=dev).to(bf)
    X = torch.randn(rows, n_in, device=dev).to(bf)
    out16 = torch.empty(n_out, n_in, device=dev, dtype=bf)
    dW = torch.zeros(n_out, n_in, device=dev)
    t_lib = timeit(lambda: torch.matmul(dY.t(), X, out=out16))
    t_own = timeit(lambda: ops.gemm(dY, X, dW, n_out, n_in, rows, n_out, n_in, n_in, transA=True, transB=True, atomic=True, splitk=sk,
                                    c_dtype=ops.F32))
    fl = 2.0 * n_out * n_in * rows
    print(f"{name:12s} {n_out:5d} x {n_in:5d} x {rows:6d}  hipBLASLt (bf16 out) {t_lib*1e6:9.1f} us {fl/t_lib/1e12:7.1f} TF | mi355x_gemm split-K {sk:2d} "
          f"{t_own*1e6:9.1f} us {fl/t_own/1e12:7.1f} TF", flush=True)
