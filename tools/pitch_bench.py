"""does the row pitch of C matter for the GEMM epilogue?  (L2-channel hot-spotting hypothesis)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nemo_amd import ops
bf = torch.bfloat16
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e6
M, N, K = 16032, 2048, 512
A = torch.randn(M, K, device="cuda").to(bf); B = torch.randn(N, K, device="cuda").to(bf)
for pad in (0, 8, 32, 64, 128, 256, 512):
    ldc = N + pad
    C = torch.empty(M, ldc, device="cuda", dtype=bf)
    us = t(lambda: ops.gemm(A, B, C, M, N, K, K, K, ldc))
    print(f"N={N} K={K} ldc={ldc:5d} ({ldc*2} B)  {us:7.1f} us  {2.0*M*N*K/us/1e6:7.1f} TFLOP/s")
for padk in (0, 8, 32, 64):
    lda = K + padk
    A2 = torch.randn(M, lda, device="cuda").to(bf); B2 = torch.randn(N, lda, device="cuda").to(bf)
    C = torch.empty(M, N + 64, device="cuda", dtype=bf)
    us = t(lambda: ops.gemm(A2, B2, C, M, N, K, lda, lda, N + 64))
    print(f"N={N} K={K} lda=ldb={lda:5d} ldc={N+64}  {us:7.1f} us  {2.0*M*N*K/us/1e6:7.1f} TFLOP/s")
