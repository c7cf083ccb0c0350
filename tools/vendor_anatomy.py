"""Calibration only (never on the product path): the vendor library's kernels (hipBLASLt through torch.matmul) next to mi355x_gemm
on the three K = 512 / 2048 shapes VERDICT r4 item 1 names, warm (one operand set) and cold (ROTATE independent sets, > the
32 MiB of L2 + 256 MiB of Infinity Cache between two uses of one set).  Run under `rocprofv3 --kernel-trace` to get the
`Cijk_...` solution names (macro-tile, depth-U, wave tiling ... are spelled out in them) and under `--pmc` for their counters."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from nemo_amd import ops

dev, bf = "cuda", torch.bfloat16
ITERS = int(os.environ.get("ITERS", "24"))
ROT = int(os.environ.get("ROTATE", "8"))
WHO = os.environ.get("WHO", "lib,own").split(",")


def timeit(fs):
    n = len(fs)
    for i in range(max(3, n)):
        fs[i % n]()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(ITERS):
        fs[i % n]()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / ITERS * 1e-3


for name, M, N, K in [("ffn1_fwd", 16032, 2048, 512), ("ffn2_fwd", 16032, 512, 2048), ("proj", 16032, 512, 512),
                      ("qkv_fwd", 16032, 1536, 512)]:
    sets = []
    for r in range(ROT):
        A = torch.randn(M, K, device=dev).to(bf)
        B = torch.randn(N, K, device=dev).to(bf)
        C = torch.empty(M, N, device=dev, dtype=bf)
        sets.append((A, B, C))
    fl = 2.0 * M * N * K
    row = [f"{name:9s} {M}x{N}x{K}"]
    for who in WHO:
        if who == "lib":
            mk = lambda A, B, C: (lambda: torch.matmul(A, B.t(), out=C))
        else:
            mk = lambda A, B, C: (lambda: ops.gemm(A, B, C, M, N, K, K, K, N))
        warm = timeit([mk(*sets[0])])
        cold = timeit([mk(*s) for s in sets])
        row.append(f"{who}: warm {warm*1e6:7.1f} us {fl/warm/1e12:6.1f} TF  cold {cold*1e6:7.1f} us {fl/cold/1e12:6.1f} TF")
    print(" | ".join(row), flush=True)
