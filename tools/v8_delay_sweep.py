"""sweep of the eighth structure's phase offset (mi355x_gemm_config(9, ticks of 10 ns)) on the K = 512 shapes of the Conformer block"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nemo_amd import ops
sys.argv = sys.argv[:1]
import tools.v8_probe as P  # noqa: E402  (make / timeit)

M = 16032
cases = [("ffn1_fwd_swish", M, 2048, 512, "swish"), ("ffn2_dgrad_dswish", M, 2048, 512, "dswish"), ("ffn1_plain", M, 2048, 512, "plain"),
         ("qkv_fwd_store", M, 1536, 512, "store")]
delays = [int(x) for x in os.environ.get("DELAYS", "0,200,400,600,800,1000,1200,1500").split(",")]
ops.gemm_config(8, 2)
for name, M_, N_, K_, epi in cases:
    call = P.make(M_, N_, K_, epi)
    out = []
    for rep in range(2):
        for d in delays:
            ops.gemm_config(9, d)
            t = P.timeit(call)
            if rep:
                out.append(f"{d * 10 / 1e3:4.1f}us:{t * 1e6:6.1f}")
    ops.gemm_config(9, 0)
    print(f"{name:20s} rot={P.ROT} " + "  ".join(out), flush=True)
