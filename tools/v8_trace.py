"""timeline of the v8 GEMM's workgroups (needs a -DV8_TRACE build: python tools/ab_build.py trace=@gemm:-DV8_TRACE, then
MI355X_ASR_LIB=nemo_amd/lib_ab/libmi355x_asr_trace.so python tools/v8_trace.py): per workgroup entry / prologue end / K-loop end /
exit, on the chip-wide 100-MHz clock (timeline) and the shader clock (durations)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nemo_amd import ops
from nemo_amd._lib import lib

dev = "cuda"
bf = torch.bfloat16
fn = lib.mi355x_gemm_debug_trace
fn.argtypes = [C.c_void_p]
M = 16032
cases = [("ffn1_swish", M, 2048, 512, "swish"), ("ffn1_plain", M, 2048, 512, "plain"), ("pw1_store", M, 1024, 512, "plain"),
         ("sq4096", 4096, 4096, 4096, "plain")]
g = torch.Generator(device=dev).manual_seed(0)
ops.gemm_config(8, 2)
for name, M_, N_, K_, epi in cases:
    A = (torch.rand(M_, K_, device=dev, generator=g) * 2 - 1).to(bf)
    B = (torch.rand(N_, K_, device=dev, generator=g) * 2 - 1).to(bf)
    Cm = torch.empty(M_, N_, device=dev, dtype=bf)
    H = torch.empty(M_, N_, device=dev, dtype=bf)
    bias = torch.randn(N_, device=dev, generator=g)
    d = ops.Dropout(0.1, 1, 1)
    nwg = ((M_ + 255) // 256) * ((N_ + 255) // 256)
    trace = torch.zeros(nwg * 16, dtype=torch.int64, device=dev)
    if epi == "swish":
        f = lambda: ops.gemm(A, B, Cm, M_, N_, K_, K_, K_, N_, bias=bias, epi=6, aux_out=H, drop=d)
    else:
        f = lambda: ops.gemm(A, B, Cm, M_, N_, K_, K_, K_, N_)
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    fn(trace.data_ptr())
    f()
    torch.cuda.synchronize()
    fn(None)
    t = trace.cpu().view(nwg, 8, 2)
    real = t[:, :, 0].double() * 10.0   # ns
    clk = t[:, :, 1].double()
    t0 = real[:, 0].min()
    real = (real - t0) / 1e3            # us since the first workgroup's entry
    order = torch.argsort(real[:, 0])
    first = order[:256] if nwg > 256 else order
    second = order[256:]
    def stats(x):
        return f"mean {x.mean():7.2f} min {x.min():7.2f} max {x.max():7.2f}"
    print(f"== {name}: {nwg} workgroups, launch span {real[:, 3].max():.1f} us")
    for label, idx in (("first-round workgroups", first), ("later workgroups", second)):
        if len(idx) == 0:
            continue
        r, c = real[idx], clk[idx]
        print(f"  {label} ({len(idx)}): entry at {stats(r[:, 0])} us")
        print(f"     prologue   {stats(r[:, 1] - r[:, 0])} us   ({(c[:, 1] - c[:, 0]).mean():9.0f} clk)")
        print(f"     K loop     {stats(r[:, 2] - r[:, 1])} us   ({(c[:, 2] - c[:, 1]).mean():9.0f} clk)")
        print(f"     epilogue   {stats(r[:, 3] - r[:, 2])} us   ({(c[:, 3] - c[:, 2]).mean():9.0f} clk)")
        print(f"     exit at    {stats(r[:, 3])} us")
        print(f"     epilogue round 0: LDS writes done {(c[:, 4] - c[:, 2]).mean():7.0f} clk | barrier {(c[:, 5] - c[:, 4]).mean():7.0f} | read + arithmetic + "
              f"store issue {(c[:, 6] - c[:, 5]).mean():7.0f} | stores landed {(c[:, 7] - c[:, 6]).mean():7.0f}")
