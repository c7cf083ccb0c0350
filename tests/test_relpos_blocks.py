"""The score-gradient hand-over format between the dQ kernel and the linear_pos gradient kernel (un-shifted `matrix_bd`
blocks, include/mi355x_asr.h): CPU checks of the layout restatement (oracle/relpos_blocks_ref.py) against the definition, of
the C-ABI size helpers, and of the ISA lint that guards the inline-asm loads; the GPU check compares the blocks the dQ kernel
writes through the C ABI with that restatement fed by autograd's dS."""
import math
import os
import subprocess
import sys
import tempfile

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import relpos_blocks_ref as RB  # noqa: E402


@pytest.mark.parametrize("T", [1, 5, 32, 33, 45, 70])
def test_block_layout_reproduces_the_definition(T):
    rng = np.random.default_rng(T)
    B, H, dk = 3, 2, 4
    lens = [T, max(1, T // 2 + 3) if T > 6 else T, 1]
    dS = np.zeros((B, H, T, T))
    for b in range(B):
        L = min(T, lens[b])
        dS[b, :, :L, :L] = rng.standard_normal((H, L, L))
    qv = rng.standard_normal((B, T, H, dk))
    X, written = RB.ds_to_blocks(dS, lens)
    got = RB.dpos_from_blocks(X, written, qv, T)
    want = RB.dpos_direct(dS, qv)
    assert np.abs(got - want).max() <= 1e-12 * max(1.0, np.abs(want).max())
    # every non-zero of dS sits in exactly one written block
    assert np.isclose(np.abs(X).sum(), np.abs(dS).sum())


def test_size_helpers_of_the_c_abi_match_the_layout():
    from nemo_amd._lib import lib
    for B, H, T in [(1, 1, 1), (3, 2, 45), (32, 8, 501), (2, 4, 512), (5, 3, 513)]:
        assert lib.mi355x_relpos_ds_elems(B, H, T) == RB.ds_elems(B, H, T)
        nT = RB.n_tiles(T)
        bchunk = 4 if B >= 8 else 1
        # (64 positions x the widest fused head, d_k' = 128: one scratch size for both head widths)
        assert lib.mi355x_relpos_dpos_partial_elems(B, H, T) == ((B + bchunk - 1) // bchunk) * nT * H * 64 * 128


CLEAN = """
_Z1kv:
\t;;#ASMSTART
\tds_read_b64_tr_b16 v[10:11], v5 offset:0
\t;;#ASMEND
\tv_add_u32_e32 v1, v2, v3
\t;;#ASMSTART
\ts_waitcnt lgkmcnt(0)
\t;;#ASMEND
\tv_mfma_f32_32x32x16_bf16 v[20:35], v[10:13], v[14:17], v[20:35]
\ts_endpgm
"""
DIRTY = CLEAN.replace("v_add_u32_e32 v1, v2, v3", "v_mov_b32_e32 v40, v10")  # copies the destination before the wait


@pytest.mark.parametrize("text,rc", [(CLEAN, 0), (DIRTY, 1)])
def test_isa_lint_flags_a_copy_of_an_unwaited_asm_load(text, rc):
    with tempfile.NamedTemporaryFile("w", suffix=".s", delete=False) as f:
        f.write(text)
    try:
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_asm_loads.py"), f.name], capture_output=True, text=True)
        assert r.returncode == rc, r.stdout
    finally:
        os.unlink(f.name)


@pytest.mark.gpu
@pytest.mark.parametrize("T", [45, 70])
def test_dq_kernel_writes_the_unshifted_blocks(T):
    from nemo_amd import ops as o
    dev = torch.device("cuda", 0)
    B, H, dk = 3, 2, 64
    d = H * dk
    g = torch.Generator().manual_seed(40 + T)
    bf = lambda x: x.to(torch.bfloat16).float()
    qkv = bf(torch.randn(B * T, 3 * d, generator=g) * 0.7)
    pos = bf(torch.randn(2 * T - 1, d, generator=g) * 0.7)
    u = torch.randn(d, generator=g) * 0.3; v = torch.randn(d, generator=g) * 0.3
    lens = torch.tensor([T, max(1, T // 2 + 3), 1])
    dO = bf(torch.randn(B * T, d, generator=g))
    scale = 1.0 / math.sqrt(dk)
    # fp32 reference with autograd; the gradient w.r.t. (ac + bd) is the kernels' dS
    q = qkv[:, :d].view(B, T, H, dk)
    qu = bf(q + u.view(H, dk)); qv = bf(q + v.view(H, dk))
    k = qkv[:, d:2 * d].view(B, T, H, dk); vv = qkv[:, 2 * d:].view(B, T, H, dk)
    p = pos.view(2 * T - 1, H, dk)
    ac = qu.transpose(1, 2) @ k.transpose(1, 2).transpose(-1, -2)
    bdf = qv.transpose(1, 2) @ p.transpose(0, 1).transpose(-1, -2).unsqueeze(0)
    ii = torch.arange(T)[:, None]; jj = torch.arange(T)[None]
    pre = (ac + bdf[:, :, ii, T - 1 + jj - ii]).requires_grad_(True)
    valid = torch.arange(T)[None] < lens[:, None]
    masked = ~(valid[:, :, None] & valid[:, None, :])[:, None]
    attn = torch.softmax((pre * scale).masked_fill(masked, -10000.0), -1).masked_fill(masked, 0.0)
    ctx_ref = (attn @ vv.transpose(1, 2)).transpose(1, 2).reshape(B * T, d)
    ctx_ref.backward(dO)
    dS_ref = pre.grad.masked_fill(masked, 0.0).numpy().astype(np.float64)  # [B,H,T,T]
    X_ref, written = RB.ds_to_blocks(dS_ref, lens.tolist())

    Tp = (T + 7) // 8 * 8
    qkv_d, pos_d, lens_d, dO_d = (t.to(dev) for t in (qkv.to(torch.bfloat16), pos.to(torch.bfloat16), lens, dO.to(torch.bfloat16)))
    ctx = torch.empty(B * T, d, device=dev, dtype=torch.bfloat16); lse = torch.zeros(B, H, T, device=dev)
    o.relpos_flash_fwd(qkv_d, 3 * d, pos_d, d, u.to(dev), v.to(dev), lens_d, ctx, d, lse, B, H, T, dk, Tp, scale)
    qu_d = torch.empty(B * T, d, device=dev, dtype=torch.bfloat16); qv_d = torch.empty_like(qu_d)
    o.qbias(qkv_d, 3 * d, u.to(dev), v.to(dev), qu_d, qv_d, B * T, d)
    delta = torch.zeros(B, H, T, device=dev)
    o.attn_delta(dO_d, ctx, delta, B, H, T, d)
    dqu = torch.empty_like(qu_d); dqv = torch.empty_like(qu_d)
    dS = o.relpos_ds_buffer(B, H, T, dev, fill=float("nan"))
    o.relpos_flash_bwd_dq(qu_d, qv_d, qkv_d, 3 * d, pos_d, d, lens_d, dO_d, lse, delta, dqu, dqv, B, H, T, dk, scale, ds_out=dS)
    torch.cuda.synchronize()
    nT = RB.n_tiles(T)
    X = dS.float().cpu().numpy().reshape(H, B, nT, nT + 1, 32, 32)
    w = written[..., None, None] & np.ones((32, 32), dtype=bool)
    assert np.all(np.isfinite(X[w])), "a block the gradient kernel reads was not written"
    assert np.all(np.isnan(X[~w])), "blocks past ceil(len/32) are never touched"
    err = np.linalg.norm(X[w] - X_ref[w]) / np.linalg.norm(X_ref[w])
    print(f"[relpos blocks T={T}] dS blocks vs autograd: relative L2 error {err:.2e}")
    assert err < 5e-3, err  # (measured 1.7e-3: bf16 storage of dS)
    # and the gradient kernel agrees with the plain product on the kernel's own blocks
    dp = torch.zeros(2 * T - 1, d, device=dev)
    o.relpos_flash_bwd_dpos(qv_d, dS, lens_d, dp, B, H, T, dk)
    torch.cuda.synchronize()
    want = RB.dpos_from_blocks(np.where(w, X, 0.0), written, qv.numpy(), T).reshape(2 * T - 1, d)
    got = dp.cpu().numpy()
    e2 = np.linalg.norm(got - want) / np.linalg.norm(want)
    print(f"[relpos blocks T={T}] gradient kernel vs plain product on its own blocks: {e2:.2e}")
    assert e2 < 1e-5, e2  # (measured 4e-8: the same products in f32 instead of f64)
