"""TEST INFRASTRUCTURE ONLY -- numpy restatement of the score-gradient layout that `mi355x_relpos_flash_bwd_dq` hands to
`mi355x_relpos_flash_bwd_dpos` (include/mi355x_asr.h), and of the linear_pos gradient formed from it.

The reference computes `matrix_bd = q_with_bias_v @ p^T` as a [B,H,T,2T-1] tensor and then `rel_shift`s it onto the score
matrix (nemo/collections/asr/parts/submodules/multi_head_attention.py:259-270, 296-300); the gradient w.r.t. the UN-shifted
matrix is therefore dS[b,h,i,j] placed at position c = T-1+j-i, and

    d pos[c, h, :] = sum_{b,i} dS[b,h,i, c-(T-1)+i] * (q[b,i,h,:] + pos_bias_v[h,:]).

The kernels cut that un-shifted matrix into 32 x 32 blocks per (head, utterance, query tile `it`):
    X[h][b][it][s][q][cl] = dS[b,h, i = 32*it+q, j]   at   c = T-1+j-i = T-32 + 32*(s-it) + cl,   s = 0 .. ceil(T/32),
slots s <= ceil(len[b]/32) are written (everything else is never read).  Nothing under nemo_amd/ imports this file.
"""
from __future__ import annotations

import numpy as np


def n_tiles(T: int) -> int:
    return (T + 31) // 32


def ds_elems(B: int, H: int, T: int) -> int:
    nT = n_tiles(T)
    return H * B * nT * (nT + 1) * 1024


def ds_to_blocks(dS: np.ndarray, lens) -> tuple[np.ndarray, np.ndarray]:
    """dS [B,H,T,T] (zero outside the valid [len, len] corner) -> (X [H,B,nT,nT+1,32,32], written [H,B,nT,nT+1] bool)"""
    B, H, T, _ = dS.shape
    nT = n_tiles(T)
    X = np.zeros((H, B, nT, nT + 1, 32, 32), dtype=np.float64)
    written = np.zeros((H, B, nT, nT + 1), dtype=bool)
    for b in range(B):
        L = min(T, int(lens[b]))
        nkt = (L + 31) // 32
        written[:, b, :, : nkt + 1] = True
        for i in range(L):
            it, q = divmod(i, 32)
            for j in range(L):
                c = T - 1 + j - i
                rel = c - (T - 32) + 32 * it  # = 32 * s + cl
                s, cl = divmod(rel, 32)
                assert 0 <= s <= nkt
                X[:, b, it, s, q, cl] = dS[b, :, i, j]
    return X, written


def dpos_from_blocks(X: np.ndarray, written: np.ndarray, qv: np.ndarray, T: int) -> np.ndarray:
    """X as above, qv [B,T,H,dk] -> d pos [2T-1, H, dk]: a plain product per block, blocks that were not written are skipped"""
    H, B, nT = X.shape[0], X.shape[1], X.shape[2]
    dk = qv.shape[-1]
    out = np.zeros((2 * T - 1, H, dk), dtype=np.float64)
    for h in range(H):
        for b in range(B):
            for it in range(nT):
                rows = np.minimum(32 * it + np.arange(32), T - 1)  # the kernels clamp rows past T (their X rows are zero)
                q_t = qv[b, rows, h, :].astype(np.float64)  # [32, dk]
                for s in range(nT + 1):
                    if not written[h, b, it, s]:
                        continue
                    c0 = T - 32 + 32 * (s - it)
                    blk = X[h, b, it, s].T @ q_t  # [cl, dk]
                    for cl in range(32):
                        c = c0 + cl
                        if 0 <= c < 2 * T - 1:
                            out[c, h] += blk[cl]
                        else:
                            assert not np.any(X[h, b, it, s][:, cl])
    return out


def dpos_direct(dS: np.ndarray, qv: np.ndarray) -> np.ndarray:
    """the definition: d pos[c,h,:] = sum_{b,i,j: c = T-1+j-i} dS[b,h,i,j] * qv[b,i,h,:]"""
    B, H, T, _ = dS.shape
    out = np.zeros((2 * T - 1, H, qv.shape[-1]), dtype=np.float64)
    for i in range(T):
        for j in range(T):
            out[T - 1 + j - i] += np.einsum("bh,bhd->hd", dS[:, :, i, j], qv[:, i])
    return out
