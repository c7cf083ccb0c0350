"""GEMM-operand images of the fp32 master weights, produced by ONE `mi355x_pack_weights` launch per optimizer step.

The state-dict (the on-disk ABI) keeps the reference's shapes -- e.g. `pre_encode.conv.2.weight`
[C,C,3,3], `self_attn.linear_{q,k,v}.weight` separately, `pre_encode.out.weight` columns ordered (c, f) -- while the
MFMA kernels want K-contiguous bf16 operands: q|k|v concatenated, conv weights as [co][(kh,kw,ci)], `out` columns
ordered (f, c) to match the channels-last conv output, plus a transposed copy of every weight for the dgrad GEMMs.
A PackPlan records those index maps once; `run()` re-materialises all images from the current master weights.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Tuple

import numpy as np
import torch

from . import ops
from ._lib import PackEntry


def _pad8(n: int) -> int:
    return (n + 7) // 8 * 8


class PackPlan:
    def __init__(self, dtype: torch.dtype, device):
        self.dtype = dtype
        self.device = torch.device(device)
        self._images: Dict[str, Tuple[int, int, int]] = {}  # name -> (offset, rows, pitch)
        self._pending: List[tuple] = []
        self._size = 0
        self.arena = None
        self._table = None
        self._n = 0
        self._tiles = 0
        self._views: Dict[str, torch.Tensor] = {}
        self._ffn: List[tuple] = []
        self._ffn_table = None

    # ------------------------------------------------------------------ image declaration
    def new_image(self, name: str, rows: int, cols: int) -> None:
        pitch = _pad8(cols)
        self._images[name] = (self._size, rows, pitch)
        self._size += (rows * pitch + 63) // 64 * 64

    def add_block(self, name: str, src: torch.Tensor, rows: int, cols: int, *, row_off=0, col_off=0, nr2=1, nc2=1,
                  sr1=0, sr2=0, sc1=0, sc2=0) -> None:
        """dst[row_off + r, col_off + c] = src.flat[r1*sr1 + r2*sr2 + c1*sc1 + c2*sc2], r = r1*nr2 + r2, c = c1*nc2 + c2"""
        assert src.dtype == torch.float32 and src.is_contiguous()
        self._pending.append((name, src, rows, cols, row_off, col_off, nr2, nc2, sr1, sr2, sc1, sc2))

    # ------------------------------------------------------------------ convenience declarations
    def add_matrix(self, name: str, w: torch.Tensor, transpose: bool = False):
        """w [R, Cc] (any trailing singleton dims squeezed) -> image [R, pad8(Cc)] or, transposed, [Cc, pad8(R)]"""
        R, Cc = w.shape[0], int(np.prod(w.shape[1:]))
        if not transpose:
            self.new_image(name, R, Cc)
            self.add_block(name, w, R, Cc, nr2=1, nc2=1, sr1=Cc, sc1=1)
        else:
            self.new_image(name, Cc, R)
            self.add_block(name, w, Cc, R, nr2=1, nc2=1, sr1=1, sc1=Cc)
        return name

    def add_concat(self, name: str, ws: List[torch.Tensor], transpose: bool = False):
        """rows of several [R_i, Cc] matrices stacked ([sum R, Cc]); transposed image is [Cc, sum R]"""
        Cc = int(np.prod(ws[0].shape[1:]))
        Rt = sum(w.shape[0] for w in ws)
        if not transpose:
            self.new_image(name, Rt, Cc)
        else:
            self.new_image(name, Cc, Rt)
        off = 0
        for w in ws:
            R = w.shape[0]
            if not transpose:
                self.add_block(name, w, R, Cc, row_off=off, sr1=Cc, sc1=1)
            else:
                self.add_block(name, w, Cc, R, col_off=off, sr1=1, sc1=Cc)
            off += R
        return name

    def add_conv3x3(self, name: str, w: torch.Tensor, transpose: bool = False):
        """w [co, ci, 3, 3] -> [co, (kh,kw,ci)]   (k = (kh*3+kw)*ci_n + ci);  transposed: [(kh,kw,ci), co]"""
        co, ci = w.shape[0], w.shape[1]
        if not transpose:
            self.new_image(name, co, 9 * ci)
            self.add_block(name, w, co, 9 * ci, nr2=1, nc2=ci, sr1=ci * 9, sc1=1, sc2=9)
        else:
            self.new_image(name, 9 * ci, co)
            self.add_block(name, w, 9 * ci, co, nr2=ci, nc2=1, sr1=1, sr2=9, sc1=ci * 9)
        return name

    def add_fc_permuted(self, name: str, w: torch.Tensor, C_: int, F_: int, transpose: bool = False):
        """w [d, C*F] with column c*F + f  ->  [d, F*C] with column f*C + c;  transposed: [F*C, d]"""
        d = w.shape[0]
        if not transpose:
            self.new_image(name, d, F_ * C_)
            self.add_block(name, w, d, F_ * C_, nr2=1, nc2=C_, sr1=C_ * F_, sc1=1, sc2=F_)
        else:
            self.new_image(name, F_ * C_, d)
            self.add_block(name, w, F_ * C_, d, nr2=C_, nc2=1, sr1=1, sr2=F_, sc1=C_ * F_)
        return name

    # ---- images of the fused feed-forward kernels (csrc/ffn.hip): MFMA-fragment-major, in the order the kernel's steps consume
    # the weights; written by their own launch (mi355x_ffn_pack) straight from the fp32 master weights
    def add_ffn(self, prefix: str, w1: torch.Tensor, w2: torch.Tensor) -> None:
        """linear1.weight [dff, 512] and linear2.weight [512, dff] -> images `prefix.w1p` = k512(W1), `prefix.w1tp` = kchunk(W1^T),
        `prefix.w2p` = kchunk(W2), `prefix.w2tp` = k512(W2^T), each 512 * dff bf16 (see include/mi355x_asr.h)"""
        dff = w1.shape[0]
        assert self.dtype == torch.bfloat16 and tuple(w1.shape) == (dff, 512) and tuple(w2.shape) == (512, dff) and dff % 64 == 0
        assert w1.dtype == torch.float32 and w2.dtype == torch.float32 and w1.is_contiguous() and w2.is_contiguous()
        for nm in ("w1p", "w1tp", "w2p", "w2tp"):
            self.new_image(f"{prefix}.{nm}", dff // 2, 1024)   # (the view's shape is irrelevant: the kernels take the flat image)
        self._ffn.append((prefix, w1, w2, dff))

    # ------------------------------------------------------------------ materialisation
    def finalize(self) -> None:
        self.arena = torch.zeros(max(self._size, 64), dtype=self.dtype, device=self.device)
        es = self.arena.element_size()
        base = self.arena.data_ptr()
        entries = (PackEntry * len(self._pending))()
        tiles = 0
        for i, (name, src, rows, cols, ro, co, nr2, nc2, sr1, sr2, sc1, sc2) in enumerate(self._pending):
            off, irows, pitch = self._images[name]
            assert ro + rows <= irows and co + cols <= pitch, name
            e = entries[i]
            e.src = src.data_ptr()
            assert co % 8 == 0, name  # the kernel writes whole 8-column groups (16 / 32-byte stores)
            e.dst = base + (off + ro * pitch + co) * es
            e.rows, e.cols, e.nr2, e.nc2 = rows, cols, nr2, nc2
            e.sr1, e.sr2, e.sc1, e.sc2, e.pitch = sr1, sr2, sc1, sc2, pitch
            e.tile_begin = tiles
            tiles += ((rows + 63) // 64) * ((cols + 63) // 64)
        raw = bytes(entries)
        self._table = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(self.device) if raw else None
        self._n, self._tiles = len(self._pending), tiles
        self._srcs = [p[1] for p in self._pending]  # keep sources alive
        if self._ffn:
            from ._lib import FfnPackEntry
            fe = (FfnPackEntry * (2 * len(self._ffn)))()
            for i, (prefix, w1, w2, dff) in enumerate(self._ffn):
                off = {nm: self._images[f"{prefix}.{nm}"][0] for nm in ("w1p", "w1tp", "w2p", "w2tp")}
                a, b = fe[2 * i], fe[2 * i + 1]
                a.src, a.k512, a.kchunk, a.d_ff, a.is_w2 = w1.data_ptr(), base + off["w1p"] * es, base + off["w1tp"] * es, dff, 0
                b.src, b.k512, b.kchunk, b.d_ff, b.is_w2 = w2.data_ptr(), base + off["w2tp"] * es, base + off["w2p"] * es, dff, 1
            self._ffn_table = torch.frombuffer(bytearray(bytes(fe)), dtype=torch.uint8).to(self.device)
            self._srcs += [t for f in self._ffn for t in (f[1], f[2])]
        for name, (off, rows, pitch) in self._images.items():
            self._views[name] = self.arena[off: off + rows * pitch].view(rows, pitch)

    def source_ptrs(self):
        return [s.data_ptr() for s in self._srcs]

    def run(self) -> None:
        if self._n:
            ops.pack_weights(self._table, self._n, self._tiles, ops.dt(self.dtype))
        if self._ffn:
            ops.ffn_pack(self._ffn_table, 2 * len(self._ffn), max(f[3] for f in self._ffn))

    def __getitem__(self, name: str) -> torch.Tensor:
        return self._views[name]

    def pitch(self, name: str) -> int:
        return self._images[name][2]
