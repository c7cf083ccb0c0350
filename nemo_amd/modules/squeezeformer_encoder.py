"""Drop-in `SqueezeformerEncoder` NeuralModule (BASELINE.json configs[4]) on the MI355X engine.

Replaces `nemo/collections/asr/modules/squeezeformer_encoder.py:40-400` + `parts/submodules/squeezeformer_modules.py:30-203`
behind the same constructor arguments, state-dict keys and (audio_signal, length) -> (encoded [B, D, T'], lengths) contract:

    'dw_striding' sub-sampling -> x*sqrt(d) (+dropout) -> pre_ln -> L x [ MHA -> FFN -> Conv -> FFN ], every sub-block
    x <- LayerNorm(x + dropout(f(x * scale + bias)))   (post-LN, squeezeformer_modules.py:139-181), with the temporal U-Net
    detour: at `time_reduce_idx` a masked depthwise Conv1d(k=5, s=2) + pointwise conv halves the frame rate (own positional
    table), at `time_recovery_idx` the frames are repeated x2, passed through a Linear and added to the cached full-rate
    activations (squeezeformer_encoder.py:340-361).

What is shared with the Conformer engine (conformer_encoder.py): the sub-sampling stack, the MFMA GEMM with fused epilogues,
rel-pos attention (flash kernels at d_k = 64, GEMM + fused softmax otherwise), depthwise k=31 conv / BatchNorm / Swish kernels
(run here on 2*d_model channels: 'swish' point-wise activation keeps both halves, conformer_modules.py:267-275), LayerNorm,
side-stream weight gradients, flat parameters.  What is new: csrc/squeezeformer.hip.

Odd geometries.  Squeezeformer-Medium is d_model = 324 with 4 heads: d_k = 81 and neither is a multiple of 8, while bf16 MFMA
operands need 16-byte aligned rows and heads.  Nothing is special-cased in the kernels; the operands are laid out so that the
arithmetic is unchanged:
  * heads are padded to d_k' = roundup8(d_k) INSIDE THE PACKED WEIGHT IMAGES (q|k|v, linear_pos, linear_out, pos_bias_u/v
    get zero rows / columns for the pad lanes), so q, k, v, p come out of their GEMMs with exact zeros in the pad lanes and
    every score, context vector and gradient is what the 81-wide heads give; softmax scale stays 1/sqrt(81);
  * [M, d] bf16 activations that feed a GEMM have row pitch roundup8(d) with zero pad columns (the kernels of
    squeezeformer.hip take the pitch); weight gradients of the padded heads are produced per head by batched TN GEMMs whose
    batch strides step over the pad lanes;
  * the sub-sampling stack's channel count (= d_model by default) is padded the same way: zero-padded copies of its small
    f32 weights / biases and zero rows / columns in the packed pointwise and output images make the pad channels exact
    zeros through ReLU, depthwise and pointwise stages; weight gradients land in a zeroed scratch of the padded shapes and the
    real part is added into the parameters' gradients (a handful of KB-sized adds per step).
"""
from __future__ import annotations

import math
import os
from typing import Optional

import torch
from torch import nn

from .. import ops
from ..packing import PackPlan
from .conformer_encoder import (ConformerEncoder, ConvSubsampling, RelPositionalEncoding, _FeedForward, _RelPosMHA, _Saved,
                                _pad8)
from ..core import NeuralModule


class ScaleBiasLayer(nn.Module):  # squeezeformer_modules.py:30-57
    def __init__(self, d_model: int, adaptive_scale: bool):
        super().__init__()
        self.adaptive_scale = adaptive_scale
        if adaptive_scale:
            self.scale = nn.Parameter(torch.ones(d_model))
            self.bias = nn.Parameter(torch.zeros(d_model))
        else:
            self.register_buffer("scale", torch.ones(d_model), persistent=True)
            self.register_buffer("bias", torch.zeros(d_model), persistent=True)


class _SwishConvolution(nn.Module):  # conformer_modules.py:236 ConformerConvolution(pointwise_activation='swish')
    def __init__(self, d_model, kernel_size):
        super().__init__()
        c2 = 2 * d_model  # 'swish' is in the activation registry: no GLU halving, the depthwise stage runs on 2*d channels
        self.pointwise_conv1 = nn.Conv1d(d_model, c2, 1)
        self.depthwise_conv = nn.Conv1d(c2, c2, kernel_size, padding=(kernel_size - 1) // 2, groups=c2)
        self.batch_norm = nn.BatchNorm1d(c2)
        self.pointwise_conv2 = nn.Conv1d(c2, d_model, 1)
        pw_max, dw_max = d_model ** -0.5, kernel_size ** -0.5  # reset_parameters_conv (conformer_modules.py:352-363)
        with torch.no_grad():
            for m, b in ((self.pointwise_conv1, pw_max), (self.pointwise_conv2, pw_max), (self.depthwise_conv, dw_max)):
                nn.init.uniform_(m.weight, -b, b)
                nn.init.uniform_(m.bias, -b, b)


class SqueezeformerLayer(nn.Module):  # squeezeformer_modules.py:60-203 (parameter layout only; compute lives in the encoder)
    def __init__(self, d_model, d_ff, n_heads, conv_kernel_size, adaptive_scale=True):
        super().__init__()
        self.norm_feed_forward1 = nn.LayerNorm(d_model)
        self.feed_forward1 = _FeedForward(d_model, d_ff)
        self.feed_forward1_scale = ScaleBiasLayer(d_model, adaptive_scale)
        self.norm_conv = nn.LayerNorm(d_model)
        self.conv = _SwishConvolution(d_model, conv_kernel_size)
        self.conv_scale = ScaleBiasLayer(d_model, adaptive_scale)
        self.norm_self_att = nn.LayerNorm(d_model)
        self.self_attn = _RelPosMHA(n_heads, d_model)
        self.self_attn_scale = ScaleBiasLayer(d_model, adaptive_scale)
        self.norm_feed_forward2 = nn.LayerNorm(d_model)
        self.feed_forward2 = _FeedForward(d_model, d_ff)
        self.feed_forward2_scale = ScaleBiasLayer(d_model, adaptive_scale)
        ffn1_max, ffn2_max = d_model ** -0.5, d_ff ** -0.5  # reset_parameters_ff (conformer_modules.py:400-408)
        with torch.no_grad():
            for ff in (self.feed_forward1, self.feed_forward2):
                nn.init.uniform_(ff.linear1.weight, -ffn1_max, ffn1_max); nn.init.uniform_(ff.linear1.bias, -ffn1_max, ffn1_max)
                nn.init.uniform_(ff.linear2.weight, -ffn2_max, ffn2_max); nn.init.uniform_(ff.linear2.bias, -ffn2_max, ffn2_max)


class TimeReductionModule(nn.Module):  # subsampling.py:589-656 (parameter layout only)
    def __init__(self, d_model: int, out_dim: int, kernel_size: int = 5, stride: int = 2):
        super().__init__()
        self.dw_conv = nn.Conv1d(d_model, d_model, kernel_size, stride=stride, padding=max(0, kernel_size - stride), groups=d_model)
        self.pw_conv = nn.Conv1d(d_model, out_dim, 1)
        dw_max, pw_max = kernel_size ** -0.5, d_model ** -0.5
        with torch.no_grad():
            for m, b in ((self.dw_conv, dw_max), (self.pw_conv, pw_max)):
                nn.init.uniform_(m.weight, -b, b)
                nn.init.uniform_(m.bias, -b, b)


class _Geo:
    """one frame rate of the temporal U-Net: B x T frames, valid lengths, positional table as a GEMM operand"""
    __slots__ = ("B", "T", "M", "lens", "pos", "P")


class SqueezeformerEncoder(ConformerEncoder):
    def __init__(self, feat_in: int, n_layers: int, d_model: int, feat_out: int = -1, subsampling: str = "dw_striding",
                 subsampling_factor: int = 4, subsampling_conv_channels: int = -1, ff_expansion_factor: int = 4,
                 self_attention_model: str = "rel_pos", n_heads: int = 4, att_context_size=None, xscaling: bool = True,
                 untie_biases: bool = True, pos_emb_max_len: int = 5000, conv_kernel_size: int = 31,
                 conv_norm_type: str = "batch_norm", dropout: float = 0.1, dropout_emb: float = 0.1, dropout_att: float = 0.0,
                 adaptive_scale: bool = True, time_reduce_idx: Optional[int] = None, time_recovery_idx: Optional[int] = None,
                 compute_dtype: Optional[torch.dtype] = None):
        NeuralModule.__init__(self)
        bad = []
        if not (subsampling == "dw_striding" and subsampling_factor in (4, 8)):
            bad.append(f"subsampling={subsampling} x{subsampling_factor} (implemented: dw_striding x4 / x8)")
        if self_attention_model != "rel_pos": bad.append(f"self_attention_model={self_attention_model}")
        if att_context_size not in (None, [-1, -1], (-1, -1)): bad.append("limited att_context_size")
        if not untie_biases: bad.append("tied pos biases")
        if conv_norm_type != "batch_norm": bad.append(f"conv_norm_type={conv_norm_type}")
        if feat_out > 0 and feat_out != d_model: bad.append("feat_out projection")
        if conv_kernel_size not in (5, 9, 31): bad.append(f"conv_kernel_size={conv_kernel_size}")
        if d_model % n_heads or d_model % 4: bad.append("d_model not divisible by n_heads / 4")
        if bad:
            raise NotImplementedError("MI355X SqueezeformerEncoder does not implement: " + ", ".join(bad))
        self.time_reduce_idx = time_reduce_idx
        self.time_recovery_idx = None
        if time_reduce_idx is not None:
            self.time_recovery_idx = n_layers - 1 if time_recovery_idx is None else time_recovery_idx
            if time_reduce_idx < 0 or self.time_recovery_idx >= n_layers:
                raise ValueError(f"Time reduce index must lie between [0, {n_layers})")  # squeezeformer_encoder.py:169-173
            if self.time_recovery_idx < 0 or self.time_recovery_idx >= n_layers:
                raise ValueError(f"Time recovery index must lie between [0, {n_layers})")
            if self.time_recovery_idx <= time_reduce_idx:
                raise NotImplementedError("time_recovery_idx must follow time_reduce_idx")
        d_ff = d_model * ff_expansion_factor
        self.d_model, self.n_layers, self._feat_in = d_model, n_layers, feat_in
        self.n_heads, self.d_k, self.d_ff = n_heads, d_model // n_heads, d_ff
        self.conv_kernel_size = conv_kernel_size
        self.subsampling_factor = subsampling_factor
        self.self_attention_model = self_attention_model
        self.att_context_size = [-1, -1]
        self.att_context_size_all, self.att_context_probs, self._ctx = [[-1, -1]], [1.0], (0, -1, -1)
        self.layer_drop_probs, self.capture_layers, self.captured = [0.0] * n_layers, [], {}
        self.adaptive_scale = adaptive_scale
        self.xscale = math.sqrt(d_model) if xscaling else None
        # (RelPositionalEncoding's dropout on x is `dropout` here: squeezeformer_encoder.py:207-213)
        self.dropout, self.dropout_pre_encoder, self.dropout_att, self.dropout_emb = dropout, dropout, dropout_att, dropout_emb
        if subsampling_conv_channels == -1:
            subsampling_conv_channels = d_model
        self.subsampling = subsampling
        self.pre_encode = ConvSubsampling(subsampling, subsampling_factor, feat_in, d_model, subsampling_conv_channels)
        self._reset_pre_encode()
        self._feat_out = d_model
        self.pos_emb_max_len = pos_emb_max_len
        self.pos_enc = RelPositionalEncoding(d_model, dropout, pos_emb_max_len, self.xscale, dropout_emb)
        self.layers = nn.ModuleList([SqueezeformerLayer(d_model, d_ff, n_heads, conv_kernel_size, adaptive_scale)
                                     for _ in range(n_layers)])
        self.time_reduce_layer = self.time_recovery_layer = self.time_reduce_pos_enc = None
        if time_reduce_idx is not None:
            self.time_reduce_layer = TimeReductionModule(d_model, d_model, kernel_size=5, stride=2)
            self.time_recovery_layer = nn.Linear(d_model, d_model)
            self.time_reduce_pos_enc = RelPositionalEncoding(d_model, 0.0, pos_emb_max_len, None, 0.0)
        self.pre_ln = nn.LayerNorm(d_model)
        self.out_proj = None
        self.max_audio_length = pos_emb_max_len
        self.sync_max_audio_length = True
        self._init_engine(compute_dtype, tail=None)
        self.packed_rows = False  # (the packed token chain is built for the Conformer layer stack; the time-reduction stages keep the grid)

    def _reset_pre_encode(self):
        """ConvSubsampling.reset_parameters for 'dw_striding' (subsampling.py:438-459): Squeezeformer's initialisation"""
        pe = self.pre_encode
        with torch.no_grad():
            scale = 1.0 / 3
            dw_max, pw_max = 9 ** -0.5, pe._conv_channels ** -0.5
            nn.init.uniform_(pe.conv[0].weight, -scale, scale); nn.init.uniform_(pe.conv[0].bias, -scale, scale)
            for dw, pw in pe.dw_stages():
                nn.init.uniform_(dw.weight, -dw_max, dw_max); nn.init.uniform_(dw.bias, -dw_max, dw_max)
                nn.init.uniform_(pw.weight, -pw_max, pw_max); nn.init.uniform_(pw.bias, -pw_max, pw_max)
            fc = (self.d_model * self._feat_in / pe._sampling_num) ** -0.5
            nn.init.uniform_(pe.out.weight, -fc, fc); nn.init.uniform_(pe.out.bias, -fc, fc)

    # ------------------------------------------------------------------ geometry helpers
    @property
    def input_types(self):   # squeezeformer_encoder.py:37-45 (no bypass_pre_encode port)
        from collections import OrderedDict
        from ..core import LengthsType, NeuralType, SpectrogramType
        return OrderedDict({"audio_signal": NeuralType(("B", "D", "T"), SpectrogramType()),
                            "length": NeuralType(tuple("B"), LengthsType())})

    def _geometry(self, cdt):
        """(row pitch of [M, d] GEMM operands, padded head width, padded attention width)"""
        dp, dkp = _pad8(self.d_model), _pad8(self.d_k)  # (same layout in fp32: one code path, exercised by the parity tests)
        if cdt == torch.bfloat16 and self.use_flash_attention and self.flash_pad_heads and dkp < 64:
            dkp = 64  # narrow heads ride the fused d_k' = 64 attention kernels (ConformerEncoder._geometry)
        elif cdt == torch.bfloat16 and self.use_flash_attention and self.flash_pad_heads and 64 < dkp < 128:
            dkp = 128  # Medium: d = 324, 4 heads, d_k = 81 -> the fused kernels' second width (round 5)
        return dp, dkp, self.n_heads * dkp

    def _sub_channels(self, cdt):
        """(real, padded) channel count of the sub-sampling stack"""
        C_ = self.pre_encode._conv_channels
        return C_, (_pad8(C_) if cdt == torch.bfloat16 else C_)

    def _sub_io(self, Wf, cdt, dev, backward=False):
        C_, Cp = self._sub_channels(cdt)
        if Cp == C_:
            return super()._sub_io(Wf, cdt, dev, backward)
        pe = self.pre_encode
        d, F2 = self.d_model, pe._feat_after
        n = len(pe.dw_stages())
        io = _Saved()
        io.C = Cp
        io.c0w, io.c0b = Wf["pre.c0w"], Wf["pre.c0b"]
        io.dw = [(Wf[f"pre.dw{i}w"], Wf[f"pre.dw{i}b"], Wf[f"pre.pw{i}b"]) for i in range(n)]
        io.finish = None
        if backward:
            sizes = [Cp * 9, Cp] + [Cp * 9, Cp, Cp * Cp, Cp] * n + [d * Cp * F2]
            offs = [0]
            for k in sizes:
                offs.append(offs[-1] + (k + 63) // 64 * 64)
            g = torch.zeros(offs[-1], dtype=torch.float32, device=dev)
            v = [g[offs[i]: offs[i] + sizes[i]] for i in range(len(sizes))]
            io.g_c0w, io.g_c0b = v[0], v[1]
            io.g_dw = [tuple(v[2 + 4 * i: 6 + 4 * i]) for i in range(n)]
            io.g_out = v[-1]

            def finish():  # real part of the padded gradients -> the parameters' gradients
                pe.conv[0].weight.grad.view(-1).add_(io.g_c0w[: C_ * 9])
                pe.conv[0].bias.grad.add_(io.g_c0b[:C_])
                for (dw, pw), (gw, gb, gpw, gpb) in zip(pe.dw_stages(), io.g_dw):
                    dw.weight.grad.view(-1).add_(gw[: C_ * 9])
                    dw.bias.grad.add_(gb[:C_])
                    pw.weight.grad.view(C_, C_).add_(gpw.view(Cp, Cp)[:C_, :C_])
                    pw.bias.grad.add_(gpb[:C_])
                pe.out.weight.grad.add_(io.g_out.view(d, Cp * F2)[:, : C_ * F2])
            io.finish = finish
        return io

    def _plan(self, cdt, device):
        # (the geometry -- head width of the packed attention images -- follows run-time flags: use_flash_attention, flash_pad_heads)
        key = (cdt, str(device), self._flatp.generation, self._geometry(cdt))
        plan = self._plans.get(key)
        if plan is None:
            self._plans = {}
            p = PackPlan(cdt, device)
            pf = PackPlan(torch.float32, device)
            d, H, dk = self.d_model, self.n_heads, self.d_k
            dp, dkp, dA = self._geometry(cdt)
            pe = self.pre_encode
            C_, F2 = pe._conv_channels, pe._feat_after
            Cs, Cp = self._sub_channels(cdt)
            if Cp == Cs:
                for si_, (_, pw) in enumerate(pe.dw_stages()):
                    p.add_matrix(f"pre.pw{si_}", pw.weight.data.view(C_, C_))
                    p.add_matrix(f"pre.pw{si_}t", pw.weight.data.view(C_, C_), True)
                p.add_fc_permuted("pre.out", pe.out.weight.data, C_, F2)
                p.add_fc_permuted("pre.outt", pe.out.weight.data, C_, F2, transpose=True)
            else:  # zero-padded channels (see the module docstring): dense [Cp, 9] / [Cp] f32 copies + padded GEMM images
                def dense(name, t, n_pad):
                    pf.new_image(name, 1, n_pad)
                    pf.add_block(name, t.view(-1), 1, t.numel(), sr1=0, sc1=1)
                dense("pre.c0w", pe.conv[0].weight.data, Cp * 9); dense("pre.c0b", pe.conv[0].bias.data, Cp)
                for si_, (dw, pw) in enumerate(pe.dw_stages()):
                    dense(f"pre.dw{si_}w", dw.weight.data, Cp * 9); dense(f"pre.dw{si_}b", dw.bias.data, Cp)
                    dense(f"pre.pw{si_}b", pw.bias.data, Cp)
                    w = pw.weight.data.view(C_, C_)
                    p.new_image(f"pre.pw{si_}", Cp, Cp); p.add_block(f"pre.pw{si_}", w, C_, C_, sr1=C_, sc1=1)
                    p.new_image(f"pre.pw{si_}t", Cp, Cp); p.add_block(f"pre.pw{si_}t", w, C_, C_, sr1=1, sc1=C_)
                wo_ = pe.out.weight.data  # [d, C*F2], column c*F2 + f  ->  image column f*Cp + c
                p.new_image("pre.out", d, F2 * Cp); p.new_image("pre.outt", F2 * Cp, d)
                for f in range(F2):
                    p.add_block("pre.out", wo_.view(-1)[f:], d, C_, col_off=f * Cp, sr1=C_ * F2, sc1=F2)
                    p.add_block("pre.outt", wo_.view(-1)[f:], C_, d, row_off=f * Cp, sr1=F2, sc1=C_ * F2)

            def heads_rows(name, w, row0):  # w [H*dk, d] -> rows row0 + h*dkp + (0..dk) of image [*, d]
                for h in range(H):
                    p.add_block(name, w.view(-1)[h * dk * d:], dk, d, row_off=row0 + h * dkp, sr1=d, sc1=1)

            def heads_cols(name, w, col0):  # w [H*dk, d] transposed -> columns col0 + h*dkp + (0..dk) of image [d, *]
                for h in range(H):
                    p.add_block(name, w.view(-1)[h * dk * d:], d, dk, col_off=col0 + h * dkp, sr1=1, sc1=d)

            for i, L in enumerate(self.layers):
                for ff, m in (("ff1", L.feed_forward1), ("ff2", L.feed_forward2)):
                    p.add_matrix(f"L{i}.{ff}.w1", m.linear1.weight.data); p.add_matrix(f"L{i}.{ff}.w1t", m.linear1.weight.data, True)
                    p.add_matrix(f"L{i}.{ff}.w2", m.linear2.weight.data); p.add_matrix(f"L{i}.{ff}.w2t", m.linear2.weight.data, True)
                a = L.self_attn
                qkv = [a.linear_q.weight.data, a.linear_k.weight.data, a.linear_v.weight.data]
                p.new_image(f"L{i}.att.wqkv", 3 * dA, d); p.new_image(f"L{i}.att.wqkvt", d, 3 * dA)
                p.new_image(f"L{i}.att.wpos", dA, d)
                p.new_image(f"L{i}.att.wo", d, dA); p.new_image(f"L{i}.att.wot", dA, d)
                for j, w in enumerate(qkv):
                    heads_rows(f"L{i}.att.wqkv", w, j * dA)
                    heads_cols(f"L{i}.att.wqkvt", w, j * dA)
                heads_rows(f"L{i}.att.wpos", a.linear_pos.weight.data, 0)
                wo = a.linear_out.weight.data  # [d, H*dk]: head h = columns h*dk ..
                for h in range(H):
                    p.add_block(f"L{i}.att.wo", wo.view(-1)[h * dk:], d, dk, col_off=h * dkp, sr1=d, sc1=1)
                    p.add_block(f"L{i}.att.wot", wo.view(-1)[h * dk:], dk, d, row_off=h * dkp, sr1=1, sc1=d)
                pf.new_image(f"L{i}.att.bqkv", 1, 3 * dA)
                pf.new_image(f"L{i}.att.bu", 1, dA); pf.new_image(f"L{i}.att.bv", 1, dA)
                for h in range(H):
                    for j, b in enumerate((a.linear_q.bias.data, a.linear_k.bias.data, a.linear_v.bias.data)):
                        pf.add_block(f"L{i}.att.bqkv", b[h * dk:], 1, dk, col_off=j * dA + h * dkp, sr1=0, sc1=1)
                    pf.add_block(f"L{i}.att.bu", a.pos_bias_u.data.view(-1)[h * dk:], 1, dk, col_off=h * dkp, sr1=0, sc1=1)
                    pf.add_block(f"L{i}.att.bv", a.pos_bias_v.data.view(-1)[h * dk:], 1, dk, col_off=h * dkp, sr1=0, sc1=1)
                c = L.conv
                p.add_matrix(f"L{i}.conv.pw1", c.pointwise_conv1.weight.data); p.add_matrix(f"L{i}.conv.pw1t", c.pointwise_conv1.weight.data, True)
                p.add_matrix(f"L{i}.conv.pw2", c.pointwise_conv2.weight.data); p.add_matrix(f"L{i}.conv.pw2t", c.pointwise_conv2.weight.data, True)
            if self.time_reduce_layer is not None:
                p.add_matrix("tr.pw", self.time_reduce_layer.pw_conv.weight.data); p.add_matrix("tr.pwt", self.time_reduce_layer.pw_conv.weight.data, True)
                p.add_matrix("rec.w", self.time_recovery_layer.weight.data); p.add_matrix("rec.wt", self.time_recovery_layer.weight.data, True)
            p.finalize(); pf.finalize()
            plan = (p, pf, None, -2)
            self._plans[key] = plan
        if plan[3] != self._weights_version or self._force_pack:
            plan[0].run(); plan[1].run()
            plan = (plan[0], plan[1], plan[2], self._weights_version)
            self._plans[key] = plan
        return plan[0], plan[1]

    def _geo(self, B, T, lens, pos_enc, d_emb, cdt, dev, dp):
        g = _Geo()
        g.B, g.T, g.M, g.lens, g.P = B, T, B * T, lens, 2 * T - 1
        key = (T, cdt, str(dev), dp, id(pos_enc))
        tab = self._pos_cache.get(key)
        if tab is None:
            tab = pos_enc.table(T, dev, torch.float32)
            if len(self._pos_cache) >= 32:
                self._pos_cache = {}
            self._pos_cache[key] = tab
        g.pos = self._new(g.P, dp, dtype=cdt, device=dev)
        ops.cast_pitched(tab, g.pos, g.P, self.d_model, dp, 1.0, d_emb)  # (+ dropout_emb on the table, :1097-1098)
        return g

    # ------------------------------------------------------------------ forward
    def _forward_impl(self, mel, length, save=False):
        dev = mel.device
        self._phase("f", dev)
        self._fwd_serial += 1
        cdt = self._cdt()
        training = self.training
        W, Wf = self._plan(cdt, dev)
        B, F_, T = mel.shape
        mel = mel.to(torch.float32).contiguous()
        d = self.d_model
        dp, dkp, dA = self._geometry(cdt)
        T1, F1 = (T - 1) // 2 + 1, (F_ - 1) // 2 + 1
        lens = self._lens(length, self.pre_encode._sampling_num)
        T2, F2 = T, F_
        for _ in range(self.pre_encode._sampling_num):
            T2, F2 = (T2 - 1) // 2 + 1, (F2 - 1) // 2 + 1
        M = B * T2
        self.update_max_seq_length(T2, dev)
        if training:
            self._step_seed = (self._step_seed + 1) & 0x3FFFFFFF
        seed = self._step_seed

        def drop(p, site):
            return ops.Dropout(p if training else 0.0, seed, site)

        S = _Saved()
        S.serial, S.arena = self._fwd_serial, self._arena is not None
        S.dims = (B, F_, T, T1, F1, T2, F2, M, cdt, training, seed)
        S.mel, S.len0, S.len2, S.lens_all = mel, lens[0], lens[-1], lens
        S.drop_pre = drop(self.dropout_pre_encoder, 100000)
        x = self._sub_fwd_dw(S, mel, lens, W, cdt, save, Wf=Wf)
        # ---- pre_ln
        x0, pmean, prstd = self._ln_fwd(self.pre_ln, x, M, d, torch.float32, dev)
        S.pre_ln = (x, pmean, prstd)
        g_full = self._geo(B, T2, lens[-1], self.pos_enc, drop(self.dropout_emb, 100001), cdt, dev, dp)
        C2 = 2 * d
        S.bn_stats = torch.zeros(self.n_layers, 2 * C2 + 8, dtype=torch.float64, device=dev) if training else None
        S.bn_world = self._syncbn_world() if training else 1
        S.layers, S.geos = [], []
        S.tr = S.rec = None
        g = g_full
        x = x0
        for i, L in enumerate(self.layers):
            if self.time_reduce_idx is not None and i == self.time_reduce_idx:
                Th = (g.T + 1) // 2
                lens_h = (torch.div(g.lens + 1, 2, rounding_mode="floor")).contiguous()
                dwo = self._new(B * Th, dp, dtype=cdt, device=dev)
                tr = self.time_reduce_layer
                ops.time_reduce_dwconv_fwd(x, g.lens, tr.dw_conv.weight, tr.dw_conv.bias, dwo, B, g.T, d, dp)
                xh = self._new(B * Th, d, dtype=torch.float32, device=dev)
                ops.gemm(dwo, W["tr.pw"], xh, B * Th, d, d, dp, W.pitch("tr.pw"), d, bias=tr.pw_conv.bias)
                S.tr = (x, g, dwo)
                g = self._geo(B, Th, lens_h, self.time_reduce_pos_enc, ops.NO_DROP, cdt, dev, dp)
                x = xh
            if self.time_reduce_idx is not None and i == self.time_recovery_idx:
                xs_in, g_skip, _ = S.tr
                xs_c = self._new(g.M, dp, dtype=cdt, device=dev)
                ops.scale_bias_fwd(x, None, None, xs_c, g.M, d, dp)
                ys = self._new(g.M, d, dtype=torch.float32, device=dev)
                rec = self.time_recovery_layer
                ops.gemm(xs_c, W["rec.w"], ys, g.M, d, d, dp, W.pitch("rec.w"), d, bias=rec.bias)
                xr = self._new(g_skip.M, d, dtype=torch.float32, device=dev)
                ops.time_recover_fwd(xs_in, ys, xr, B, g_skip.T, d)
                S.rec = (xs_c, g)
                g, x = g_skip, xr
            if training and S.bn_world > 1:
                S.bn_stats[i, 2 * C2] = float(g.M)
            x, sl = self._sq_layer_fwd(i, L, x, g, S, W, Wf, drop)
            S.layers.append(sl)
            S.geos.append(g)
        if training:
            torch._foreach_add_([L.conv.batch_norm.num_batches_tracked for L in self.layers], 1)
        out = x.view(B, T2, d).transpose(1, 2)
        return out, lens[-1], (S if save else None)

    def _sb_fwd(self, sb, x, M, cdt, dp):
        y = self._new(M, dp, dtype=cdt, device=x.device)
        ops.scale_bias_fwd(x, sb.scale, sb.bias, y, M, self.d_model, dp)
        return y

    def _post_ln(self, ln, r, M):
        return self._ln_fwd(ln, r, M, self.d_model, torch.float32, r.device)

    def _sq_ffn_fwd(self, pfx, ff, sb, ln, x, g, W, drop, site, cdt, dp):
        M, d, dff, dev = g.M, self.d_model, self.d_ff, x.device
        y = self._sb_fwd(sb, x, M, cdt, dp)
        h = self._new(M, dff, dtype=cdt, device=dev)
        a = self._new(M, dff, dtype=cdt, device=dev)
        d_in, d_res = drop(self.dropout, site), drop(self.dropout, site + 1)
        g_form = self.swish_g and cdt == torch.bfloat16  # `h` then holds swish'(h) * mask (ConformerEncoder.__init__)
        ops.gemm(y, W[pfx + ".w1"], a, M, dff, d, dp, W.pitch(pfx + ".w1"), dff, bias=ff.linear1.bias,
                 epi=ops.EPI_SWISH_DROP_G if g_form else ops.EPI_SWISH_DROP, aux_out=h, drop=d_in)
        if g_form:
            d_in = None
        r = self._new(M, d, dtype=torch.float32, device=dev)
        ops.gemm(a, W[pfx + ".w2"], r, M, d, dff, dff, W.pitch(pfx + ".w2"), d, bias=ff.linear2.bias, epi=ops.EPI_RESID,
                 aux_in=x, drop=d_res)  # fc_factor = 1.0 (squeezeformer_modules.py:103)
        xo, mean, rstd = self._post_ln(ln, r, M)
        return xo, (x, y, h, a, d_in, d_res, r, mean, rstd)

    def _sq_layer_fwd(self, i, L, x, g, S, W, Wf, drop):
        B, T, M = g.B, g.T, g.M
        cdt, training = S.dims[8], S.dims[9]
        dev = x.device
        d, H, dk = self.d_model, self.n_heads, self.d_k
        dp, dkp, dA = self._geometry(cdt)
        sl = _Saved()
        site = i * 16
        # ---- rel-pos multi-head self-attention
        a = L.self_attn
        y = self._sb_fwd(L.self_attn_scale, x, M, cdt, dp)
        qkv = self._new(M, 3 * dA, dtype=cdt, device=dev)
        ops.gemm(y, W[f"L{i}.att.wqkv"], qkv, M, 3 * dA, d, dp, W.pitch(f"L{i}.att.wqkv"), 3 * dA, bias=Wf[f"L{i}.att.bqkv"])
        p = self._new(g.P, dA, dtype=cdt, device=dev)
        ops.gemm(g.pos, W[f"L{i}.att.wpos"], p, g.P, dA, d, dp, W.pitch(f"L{i}.att.wpos"), dA)
        d_att, d_res = drop(self.dropout_att, site + 2), drop(self.dropout, site + 3)
        bu, bv = Wf[f"L{i}.att.bu"], Wf[f"L{i}.att.bv"]
        ctx, att_saved = self._attn_fwd(qkv, p, bu, bv, g.lens, B, T, dA, dkp, 1.0 / math.sqrt(dk), d_att, cdt, dev)
        r = self._new(M, d, dtype=torch.float32, device=dev)
        ops.gemm(ctx, W[f"L{i}.att.wo"], r, M, d, dA, dA, W.pitch(f"L{i}.att.wo"), d, bias=a.linear_out.bias, epi=ops.EPI_RESID,
                 aux_in=x, drop=d_res)
        x1, mean, rstd = self._post_ln(L.norm_self_att, r, M)
        sl.att = (x, y, qkv, p, att_saved, ctx, d_att, d_res, r, mean, rstd)
        # ---- feed forward 1
        x2, sl.ff1 = self._sq_ffn_fwd(f"L{i}.ff1", L.feed_forward1, L.feed_forward1_scale, L.norm_feed_forward1, x1, g, W, drop,
                                      site + 4, cdt, dp)
        # ---- convolution module (Swish point-wise activation: 2*d channels through depthwise conv / BatchNorm)
        c = L.conv
        k = self.conv_kernel_size
        C2 = 2 * d
        y3 = self._sb_fwd(L.conv_scale, x2, M, cdt, dp)
        pw1 = self._new(M, C2, dtype=cdt, device=dev)
        ops.gemm(y3, W[f"L{i}.conv.pw1"], pw1, M, C2, d, dp, W.pitch(f"L{i}.conv.pw1"), C2, bias=c.pointwise_conv1.bias)
        gact = self._new(M, C2, dtype=cdt, device=dev)
        fuse_act = self.fuse_glu_dwconv_fwd and C2 % (8 if cdt == torch.bfloat16 else 4) == 0   # (see ConformerEncoder._layer_fwd)
        if not fuse_act:
            ops.swish_mask_fwd(pw1, gact, g.lens, T, M, C2)
        cc = self._new(M, C2, dtype=cdt, device=dev)
        bn = c.batch_norm
        bmean = self._new(C2, dtype=torch.float32, device=dev)
        brstd = self._new(C2, dtype=torch.float32, device=dev)
        count = float(M)
        if training:
            stats = S.bn_stats[i]
            if fuse_act:
                ops.dwconv_fwd_glu(pw1, g.lens, None, gact, c.depthwise_conv.weight, c.depthwise_conv.bias, cc, stats, B, T, C2, k, act=1)
            else:
                ops.dwconv_fwd(gact, c.depthwise_conv.weight, c.depthwise_conv.bias, cc, stats, B, T, C2, k)
            if S.bn_world > 1:
                self._sync_stats(stats[: 2 * C2 + 1])
                count = stats[2 * C2: 2 * C2 + 1]
        else:
            if fuse_act:
                ops.dwconv_fwd_glu(pw1, g.lens, None, gact, c.depthwise_conv.weight, c.depthwise_conv.bias, cc, None, B, T, C2, k, act=1)
            else:
                ops.dwconv_fwd(gact, c.depthwise_conv.weight, c.depthwise_conv.bias, cc, None, B, T, C2, k)
            ops.bn_eval_stats(bn.running_mean, bn.running_var, bmean, brstd, bn.eps, C2)
        z = self._new(M, C2, dtype=cdt, device=dev)
        if training:
            # (two launches on purpose: the one-launch form, mi355x_bn_stats_swish_fwd, makes EVERY workgroup derive the
            #  coefficients of its channels from the f64 sums and measured 31 us against 15.5 us for this pair,
            #  tools/bn_bench.py)
            ops.bn_finalize(stats, count, bmean, brstd, bn.running_mean, bn.running_var, bn.momentum, bn.eps, C2)
        ops.bn_swish_fwd(cc, bmean, brstd, bn.weight, bn.bias, z, M, C2)
        r3 = self._new(M, d, dtype=torch.float32, device=dev)
        d_cres = drop(self.dropout, site + 6)
        ops.gemm(z, W[f"L{i}.conv.pw2"], r3, M, d, C2, C2, W.pitch(f"L{i}.conv.pw2"), d, bias=c.pointwise_conv2.bias,
                 epi=ops.EPI_RESID, aux_in=x2, drop=d_cres)
        x3, mean3, rstd3 = self._post_ln(L.norm_conv, r3, M)
        sl.conv = (x2, y3, pw1, gact, cc, bmean, brstd, count, z, d_cres, r3, mean3, rstd3)
        # ---- feed forward 2
        x4, sl.ff2 = self._sq_ffn_fwd(f"L{i}.ff2", L.feed_forward2, L.feed_forward2_scale, L.norm_feed_forward2, x3, g, W, drop,
                                      site + 7, cdt, dp)
        return x4, sl

    # ------------------------------------------------------------------ backward
    def _branch_grad(self, ln, dxo, r, mean, rstd, M, d_res, cdt, dp):
        """post-LN sub-block: -> (dr f32 [M,d] = d/d(x + branch), operand copy of the branch gradient [M, dp])"""
        d, dev = self.d_model, dxo.device
        dr = self._new(M, d, dtype=torch.float32, device=dev)
        ops.layernorm_bwd(dxo, r, ln.weight, mean, rstd, dr, False, ln.weight.grad, ln.bias.grad, M, d)
        df = self._new(M, dp, dtype=cdt, device=dev)
        ops.cast_pitched(dr, df, M, d, dp, 1.0, d_res)
        return dr, df

    def _sb_bwd(self, sb, dy, x, dr, M, dp):
        ad = sb.adaptive_scale
        ops.scale_bias_bwd(dy, dp, x, sb.scale, dr, sb.scale.grad if ad else None, sb.bias.grad if ad else None, M, self.d_model)
        return dr

    def _sq_ffn_bwd(self, pfx, ff, sb, ln, saved, dxo, g, W, cdt, dp):
        x, y, h, a, d_in, d_res, r, mean, rstd = saved
        M, d, dff, dev = g.M, self.d_model, self.d_ff, dxo.device
        dr, df = self._branch_grad(ln, dxo, r, mean, rstd, M, d_res, cdt, dp)
        self._wgrad(df, dp, 0, a, dff, 0, ff.linear2.weight.grad, d, dff, M, bias_grad=ff.linear2.bias.grad)
        dh = self._new(M, dff, dtype=cdt, device=dev)
        if d_in is None:
            ops.gemm(df, W[pfx + ".w2t"], dh, M, dff, d, dp, W.pitch(pfx + ".w2t"), dff, epi=ops.EPI_DSWISH_G, aux_in=h)
        else:
            ops.gemm(df, W[pfx + ".w2t"], dh, M, dff, d, dp, W.pitch(pfx + ".w2t"), dff, epi=ops.EPI_DSWISH, aux_in=h, drop=d_in)
        self._wgrad(dh, dff, 0, y, dp, 0, ff.linear1.weight.grad, dff, d, M, bias_grad=ff.linear1.bias.grad)
        dy = self._new(M, dp, dtype=cdt, device=dev)
        ops.gemm(dh, W[pfx + ".w1t"], dy, M, d, dff, dff, W.pitch(pfx + ".w1t"), dp)
        return self._sb_bwd(sb, dy, x, dr, M, dp)

    # (_heads_wgrad / _unpad_add: the padded-head weight-gradient helpers live in ConformerEncoder)
    def _sq_layer_bwd(self, i, L, dxo, g, S, sl, W, Wf):
        B, T, M = g.B, g.T, g.M
        cdt, training = S.dims[8], S.dims[9]
        dev = dxo.device
        d, H, dk = self.d_model, self.n_heads, self.d_k
        dp, dkp, dA = self._geometry(cdt)
        padded = dkp != dk
        C2 = 2 * d
        k = self.conv_kernel_size
        # ---- feed forward 2
        dx = self._sq_ffn_bwd(f"L{i}.ff2", L.feed_forward2, L.feed_forward2_scale, L.norm_feed_forward2, sl.ff2, dxo, g, W, cdt, dp)
        # ---- convolution module
        c = L.conv
        bn = c.batch_norm
        x2, y3, pw1, gact, cc, bmean, brstd, count, z, d_cres, r3, mean3, rstd3 = sl.conv
        dr, db = self._branch_grad(L.norm_conv, dx, r3, mean3, rstd3, M, d_cres, cdt, dp)
        self._wgrad(db, dp, 0, z, C2, 0, c.pointwise_conv2.weight.grad, d, C2, M, bias_grad=c.pointwise_conv2.bias.grad)
        dz = self._new(M, C2, dtype=cdt, device=dev)
        ops.gemm(db, W[f"L{i}.conv.pw2t"], dz, M, C2, d, dp, W.pitch(f"L{i}.conv.pw2t"), C2)
        sums = S.bn_sums[i]
        ops.bn_swish_bwd_reduce(dz, cc, bmean, brstd, bn.weight, bn.bias, sums, M, C2, dgamma=bn.weight.grad, dbeta=bn.bias.grad)
        if training and S.bn_world > 1:
            self._sync_stats(sums)
        dpw1 = self._new(M, C2, dtype=cdt, device=dev)
        fuse_act = self.fuse_bn_dwconv_bwd and self.fuse_glu_dwconv_bwd and C2 % (8 if cdt == torch.bfloat16 else 4) == 0
        if fuse_act:   # BatchNorm + Swish backward, depthwise backward and the pointwise Swish's backward in one launch
            ops.dwconv_bwd_bnswish(dz, cc, bmean, brstd, bn.weight, bn.bias, sums, count, training, gact, c.depthwise_conv.weight, None,
                                   c.depthwise_conv.weight.grad, c.depthwise_conv.bias.grad, B, T, C2, k, glu_in=pw1, glu_din=dpw1,
                                   glu_len=g.lens, glu_act=1)
        else:
            dg = self._new(M, C2, dtype=cdt, device=dev)
            if self.fuse_bn_dwconv_bwd:   # (see ConformerEncoder._layer_bwd)
                ops.dwconv_bwd_bnswish(dz, cc, bmean, brstd, bn.weight, bn.bias, sums, count, training, gact, c.depthwise_conv.weight, dg,
                                       c.depthwise_conv.weight.grad, c.depthwise_conv.bias.grad, B, T, C2, k)
            else:
                dcc = self._new(M, C2, dtype=cdt, device=dev)
                ops.bn_swish_bwd_apply(dz, cc, bmean, brstd, bn.weight, bn.bias, sums, count, training, dcc, M, C2)
                ops.dwconv_bwd(dcc, gact, c.depthwise_conv.weight, dg, c.depthwise_conv.weight.grad, c.depthwise_conv.bias.grad, B, T, C2, k)
            ops.swish_mask_bwd(pw1, dg, dpw1, g.lens, T, M, C2)
        self._wgrad(dpw1, C2, 0, y3, dp, 0, c.pointwise_conv1.weight.grad, C2, d, M, bias_grad=c.pointwise_conv1.bias.grad)
        dy3 = self._new(M, dp, dtype=cdt, device=dev)
        ops.gemm(dpw1, W[f"L{i}.conv.pw1t"], dy3, M, d, C2, C2, W.pitch(f"L{i}.conv.pw1t"), dp)
        dx = self._sb_bwd(L.conv_scale, dy3, x2, dr, M, dp)
        # ---- feed forward 1
        dx = self._sq_ffn_bwd(f"L{i}.ff1", L.feed_forward1, L.feed_forward1_scale, L.norm_feed_forward1, sl.ff1, dx, g, W, cdt, dp)
        # ---- self-attention
        a = L.self_attn
        x, y, qkv, p, att_saved, ctx, d_att, d_res, r, mean, rstd = sl.att
        dr, dao = self._branch_grad(L.norm_self_att, dx, r, mean, rstd, M, d_res, cdt, dp)
        bf16 = cdt == torch.bfloat16
        if not padded:
            self._wgrad(dao, dp, 0, ctx, dA, 0, a.linear_out.weight.grad, d, d, M, bias_grad=a.linear_out.bias.grad)
        else:  # d linear_out.weight[:, h*dk:(h+1)*dk] += dao^T @ ctx[:, h*dkp : +dk]: batch over heads
            with self._wgrad_scope(dao, ctx):
                ops.gemm(dao, ctx, a.linear_out.weight.grad, d, dk, M, dp, dA, d, transA=True, transB=True, atomic=True,
                         splitk=self._splitk(self._tiles(d, dk, bf16) * H, M), batch=H, nb0=H, sB=(dkp, 0), sC=(dk, 0),
                         c_dtype=ops.F32)
                ops.colsum(dao, a.linear_out.bias.grad, M, d, ld=dp)
        dctx = self._new(M, dA, dtype=cdt, device=dev)
        ops.gemm(dao, W[f"L{i}.att.wot"], dctx, M, dA, d, dp, W.pitch(f"L{i}.att.wot"), dA)
        dpos = torch.zeros(g.P, dA, dtype=torch.float32, device=dev)
        dpos_c = self._new(g.P, dA, dtype=cdt, device=dev)
        bu, bv = Wf[f"L{i}.att.bu"], Wf[f"L{i}.att.bv"]
        dqkv, dqu, dqv = self._attn_bwd(att_saved, qkv, p, bu, bv, ctx, dctx, g.lens, B, T, dA, dkp, 1.0 / math.sqrt(dk), d_att,
                                        cdt, dev, dpos, dpos_c)
        gu, gv_ = a.pos_bias_u.grad, a.pos_bias_v.grad
        if not padded:
            ops.colsum(dqu, gu, M, dA)
            ops.colsum(dqv, gv_, M, dA)
        else:
            sc = torch.zeros(2, dA, dtype=torch.float32, device=dev)
            ops.colsum(dqu, sc[0], M, dA)
            ops.colsum(dqv, sc[1], M, dA)
            self._unpad_add(gu.view(-1), sc[0], dkp)
            self._unpad_add(gv_.view(-1), sc[1], dkp)
        ops.add2(dqu, dqv, dqkv, 3 * dA, M, dA)
        # q / k / v / linear_pos weight gradients (the flash path produced dpos_c on the weight-gradient stream: its consumer runs
        # there too; the GEMM path produced it on the main stream, which _wgrad_scope waits for)
        lins = (a.linear_q, a.linear_k, a.linear_v)
        if not padded:
            for j, lin in enumerate(lins):
                self._wgrad(dqkv, 3 * dA, j * dA, y, dp, 0, lin.weight.grad, d, d, M, bias_grad=lin.bias.grad)
            self._wgrad(dpos_c, dA, 0, g.pos, dp, 0, a.linear_pos.weight.grad, d, d, g.P)
        else:
            gq, gk, gvw = (lin.weight.grad for lin in lins)
            sw = (gk.data_ptr() - gq.data_ptr()) // 4
            if sw > 0 and (gvw.data_ptr() - gk.data_ptr()) // 4 == sw:
                self._heads_wgrad(dqkv, 3 * dA, 0, y, dp, gq, M, 3, dA, sw)
            else:
                for j, lin in enumerate(lins):
                    self._heads_wgrad(dqkv, 3 * dA, j * dA, y, dp, lin.weight.grad, M, 1, 0, 0)
            sc = torch.zeros(3 * dA, dtype=torch.float32, device=dev)
            with self._wgrad_scope(dqkv, sc):
                ops.colsum(dqkv, sc, M, 3 * dA)
                for j, lin in enumerate(lins):
                    self._unpad_add(lin.bias.grad, sc[j * dA:(j + 1) * dA], dkp)
            self._heads_wgrad(dpos_c, dA, 0, g.pos, dp, a.linear_pos.weight.grad, g.P, 1, 0, 0)
        dy = self._new(M, dp, dtype=cdt, device=dev)
        ops.gemm(dqkv, W[f"L{i}.att.wqkvt"], dy, M, d, 3 * dA, 3 * dA, W.pitch(f"L{i}.att.wqkvt"), dp)
        return self._sb_bwd(L.self_attn_scale, dy, x, dr, M, dp)

    def _backward_impl(self, S, dout):
        B, F_, T, T1, F1, T2, F2, M, cdt, training, seed = S.dims
        dev = dout.device
        self._check_serial(S)
        self._phase("b", dev)
        d = self.d_model
        W, Wf = self._plan(cdt, dev)
        fp = self._flatp
        dp, dkp, dA = self._geometry(cdt)
        dx = dout.transpose(1, 2).contiguous().view(M, d).to(torch.float32)
        C2 = 2 * d
        S.bn_sums = torch.zeros(self.n_layers, 2, C2, dtype=torch.float64, device=dev)
        # (the grouped launch takes any n_out / n_in: its operands are pitched, its epilogue is per-element atomics)
        self._wg_pending = [] if (self.wgrad_grouped and cdt == torch.bfloat16) else None
        dskip = None
        for i in range(self.n_layers - 1, -1, -1):
            g = S.geos[i]
            dx = self._sq_layer_bwd(i, self.layers[i], dx, g, S, S.layers[i], W, Wf)
            self._wgrad_flush()
            S.layers[i] = None
            if self.time_reduce_idx is not None and i == self.time_recovery_idx:
                # x = skip + Linear(repeat_interleave(x_small, 2)[:, :T]): the skip gradient waits for the time-reduction backward
                xs_c, g_small = S.rec
                rec = self.time_recovery_layer
                dskip = dx
                dys = self._new(g_small.M, dp, dtype=cdt, device=dev)
                ops.time_recover_bwd(dx, dys, B, g.T, d, dp)
                self._wgrad(dys, dp, 0, xs_c, dp, 0, rec.weight.grad, d, d, g_small.M, bias_grad=rec.bias.grad)
                self._wgrad_flush()
                dx = self._new(g_small.M, d, dtype=torch.float32, device=dev)
                ops.gemm(dys, W["rec.wt"], dx, g_small.M, d, d, dp, W.pitch("rec.wt"), d)
            if self.time_reduce_idx is not None and i == self.time_reduce_idx:
                x_in, g_full, dwo = S.tr
                tr = self.time_reduce_layer
                Mh = g.M
                dxc = self._new(Mh, dp, dtype=cdt, device=dev)
                ops.cast_pitched(dx, dxc, Mh, d, dp)
                self._wgrad(dxc, dp, 0, dwo, dp, 0, tr.pw_conv.weight.grad, d, d, Mh, bias_grad=tr.pw_conv.bias.grad)
                self._wgrad_flush()
                ddw = self._new(Mh, dp, dtype=cdt, device=dev)
                ops.gemm(dxc, W["tr.pwt"], ddw, Mh, d, d, dp, W.pitch("tr.pwt"), dp)
                ops.time_reduce_dwconv_bwd(ddw, dp, x_in, g_full.lens, tr.dw_conv.weight, dskip, tr.dw_conv.weight.grad,
                                           tr.dw_conv.bias.grad, B, g_full.T, d)
                dx, dskip = dskip, None
            if self.grad_ready_hook is not None:
                if self._wgrad_join_per_layer:
                    self._wgrad_join()
                self._hook(*fp.range_of(f"layers.{i}."))
        self._wg_pending = None
        x_pre, pmean, prstd = S.pre_ln
        dpre = self._new(M, d, dtype=torch.float32, device=dev)
        ops.layernorm_bwd(dx, x_pre, self.pre_ln.weight, pmean, prstd, dpre, False, self.pre_ln.weight.grad, self.pre_ln.bias.grad, M, d)
        self._wgrad_join()
        if self.grad_ready_hook is not None:
            for pfx in ("pre_ln.", "time_reduce_layer.", "time_recovery_layer."):
                try:
                    self._hook(*fp.range_of(pfx))
                except KeyError:
                    pass
        return self._sub_bwd_dw(S, dpre, W, cdt, Wf=Wf)
