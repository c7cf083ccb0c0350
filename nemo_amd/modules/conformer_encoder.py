"""Drop-in for `nemo.collections.asr.modules.ConformerEncoder` (modules/conformer_encoder.py:62) on MI355X.

Same constructor kwargs (:289-329), typed I/O (:221-260), attributes callers touch (`_feat_out`, `subsampling_factor`,
`pre_encode`, `layers`, `pos_enc`, `set_max_audio_length`, ...) and -- the on-disk ABI -- the same parameter / buffer
names and shapes, so a reference `.nemo` state-dict loads with strict=True.  The arithmetic is NOT torch's: forward and
backward are sequenced here in Python over the hand-written HIP kernels of libmi355x_asr.so (one autograd node for the
whole encoder; activations are channels-last [B*T, d]; gradients are written straight into the flat gradient buffer).

Implemented configuration = what the Conformer-CTC recipes use (examples/asr/conf/conformer/conformer_ctc_bpe.yaml):
subsampling='striding' (x4), self_attention_model='rel_pos', untied biases, xscaling, batch_norm conv module, full
attention context.  Anything else raises NotImplementedError at construction (there is no fallback path).
"""
from __future__ import annotations

import math
import os
from collections import OrderedDict
from typing import Optional

import torch
from torch import nn

from .. import ops
from ..core import (AcousticEncodedRepresentation, BoolType, ChannelType, LengthsType, NeuralModule, NeuralType, SpectrogramType,
                    typecheck)
from ..flat import FlatParams
from ..packing import PackPlan

INF_VAL = 10000.0


def _pad8(n):
    return (n + 7) // 8 * 8


# ------------------------------------------------------------------------------------------------ parameter containers
class _FeedForward(nn.Module):  # conformer_modules.py:366 ConformerFeedForward
    def __init__(self, d_model, d_ff, use_bias=True):
        super().__init__()
        self.linear1 = nn.Linear(d_model, d_ff, bias=use_bias)
        self.linear2 = nn.Linear(d_ff, d_model, bias=use_bias)


class _Convolution(nn.Module):  # conformer_modules.py:236 ConformerConvolution
    def __init__(self, d_model, kernel_size, norm_type="batch_norm"):
        super().__init__()
        self.pointwise_conv1 = nn.Conv1d(d_model, d_model * 2, kernel_size=1)
        self.depthwise_conv = nn.Conv1d(d_model, d_model, kernel_size, groups=d_model, padding=0)
        # conformer_modules.py:293-306: the attribute is called batch_norm whatever the norm (state-dict key)
        self.batch_norm = nn.LayerNorm(d_model) if norm_type == "layer_norm" else nn.BatchNorm1d(d_model)
        self.pointwise_conv2 = nn.Conv1d(d_model, d_model, kernel_size=1)


class _RelPosMHA(nn.Module):  # multi_head_attention.py:212 RelPositionMultiHeadAttention
    def __init__(self, n_head, n_feat):
        super().__init__()
        self.h, self.d_k = n_head, n_feat // n_head
        self.linear_q = nn.Linear(n_feat, n_feat)
        self.linear_k = nn.Linear(n_feat, n_feat)
        self.linear_v = nn.Linear(n_feat, n_feat)
        self.linear_out = nn.Linear(n_feat, n_feat)
        self.linear_pos = nn.Linear(n_feat, n_feat, bias=False)
        self.pos_bias_u = nn.Parameter(torch.zeros(n_head, self.d_k))
        self.pos_bias_v = nn.Parameter(torch.zeros(n_head, self.d_k))


class ConformerLayer(nn.Module):  # conformer_modules.py:35 (parameter layout only; compute lives in ConformerEncoder)
    def __init__(self, d_model, d_ff, n_heads, conv_kernel_size, conv_norm_type="batch_norm"):
        super().__init__()
        self.fc_factor = 0.5
        self.norm_feed_forward1 = nn.LayerNorm(d_model)
        self.feed_forward1 = _FeedForward(d_model, d_ff)
        self.norm_conv = nn.LayerNorm(d_model)
        self.conv = _Convolution(d_model, conv_kernel_size, conv_norm_type)
        self.norm_self_att = nn.LayerNorm(d_model)
        self.self_attn = _RelPosMHA(n_heads, d_model)
        self.norm_feed_forward2 = nn.LayerNorm(d_model)
        self.feed_forward2 = _FeedForward(d_model, d_ff)
        self.norm_out = nn.LayerNorm(d_model)


class ConvSubsampling(nn.Module):  # subsampling.py:62 ('striding' :217-253 and 'dw_striding' :142-215; parameter layout only)
    def __init__(self, subsampling, subsampling_factor, feat_in, feat_out, conv_channels, causal=False):
        super().__init__()
        # causal = CausalConv2D stages (causal_convs.py:24-72): two zero rows / columns in front of the grid, one behind it, no
        # symmetric padding -- same parameters and state-dict keys, another sampling grid: extents floor(n / 2) + 1 per stage
        self.is_causal, self._pad = bool(causal), (2 if causal else 1)
        self._subsampling = subsampling
        self.subsampling_factor = subsampling_factor
        self._conv_channels = conv_channels
        self._sampling_num = int(math.log2(subsampling_factor))
        C = conv_channels
        layers = [nn.Conv2d(1, C, 3, stride=2, padding=1), nn.ReLU(True)]
        for _ in range(self._sampling_num - 1):
            if subsampling == "striding":
                layers += [nn.Conv2d(C, C, 3, stride=2, padding=1), nn.ReLU(True)]
            else:  # depthwise 3x3 stride 2 + pointwise 1x1: sequential indices 2,3,(4=ReLU), 5,6,(7) -- the reference's keys
                layers += [nn.Conv2d(C, C, 3, stride=2, padding=1, groups=C), nn.Conv2d(C, C, 1), nn.ReLU(True)]
        self.conv = nn.Sequential(*layers)
        f = feat_in
        for _ in range(self._sampling_num):
            f = (f + self._pad - 2) // 2 + 1
        self._feat_after = f
        self.out = nn.Linear(conv_channels * f, feat_out)

    def dw_stages(self):
        """[(depthwise conv, pointwise conv)] of the 'dw_striding' stack"""
        mods = list(self.conv)
        return [(mods[i], mods[i + 1]) for i in range(2, len(mods), 3)]


class RelPositionalEncoding(nn.Module):  # multi_head_attention.py:1056
    def __init__(self, d_model, dropout_rate, max_len=5000, xscale=None, dropout_rate_emb=0.0):
        super().__init__()
        self.d_model, self.xscale, self.max_len = d_model, xscale, max_len
        self.dropout_rate, self.dropout_rate_emb = dropout_rate, dropout_rate_emb

    def table(self, T: int, device, dtype) -> torch.Tensor:
        """pos_emb [2T-1, d]: row r <-> relative position T-1-r (multi_head_attention.py:1015-1035,1067-1100).
        A row depends on its relative position only, so -- exactly as the reference does with its `pe` buffer (extend_pe builds
        positions L-1 ... -(L-1) for L = max_len once, forward slices `pe[:, center - T + 1 : center + T]`) -- ONE table for the
        longest supported length lives on the device per dtype and a length-T table is a contiguous row slice of it: no host
        arithmetic, no host-to-device copy (which would also drain the launch queue) when a batch brings a new length."""
        key = (str(device), dtype)
        full = self._full.get(key) if hasattr(self, "_full") else None
        # rebuilt only when the table is too SHORT, and then for the longest length seen so far: it grows monotonically (the
        # previous form rebuilt on every change of max(max_len, T), i.e. at every batch after one longer than max_len)
        if full is None or full.shape[0] < 2 * T - 1:
            if not hasattr(self, "_full"):
                self._full, self._full_L = {}, {}
            L = max(int(self.max_len), T, self._full_L.get(key, 0))
            d = self.d_model
            pos = torch.arange(L - 1, -L, -1, dtype=torch.float32).unsqueeze(1)
            div = torch.exp(torch.arange(0, d, 2, dtype=torch.float32) * -(math.log(INF_VAL) / d))
            pe = torch.zeros(2 * L - 1, d)
            pe[:, 0::2] = torch.sin(pos * div)
            pe[:, 1::2] = torch.cos(pos * div)
            full = pe.to(device=device, dtype=dtype).contiguous()
            self._full[key], self._full_L[key] = full, L
        L = self._full_L[key]
        return full[L - T: L + T - 1]


class _Saved:
    """plain attribute bag for activations kept for backward"""
    pass


class _EncoderFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mel, length, token, module):
        ctx.module, ctx.saved, ctx.gset = module, None, None
        res = module._graphed_forward(mel, length)  # None: this call runs eagerly (warm-up, unstable shapes, graphs off)
        ctx.timed_gs = module._cur_gs
        caps = ()
        if res is not None:
            out, enc_len, ctx.gset, ctx.gen = res
        else:
            out, enc_len, ctx.saved = module._forward_impl(mel, length, save=True)
            caps = tuple(getattr(ctx.saved, "cap_out", ())) if ctx.saved is not None else ()
        ctx.mark_non_differentiable(enc_len)
        return (out, enc_len) + caps   # (+ the InterCTC captures, differentiable)

    @staticmethod
    def backward(ctx, dout, _, *dcaps):
        if ctx.gset is not None:
            ctx.module._graphed_backward(ctx.gset, ctx.gen, dout)
        else:
            if dcaps:
                ctx.module._backward_impl(ctx.saved, dout, dcaps)
            else:
                ctx.module._backward_impl(ctx.saved, dout)
        ctx.module._auto_end(ctx.timed_gs)
        ctx.saved = None
        return None, None, None, None


class _GraphSet:
    """the recorded forward / backward launch sequences of ONE (shape, configuration) key, their static inputs / outputs"""

    def __init__(self):
        self.calls = 0          # training forwards seen with this key (the first ones run eagerly: they also warm the caches up)
        self.fwd = self.bwd = None
        self.S = None           # saved activations between the forward and the backward CAPTURE (dropped afterwards)
        self.mel = self.length = self.out = self.enc_len = self.dout = None
        self.gen = 0            # forward replays so far; a backward must belong to the latest one
        self.failed = False
        # "auto" mode: the launch sequence is timed on the device (event pairs: forward start -> backward end) both ways and
        # the faster one is kept -- replaying ~1 000 graph nodes is not always faster than issuing them live (on ROCm 7 a node
        # costs the GPU's command processor about as much as a launch, and a step whose live issue time is well below its GPU
        # time has nothing to gain: Conformer-CTC-Large replays 0.7 ms SLOWER than it launches, Squeezeformer-Medium 2.6 ms faster)
        self.samples = {"eager": [], "graph": []}   # (start, end) event pairs
        self.open_pair = None   # [mode, start event, None] of the step in flight
        self.decided = None     # None (still measuring) | "eager" | "graph"
        self.inherited = False  # the decision was taken over from another padded length of the same configuration (no trial)


class ConformerEncoder(NeuralModule):
    @property
    def input_types(self):
        return OrderedDict({
            "audio_signal": NeuralType(("B", "D", "T"), SpectrogramType()),
            "length": NeuralType(tuple("B"), LengthsType()),
            "bypass_pre_encode": NeuralType(tuple(), BoolType(), optional=True),   # conformer_encoder.py:231
        })

    @property
    def output_types(self):
        return OrderedDict({
            "outputs": NeuralType(("B", "D", "T"), AcousticEncodedRepresentation()),
            "encoded_lengths": NeuralType(tuple("B"), LengthsType()),
        })

    def __init__(self, feat_in, n_layers, d_model, feat_out=-1, causal_downsampling=False, subsampling="striding",
                 subsampling_factor=4, subsampling_conv_chunking_factor=1, subsampling_conv_channels=-1, reduction=None,
                 reduction_position=None, reduction_factor=1, ff_expansion_factor=4, self_attention_model="rel_pos",
                 n_heads=4, att_context_size=None, att_context_probs=None, att_context_style="regular", xscaling=True,
                 untie_biases=True, pos_emb_max_len=5000, conv_kernel_size=31, conv_norm_type="batch_norm",
                 conv_context_size=None, use_bias=True, dropout=0.1, dropout_pre_encoder=0.1, dropout_emb=0.1,
                 dropout_att=0.0, stochastic_depth_drop_prob: float = 0.0, stochastic_depth_mode: str = "linear",
                 stochastic_depth_start_layer: int = 1, global_tokens: int = 0, global_tokens_spacing: int = 1,
                 global_attn_separate: bool = False, use_pytorch_sdpa: bool = False, use_pytorch_sdpa_backends=None,
                 sync_max_audio_length: bool = True, compute_dtype: Optional[torch.dtype] = None):
        super().__init__()
        bad = []
        if not ((subsampling == "striding" and subsampling_factor == 4) or
                (subsampling == "dw_striding" and subsampling_factor in (4, 8))):
            bad.append(f"subsampling={subsampling} x{subsampling_factor} (implemented: striding x4, dw_striding x4 / x8)")
        if self_attention_model not in ("rel_pos", "rel_pos_local_attn"):
            bad.append(f"self_attention_model={self_attention_model} (implemented: rel_pos, rel_pos_local_attn)")
        if self_attention_model == "rel_pos_local_attn" and global_tokens:
            bad.append("global_tokens > 0")
        if not untie_biases: bad.append("tied pos biases")
        if conv_norm_type not in ("batch_norm", "layer_norm"): bad.append(f"conv_norm_type={conv_norm_type} (implemented: batch_norm, layer_norm)")
        if not use_bias: bad.append("use_bias=False")
        if reduction: bad.append("reduction")
        if conv_kernel_size not in (3, 5, 9, 31): bad.append(f"conv_kernel_size={conv_kernel_size}")
        if d_model % n_heads or d_model % 4: bad.append("d_model not divisible by n_heads / 4")
        if bad:
            raise NotImplementedError("MI355X ConformerEncoder does not implement: " + ", ".join(bad))
        d_ff = d_model * ff_expansion_factor
        self.d_model, self.n_layers, self._feat_in = d_model, n_layers, feat_in
        self.n_heads, self.d_k, self.d_ff = n_heads, d_model // n_heads, d_ff
        self.conv_kernel_size = conv_kernel_size
        self.subsampling_factor = subsampling_factor
        self.att_context_style, self.self_attention_model = att_context_style, self_attention_model
        # limited attention context (conformer_encoder.py:863-894 `_calc_context_sizes`): a list of [left, right] windows, the first
        # one is the default (evaluation); training draws one per step with att_context_probs when there are several (:620-625)
        self.att_context_size_all, self.att_context_size, self.att_context_probs = self._calc_att_context(
            att_context_size, att_context_probs, att_context_style)
        if self_attention_model == "rel_pos_local_attn":
            # Longformer-style sliding window with the relative-position term inside the window
            # (RelPositionMultiHeadAttentionLongformer, multi_head_attention.py:357-586; LocalAttRelPositionalEncoding :1103-1148).
            # Its overlapping-chunk arithmetic computes the dense banded attention |j - i| <= w with the SAME positional rows
            # p(i - j) the full table holds (the sinusoid depends on the position value only), zeroed rows for padded queries --
            # i.e. the 'regular' limited context [w, w] of the rel_pos model, which is how it runs here (the band test inside
            # mi355x_relpos_softmax_fwd_ctx).  The reference's diagonal-wise positional sum is only consistent for left == right
            # (what the recipes use: conf/fastconformer/long_fastconformer/*.yaml), other windows are refused.
            if max(self.att_context_size) <= 0:
                raise ValueError("When using local attention, context size must be set > 0")
            if len(self.att_context_size_all) != 1 or self.att_context_size[0] != self.att_context_size[1]:
                raise NotImplementedError("MI355X ConformerEncoder: rel_pos_local_attn needs one window att_context_size = [w, w]")
            self.att_context_style = "regular"
        self._ctx = (0, -1, -1)   # (style id, left, right) of the forward in flight
        self.sync_max_audio_length = sync_max_audio_length
        self.xscale = math.sqrt(d_model) if xscaling else None
        self.dropout, self.dropout_pre_encoder, self.dropout_att = dropout, dropout_pre_encoder, dropout_att
        self.dropout_emb = dropout_emb
        if subsampling_conv_channels == -1:
            subsampling_conv_channels = d_model
        self.subsampling = subsampling
        self.pre_encode = ConvSubsampling(subsampling, subsampling_factor, feat_in, d_model, subsampling_conv_channels,
                                          causal=causal_downsampling)
        self._feat_out = d_model
        self.pos_emb_max_len = pos_emb_max_len
        self.pos_enc = RelPositionalEncoding(d_model, dropout_pre_encoder, pos_emb_max_len, self.xscale, dropout_emb)
        # conv module options: LayerNorm instead of BatchNorm (conformer_modules.py:293-306, 335-340) and CausalConv1D's asymmetric
        # padding (conv_context_size = 'causal' | [left, right], left + right + 1 = kernel; conformer_encoder.py:896-907, causal_convs.py:89-150)
        self.conv_norm_type = conv_norm_type
        if conv_context_size is None:
            self.conv_context_size, self.conv_pad_left = None, -1
        else:
            if isinstance(conv_context_size, str):
                if conv_context_size != "causal":
                    raise ValueError("Invalid conv_context_size! It should be the string 'causal' or a list of two integers.")
                cc_ = [conv_kernel_size - 1, 0]
            else:
                cc_ = [int(v) for v in conv_context_size]
                if len(cc_) != 2 or cc_[0] + cc_[1] + 1 != conv_kernel_size:
                    raise ValueError(f"Invalid conv_context_size: {conv_context_size}!")
            self.conv_context_size = cc_
            self.conv_pad_left = -1 if cc_[0] == cc_[1] else cc_[0]
        self.layers = nn.ModuleList([ConformerLayer(d_model, d_ff, n_heads, conv_kernel_size, conv_norm_type) for _ in range(n_layers)])
        # feat_out projection (conformer_encoder.py:474-479, 738-739): a Linear(d_model, feat_out) behind the last layer
        if feat_out > 0 and feat_out != d_model:
            self.out_proj = nn.Linear(d_model, feat_out)
            self._feat_out = feat_out
        else:
            self.out_proj = None
        # stochastic depth (conformer_encoder.py:486-488, 696-707; parts/utils/regularization_utils.py:18-64)
        self.layer_drop_probs = self._stochastic_depth_probs(n_layers, stochastic_depth_drop_prob, stochastic_depth_mode,
                                                             stochastic_depth_start_layer)
        # InterCTC (parts/mixins/interctc_mixin.py; conformer_encoder.py:724-736): the model lists the layers whose outputs it wants;
        # `captured[l]` = [B, D, T'] output of layer l (through out_proj), part of the autograd graph of the forward that made it
        self.capture_layers = []
        self.captured = {}
        self.max_audio_length = pos_emb_max_len
        self._init_engine(compute_dtype, tail=lambda n: n.endswith("self_attn.linear_pos.weight"))

    def _init_engine(self, compute_dtype, tail=None):
        """engine state shared by the encoders built on this class (not part of the state-dict)"""
        self.compute_dtype = compute_dtype  # None: bf16 under torch autocast(bf16), else fp32
        # SyncBatchNorm semantics across data-parallel ranks (trainer.sync_batchnorm: true in the recipe)
        self.sync_batchnorm = True
        # The element count of the synchronised statistics is the SUM of the ranks' own B*T' (torch.nn.SyncBatchNorm gathers
        # the per-rank counts; ranks padded to different lengths hold different counts): it rides as one more f64 element
        # behind the [2,d] sums through the same all-reduce and is read from device memory by the BatchNorm kernels
        # (mi355x_bn_finalize_dev_count / mi355x_bn_swish_bwd_apply_dev_count) -- exact for ragged ranks, no host round trip.
        self._syncbn_group = None
        self._syncbn_mailbox = None  # nemo_amd.mailbox.StatsMailbox when MI355X_SYNCBN_MAILBOX=1 and every rank could map its peers
        self._syncbn_mailbox_tried = False
        self.syncbn_profile = None  # a list while bench.py measures the exposed time of the statistics exchanges
        self.use_flash_attention = True  # bf16 + (padded) d_k == 64: fused kernels; otherwise the GEMM + softmax-kernel path
        self.flash_pad_heads = os.environ.get("MI355X_FLASH_PAD_HEADS", "1") != "0"  # d_k < 64 -> heads zero-padded to 64 (_geometry)
        # PACKED ROWS ("length-aware kernels skipping padded frames"): the row-wise chain of the layers -- LayerNorms,
        # feed-forward / projection / pointwise-conv GEMMs, residuals, every weight gradient -- runs on the sum_b L_b valid frames
        # of a ragged batch instead of on B * T'_max rows; the fused attention addresses utterance b at row_offsets[b]; only the
        # depthwise conv + BatchNorm + Swish core of the conv module keeps the padded [B, T', d] grid, because the reference's
        # batch statistics run over padded frames too (conformer_modules.py:297,330-331; oracle/packed_ref.py states the semantics).
        # "auto": pack when the caller handed the lengths over on the host (`length.host_lengths`, no device sync) and at least
        # `packed_min_padding` of the rows are padding; True: always (reads the lengths back if it has to); False: never.
        _pk = os.environ.get("MI355X_PACKED", "auto").lower()
        self.packed_rows = {"0": False, "false": False, "1": True, "true": True}.get(_pk, "auto")
        self.packed_min_padding = float(os.environ.get("MI355X_PACKED_MIN_PADDING", "0.02"))
        self.packed_last = None   # diagnostics: (packed rows, padded rows) of the last forward that packed, else None
        # one launch for a layer's norm_out and the next layer's norm_feed_forward1 (d = 512; MI355X_LN2=0: two launches)
        self.fuse_layer_boundary_norms = os.environ.get("MI355X_LN2", "1") != "0"
        self.fuse_boundary_bwd = os.environ.get("MI355X_LN2_BWD", "1") != "0"  # ... and their backwards (mi355x_layernorm2_bwd)
        self.flash_delta_residual = os.environ.get("MI355X_FLASH_DELTA_LO", "1") != "0"  # (A/B switch of the delta fix)
        self.grad_ready_hook = None  # callable(start, end) on the flat gradient buffer (data-parallel bucketing)
        # linear_pos weights of all layers sit together at the tail: their gradients come from ONE batched GEMM
        self._flatp = FlatParams(self, tail=tail)
        self.wgrad_side_stream = os.environ.get("MI355X_WGRAD_STREAM", "1") != "0"
        self._wg_stream = None
        self._wgrad_join_per_layer = True
        self.wgrad_grouped = os.environ.get("MI355X_WGRAD_GROUPED", "1") != "0"
        # where a layer's grouped weight-gradient launch enters the side stream: 0 = at the end of the layer's own backward (it then
        # runs beside the NEXT layer's first GEMMs), 1 / 2 = inside the next layer's backward, in front of its conv-module
        # elementwise block / its attention backward (_defer_point; profiles/r5_wgrad_lane.md)
        # Round 5, same box, alternating: live launches 40.15 -> 39.68 ms with 1 (39.93 with 2), launch tapes 40.33 -> 40.08 ms.
        self.wgrad_defer = int(os.environ.get("MI355X_WGRAD_DEFER", "1"))
        # how many layers' weight gradients go out as ONE grouped launch (with wgrad_defer): larger groups fill whole rounds of the
        # chip with longer K loops per workgroup and meet the main chain less often (1: 39.29 ms, 2: 39.17 ms same box, in-process)
        self.wgrad_layers = int(os.environ.get("MI355X_WGRAD_LAYERS", "2"))
        self.posproj_side = os.environ.get("MI355X_POSPROJ_SIDE", "1") != "0"  # linear_pos weight gradients behind their producers on the side stream
        self._wg_pending, self._wg_rows = None, None
        self.conv2_implicit = os.environ.get("MI355X_CONV2_IMPLICIT", "1") != "0"
        # conv module backward: BatchNorm + Swish backward inside the depthwise backward's tile staging (MI355X_BN_DW_FUSE=0: two launches)
        self.fuse_bn_dwconv_bwd = os.environ.get("MI355X_BN_DW_FUSE", "1") != "0"
        self.fuse_glu_dwconv_bwd = os.environ.get("MI355X_GLU_DW_FUSE", "1") != "0"   # ... and the GLU backward in its write-out
        self.tap_reduce_side = os.environ.get("MI355X_TAP_REDUCE_SIDE", "1") != "0"   # second stage of the depthwise tap gradients on the side stream
        # sub-sampling backward: GEMM tiles / K-tiles that lie entirely beyond an utterance's length are skipped (row_len hints)
        self.pad_tile_skip = os.environ.get("MI355X_PAD_SKIP", "1") != "0"
        self.fuse_glu_dwconv_fwd = os.environ.get("MI355X_GLU_DW_FUSE_FWD", "1") != "0"   # forward: GLU in the depthwise conv's tile staging
        self.ln_cast_fuse = os.environ.get("MI355X_LN_CAST_FUSE", "1") != "0"
        # one-launch feed-forward blocks (csrc/ffn.hip).  OFF by default: parity-green and 10 % faster than the GEMM pair in the
        # forward direction, but inside the training step the pair of fused launches measured +1.1 ms (40.65 vs 39.55 ms, same box,
        # profiles/r4_ffn_fused.md): at 64 tokens per workgroup the weight stream through the 64 B/clk vector-memory path, the MFMA
        # work and the Swish / dropout VALU work each cost ~28 us per launch and overlap only partly.  MI355X_FFN_FUSED=1 enables.
        self.ffn_fused = os.environ.get("MI355X_FFN_FUSED", "0") != "0"
        # bf16 feed-forward pair: linear1's epilogue stores swish'(h) * dropout mask instead of h, the Swish-gradient GEMM's epilogue
        # is one multiply (MI355X_EPI_SWISH_DROP_G / _DSWISH_G; MI355X_SWISH_G=0: the pre-activation form)
        self.swish_g = os.environ.get("MI355X_SWISH_G", "1") != "0"
        self._saving = False        # the forward in progress keeps its activations for a backward
        self.dpos_side_stream = os.environ.get("MI355X_DPOS_STREAM", "1") != "0"
        self.sub_wgrad_side_stream = os.environ.get("MI355X_SUB_WGRAD_STREAM", "1") != "0"
        self._plans = {}
        self._ws = {}
        self._ws_retired = []
        self._pos_cache = {}
        self._step_seed = 0
        self._weights_version = -1
        self._token = None
        # ---- replayable launch sequences (nemo_amd/graphs.py).  MI355X_GRAPHS=0 keeps every step on the eager sequencer.
        # MI355X_GRAPHS: 0 = always live launches, 1 = always replay, auto (default) = time both on the device, keep the faster
        _g = os.environ.get("MI355X_GRAPHS", "auto")
        self.use_graphs = _g != "0"
        self.graph_auto = _g not in ("0", "1", "2")
        # how a recorded segment runs again: as a LAUNCH TAPE (csrc/tape.hip: the captured nodes re-issued as live launches from
        # one C loop; default) or, MI355X_TAPE=0, as a hipGraph replay (3-4 % slower on the device timeline, r3_host_issue.md)
        self.graph_tape = os.environ.get("MI355X_TAPE", "1") != "0"
        # MI355X_GRAPHS_BWD_LIVE=1: only the FORWARD sequence is replayed (its ~300 short launches are where the Python sequencer
        # cannot keep ahead of the device); the backward keeps live launches, whose issue order in time -- weight gradients entering
        # the side stream when the host gets there -- measured better on the device than the replay's (profiles/r5_launch_modes.md)
        self.graph_bwd_live = os.environ.get("MI355X_GRAPHS_BWD_LIVE", "0") != "0"
        self.graph_warmup = 2       # eager training forwards per key before its launch sequence is captured
        self.graph_trials = 4       # auto: timed steps per mode, alternating
        # distinct (shape, configuration) keys kept.  A recorded sequence costs host memory only (its launches point into the ONE
        # step arena every padded length shares), so a duration-bucketed / `pad_to`-quantised loader keeps one per padded length
        self.max_graph_sets = int(os.environ.get("MI355X_GRAPH_SETS", "64"))
        self.graph_warmup_inherited = 1   # ... for a padded length whose configuration has already been decided by another length
        self.replayed_steps = self.live_steps = 0   # diagnostics (bench.py): training forwards served from a recording / issued live
        self._graph_sets = OrderedDict()
        self._cur_gs = None
        # ---- step-scoped bump allocation of the sequencer's tensors (nemo_amd/arena.py); MI355X_ARENA=0: torch's caching allocator
        from ..arena import Arena
        # Both the arena and the recorded launch sequences keep ONE set of activations per encoder: they need the forward ->
        # backward -> next forward discipline of a training loop.  `step_scope()` (entered by the model's fit_step) switches them on;
        # a module driven by hand (two stochastic forwards, then backward through the first -- legal torch) stays on torch's allocator
        # and live launches unless MI355X_ARENA=1 / MI355X_GRAPHS=1 force them.
        _a = os.environ.get("MI355X_ARENA", "auto")
        self.use_arena = _a != "0"
        self.arena_forced = _a == "1"
        self._in_step = False
        self._arena_f, self._arena_b, self._arena = Arena("forward"), Arena("backward"), None
        self._fwd_serial = 0        # forwards so far: a backward must belong to the latest one (its activations live in the arena)
        self._capture = None        # the SegmentedCapture while a sequence is being recorded
        self._wg_forked = False     # recording: the weight-gradient stream has joined the capture and not re-joined yet
        self._force_pack = False    # record the weight-image pack unconditionally (a replayed forward always re-packs)
        self._step_word = None      # device-side dropout step word (int32), advanced by the forward graph

    # ------------------------------------------------------------------ reference API surface
    def set_max_audio_length(self, max_audio_length):
        self.max_audio_length = max_audio_length

    def update_max_seq_length(self, seq_length: int, device):
        """conformer_encoder.py:761-779.  The reference all-reduces MAX + .item() (a host sync) every forward to size a
        cached table; our positional table is built per T on demand, so only the bookkeeping remains."""
        if seq_length > self.max_audio_length:
            self.set_max_audio_length(seq_length)

    def flat_parameters(self) -> FlatParams:
        self._flatp.ensure()
        return self._flatp

    def weights_updated(self):
        """call after an optimizer step / load_state_dict: GEMM operand images are re-packed on next forward"""
        self._weights_version += 1

    def load_state_dict(self, *a, **k):
        r = super().load_state_dict(*a, **k)
        self.weights_updated()
        return r

    # ------------------------------------------------------------------ forward (typed)
    @staticmethod
    def _calc_att_context(att_context_size, att_context_probs, att_context_style):
        """the reference's normalisation and checks of att_context_size / att_context_probs (conformer_encoder.py:863-894)"""
        if att_context_style not in ("regular", "chunked_limited"):
            raise ValueError(f"att_context_style={att_context_style}")
        if att_context_size:
            all_ = [list(x) for x in ([att_context_size] if isinstance(list(att_context_size)[0], int) else list(att_context_size))]
            for i, cs in enumerate(all_):
                if att_context_style == "chunked_limited":
                    if cs[0] > 0 and cs[0] % (cs[1] + 1) > 0:
                        raise ValueError(f"att_context_size[{i}][0] % (att_context_size[{i}][1] + 1) should be zero!")
                    if cs[1] < 0 and len(all_) <= 1:
                        raise ValueError(f"Right context (att_context_size[{i}][1]) can not be unlimited for chunked_limited style!")
        else:
            all_ = [[-1, -1]]
        if att_context_probs:
            if len(att_context_probs) != len(all_):
                raise ValueError("The size of the att_context_probs should be the same as att_context_size.")
            probs = list(att_context_probs)
            if sum(probs) != 1:
                raise ValueError("The sum of numbers in att_context_probs should be equal to one to be a distribution.")
        else:
            probs = [1.0 / len(all_)] * len(all_)
        return all_, all_[0], probs

    @staticmethod
    def _stochastic_depth_probs(n, p, mode, start):
        """compute_stochastic_depth_drop_probs (parts/utils/regularization_utils.py:18-64)"""
        if not (0 <= p < 1.0):
            raise ValueError("stochastic_depth_drop_prob has to be in [0, 1).")
        if not (1 <= start <= n):
            raise ValueError("stochastic_depth_start_layer has to be in [1, num layers].")
        probs = [0.0] * start
        L = n - start
        if L > 0:
            if mode == "linear":
                probs += [l / L * p for l in range(1, L + 1)]
            elif mode == "uniform":
                probs += [p] * L
            else:
                raise ValueError(f'stochastic_depth_mode has to be one of ["linear", "uniform"]. Current value: {mode}')
        return probs

    def _live_only(self):
        """options whose launch sequence changes from step to step (a layer dropped at random) or that hand extra differentiable
        outputs to the caller (InterCTC captures): issued live, never from a recorded sequence"""
        return ((self.training and any(p > 0.0 for p in self.layer_drop_probs)) or bool(self.capture_layers)
                or getattr(self, "_bypass_now", False))

    def _out_proj_fwd(self, x, M, dev):
        """y [M, feat_out] = x [M, d] @ W^T + b, fp32 (exact-fp32 MFMA GEMM) whatever the compute dtype"""
        fo, d = self._feat_out, self.d_model
        y = self._new(M, fo, dtype=torch.float32, device=dev)
        ops.gemm(x, self.out_proj.weight, y, M, fo, d, d, d, fo, bias=self.out_proj.bias)
        return y

    def _out_proj_bwd(self, dy, x, M, dev):
        """dx [M, d] = dy @ W; W.grad += dy^T x; b.grad += column sums of dy"""
        fo, d = self._feat_out, self.d_model
        dx = self._new(M, d, dtype=torch.float32, device=dev)
        ops.gemm(dy, self.out_proj.weight, dx, M, d, fo, fo, d, d, transB=True)
        ops.gemm(dy, x, self.out_proj.weight.grad, fo, d, M, fo, d, d, transA=True, transB=True, atomic=True, c_dtype=ops.F32)
        ops.colsum(dy, self.out_proj.bias.grad, M, fo)
        return dx

    def set_default_att_context_size(self, att_context_size):
        """conformer_encoder.py:896-910"""
        if att_context_size is not None:
            self.att_context_size = list(att_context_size)

    def _ctx_limited_any(self):
        """does any configured window limit the context?  Then attention runs on the GEMM + softmax-kernel path (the window is a
        mask inside mi355x_relpos_softmax_fwd_ctx); the fused kernels implement the unlimited case."""
        return any(l >= 0 or r >= 0 for l, r in self.att_context_size_all + [self.att_context_size])

    def _flash_ok(self):
        return self.use_flash_attention and not self._ctx_limited_any()

    def _pick_ctx(self):
        import random
        if self.training and len(self.att_context_size_all) > 1:
            cs = random.choices(self.att_context_size_all, weights=self.att_context_probs)[0]
        else:
            cs = self.att_context_size
        limited = cs[0] >= 0 or cs[1] >= 0
        self._ctx = ((1 if self.att_context_style == "regular" else 2) if limited else 0, int(cs[0]), int(cs[1]))

    @typecheck()
    def forward(self, audio_signal, length, cache_last_channel=None, cache_last_time=None, cache_last_channel_len=None,
                bypass_pre_encode=False):
        if cache_last_channel is not None:
            raise NotImplementedError("streaming caches are not on the training hot path")
        if bypass_pre_encode and audio_signal.shape[-1] != self.d_model:
            raise ValueError(f"If bypass_pre_encode is True, audio_signal should have shape (batch, n_frame, {self.d_model}) "
                             f"but got last dimension {audio_signal.shape[-1]}.")  # conformer_encoder.py:563-568
        if not bypass_pre_encode and audio_signal.shape[-2] != self._feat_in:
            raise ValueError(f"If bypass_pre_encode is False, audio_signal should have shape (batch, {self._feat_in}, "
                             f"n_frame) but got last dimension {audio_signal.shape[-2]}.")  # conformer_encoder.py:569-578
        self._bypass_now = bool(bypass_pre_encode)   # pre-encoded frames [B, T', d_model]: the sub-sampling stack is skipped
        if length is None:
            length = audio_signal.new_full((audio_signal.size(0),), audio_signal.size(1 if bypass_pre_encode else -1), dtype=torch.int64)
        self._flatp.ensure(audio_signal.device)
        self._pick_ctx()
        if self._token is None or self._token.device != audio_signal.device:
            self._token = torch.zeros(1, device=audio_signal.device, requires_grad=True)
        need_grad = torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters())
        if need_grad:
            res = _EncoderFn.apply(audio_signal, length, self._token, self)
            self.captured = dict(zip(self.capture_layers, res[2:]))
            return res[0], res[1]
        out, enc_len, S_ = self._forward_impl(audio_signal, length, save=bool(self.capture_layers))
        self.captured = dict(zip(self.capture_layers, getattr(S_, "cap_out", ()))) if S_ is not None else {}
        return out, enc_len

    # ------------------------------------------------------------------ helpers
    def _cdt(self):
        if self.compute_dtype is not None:
            return self.compute_dtype
        if torch.is_autocast_enabled() and torch.get_autocast_gpu_dtype() == torch.bfloat16:
            return torch.bfloat16
        return torch.float32

    def _new(self, *shape, dtype, device):
        """a tensor of this step's forward / backward: from the phase's arena when one is active, else from torch's allocator"""
        a = self._arena
        if a is None:
            return torch.empty(*shape, dtype=dtype, device=device)
        return a.take(shape, dtype, device)

    def _phase(self, which, device):
        """start of a forward ('f') / backward ('b') pass: rewind that phase's arena and make it the current one"""
        if not (self.use_arena and (self._in_step or self.arena_forced) and torch.device(device).type == "cuda"):
            self._arena = None
            return
        a = self._arena_f if which == "f" else self._arena_b
        a.rewind(torch.device(device))
        self._arena = None if a.disabled else a

    def _buf(self, name, shape, dtype, device, zero=False):
        """persistent workspace `name`: ONE allocation per (name, dtype) that grows to the largest size ever asked for and is handed
        out as a view -- with variable-length batches every step brings new shapes, and a workspace per shape meant a fresh
        (up to 1.3 GB) allocation per step; `zero` only guarantees zeros at the first use of a fresh allocation"""
        n = 1
        for k in shape:
            n *= int(k)
        key = (name, dtype)
        t = self._ws.get(key)
        if t is None or t.device != device or t.numel() < n:
            cap = n if t is None or t.device != device else max(n, int(t.numel() * 1.25))
            if t is not None:
                # a recorded launch sequence (hipGraph) may hold the old address: the outgrown buffer is kept, not freed
                # (geometric growth bounds what accumulates to a few times the largest size)
                self._ws_retired.append(t)
            t = torch.zeros(cap, dtype=dtype, device=device) if zero else torch.empty(cap, dtype=dtype, device=device)
            self._ws[key] = t
        return t[:n].view(shape)

    def _plan(self, cdt, device):
        # (the geometry -- head width of the packed attention images -- follows run-time flags: use_flash_attention, flash_pad_heads)
        key = (cdt, str(device), self._flatp.generation, self._geometry(cdt))
        plan = self._plans.get(key)
        if plan is None:
            self._plans = {}  # parameters moved: drop images of the old storage
            p = PackPlan(cdt, device)
            pf = PackPlan(torch.float32, device)
            srcs = []
            pe = self.pre_encode
            C_, F2 = pe._conv_channels, pe._feat_after
            if self.subsampling == "striding":
                p.add_conv3x3("pre.w2", pe.conv[2].weight.data); p.add_conv3x3("pre.w2t", pe.conv[2].weight.data, transpose=True)
                # conv2 input-gradient as four implicit GEMMs, one per (t1, f1) parity class: image [ci][(slot, co)] of the
                # taps that reach the class (a stride-2 3x3 conv touches an even position through k = 1 only, an odd one
                # through 0, 2)
                w2flat = pe.conv[2].weight.data.view(-1)
                for par_t in (0, 1):
                    for par_f in (0, 1):
                        slots = self._dgrad_slots(par_t, par_f, pe._pad)
                        name = f"pre.w2d{par_t}{par_f}"
                        p.new_image(name, C_, len(slots) * C_)
                        for si_, (kh, kw) in enumerate(slots):
                            p.add_block(name, w2flat[kh * 3 + kw:], C_, C_, col_off=si_ * C_, sr1=9, sc1=9 * C_)
            else:  # 'dw_striding': the pointwise convolutions are plain [C, C] GEMM operands
                for si_, (_, pw) in enumerate(pe.dw_stages()):
                    p.add_matrix(f"pre.pw{si_}", pw.weight.data.view(C_, C_))
                    p.add_matrix(f"pre.pw{si_}t", pw.weight.data.view(C_, C_), True)
            p.add_fc_permuted("pre.out", pe.out.weight.data, C_, F2)
            p.add_fc_permuted("pre.outt", pe.out.weight.data, C_, F2, transpose=True)
            d_, H_, dk = self.d_model, self.n_heads, self.d_k
            _, dkp, dA = self._geometry(cdt)
            for i, L in enumerate(self.layers):
                for ff, m in (("ff1", L.feed_forward1), ("ff2", L.feed_forward2)):
                    if self._ffn_fused_ok(cdt):
                        # the fused feed-forward kernels (csrc/ffn.hip) stream their weights in the order their steps consume them
                        p.add_ffn(f"L{i}.{ff}", m.linear1.weight.data, m.linear2.weight.data)
                        continue
                    p.add_matrix(f"L{i}.{ff}.w1", m.linear1.weight.data); p.add_matrix(f"L{i}.{ff}.w1t", m.linear1.weight.data, True)
                    p.add_matrix(f"L{i}.{ff}.w2", m.linear2.weight.data); p.add_matrix(f"L{i}.{ff}.w2t", m.linear2.weight.data, True)
                a = L.self_attn
                qkv = [a.linear_q.weight.data, a.linear_k.weight.data, a.linear_v.weight.data]
                if dkp == dk:
                    p.add_concat(f"L{i}.att.wqkv", qkv); p.add_concat(f"L{i}.att.wqkvt", qkv, transpose=True)
                    p.add_matrix(f"L{i}.att.wo", a.linear_out.weight.data); p.add_matrix(f"L{i}.att.wot", a.linear_out.weight.data, True)
                    p.add_matrix(f"L{i}.att.wpos", a.linear_pos.weight.data)
                    pf.new_image(f"L{i}.att.bqkv", 1, 3 * d_)
                    for j, b in enumerate((a.linear_q.bias.data, a.linear_k.bias.data, a.linear_v.bias.data)):
                        pf.add_block(f"L{i}.att.bqkv", b, 1, d_, col_off=j * d_, sr1=0, sc1=1)
                else:  # zero-padded heads (see _geometry): head h = rows / columns h*dkp .. h*dkp + dk of the images
                    p.new_image(f"L{i}.att.wqkv", 3 * dA, d_); p.new_image(f"L{i}.att.wqkvt", d_, 3 * dA)
                    p.new_image(f"L{i}.att.wpos", dA, d_)
                    p.new_image(f"L{i}.att.wo", d_, dA); p.new_image(f"L{i}.att.wot", dA, d_)
                    wo = a.linear_out.weight.data  # [d, H*dk]
                    pf.new_image(f"L{i}.att.bqkv", 1, 3 * dA)
                    pf.new_image(f"L{i}.att.bu", 1, dA); pf.new_image(f"L{i}.att.bv", 1, dA)
                    for h in range(H_):
                        for j, w in enumerate(qkv):
                            p.add_block(f"L{i}.att.wqkv", w.view(-1)[h * dk * d_:], dk, d_, row_off=j * dA + h * dkp, sr1=d_, sc1=1)
                            p.add_block(f"L{i}.att.wqkvt", w.view(-1)[h * dk * d_:], d_, dk, col_off=j * dA + h * dkp, sr1=1, sc1=d_)
                        p.add_block(f"L{i}.att.wpos", a.linear_pos.weight.data.view(-1)[h * dk * d_:], dk, d_, row_off=h * dkp,
                                    sr1=d_, sc1=1)
                        p.add_block(f"L{i}.att.wo", wo.view(-1)[h * dk:], d_, dk, col_off=h * dkp, sr1=d_, sc1=1)
                        p.add_block(f"L{i}.att.wot", wo.view(-1)[h * dk:], dk, d_, row_off=h * dkp, sr1=1, sc1=d_)
                        for j, b in enumerate((a.linear_q.bias.data, a.linear_k.bias.data, a.linear_v.bias.data)):
                            pf.add_block(f"L{i}.att.bqkv", b[h * dk:], 1, dk, col_off=j * dA + h * dkp, sr1=0, sc1=1)
                        pf.add_block(f"L{i}.att.bu", a.pos_bias_u.data.view(-1)[h * dk:], 1, dk, col_off=h * dkp, sr1=0, sc1=1)
                        pf.add_block(f"L{i}.att.bv", a.pos_bias_v.data.view(-1)[h * dk:], 1, dk, col_off=h * dkp, sr1=0, sc1=1)
                c = L.conv
                p.add_matrix(f"L{i}.conv.pw1", c.pointwise_conv1.weight.data); p.add_matrix(f"L{i}.conv.pw1t", c.pointwise_conv1.weight.data, True)
                p.add_matrix(f"L{i}.conv.pw2", c.pointwise_conv2.weight.data); p.add_matrix(f"L{i}.conv.pw2t", c.pointwise_conv2.weight.data, True)
            p.finalize(); pf.finalize()
            plan = (p, pf, None, -2)
            self._plans[key] = plan
        if plan[3] != self._weights_version or self._force_pack:
            plan[0].run(); plan[1].run()
            plan = (plan[0], plan[1], plan[2], self._weights_version)
            self._plans[key] = plan
        return plan[0], plan[1]

    # ------------------------------------------------------------------ replayable launch sequences (nemo_amd/graphs.py)
    def _eager_point(self, fn):
        """a step of the forward / backward sequence that must stay a live host call (collective, user hook): executed in place
        on the eager path, recorded between two graph segments while a sequence is being captured"""
        if self._capture is None:
            fn()
        else:
            self._capture.cut(fn)

    def _hook(self, lo, hi):
        """grad_ready_hook(lo, hi), looked up when it RUNS (a replayed sequence calls whatever hook is installed then)"""
        self._eager_point(lambda: self.grad_ready_hook(lo, hi) if self.grad_ready_hook is not None else None)

    def _graph_key(self, mel, length):
        return (tuple(mel.shape), self._cdt(), str(mel.device), self._flatp.generation, self._syncbn_world(),
                self.grad_ready_hook is not None, self._wgrad_join_per_layer, self.wgrad_side_stream, self.wgrad_grouped,
                self.dpos_side_stream, self.sub_wgrad_side_stream, self.conv2_implicit, self.ln_cast_fuse, self.fuse_bn_dwconv_bwd, self.fuse_glu_dwconv_bwd, self.fuse_glu_dwconv_fwd, self.tap_reduce_side, self.pad_tile_skip,
                self.use_flash_attention, self.flash_delta_residual, self.syncbn_profile is not None, self.graph_tape, self.swish_g, self.dropout, self.dropout_att, self.dropout_emb, self.dropout_pre_encoder,
                self.wgrad_defer, self.graph_bwd_live, self.posproj_side, self.wgrad_layers, self._ctx)

    def _auto_begin(self, gs, mode):
        e0 = torch.cuda.Event(enable_timing=True)
        e0.record()
        gs.open_pair = [mode, e0, None]

    def _auto_end(self, gs):
        """backward of the step has been issued: stamp the end of the timed interval"""
        op = gs.open_pair if gs is not None else None
        if op is not None and op[2] is None:
            op[2] = torch.cuda.Event(enable_timing=True)
            op[2].record()
            gs.samples[op[0]].append((op[1], op[2]))
            gs.open_pair = None

    def _auto_decide(self, gs):
        """both ways have been timed `graph_trials` times, alternating step by step (drift and clock changes hit both alike):
        keep the faster one.  One host wait on the last end event -- once per shape, outside any timed region."""
        for v in gs.samples.values():
            for _, e1 in v:
                e1.synchronize()
        med = lambda v: sorted(v)[len(v) // 2]
        t = {m: med([e0.elapsed_time(e1) for e0, e1 in v]) for m, v in gs.samples.items()}
        gs.decided = "graph" if t["graph"] < t["eager"] else "eager"
        gs.auto_ms = {m: round(x, 3) for m, x in t.items()}
        gs.samples = {"eager": [], "graph": []}
        if gs.decided == "eager":
            gs.fwd = gs.bwd = gs.S = gs.out = gs.enc_len = gs.mel = gs.dout = None  # the pool goes with the graphs

    def _graphed_forward(self, mel, length):
        """-> (out, enc_len, graph set, generation) from the recorded sequence, or None when this call has to run eagerly"""
        self._cur_gs = None
        if not (self.use_graphs and self.training and mel.is_cuda) or ops.GEMM_PROFILE is not None or self._live_only():
            self.live_steps += 1
            return None
        if self._packing_plan(length, mel.shape[0], mel.shape[2], peek=True) is not None:
            self.live_steps += 1
            if not getattr(self, "_packed_over_graphs_noted", False) and self.packed_rows == "auto":
                self._packed_over_graphs_noted = True
                import logging
                logging.getLogger(__name__).info(
                    "ConformerEncoder: packed rows (packed_rows='auto', host lengths attached to the length tensor) pre-empt recorded "
                    "launch sequences for ragged batches; set encoder.packed_rows = False to keep the replay path")
            return None  # packed rows: the row count changes with every batch, a recorded sequence holds one shape
        key = self._graph_key(mel, length)
        gs = self._graph_sets.get(key)
        auto = self.graph_auto
        if gs is None:
            gs = self._graph_sets[key] = _GraphSet()
            while len(self._graph_sets) > self.max_graph_sets:
                self._graph_sets.popitem(last=False)
            if not auto:
                # (forced replay: a later padded length of a configuration that has already been recorded once needs one live
                #  visit -- the per-length caches -- instead of graph_warmup of them)
                gs.inherited = any(g2 is not gs and g2.bwd is not None and k2[1:] == key[1:] for k2, g2 in self._graph_sets.items())
        else:
            self._graph_sets.move_to_end(key)
        gs.calls += 1
        self._cur_gs = gs
        if auto and gs.decided is None and gs.open_pair is None:
            # Variable-length loaders: whether a recorded sequence beats live launches is a property of the recipe (is the host or the
            # device the bound?), not of the padded length -- the first length that finishes its trial decides for every other
            # length of the same configuration (same key but for the shape), which then needs no 2 x graph_trials visits of its own
            for k2, g2 in self._graph_sets.items():
                if g2 is not gs and g2.decided is not None and not g2.inherited and k2[1:] == key[1:]:
                    gs.decided, gs.inherited = g2.decided, True
                    gs.samples = {"eager": [], "graph": []}
                    if gs.decided == "eager":
                        gs.fwd = gs.bwd = gs.S = gs.out = gs.enc_len = gs.mel = gs.dout = None
                    break
        if auto and gs.decided is None and self._dp_world() > 1:
            # data-parallel runs keep live launches unless MI355X_GRAPHS=1 asks for the replay: the trial would time recorded
            # segments with live RCCL collectives between them -- a combination that has only ever run on gloo -- for a replay that
            # measured 3-4 % slower than live launches on one GPU (profiles/r3_host_issue.md); no trial also means every rank needs
            # the same number of warm-up steps by construction
            gs.decided = "eager"
        if auto and (gs.decided == "eager" or not self._in_step):
            self.live_steps += 1
            return None
        if gs.failed or gs.calls <= (self.graph_warmup_inherited if gs.inherited else self.graph_warmup):
            self.live_steps += 1
            return None
        if gs.fwd is None:
            try:
                self._capture_forward(gs, mel, length)
            except Exception as e:  # noqa: BLE001 -- a capture executes nothing, so the eager sequencer can simply take this call
                if os.environ.get("MI355X_GRAPHS") == "2":
                    raise
                import warnings
                warnings.warn(f"recording the encoder forward as hipGraphs failed ({type(e).__name__}: {e}); this shape keeps "
                              "running on the eager sequencer")
                gs.failed, gs.fwd, gs.S = True, None, None
                self.live_steps += 1
                return None
        elif auto and gs.decided is None and gs.bwd is not None:
            # the trial: live launches and replay ALTERNATE step by step, each timed on the device from here to the end of backward
            n_e, n_g = len(gs.samples["eager"]), len(gs.samples["graph"])
            if n_e >= self.graph_trials and n_g >= self.graph_trials:
                self._auto_decide(gs)
                if gs.decided == "eager":
                    self.live_steps += 1
                    return None
            elif n_e <= n_g:
                self._auto_begin(gs, "eager")
                self.live_steps += 1
                return None
            else:
                self._auto_begin(gs, "graph")
        gs.mel.copy_(mel)
        gs.length.copy_(length)
        gs.fwd.replay()
        self.replayed_steps += 1
        gs.gen += 1
        self._fwd_serial += 1       # (the recorded tensors live in the shared arena: ANY later forward overwrites them)
        gs.serial = self._fwd_serial
        # fresh tensor objects on the static storage (autograd attaches this call's node to what a Function returns)
        return gs.out.detach(), gs.enc_len.detach(), gs, gs.gen

    def _capture_forward(self, gs, mel, length):
        from ..graphs import SegmentedCapture
        dev = mel.device
        gs.mel = torch.empty(tuple(mel.shape), dtype=torch.float32, device=dev)
        gs.length = torch.empty(tuple(length.shape), dtype=torch.int64, device=dev)
        gs.mel.copy_(mel)
        gs.length.copy_(length)
        if self._step_word is None or self._step_word.device != dev:
            self._step_word = torch.zeros(1, dtype=torch.int32, device=dev)
        cap = SegmentedCapture(dev, tape=self.graph_tape)
        self._force_pack = True
        ops.set_step_counter(self._step_word)
        try:
            with cap.capturing(before_cut=self._wgrad_join):
                self._capture = cap
                self._step_word.add_(ops.STEP_WORD_INC)  # recorded: every replayed step draws fresh dropout masks
                gs.out, gs.enc_len, gs.S = self._forward_impl(gs.mel, gs.length, save=True)
        finally:
            self._capture = None
            self._force_pack = False
            self._wg_forked = False
            ops.set_step_counter(None)
        gs.fwd = cap
        gs.pool = cap.pool

    def _graphed_backward(self, gs, gen, dout):
        from ..graphs import SegmentedCapture
        if gen != gs.gen or getattr(gs, "serial", self._fwd_serial) != self._fwd_serial:
            raise RuntimeError("backward through an encoder forward whose saved activations were overwritten by a later forward "
                               "of the same shape (recorded launch sequences keep ONE set of activations per shape); set "
                               "encoder.use_graphs = False (or MI355X_GRAPHS=0) for several forwards per backward")
        if self.graph_bwd_live and gs.bwd is None:
            # live backward on the activations the replayed forward has just refilled (the recording's tensors keep their addresses);
            # the sequencer consumes its bag of saved tensors, so it gets a copy of the bag
            import copy
            S = copy.copy(gs.S)
            S.layers = list(gs.S.layers)
            S.serial = self._fwd_serial
            ops.set_step_counter(self._step_word)   # dropout keys: the device-side step word the recorded forward advanced
            try:
                self._backward_impl(S, dout)
            finally:
                ops.set_step_counter(None)
            return
        if gs.bwd is None:
            gs.dout = torch.empty_like(dout)
            gs.dout.copy_(dout)
            cap = SegmentedCapture(dout.device, pool=gs.pool, tape=self.graph_tape)
            ops.set_step_counter(self._step_word)
            try:
                gs.S.serial = self._fwd_serial  # (the replay that filled these activations is the latest forward)
                with cap.capturing(before_cut=self._wgrad_join):
                    self._capture = cap
                    self._backward_impl(gs.S, gs.dout)
            except Exception as e:  # noqa: BLE001
                self._capture = None
                ops.set_step_counter(None)
                raise RuntimeError("recording the encoder backward as hipGraphs failed after its forward had been recorded "
                                   f"({type(e).__name__}: {e}); run with MI355X_GRAPHS=0") from e
            finally:
                self._capture = None
                self._wg_forked = False
                ops.set_step_counter(None)
            gs.bwd = cap
            gs.S = None  # the recorded launches hold the addresses; the pool keeps the memory
        else:
            gs.dout.copy_(dout)
        # (the live sequencer joins the weight-gradient stream per layer only when nobody else orders behind it)
        gs.bwd.replay(join_between=self._wgrad_join_per_layer)

    def graphs_settled(self) -> bool:
        """auto mode: has every shape seen so far finished its eager-vs-replay trial?  (bench.py keeps running un-timed steps
        until this holds, so that neither a recording step nor a trial step falls into the timed region)"""
        if not (self.use_graphs and self.graph_auto):
            return True
        return all(gs.decided is not None or gs.failed for gs in self._graph_sets.values())

    def graphs_recorded(self) -> bool:
        """has every padded length seen so far reached its final way of running -- live launches by decision, or a recorded forward AND
        backward?  (bench.py --var-len pre-visits the padded lengths of its timed batches until this holds: the state a training run
        reaches after its first pass over the duration buckets)"""
        if not self.use_graphs:
            return True
        for gs in self._graph_sets.values():
            if gs.failed or gs.decided == "eager":
                continue
            if gs.bwd is None and not (self.graph_bwd_live and gs.fwd is not None):
                return False
            if self.graph_auto and gs.decided is None:
                return False
        return True

    def graph_info(self):
        """diagnostics (bench.py): recorded keys, graph segments and live host calls per forward / backward"""
        out = []
        for key, gs in self._graph_sets.items():
            if gs.fwd is not None:
                out.append({"mel_shape": list(key[0]), "fwd_graphs": gs.fwd.n_graphs(), "fwd_host_calls": len(gs.fwd.seq) - gs.fwd.n_graphs(),
                            "bwd_graphs": gs.bwd.n_graphs() if gs.bwd is not None else None,
                            "bwd_host_calls": (len(gs.bwd.seq) - gs.bwd.n_graphs()) if gs.bwd is not None else None,
                            "auto": getattr(gs, "auto_ms", None), "decided": gs.decided,
                            "replay": "launch tape" if gs.fwd.tape else "hipGraph",
                            "fwd_tape": gs.fwd.tape_info(), "bwd_tape": gs.bwd.tape_info() if gs.bwd is not None else None})
            elif gs.decided == "eager":
                out.append({"mel_shape": list(key[0]), "decided": "eager (live launches measured faster than the replay)",
                            "auto": getattr(gs, "auto_ms", None)})
        return out

    @staticmethod
    def _dgrad_slots(pt, pf, pad=1):
        """taps (kh, kw) through which a stride-2 3x3 conv reads an input position of parity (pt, pf): t1 = 2 t2 - pad + kh, so
        kh has the parity of t1 + pad (padding 1: an even position through k = 1 only, an odd one through 0 and 2; CausalConv2D's
        front padding of 2: the other way round)"""
        par = lambda q: [1] if (q + pad) % 2 else [0, 2]
        return [(kh, kw) for kh in par(pt) for kw in par(pf)]

    def _conv2_implicit(self, cdt, C_, M2):
        """implicit-GEMM conv2 (gathered A operand) needs the bf16 LDS-DMA GEMM structures and whole K-tiles per tap"""
        return cdt == torch.bfloat16 and C_ % 64 == 0 and C_ >= 192 and M2 >= 192  # (C is also the wgrad's M and N)

    @staticmethod
    def _splitk(tiles, K, strided_c=False):
        """split-K factor for `tiles` output tiles: the factor (<= 16, at least 16 K-tiles per workgroup) whose workgroup
        count fills whole rounds of the 256 CUs best, smaller factors preferred (every extra slice is one more atomic
        pass over the output)"""
        nk = (K + 63) // 64
        if strided_c:  # atomics into a column-strided C (reference weight layouts) are expensive: only fill the chip once
            return max(1, min(nk // 4 if nk >= 8 else 1, 256 // max(tiles, 1), 16))
        if tiles < 16:
            # skinny problems (e.g. the [C, C] pointwise-conv weight gradients of 'dw_striding' over 320 000 positions: 2 tiles):
            # 16 slices would leave them on 32 CUs (measured: 9.5 ms for 42 GFLOP); one round of the chip, >= 16 K-tiles each
            return max(1, min(256 // max(tiles, 1), nk // 16, 128))
        best, best_score = 1, -1.0
        for c in range(1, 17):
            if c > 1 and nk // c < 16:
                break
            blocks = tiles * c
            eff = blocks / (((blocks + 255) // 256) * 256)
            score = eff - 0.01 * c
            if score > best_score + 1e-9:
                best, best_score = c, score
        return best

    @staticmethod
    def _tiles(n_out, n_in, bf16):
        return ((n_out + 255) // 256) * ((n_in + 127) // 128) if bf16 else ((n_out + 63) // 64) * ((n_in + 63) // 64)

    def _wgrad(self, dY, ldy, y_off, X, ldx, x_off, dW, n_out, n_in, rows, bias_grad=None):
        """dW[n_out, n_in] += dY[:, y_off:y_off+n_out]^T @ X[:, x_off:x_off+n_in]   (TN GEMM, atomic split-K);
        bias_grad[n_out] += column sums of dY -- fused into the same kernel on the bf16 path."""
        bf16 = dY.dtype == torch.bfloat16
        if (self._wg_pending is not None and bf16 and n_out >= 192 and n_in >= 96 and ldy % 8 == 0 and ldx % 8 == 0
                and (self._wg_rows is None or self._wg_rows == rows)):
            # deferred: all weight gradients of the layer go out as ONE grouped launch (_wgrad_flush)
            self._wg_pending.append((dY, ldy, y_off, X, ldx, x_off, dW, n_out, n_in, bias_grad))
            self._wg_rows = rows
            if len(self._wg_pending) == 12:   # (one layer never collects that many; mi355x_gemm_grouped takes up to 40)
                self._wgrad_flush()
            return
        tiles = self._tiles(n_out, n_in, bf16)
        with self._wgrad_scope(dY, X):
            if bias_grad is not None and not bf16:
                ops.colsum(dY, bias_grad, rows, n_out, ld=ldy, x_off=y_off)
            ops.gemm(dY, X, dW, n_out, n_in, rows, ldy, ldx, n_in, transA=True, transB=True, atomic=True,
                     splitk=self._splitk(tiles, rows), a_off=y_off, b_off=x_off, c_dtype=ops.F32,
                     colsum_out=bias_grad if bf16 else None)

    # ---- weight gradients on a side stream.  Nothing downstream in backward depends on a weight gradient, and most wgrad
    # launches leave half the CUs idle (8-32 output tiles x split-K); on their own stream they run next to the HBM-bound
    # kernels of the main chain (LayerNorm backward, dropout casts, conv-module elementwise) instead of in line with them.
    def _wgrad_scope(self, *tensors):
        import contextlib
        if not self.wgrad_side_stream or not tensors[0].is_cuda:
            return contextlib.nullcontext()
        dev = tensors[0].device
        if self._wg_stream is None or self._wg_stream.device != dev:
            # MI355X_WGRAD_PRIO: -1 = high, 0 = default, 1 = low (HIP stream priorities): nothing on the backward chain waits
            # for a weight gradient, so the chain's kernels should win the CUs whenever both streams have workgroups pending
            prio = int(os.environ.get("MI355X_WGRAD_PRIO", "0"))
            from ..streams import private_stream  # (torch's pooled streams are shared beyond 32 creations per process)
            self._wg_stream = private_stream(dev, priority=prio)
        side = self._wg_stream
        if torch.cuda.current_stream(dev) == side:
            # nested scope (the padded-heads linear_pos weight gradients inside the posproj_side scope): already on the side lane.
            # A stream waiting for its own event is a no-op live, but inside a stream capture it makes the side stream its own
            # parallel capture stream and hip::Stream::EndCapture recurses until the stack ends (SIGSEGV at the Small geometry)
            return contextlib.nullcontext()
        side.wait_stream(torch.cuda.current_stream(dev))  # operands are produced on the main stream
        if self._capture is not None:
            self._wg_forked = True  # the side stream is part of the capture now: it must re-join before the segment ends
        for t in tensors:
            t.record_stream(side)  # the caching allocator must not hand the storage out again before the side stream is done
        return torch.cuda.stream(side)

    def _sub_wgrad_scope(self, *tensors):
        """the sub-sampling weight gradients also go to the side stream (they overlap conv1's HBM-bound backward)"""
        import contextlib
        return self._wgrad_scope(*tensors) if self.sub_wgrad_side_stream else contextlib.nullcontext()

    def _wgrad_flush(self):
        """launch the collected weight gradients of a layer as one grouped TN GEMM (side stream)"""
        pend = self._wg_pending
        if not pend:
            return
        rows = self._wg_rows
        tiles = sum(((q[7] + 255) // 256) * ((q[8] + 127) // 128) for q in pend)
        nk = (rows + 63) // 64
        env = os.environ.get("MI355X_WGRAD_SK")
        sk = int(env) if env else self._splitk(tiles, rows)
        tensors = [t for q in pend for t in (q[0], q[3])]
        with self._wgrad_scope(*tensors):
            ops.wgrad_grouped(pend, rows, sk)
        self._wg_pending, self._wg_rows = [], None

    def _defer_point(self, k):
        """see _backward_impl: point k of a layer's backward (1 = in front of the conv module's BatchNorm / depthwise / GLU
        backward, 2 = in front of the attention backward, 3 = in front of norm_conv's backward, 5 / 6 = inside the conv module: behind
        the BatchNorm reduction / behind the depthwise backward)"""
        f = getattr(self, "_defer_flush", None)
        if f is None:
            return
        if self.wgrad_defer == k or (self.wgrad_defer in (3, 4) and k == 1):
            f()
        elif (self.wgrad_defer == 3 and k == 2) or (self.wgrad_defer == 4 and k == 3):
            # split: what this layer has collected so far (feed-forward 2 + conv module) goes now, beside its own attention backward /
            # its own LayerNorm; the rest (attention + feed-forward 1) waits for the next layer's point 1
            self._wgrad_flush()

    def _wgrad_join(self, consume=False):
        """consume=True: the MAIN chain itself reads, right after this call, something the side stream produced"""
        if self._capture is not None:
            # recording: a capturing stream may only wait for work of its own capture.  The side stream is joined if (and only
            # if) this segment forked it; an un-forked side stream holds nothing THIS SEGMENT depends on.
            if not self._wg_forked:
                # ... but an EARLIER segment may have left work on it: a cut re-joins the side stream for the capture's sake
                # (before_cut) and clears the fork flag, while a launch tape replayed with join_between=False (the default
                # single-GPU path: optimizer-in-backward clears _wgrad_join_per_layer) leaves those segment-end joins out.
                # The consumer's ordering then has to be a LIVE step of the sequence: the tape's side lane IS _wg_stream.
                if consume and self._capture.tape and not self._wgrad_join_per_layer and self._wg_stream is not None:
                    self._capture.cut(self._live_wgrad_join)
                return
            self._wg_forked = False
        if self._wg_stream is not None and torch.cuda.current_stream(self._wg_stream.device) != self._wg_stream:
            torch.cuda.current_stream(self._wg_stream.device).wait_stream(self._wg_stream)   # (never a stream on itself: _wgrad_scope)

    def _live_wgrad_join(self):
        if self._wg_stream is not None and torch.cuda.current_stream(self._wg_stream.device) != self._wg_stream:
            torch.cuda.current_stream(self._wg_stream.device).wait_stream(self._wg_stream)

    def _packing_plan(self, length, B, T_mel, lens=None, peek=False):
        """None (the padded path) or the plan of a packed forward: pk.cu = i64 [B+1] row offsets on the device, pk.Mp = number of
        valid frames after sub-sampling.  The row count is a HOST number (it sizes every launch), so the lengths have to be known on
        the host: `length.host_lengths` (a CPU tensor / list the caller attached to the length tensor -- the input pipeline has them
        anyway) costs nothing; without it packed_rows=True reads them back (one device sync per step), "auto" stays padded.
        peek=True: decision only (no device work, no sync)."""
        mode = self.packed_rows
        if mode is False or not length.is_cuda or getattr(self, "_bypass_now", False):
            return None
        host = getattr(length, "host_lengths", None)
        if host is None:
            if mode == "auto":
                return None
            if peek:
                return _Saved()   # forced packing without host lengths: the decision is "packed", the numbers come with the sync
            host = length.detach().to("cpu")
        hl = torch.as_tensor(host, dtype=torch.int64).clamp(min=0, max=T_mel)
        T2 = T_mel
        sp = getattr(self.pre_encode, "_pad", 1)
        for _ in range(self.pre_encode._sampling_num):  # the same recurrence as _lens / calc_length (subsampling.py:576-586)
            hl = (torch.div(hl + (sp - 2), 2, rounding_mode="floor") + 1).clamp_(min=0)
            T2 = (T2 + sp - 2) // 2 + 1
        hl = hl.clamp(max=T2)
        Mp, M = int(hl.sum()), B * T2
        if Mp <= 0 or (mode == "auto" and (M - Mp) < self.packed_min_padding * M):
            return None
        pk = _Saved()
        pk.Mp = Mp
        if peek:
            return pk
        cu = torch.zeros(B + 1, dtype=torch.int64)
        cu[1:] = torch.cumsum(hl, 0)
        pk.cu = cu.to(length.device, non_blocking=True)
        pk.host_lens = hl
        return pk

    def _lens(self, length, n_stages=2):
        """valid lengths after 0, 1, ..., n stride-2 stages: floor((n + all_paddings - 3)/2) + 1 each, all_paddings = 2 (or 3 with
        causal_downsampling), subsampling.py:576-586"""
        sp = getattr(self.pre_encode, "_pad", 1)
        out = [length.to(torch.int64).contiguous()]
        for _ in range(n_stages):
            out.append((torch.div(out[-1] + (sp - 2), 2, rounding_mode="floor") + 1).clamp_(min=0).contiguous())
        return out

    # ------------------------------------------------------------------ forward implementation
    def _forward_impl(self, mel, length, save=False):
        self._saving = bool(save)
        dev = mel.device
        self._phase("f", dev)
        self._fwd_serial += 1
        cdt = self._cdt()
        if cdt == torch.bfloat16 and self.d_model % 8:
            # the bf16 operand path moves 16-byte pieces.  Heads that do not start on an 8-element boundary (the recipe table's
            # Small: d = 176, 4 heads, d_k = 44) are zero-padded inside the packed weight images (_geometry); a d_model that is
            # not a multiple of 8 would need padded activation rows as well (SqueezeformerEncoder has them)
            raise NotImplementedError(f"bf16 compute needs d_model divisible by 8 (got d_model={self.d_model}); use "
                                      f"compute_dtype=torch.float32 for this geometry")
        training = self.training
        W, Wf = self._plan(cdt, dev)
        bypass = bool(getattr(self, "_bypass_now", False))  # bypass_pre_encode: `mel` holds pre-encoded frames [B, T', d_model]
        mel = mel.to(torch.float32).contiguous()
        d, H, dk, dff, C_ = self.d_model, self.n_heads, self.d_k, self.d_ff, self.pre_encode._conv_channels
        if bypass:
            B, T, _ = mel.shape
            F_ = T1 = F1 = F2 = 0
            T2 = T
            ln_ = length.to(torch.int64).contiguous()
            lens = [ln_] * (self.pre_encode._sampling_num + 1)
        else:
            B, F_, T = mel.shape
            sp = self.pre_encode._pad
            T1, F1 = ops.half_len(T, sp), ops.half_len(F_, sp)
            lens = self._lens(length, self.pre_encode._sampling_num)  # [len0, len1, ..., len_final]
            T2, F2 = T, F_
            for _ in range(self.pre_encode._sampling_num):
                T2, F2 = ops.half_len(T2, sp), ops.half_len(F2, sp)
        len0, len1, len2 = lens[0], lens[1], lens[-1]                # (`len2` / `T2` / `F2` name the FINAL grid everywhere below)
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
        S.mel, S.len0, S.len2 = mel, len0, len2
        pe = self.pre_encode
        S.lens_all = lens
        S.drop_pre = drop(self.dropout_pre_encoder, 100000)
        S.bypass = bypass
        if bypass:
            # RelPositionalEncoding's part of the front (multi_head_attention.py:1087-1098): x * xscale, dropout
            x = self._new(M, d, dtype=torch.float32, device=dev)
            ops.drop_scale_cast(mel.view(M, d), x, M * d, (self.xscale or 1.0), S.drop_pre)
        elif self.subsampling == "dw_striding":
            x = self._sub_fwd_dw(S, mel, lens, W, cdt, save)
        else:
            # ---- sub-sampling: conv1 (direct) -> conv2 (implicit MFMA GEMM, ReLU+mask epilogue) -> out Linear (+xscale, dropout)
            S.out1 = self._new(B, T1, F1, C_, dtype=cdt, device=dev)
            ops.conv1_fwd(mel, pe.conv[0].weight, pe.conv[0].bias, S.out1, len0, len1, C_, pad=pe._pad)
            # (channel counts the gather does not cover fall back to an im2col image, kept alive for the weight gradient)
            implicit = (self.conv2_implicit and self._conv2_implicit(cdt, C_, B * T2 * F2)
                        and B * T1 * F1 * C_ < 2 ** 31)  # the gathered weight gradient addresses the grid with 32-bit offsets
            S.col = None
            if not implicit:
                col = self._buf("col", (B * T2 * F2, 9 * C_), cdt, dev)
                ops.im2col(S.out1, col, B, T1, F1, C_, pad=pe._pad)
                self._col_gen = getattr(self, "_col_gen", 0) + 1
                S.col, S.col_gen = (col if save else None), self._col_gen
            S.out2 = self._new(B * T2 * F2, C_, dtype=cdt, device=dev)
            if implicit:
                # implicit GEMM: the A rows are gathered from out1 by the LDS-DMA (tap (kh-1, kw-1) per 512-wide K block);
                # forward, weight gradient and input gradient all gather -- no im2col image, no col2im pass
                ops.gemm(S.out1, W["pre.w2"], S.out2, B * T2 * F2, C_, 9 * C_, C_, W.pitch("pre.w2"), C_, bias=pe.conv[2].bias,
                         epi=ops.EPI_RELU_MASK, row_len=len2, rows_per_b=T2 * F2, rows_inner=F2,
                         gather=dict(nI=T2, nJ=F2, SI=T1, SJ=F1, C=C_, si=2, sj=2,
                                     taps=[(kh - pe._pad, kw - pe._pad) for kh in range(3) for kw in range(3)]))
            else:
                ops.gemm(col, W["pre.w2"], S.out2, B * T2 * F2, C_, 9 * C_, 9 * C_, W.pitch("pre.w2"), C_, bias=pe.conv[2].bias,
                         epi=ops.EPI_RELU_MASK, row_len=len2, rows_per_b=T2 * F2, rows_inner=F2)
            x = self._new(M, d, dtype=torch.float32, device=dev)
            ops.gemm(S.out2, W["pre.out"], x, M, d, F2 * C_, F2 * C_, W.pitch("pre.out"), d, bias=pe.out.bias,
                     alpha=(self.xscale or 1.0), drop=S.drop_pre)
        # ---- packed rows: from here to the end of the layer stack only the valid frames exist
        S.pk = pk = self._packing_plan(length, B, T, lens=lens)
        self.packed_last = (pk.Mp, M) if pk is not None else None
        if pk is not None:
            xp = self._new(pk.Mp, d, dtype=torch.float32, device=dev)
            ops.rows_pack(x, xp, len2, pk.cu, T2, M, d)
            x = xp
        # ---- relative positional table (constant)
        pkey = (T2, cdt, str(dev))
        pos = self._pos_cache.get(pkey)
        if pos is None:
            pos = self.pos_enc.table(T2, dev, cdt)
            if len(self._pos_cache) >= 16:  # (a duration-bucketed loader alternates between a handful of lengths)
                self._pos_cache = {}
            self._pos_cache[pkey] = pos
        d_emb = drop(self.dropout_emb, 100001)
        if d_emb.threshold:  # dropout on the positional table (multi_head_attention.py:1097-1098)
            posd = torch.empty_like(pos)
            ops.drop_scale_cast(pos, posd, pos.numel(), 1.0, d_emb)
            pos = posd
        S.pos = pos
        S.p_all = self._pos_proj_fwd(pos, W, cdt, dev)
        # per-layer f64 BatchNorm sums: one allocation + one fill for all layers (18 tiny fill launches otherwise)
        # (row layout [sum | sum of squares | element count | pad]: under SyncBatchNorm the count travels with the sums)
        S.bn_stats = torch.zeros(self.n_layers, 2 * d + 8, dtype=torch.float64, device=dev) if training else None
        S.bn_world = self._syncbn_world() if training else 1
        if training and S.bn_world > 1:
            S.bn_stats[:, 2 * d] = float(M)
        S.layers = []
        S.sd, S.cap_idx, S.cap_out, S.proj_in = [], [], [], {}

        def on_grid(t):  # a layer output on the reference's [B, T', d] grid (packed rows: frames beyond an utterance are zeros)
            if pk is None:
                return t
            xo = self._new(M, d, dtype=torch.float32, device=dev)
            ops.rows_unpack(t, xo, len2, pk.cu, T2, M, d)
            return xo

        def projected(t, key):  # [M, d] -> [B, D, T'] through out_proj (D = feat_out) or as it is
            if self.out_proj is None:
                return t.view(B, T2, d).transpose(1, 2)
            S.proj_in[key] = t
            return self._out_proj_fwd(t, M, dev).view(B, T2, self._feat_out).transpose(1, 2)

        for i, L in enumerate(self.layers):
            x_in = x
            x, sl = self._layer_fwd(i, L, x, S, W, Wf, drop)
            S.layers.append(sl)
            sd = None
            if training and self.layer_drop_probs[i] > 0.0:
                # stochastic depth (conformer_encoder.py:696-707): one torch.rand(1) per droppable layer from the global generator, at
                # the reference's point in the sequence.  A dropped layer has run (BatchNorm statistics, dropout counters move on)
                # and contributes nothing; a kept one is rescaled: x_in + (x - x_in) / (1 - p)
                p_ = self.layer_drop_probs[i]
                if bool(torch.rand(1) < p_):
                    sd, x = ("drop", 0.0), x_in
                else:
                    a_ = 1.0 / (1.0 - p_)
                    sd, x = ("keep", a_), torch.add(x_in, x - x_in, alpha=a_)
            S.sd.append(sd)
            if i in self.capture_layers:
                S.cap_idx.append(i)
                S.cap_out.append(projected(on_grid(x), i))
        if training:  # nn.BatchNorm1d bookkeeping, one multi-tensor launch
            if self.conv_norm_type == "batch_norm":
                torch._foreach_add_([L.conv.batch_norm.num_batches_tracked for L in self.layers], 1)
        x = on_grid(x)  # back to the reference's [B, T', d] grid (frames beyond an utterance: zeros -- nothing downstream reads them)
        out = projected(x, "final")
        order = {l: k for k, l in enumerate(S.cap_idx)}   # captures in the order the caller listed the layers
        S.cap_out = [S.cap_out[order[l]] for l in self.capture_layers if l in order]
        S.cap_layers = [l for l in self.capture_layers if l in order]
        return out, len2, (S if save else None)

    # ------------------------------------------------------------------ 'dw_striding' sub-sampling (FastConformer, Squeezeformer)
    def _sub_io(self, Wf, cdt, dev, backward=False):
        """what the 'dw_striding' stack reads and where its gradients go: here the parameters and their `.grad` views.
        (SqueezeformerEncoder substitutes zero-padded images / scratch gradients when the channel count is not a multiple of 8.)"""
        pe = self.pre_encode
        io = _Saved()
        io.C = pe._conv_channels
        io.c0w, io.c0b = pe.conv[0].weight, pe.conv[0].bias
        io.dw = [(dw.weight, dw.bias, pw.bias) for dw, pw in pe.dw_stages()]
        if backward:
            io.g_c0w, io.g_c0b = pe.conv[0].weight.grad, pe.conv[0].bias.grad
            io.g_dw = [(dw.weight.grad, dw.bias.grad, pw.weight.grad, pw.bias.grad) for dw, pw in pe.dw_stages()]
            io.g_out = pe.out.weight.grad
        io.finish = None
        return io

    def _sub_fwd_dw(self, S, mel, lens, W, cdt, save, Wf=None):
        """conv(1->C, 3x3, s2) ReLU -> [depthwise 3x3 s2 -> pointwise 1x1 -> ReLU] x (log2(factor) - 1) -> Linear, every layer
        on a time-masked input (subsampling.py:142-215, 385-436, 725-759).  conv1: the direct kernel of the 'striding' path;
        depthwise: mi355x_dwconv2d_s2_*; pointwise: MFMA GEMM with the ReLU + time-mask epilogue; channels-last throughout."""
        B, F_, T, T1, F1, T2, F2, M, _, training, seed = S.dims
        dev = mel.device
        pe = self.pre_encode
        io = self._sub_io(Wf, cdt, dev)
        C_, d = io.C, self.d_model
        out0 = self._new(B, T1, F1, C_, dtype=cdt, device=dev)
        ops.conv1_fwd(mel, io.c0w, io.c0b, out0, lens[0], lens[1], C_, pad=pe._pad)
        cur, Tc, Fc = out0, T1, F1
        S.dw = []
        for si_, (dww, dwb, pwb) in enumerate(io.dw):
            Tn, Fn = ops.half_len(Tc, pe._pad), ops.half_len(Fc, pe._pad)
            dwo = self._new(B * Tn * Fn, C_, dtype=cdt, device=dev)
            ops.dwconv2d_s2_fwd(cur, dww, dwb, dwo, B, Tc, Fc, C_, pad=pe._pad)
            pwo = self._new(B * Tn * Fn, C_, dtype=cdt, device=dev)
            ops.gemm(dwo, W[f"pre.pw{si_}"], pwo, B * Tn * Fn, C_, C_, C_, W.pitch(f"pre.pw{si_}"), C_, bias=pwb,
                     epi=ops.EPI_RELU_MASK, row_len=lens[si_ + 2], rows_per_b=Tn * Fn, rows_inner=Fn)
            S.dw.append((cur, Tc, Fc, dwo, pwo, Tn, Fn))
            cur, Tc, Fc = pwo, Tn, Fn
        x = self._new(M, d, dtype=torch.float32, device=dev)
        ops.gemm(cur, W["pre.out"], x, M, d, F2 * C_, F2 * C_, W.pitch("pre.out"), d, bias=pe.out.bias,
                 alpha=(self.xscale or 1.0), drop=S.drop_pre)
        S.out1, S.out2, S.col = out0, cur, None
        return x

    def _sub_bwd_dw(self, S, dx, W, cdt, Wf=None):
        B, F_, T, T1, F1, T2, F2, M, _, training, seed = S.dims
        dev = dx.device
        pe = self.pre_encode
        io = self._sub_io(Wf, cdt, dev, backward=True)
        C_, d = io.C, self.d_model
        bf16 = cdt == torch.bfloat16
        ldx = _pad8(d) if bf16 else d  # (bf16 operand rows start on 16-byte boundaries)
        dxs = self._new(M, ldx, dtype=cdt, device=dev)
        if ldx != d or (M * d) % 8:
            ops.cast_pitched(dx, dxs, M, d, ldx, (self.xscale or 1.0), S.drop_pre)
        else:
            ops.drop_scale_cast(dx, dxs, M * d, (self.xscale or 1.0), S.drop_pre)
        with self._sub_wgrad_scope(dxs):
            ops.colsum(dxs, pe.out.bias.grad, M, d, ld=ldx)
        last = S.out2  # [B*T2*F2, C] = [M, F2*C]
        tiles = self._tiles(d, C_, bf16) * F2
        with self._sub_wgrad_scope(dxs, last):  # d out.weight in the reference's (c, f) column order (batch over f)
            ops.gemm(dxs, last, io.g_out, d, C_, M, ldx, F2 * C_, C_ * F2, transA=True, transB=True, atomic=True,
                     splitk=self._splitk(tiles, M, strided_c=True), batch=F2, nb0=F2, sB=(C_, 0), sC=(1, 0), c_col_stride=F2,
                     c_dtype=ops.F32)
        dcur = self._new(B * T2 * F2, C_, dtype=cdt, device=dev)
        ops.gemm(dxs, W["pre.outt"], dcur, M, F2 * C_, d, ldx, W.pitch("pre.outt"), F2 * C_, epi=ops.EPI_MUL_POS, aux_in=last)
        for si_ in range(len(S.dw) - 1, -1, -1):
            cur_in, Tc, Fc, dwo, pwo, Tn, Fn = S.dw[si_]
            dww, dwb, pwb = io.dw[si_]
            g_dww, g_dwb, g_pww, g_pwb = io.g_dw[si_]
            Ms = B * Tn * Fn
            # pointwise conv: bias / weight gradients, then the gradient w.r.t. the depthwise output (no gate: no ReLU there)
            if bf16 and C_ >= 192:
                with self._sub_wgrad_scope(dcur, dwo):
                    ops.gemm(dcur, dwo, g_pww, C_, C_, Ms, C_, C_, C_, transA=True, transB=True, atomic=True,
                             splitk=self._splitk(self._tiles(C_, C_, True), Ms), c_dtype=ops.F32, colsum_out=g_pwb)
            else:
                with self._sub_wgrad_scope(dcur, dwo):
                    ops.colsum(dcur, g_pwb, Ms, C_)
                    ops.gemm(dcur, dwo, g_pww, C_, C_, Ms, C_, C_, C_, transA=True, transB=True, atomic=True,
                             splitk=self._splitk(self._tiles(C_, C_, bf16), Ms), c_dtype=ops.F32)
            ddw = self._new(Ms, C_, dtype=cdt, device=dev)
            ops.gemm(dcur, W[f"pre.pw{si_}t"], ddw, Ms, C_, C_, C_, W.pitch(f"pre.pw{si_}t"), C_)
            din = self._new(B * Tc * Fc, C_, dtype=cdt, device=dev)
            ops.dwconv2d_s2_bwd(ddw, cur_in, dww, din, g_dww, g_dwb, B, Tc, Fc, C_, pad=pe._pad)
            dcur = din
        ops.conv1_bwd(dcur, S.mel, S.len0, io.g_c0w, io.g_c0b, C_, pad=pe._pad)
        self._wgrad_join(consume=io.finish is not None)
        if io.finish is not None:
            io.finish()
        if self.grad_ready_hook is not None:
            self._hook(*self._flatp.range_of("pre_encode."))

    def _ln_fwd(self, ln, x, M, d, out_dtype, dev):
        y = self._new(M, d, dtype=out_dtype, device=dev)
        mean = self._new(M, dtype=torch.float32, device=dev)
        rstd = self._new(M, dtype=torch.float32, device=dev)
        ops.layernorm_fwd(x, ln.weight, ln.bias, y, mean, rstd, M, d, ln.eps)
        return y, mean, rstd

    def _ffn_fused_ok(self, cdt):
        """the one-launch feed-forward kernels cover bf16, d_model = 512, 128 <= d_ff <= 2048, d_ff % 64 == 0 (Conformer-CTC-Large,
        FastConformer-Large); MI355X_FFN_FUSED=0 keeps the GEMM pair (A/B switch)"""
        return (self.ffn_fused and cdt == torch.bfloat16 and self.d_model == 512 and self.d_ff % 64 == 0
                and 128 <= self.d_ff <= 2048)

    def _ffn_fwd(self, pfx, ff, x, ln, S, sl, W, drop, site, M, d, dff, cdt, dev, tag):
        pre = getattr(S, "pre_ln", None)
        if pre is not None and tag == "ff1":  # the previous layer's output norm already normalised this layer's input
            y, mean, rstd = pre
            S.pre_ln = None
        else:
            y, mean, rstd = self._ln_fwd(ln, x, M, d, cdt, dev)
        h = self._new(M, dff, dtype=cdt, device=dev)
        d_in = drop(self.dropout, site)
        d_res = drop(self.dropout, site + 1)
        if self._ffn_fused_ok(cdt):
            # conformer_modules.py:366-387 + the macaron residual (:174-181, :209-215) in ONE launch; the activated hidden never
            # reaches memory (backward recomputes it from h for the linear2 weight gradient)
            r = self._new(M, d, dtype=torch.float32, device=dev)
            ops.ffn_fwd(y, W[pfx + ".w1p"], ff.linear1.bias, W[pfx + ".w2p"], ff.linear2.bias, x, h, r, M, d, dff, 0.5, d_in, d_res)
            setattr(sl, tag, (x, y, mean, rstd, h, None, d_in, d_res))
            return r
        a = self._new(M, dff, dtype=cdt, device=dev)
        g_form = self.swish_g and cdt == torch.bfloat16  # `h` then holds swish'(h) * mask (see __init__)
        ops.gemm(y, W[pfx + ".w1"], a, M, dff, d, d, W.pitch(pfx + ".w1"), dff, bias=ff.linear1.bias,
                 epi=ops.EPI_SWISH_DROP_G if g_form else ops.EPI_SWISH_DROP, aux_out=h, drop=d_in)
        r = self._new(M, d, dtype=torch.float32, device=dev)
        ops.gemm(a, W[pfx + ".w2"], r, M, d, dff, dff, W.pitch(pfx + ".w2"), d, bias=ff.linear2.bias, alpha=0.5,
                 epi=ops.EPI_RESID, aux_in=x, drop=d_res)
        setattr(sl, tag, (x, y, mean, rstd, h, a, None if g_form else d_in, d_res))
        return r

    def _geometry(self, cdt):
        """(row pitch of [M, d] GEMM operands, head width in the attention operands, attention width H * head width).  bf16
        operands move in 16-byte pieces, so a head must start on an 8-element boundary: with d_k % 8 != 0 (Conformer-Small:
        d = 176, 4 heads, d_k = 44 -> 48) the heads are zero-padded INSIDE the packed weight images (q | k | v rows,
        linear_pos rows, linear_out columns, pos_bias lanes); activations outside the attention block keep width d."""
        dk = self.d_k
        dkp = _pad8(dk) if cdt == torch.bfloat16 else dk
        if cdt == torch.bfloat16 and self._flash_ok() and self.flash_pad_heads and 64 < dkp < 128:
            dkp = 128   # ... and heads of 65..127 lanes to the kernels' second width (d_k' = 128: 8 k-steps, one workgroup per CU)
        if cdt == torch.bfloat16 and self._flash_ok() and self.flash_pad_heads and dkp < 64:
            # round 5: heads narrower than the fused kernels' width (Conformer-Small: 44) are padded up to 64 instead of to the
            # next multiple of 8, so that they take the fused rel-pos attention (csrc/attention.hip, d_k' = 64) instead of
            # materialising the [H, B, T', T'] scores and the [H, B, T', 2T'-1] positional matrix in HBM
            # (multi_head_attention.py:259-354).  The zero lanes add nothing to q.k, (q+v).p or P.V; 1/sqrt(d_k) stays the true one.
            dkp = 64
        return self.d_model, dkp, self.n_heads * dkp

    def _heads_wgrad(self, dY, ldy, y_off, X, ldx, dW, rows, n_groups, group_stride_y, group_stride_w):
        """dW_g[h*dk:(h+1)*dk, :] += dY[:, y_off + g*group_stride_y + h*dkp : +dk]^T @ X for every head h and group g (q, k, v):
        one batched TN GEMM whose batch strides step over the heads' pad lanes"""
        d, H, dk = self.d_model, self.n_heads, self.d_k
        dkp = self._geometry(dY.dtype)[1]
        bf16 = dY.dtype == torch.bfloat16
        tiles = self._tiles(dk, d, bf16) * H * n_groups
        with self._wgrad_scope(dY, X):
            ops.gemm(dY, X, dW, dk, d, rows, ldy, ldx, d, transA=True, transB=True, atomic=True,
                     splitk=self._splitk(tiles, rows), batch=H * n_groups, nb0=H, sA=(dkp, group_stride_y),
                     sC=(dk * d, group_stride_w), a_off=y_off, c_dtype=ops.F32)

    def _unpad_add(self, dst, src, dkp):
        """dst [.., H*dk] += src [.., H*dkp] without the pad lanes (tiny: bias-sized vectors)"""
        H, dk = self.n_heads, self.d_k
        dst.view(*dst.shape[:-1], H, dk).add_(src.view(*src.shape[:-1], H, dkp)[..., :dk])

    def _pos_proj_fwd(self, pos, W, cdt, dev):
        """p_l = linear_pos_l(pos_emb) for all layers (multi_head_attention.py:309): the input is the same table, so the
        18 [2T-1, d] x [d, d] products are one batched GEMM (288 tiles) instead of 18 launches of 16 tiles."""
        nl, d = self.n_layers, self.d_model
        dA = self._geometry(cdt)[2]
        P = pos.shape[0]
        p_all = self._new(nl, P, dA, dtype=cdt, device=dev)
        es = W["L0.att.wpos"].element_size()
        stride = (W["L1.att.wpos"].data_ptr() - W["L0.att.wpos"].data_ptr()) // es if nl > 1 else 0
        uniform = all(W[f"L{i}.att.wpos"].data_ptr() - W["L0.att.wpos"].data_ptr() == i * stride * es for i in range(nl))
        if uniform and stride >= 0:
            ops.gemm(pos, W["L0.att.wpos"], p_all, P, dA, d, d, W.pitch("L0.att.wpos"), dA, batch=nl, nb0=nl,
                     sB=(stride, 0), sC=(P * dA, 0))
        else:
            for i in range(nl):
                ops.gemm(pos, W[f"L{i}.att.wpos"], p_all[i], P, dA, d, d, W.pitch(f"L{i}.att.wpos"), dA)
        return p_all

    def _pos_proj_wgrad(self, dp_all, pos, P, cdt):
        """d linear_pos_l.weight += dp_l^T @ pos_emb for all layers: one batched TN GEMM into the tail of the flat
        gradient buffer (FlatParams(tail=...) keeps the 18 gradients equally spaced)."""
        nl, d = self.n_layers, self.d_model
        grads = [L.self_attn.linear_pos.weight.grad for L in self.layers]
        _, dkp, dA = self._geometry(cdt)
        if dkp != self.d_k:  # padded heads: per layer one TN GEMM batched over the heads (strides step over the pad lanes)
            for i in range(nl):
                self._heads_wgrad(dp_all[i], dA, 0, pos, d, grads[i], P, 1, 0, 0)
            return
        stride = (grads[1].data_ptr() - grads[0].data_ptr()) // 4 if nl > 1 else 0
        uniform = all(g.data_ptr() - grads[0].data_ptr() == i * stride * 4 for i, g in enumerate(grads))
        if uniform and stride >= 0:
            ops.gemm(dp_all, pos, grads[0], d, d, P, d, d, d, transA=True, transB=True, atomic=True, batch=nl, nb0=nl,
                     sA=(P * d, 0), sC=(stride, 0), c_dtype=ops.F32)
        else:
            for i in range(nl):
                self._wgrad(dp_all[i], d, 0, pos, d, 0, grads[i], d, d, P)

    # ------------------------------------------------------------------ rel-pos attention core (shared with Squeezeformer)
    def _attn_fwd(self, qkv, p, bias_u, bias_v, lens, B, T, dA, dk, scale, d_att, cdt, dev, pk=None):
        """qkv [B*T, 3*dA] (q | k | v, heads of width dk = dA / H, possibly zero-padded heads), p [2T-1, dA] = linear_pos of
        the table, bias_u / bias_v [dA] -> ctx [B*T, dA] and what backward needs.  `scale` = 1/sqrt(true d_k).
        pk (packed rows): qkv / ctx hold the valid frames only ([pk.Mp, .]); the fused kernels address utterance b at pk.cu[b],
        the un-fused path (fp32, other head widths) runs on a padded copy."""
        H = self.n_heads
        M, P = B * T, 2 * T - 1
        Tp, Pp = _pad8(T), _pad8(P)
        flash = self._flash_ok() and cdt == torch.bfloat16 and dk in (64, 128)
        cu = pk.cu if pk is not None else None
        if pk is not None and not flash:
            qkv_p = self._new(M, 3 * dA, dtype=cdt, device=dev)
            ops.rows_unpack(qkv, qkv_p, lens, cu, T, M, 3 * dA)
            ctx_p, saved = self._attn_fwd(qkv_p, p, bias_u, bias_v, lens, B, T, dA, dk, scale, d_att, cdt, dev)
            ctx = self._new(pk.Mp, dA, dtype=cdt, device=dev)
            ops.rows_pack(ctx_p, ctx, lens, cu, T, M, dA)
            return ctx, saved
        ctx = self._new(pk.Mp if pk is not None else M, dA, dtype=cdt, device=dev)
        if flash:
            # fused rel-pos flash attention: scores / positional matrix never touch HBM; only the log-sum-exp is kept
            lse = self._new(B, H, T, dtype=torch.float32, device=dev)
            # training: also the bf16 rounding residual of the context (backward's delta = sum dO * O needs more than the 8
            # mantissa bits of the stored operand -- see mi355x_relpos_flash_fwd); it travels in the first slot of the saved tuple
            # (kept whenever activations are saved for a backward -- also for an eval-mode / frozen encoder that is
            #  differentiated through: delta from the rounded O alone put a 36 % error on layer-0 q / k gradients)
            ctx_lo = self._new(ctx.shape[0], dA, dtype=cdt, device=dev) if (self._saving and self.flash_delta_residual) else None
            ops.relpos_flash_fwd(qkv, 3 * dA, p, dA, bias_u, bias_v, lens, ctx, dA, lse, B, H, T, dk, Tp, scale, d_att,
                                 ctx_lo=ctx_lo, cu=cu)
            return ctx, (ctx_lo, None, None, None, lse)
        qu = self._new(M, dA, dtype=cdt, device=dev)
        qv = self._new(M, dA, dtype=cdt, device=dev)
        ops.qbias(qkv, 3 * dA, bias_u, bias_v, qu, qv, M, dA)
        ac = self._buf("ac", (H, B, T, Tp), torch.float32, dev)
        bdf = self._buf("bdf", (H, B, T, Pp), torch.float32, dev)
        # ac[h,b] = qu_bh @ k_bh^T ; bdf[h,b] = qv_bh @ p_h^T      (z0 = b, z1 = h)
        ops.gemm(qu, qkv, ac, T, T, dk, dA, 3 * dA, Tp, batch=H * B, nb0=B, sA=(T * dA, dk), sB=(T * 3 * dA, dk),
                 sC=(T * Tp, B * T * Tp), b_off=dA)
        ops.gemm(qv, p, bdf, T, P, dk, dA, dA, Pp, batch=H * B, nb0=B, sA=(T * dA, dk), sB=(0, dk), sC=(T * Pp, B * T * Pp))
        s_ = self._new(H, B, T, Tp, dtype=cdt, device=dev)
        pd = self._new(H, B, T, Tp, dtype=cdt, device=dev) if d_att.threshold else None
        ops.relpos_softmax_fwd(ac, bdf, s_, pd, lens, H, B, T, Tp, Pp, scale, d_att, ctx=self._ctx)
        if pd is None:
            pd = s_
        # ctx_bh = pd_bh [T,T] @ v_bh [T,dk]   (NN: v is reduction-major inside qkv)
        ops.gemm(pd, qkv, ctx, T, dk, T, Tp, 3 * dA, dA, transB=True, batch=H * B, nb0=B, sA=(T * Tp, B * T * Tp),
                 sB=(T * 3 * dA, dk), sC=(T * dA, dk), b_off=2 * dA)
        return ctx, (qu, qv, s_, pd, None)

    def _attn_bwd(self, saved, qkv, p, bias_u, bias_v, ctx, dctx, lens, B, T, dA, dk, scale, d_att, cdt, dev, dp, dp_cast,
                  bias_grads=None, pk=None):
        """-> (dqkv [M, 3*dA] with the k and v thirds filled, dqu, dqv [M, dA]); dp f32 [2T-1, dA] += d linear_pos output,
        dp_cast (compute dtype) = its GEMM-operand copy for the linear_pos weight gradient.  `bias_grads` (fused path only):
        f32 [2 * dA] = pos_bias_u.grad | pos_bias_v.grad in one piece -- then the dQ kernel writes dq = dqu + dqv into the q third
        of dqkv and the bias gradients itself, and (dqkv, None, None) comes back."""
        qu, qv, s_, pd, lse = saved
        H = self.n_heads
        M, P = B * T, 2 * T - 1
        Tp, Pp = _pad8(T), _pad8(P)
        cu = pk.cu if pk is not None else None
        if pk is not None and lse is None:
            # packed rows, un-fused path (fp32 / other head widths): the batched GEMMs stride over a padded grid -- run them on padded
            # copies and hand the valid rows back
            qkv_p = self._new(M, 3 * dA, dtype=cdt, device=dev)
            dctx_p = self._new(M, dA, dtype=cdt, device=dev)
            ops.rows_unpack(qkv, qkv_p, lens, cu, T, M, 3 * dA)
            ops.rows_unpack(dctx, dctx_p, lens, cu, T, M, dA)
            dqkv_p, dqu_p, dqv_p = self._attn_bwd(saved, qkv_p, p, bias_u, bias_v, None, dctx_p, lens, B, T, dA, dk, scale, d_att, cdt,
                                                  dev, dp, dp_cast)
            out = []
            for t_p, w in ((dqkv_p, 3 * dA), (dqu_p, dA), (dqv_p, dA)):
                t = self._new(pk.Mp, w, dtype=cdt, device=dev)
                ops.rows_pack(t_p, t, lens, cu, T, M, w)
                out.append(t)
            return tuple(out)
        if pk is not None:
            M = pk.Mp   # (fused path: the kernels address utterance b at cu[b]; lse / delta / dS keep their padded [B, H, T] layout)
        dqkv = self._new(M, 3 * dA, dtype=cdt, device=dev)
        fuse_dq = lse is not None and bias_grads is not None
        dqu = None if fuse_dq else self._new(M, dA, dtype=cdt, device=dev)
        dqv = None if fuse_dq else self._new(M, dA, dtype=cdt, device=dev)
        if lse is not None:
            ctx_lo = qu  # (fused path: the first slot carries the context's rounding residual, q + u is recomputed here)
            qu = self._new(M, dA, dtype=cdt, device=dev)
            qv = self._new(M, dA, dtype=cdt, device=dev)
            dlt = self._new(B, H, T, dtype=torch.float32, device=dev)
            if (bias_u.data_ptr() | bias_v.data_ptr()) & 15:
                ops.qbias(qkv, 3 * dA, bias_u, bias_v, qu, qv, M, dA)
                ops.attn_delta(dctx, ctx, dlt, B, H, T, dA, O_lo=ctx_lo, lens=lens, cu=cu)
            else:  # q + u, q + v and delta = sum dO * O in one pass over the rows
                ops.attn_bwd_prep(dctx, ctx, dlt, qkv, 3 * dA, bias_u, bias_v, qu, qv, B, H, T, dA, O_lo=ctx_lo, lens=lens, cu=cu)
            # transient dS (un-shifted 32 x 32 blocks): dQ kernel -> linear_pos gradient kernel.  The latter feeds only the
            # (batched, end-of-backward) linear_pos weight gradient, so with the side stream it leaves the critical path; dS
            # then comes from the caching allocator (record_stream keeps the next layer's dQ kernel from overwriting it too early).
            side_pos = self.dpos_side_stream and self.wgrad_side_stream
            n_ds = ops.lib.mi355x_relpos_ds_elems(B, H, T)
            dS = (self._new(n_ds, dtype=cdt, device=dev) if side_pos else self._buf("dS", (n_ds,), cdt, dev))
            ops.relpos_flash_bwd_dq(qu, qv, qkv, 3 * dA, p, dA, lens, dctx, lse, dlt, dqu, dqv, B, H, T, dk, scale, d_att,
                                    ds_out=dS, dq_out=dqkv if fuse_dq else None, ld_dq=3 * dA, bias_grads=bias_grads, cu=cu)
            if side_pos:
                with self._wgrad_scope(qv, dS):
                    ops.relpos_flash_bwd_dpos(qv, dS, lens, dp, B, H, T, dk, dpos_cast=dp_cast, cu=cu)
            ops.relpos_flash_bwd_dkv(qu, qv, qkv, 3 * dA, p, dA, lens, dctx, lse, dlt, dqkv, 3 * dA, B, H, T, dk, Tp, scale, d_att,
                                     cu=cu)
            if not side_pos:
                ops.relpos_flash_bwd_dpos(qv, dS, lens, dp, B, H, T, dk, dpos_cast=dp_cast, cu=cu)
            return dqkv, dqu, dqv
        # dpd[h,b] = dctx_bh @ v_bh^T  -> reuse the f32 score workspace
        dpd = self._buf("ac", (H, B, T, Tp), torch.float32, dev)
        ops.gemm(dctx, qkv, dpd, T, T, dk, dA, 3 * dA, Tp, batch=H * B, nb0=B, sA=(T * dA, dk), sB=(T * 3 * dA, dk),
                 sC=(T * Tp, B * T * Tp), b_off=2 * dA)
        # dv_bh[j,e] = sum_i pd[i,j] dctx[i,e]
        ops.gemm(pd, dctx, dqkv, T, dk, T, Tp, dA, 3 * dA, transA=True, transB=True, batch=H * B, nb0=B,
                 sA=(T * Tp, B * T * Tp), sB=(T * dA, dk), sC=(T * 3 * dA, dk), c_off=2 * dA)
        dscore = self._buf("dscore", (H, B, T, Tp), cdt, dev)
        dbdf = self._buf("dbdf", (H, B, T, Pp), cdt, dev)
        ops.relpos_softmax_bwd(dpd, s_, dscore, dbdf, H, B, T, Tp, Pp, scale, d_att)
        # dqu_bh = dscore_bh [T,T] @ k_bh [T,dk]  (NN)
        ops.gemm(dscore, qkv, dqu, T, dk, T, Tp, 3 * dA, dA, transB=True, batch=H * B, nb0=B, sA=(T * Tp, B * T * Tp),
                 sB=(T * 3 * dA, dk), sC=(T * dA, dk), b_off=dA)
        # dk_bh[j,e] = sum_i dscore[i,j] qu[i,e]
        ops.gemm(dscore, qu, dqkv, T, dk, T, Tp, dA, 3 * dA, transA=True, transB=True, batch=H * B, nb0=B,
                 sA=(T * Tp, B * T * Tp), sB=(T * dA, dk), sC=(T * 3 * dA, dk), c_off=dA)
        # dqv_bh = dbdf_bh [T,P] @ p_h [P,dk]  (NN)
        ops.gemm(dbdf, p, dqv, T, dk, P, Pp, dA, dA, transB=True, batch=H * B, nb0=B, sA=(T * Pp, B * T * Pp),
                 sB=(0, dk), sC=(T * dA, dk))
        # dp_h[c,e] = sum_{b,i} dbdf[h,b,i,c] qv[b,i,h,e]   (reduction over all B*T rows of head h)
        tiles = self._tiles(P, dk, cdt == torch.bfloat16) * H
        ops.gemm(dbdf, qv, dp, P, dk, B * T, Pp, dA, dA, transA=True, transB=True, atomic=True,
                 splitk=self._splitk(tiles, B * T), batch=H, nb0=H, sA=(B * T * Pp, 0), sB=(dk, 0), sC=(dk, 0))
        ops.drop_scale_cast(dp, dp_cast, P * dA, 1.0)  # linear_pos weight gradients: one batched GEMM after the loop
        return dqkv, dqu, dqv

    def _layer_fwd(self, i, L, x, S, W, Wf, drop):
        B, F_, T, T1, F1, T2, F2, M, cdt, training, seed = S.dims
        dev = x.device
        d, H, dk, dff = self.d_model, self.n_heads, self.d_k, self.d_ff
        P = 2 * T2 - 1
        Tp, Pp = _pad8(T2), _pad8(P)
        sl = _Saved()
        site = i * 16
        pk = getattr(S, "pk", None)
        Mg, cu = M, (pk.cu if pk is not None else None)   # Mg: rows of the padded [B, T'] grid (conv core, statistics)
        if pk is not None:
            M = pk.Mp                                     # rows of the packed chain
        # ---- macaron FFN 1
        r1 = self._ffn_fwd(f"L{i}.ff1", L.feed_forward1, x, L.norm_feed_forward1, S, sl, W, drop, site, M, d, dff, cdt, dev, "ff1")
        # ---- rel-pos multi-head self-attention
        a = L.self_attn
        y2, mean2, rstd2 = self._ln_fwd(L.norm_self_att, r1, M, d, cdt, dev)
        _, dkp, dA = self._geometry(cdt)  # (dkp = dk, dA = d unless the heads are padded)
        qkv = self._new(M, 3 * dA, dtype=cdt, device=dev)
        ops.gemm(y2, W[f"L{i}.att.wqkv"], qkv, M, 3 * dA, d, d, W.pitch(f"L{i}.att.wqkv"), 3 * dA, bias=Wf[f"L{i}.att.bqkv"])
        p = S.p_all[i]  # linear_pos(pos_emb) of every layer was computed by one batched GEMM (same input, 18 weights)
        d_att = drop(self.dropout_att, site + 2)
        bu, bv = (a.pos_bias_u, a.pos_bias_v) if dkp == dk else (Wf[f"L{i}.att.bu"], Wf[f"L{i}.att.bv"])
        ctx, (qu, qv, s_, pd, lse) = self._attn_fwd(qkv, p, bu, bv, S.len2, B, T2, dA, dkp, 1.0 / math.sqrt(dk), d_att, cdt, dev,
                                                    pk=pk)
        r2 = self._new(M, d, dtype=torch.float32, device=dev)
        d_ares = drop(self.dropout, site + 3)
        ops.gemm(ctx, W[f"L{i}.att.wo"], r2, M, d, dA, dA, W.pitch(f"L{i}.att.wo"), d, bias=a.linear_out.bias, epi=ops.EPI_RESID,
                 aux_in=r1, drop=d_ares)
        sl.att = (r1, y2, mean2, rstd2, qkv, p, qu, qv, s_, pd, ctx, d_att, d_ares, lse)
        # ---- convolution module
        c = L.conv
        k = self.conv_kernel_size
        y3, mean3, rstd3 = self._ln_fwd(L.norm_conv, r2, M, d, cdt, dev)
        pw1 = self._new(M, 2 * d, dtype=cdt, device=dev)
        ops.gemm(y3, W[f"L{i}.conv.pw1"], pw1, M, 2 * d, d, d, W.pitch(f"L{i}.conv.pw1"), 2 * d, bias=c.pointwise_conv1.bias)
        g = self._new(Mg, d, dtype=cdt, device=dev)   # the conv core stays on the padded grid (BatchNorm counts padded frames)
        padl = self.conv_pad_left                      # -1: symmetric padding; else CausalConv1D's left pad (the fused GLU form is symmetric)
        fuse_glu = self.fuse_glu_dwconv_fwd and d % (8 if cdt == torch.bfloat16 else 4) == 0 and padl < 0
        if not fuse_glu:
            ops.glu_fwd(pw1, g, S.len2, T2, Mg, d, cu=cu)
        cc = self._new(Mg, d, dtype=cdt, device=dev)
        bn = c.batch_norm
        bmean = self._new(d, dtype=torch.float32, device=dev)
        brstd = self._new(d, dtype=torch.float32, device=dev)
        count = float(Mg)
        if self.conv_norm_type == "layer_norm":
            # nn.LayerNorm over the channels of every frame instead of BatchNorm (conformer_modules.py:335-338): depthwise conv without
            # statistics -> LayerNorm -> Swish; per-frame mean / rstd and the normalised frames travel in the BatchNorm slots
            if fuse_glu:
                ops.dwconv_fwd_glu(pw1, S.len2, cu, g, c.depthwise_conv.weight, c.depthwise_conv.bias, cc, None, B, T2, d, k)
            else:
                ops.dwconv_fwd(g, c.depthwise_conv.weight, c.depthwise_conv.bias, cc, None, B, T2, d, k, pad_left=padl)
            count, bmean, brstd = self._ln_fwd(bn, cc, Mg, d, cdt, dev)   # (yln, mean, rstd)
            z = self._new(Mg, d, dtype=cdt, device=dev)
            ops.swish_mask_fwd(count, z, None, T2, Mg, d)
        elif training:
            stats = S.bn_stats[i]
            if fuse_glu:   # GLU + pad mask applied while the depthwise forward stages its tile (g is written for backward, not re-read)
                ops.dwconv_fwd_glu(pw1, S.len2, cu, g, c.depthwise_conv.weight, c.depthwise_conv.bias, cc, stats, B, T2, d, k)
            else:
                ops.dwconv_fwd(g, c.depthwise_conv.weight, c.depthwise_conv.bias, cc, stats, B, T2, d, k, pad_left=padl)
            if S.bn_world > 1:  # sums and count in one exchange; the global count stays on the device
                self._sync_stats(stats[: 2 * d + 1])
                count = stats[2 * d: 2 * d + 1]
        else:
            if fuse_glu:
                ops.dwconv_fwd_glu(pw1, S.len2, cu, g, c.depthwise_conv.weight, c.depthwise_conv.bias, cc, None, B, T2, d, k)
            else:
                ops.dwconv_fwd(g, c.depthwise_conv.weight, c.depthwise_conv.bias, cc, None, B, T2, d, k, pad_left=padl)
            ops.bn_eval_stats(bn.running_mean, bn.running_var, bmean, brstd, bn.eps, d)
        if self.conv_norm_type == "batch_norm":
            z = self._new(Mg, d, dtype=cdt, device=dev)
            if training:
                # (two launches on purpose: the one-launch form, mi355x_bn_stats_swish_fwd, makes EVERY workgroup derive the
                #  coefficients of its channels from the f64 sums and measured 31 us against 15.5 us for this pair,
                #  tools/bn_bench.py)
                ops.bn_finalize(stats, count, bmean, brstd, bn.running_mean, bn.running_var, bn.momentum, bn.eps, d)
            ops.bn_swish_fwd(cc, bmean, brstd, bn.weight, bn.bias, z, Mg, d)
        if pk is not None:   # the valid frames of the core's output rejoin the packed chain (z is also the pw2 weight gradient's operand)
            zp = self._new(M, d, dtype=cdt, device=dev)
            ops.rows_pack(z, zp, S.len2, cu, T2, Mg, d)
            z = zp
        r3 = self._new(M, d, dtype=torch.float32, device=dev)
        d_cres = drop(self.dropout, site + 4)
        ops.gemm(z, W[f"L{i}.conv.pw2"], r3, M, d, d, d, W.pitch(f"L{i}.conv.pw2"), d, bias=c.pointwise_conv2.bias,
                 epi=ops.EPI_RESID, aux_in=r2, drop=d_cres)
        sl.conv = (r2, y3, mean3, rstd3, pw1, g, cc, bmean, brstd, count, z, d_cres)
        # ---- macaron FFN 2 + output norm
        r4 = self._ffn_fwd(f"L{i}.ff2", L.feed_forward2, r3, L.norm_feed_forward2, S, sl, W, drop, site + 5, M, d, dff, cdt, dev, "ff2")
        nxt = self.layers[i + 1] if i + 1 < self.n_layers else None
        if nxt is not None and d == 512 and self.fuse_layer_boundary_norms:
            # norm_out of this layer and norm_feed_forward1 of the next one in one pass over the rows (mi355x_layernorm2_fwd)
            ln1, ln2 = L.norm_out, nxt.norm_feed_forward1
            xo = self._new(M, d, dtype=torch.float32, device=dev)
            yn = self._new(M, d, dtype=cdt, device=dev)
            mean5, rstd5, mean_n, rstd_n = (self._new(M, dtype=torch.float32, device=dev) for _ in range(4))
            ops.layernorm2_fwd(r4, ln1.weight, ln1.bias, xo, mean5, rstd5, ln2.weight, ln2.bias, yn, mean_n, rstd_n, M, d, ln1.eps)
            S.pre_ln = (yn, mean_n, rstd_n)
        else:
            xo, mean5, rstd5 = self._ln_fwd(L.norm_out, r4, M, d, torch.float32, dev)
        sl.out = (r4, mean5, rstd5)
        return xo, sl

    @staticmethod
    def _dp_world():
        import torch.distributed as dist
        return dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1

    def _syncbn_world(self):
        if self.sync_batchnorm and torch.distributed.is_available() and torch.distributed.is_initialized():
            return torch.distributed.get_world_size()
        return 1

    def setup_process_groups(self):
        """Collective: every rank must call it at the same point (EncDecCTCModel._grad_syncs does, before the first step).
        MI355X_SYNCBN_OWN_GROUP=1 gives the SyncBatchNorm exchanges their own process group (= their own RCCL communicator and
        stream), so that the 8-KB latency-bound calls do not queue behind the 64-MiB gradient buckets in flight during
        backward.  Off by default: two communicators used concurrently from one process have not been validated on a multi-GPU
        RCCL run yet (every 2-rank test here runs over gloo), and the default group is the conservative choice until then."""
        import torch.distributed as dist
        if self._syncbn_group is None and dist.is_available() and dist.is_initialized():
            own = os.environ.get("MI355X_SYNCBN_OWN_GROUP", "0") == "1" and dist.get_world_size() > 1
            self._syncbn_group = dist.new_group(backend=dist.get_backend()) if own else dist.group.WORLD
        if (not self._syncbn_mailbox_tried and dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
                and os.environ.get("MI355X_SYNCBN_MAILBOX", "0") == "1"):
            # MI355X_SYNCBN_MAILBOX=1: the statistics exchanges leave the process group altogether -- one kernel launch each over
            # peer-mapped mailboxes (nemo_amd/mailbox.py, csrc/mailbox.hip).  All-or-nothing across the ranks; a job whose ranks
            # cannot map each other's memory keeps the process-group path above.
            self._syncbn_mailbox_tried = True
            dev = next(self.parameters()).device
            if dev.type == "cuda":
                from ..mailbox import StatsMailbox
                self._syncbn_mailbox = StatsMailbox.create(dev, n_max=max(8193, 4 * self.d_model + 1))
        return self._syncbn_group

    def _sync_stats(self, stats):
        """SyncBatchNorm: all-reduce the raw f64 sums (+ count) over the data-parallel ranks (torch.nn.SyncBatchNorm
        semantics).  A live host call also when the launch sequence is replayed from graphs (_eager_point)."""
        import torch.distributed as dist
        group = self._syncbn_group
        if group is None:
            # (an encoder driven without the model class: the own-group option then creates its group here, in the first training
            # forward -- new_group() is a collective, so every rank has to reach this forward)
            group = self.setup_process_groups()
        mb = self._syncbn_mailbox
        if mb is not None and stats.is_cuda:
            exchange = lambda: mb.all_reduce_(stats)
        else:
            exchange = lambda: dist.all_reduce(stats, group=group)
        if self.syncbn_profile is None:
            self._eager_point(exchange)
            return

        def timed():  # diagnostics (bench.py): the exchange blocks the chain -- its stream time IS exposed time
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            exchange()
            e1.record()
            if self.syncbn_profile is not None:
                self.syncbn_profile.append((e0, e1))
        self._eager_point(timed)

    # ------------------------------------------------------------------ backward implementation
    def step_scope(self):
        """with encoder.step_scope(): one forward, then its backward, before the next forward -- what a training loop does.
        Inside it the sequencer's tensors come from the step-scoped arena and (in auto mode) the launch sequence may be recorded."""
        import contextlib

        @contextlib.contextmanager
        def scope():
            prev, self._in_step = self._in_step, True
            try:
                yield self
            finally:
                self._in_step = prev
        return scope()

    def _check_serial(self, S):
        if getattr(S, "arena", False) and S.serial != self._fwd_serial:
            raise RuntimeError("backward through an encoder forward whose saved activations were overwritten by a later forward "
                               "(the sequencer's tensors live in a step-scoped arena); set MI355X_ARENA=0 (and MI355X_GRAPHS=0) for "
                               "several forwards per backward")

    def _backward_impl(self, S, dout, dcaps=()):
        B, F_, T, T1, F1, T2, F2, M, cdt, training, seed = S.dims
        dev = dout.device
        self._check_serial(S)
        self._phase("b", dev)
        d, C_ = self.d_model, self.pre_encode._conv_channels
        W, Wf = self._plan(cdt, dev)
        fp = self._flatp
        pk = getattr(S, "pk", None)

        def to_rows(g, key):  # gradient of a [B, D, T'] output -> rows of the layers' residual stream (through out_proj, packed)
            if self.out_proj is not None:
                gy = g.transpose(1, 2).contiguous().view(M, self._feat_out).to(torch.float32)
                gx = self._out_proj_bwd(gy, S.proj_in[key], M, dev)
            else:
                gx = g.transpose(1, 2).contiguous().view(M, d).to(torch.float32)  # no copy when g is a [B,T,d] view
            if pk is not None:  # packed rows: the layers' backward runs on the valid frames (the others never reached the loss)
                gp = self._new(pk.Mp, d, dtype=torch.float32, device=dev)
                ops.rows_pack(gx, gp, S.len2, pk.cu, T2, M, d)
                gx = gp
            return gx

        dx = to_rows(dout, "final")
        cap_grads = {l: g for l, g in zip(getattr(S, "cap_layers", []), dcaps) if g is not None}
        P = 2 * T2 - 1
        dA = self._geometry(cdt)[2]
        S.dp_all = self._buf("dp_all", (self.n_layers, P, dA), cdt, dev)
        S.bn_sums = torch.zeros(self.n_layers, 2, d, dtype=torch.float64, device=dev)
        S.dpos_f32 = torch.zeros(self.n_layers, P, dA, dtype=torch.float32, device=dev)
        self._wg_pending = [] if (self.wgrad_grouped and cdt == torch.bfloat16) else None
        defer = self.wgrad_defer if self._wg_pending is not None else 0
        self._wg_deferred = None

        def layer_done(j):
            if self.grad_ready_hook is not None:
                # the layer's gradients are final only when its side-stream wgrads have run: either the consumer waits for
                # that stream itself (GradSync.producer_streams) or the backward chain joins here
                if self._wgrad_join_per_layer:
                    self._wgrad_join()
                self._hook(*fp.range_of(f"layers.{j}."))

        nlay = max(1, min(4, self.wgrad_layers)) if defer else 1   # layers' weight gradients per grouped launch
        pair = nlay > 1

        def flush_deferred(force=True):
            # the grouped weight-gradient launch of the layer(s) ABOVE, issued at the point of this layer's backward where the main
            # chain runs kernels that share a CU with it (self.wgrad_defer) -- not beside the next layer's first GEMMs
            dfr = self._wg_deferred
            if dfr is None or (pair and not force and len(dfr[2]) < nlay):
                return
            self._wg_deferred = None
            cur = (self._wg_pending, self._wg_rows)
            self._wg_pending, self._wg_rows, js = dfr
            self._wgrad_flush()
            self._wg_pending, self._wg_rows = cur
            for j in js:
                layer_done(j)
        self._defer_flush = (lambda: flush_deferred(force=not pair)) if defer else None
        for i in range(self.n_layers - 1, -1, -1):
            if i in cap_grads:  # InterCTC: the gradient of the captured output of layer i joins the stream here
                dx = dx + to_rows(cap_grads[i], i)
            sd = S.sd[i] if getattr(S, "sd", None) else None
            if sd is None:
                dx = self._layer_bwd(i, self.layers[i], dx, S, S.layers[i], W, Wf)
            elif sd[0] == "drop":
                # a dropped layer contributed x * 0: its backward runs on a zero gradient (every weight gets its -- zero --
                # gradient, the per-layer hooks and accumulators see the usual sequence), the stream's gradient passes by
                self._layer_bwd(i, self.layers[i], torch.zeros_like(dx), S, S.layers[i], W, Wf)
            else:
                # kept and rescaled: x_in + a (L(x_in) - x_in)
                din = self._layer_bwd(i, self.layers[i], dx * sd[1], S, S.layers[i], W, Wf)
                dx = torch.add(din, dx, alpha=1.0 - sd[1])
            S.layers[i] = None
            if defer:
                dfr = self._wg_deferred
                if pair and dfr is not None and len(dfr[2]) < nlay and dfr[1] == self._wg_rows:
                    self._wg_deferred = (dfr[0] + self._wg_pending, dfr[1], dfr[2] + [i])
                else:
                    flush_deferred()  # (a layer whose backward never reached the defer point)
                    self._wg_deferred = (self._wg_pending, self._wg_rows, [i])
                self._wg_pending, self._wg_rows = [], None
                continue
            self._wgrad_flush()
            layer_done(i)
        if defer:
            flush_deferred()
        self._defer_flush = None
        self._wg_pending = None  # (the remaining weight gradients have their own shapes / layouts)
        if pk is not None:  # back onto the [B, T', d] grid of the sub-sampling stack (zero gradient beyond the utterances)
            dxu = self._new(M, d, dtype=torch.float32, device=dev)
            ops.rows_unpack(dx, dxu, S.len2, pk.cu, T2, M, d)
            dx = dxu
        # the linear_pos weight gradients consume dp_all, which the side stream produced (dpos kernels): they run THERE, behind
        # their producers, instead of making the main chain wait for the whole weight-gradient lane in front of the sub-sampling
        # backward (a 330-us bubble per step on the main stream, profiles/r5_launch_modes.md).  Without a side stream the scope is
        # empty and everything is in stream order anyway.
        if self.posproj_side:
            with self._wgrad_scope(S.dp_all, S.pos):
                self._pos_proj_wgrad(S.dp_all, S.pos, P, cdt)
        else:
            self._wgrad_join(consume=True)
            self._pos_proj_wgrad(S.dp_all, S.pos, P, cdt)
        if self.grad_ready_hook is not None:
            if self.posproj_side and self._wgrad_join_per_layer:
                # the tail range holds the linear_pos weight gradients, which the side stream is still writing: a hook consumer
                # that does not itself wait on the weight-gradient stream (a custom hook, GradSync without producer_streams) must
                # not see the range before they landed -- the same join the per-layer hooks get (layer_done)
                self._wgrad_join(consume=True)
            self._hook(*fp.tail_range())
        if getattr(S, "bypass", False):   # pre-encoded input: no sub-sampling stack behind the layers (its gradients stay zero)
            return None
        # ---- sub-sampling backward
        pe = self.pre_encode
        if self.subsampling == "dw_striding":
            return self._sub_bwd_dw(S, dx, W, cdt)
        dxs = self._new(M, d, dtype=cdt, device=dev)
        ops.drop_scale_cast(dx, dxs, M * d, (self.xscale or 1.0), S.drop_pre)
        with self._sub_wgrad_scope(dxs):
            ops.colsum(dxs, pe.out.bias.grad, M, d)
        # d out.weight in the reference's (c, f) column order: batch over f, C column stride F2
        tiles = self._tiles(d, C_, cdt == torch.bfloat16) * F2
        with self._sub_wgrad_scope(dxs, S.out2):
            ops.gemm(dxs, S.out2, pe.out.weight.grad, d, C_, M, d, F2 * C_, C_ * F2, transA=True, transB=True, atomic=True,
                     splitk=self._splitk(tiles, M, strided_c=True), batch=F2, nb0=F2, sB=(C_, 0), sC=(1, 0), c_col_stride=F2,
                     c_dtype=ops.F32)
        dout2 = self._new(B * T2 * F2, C_, dtype=cdt, device=dev)
        # (row_len: out2 -- the ReLU gate -- is zero beyond an utterance's len2 frames, row tiles that lie there are zero-filled)
        ops.gemm(dxs, W["pre.outt"], dout2, M, F2 * C_, d, d, W.pitch("pre.outt"), F2 * C_, epi=ops.EPI_MUL_POS, aux_in=S.out2,
                 row_len=S.len2 if self.pad_tile_skip else None, rows_per_b=T2, rows_inner=1)
        M2 = B * T2 * F2
        with self._sub_wgrad_scope(dout2):
            ops.colsum(dout2, pe.conv[2].bias.grad, M2, C_)  # 328 MB stream: next to the GEMMs, not in line with them
        implicit = self.conv2_implicit and self._conv2_implicit(cdt, C_, M2) and S.col is None
        # d conv2.weight [co, ci, 3, 3]: batch over the 9 taps, column stride 9
        tiles = self._tiles(C_, C_, cdt == torch.bfloat16) * 9
        if implicit:
            with self._sub_wgrad_scope(dout2, S.out1):
                ops.gemm(dout2, S.out1, pe.conv[2].weight.grad, C_, C_, M2, C_, C_, 9 * C_, transA=True, transB=True,
                         atomic=True, splitk=self._splitk(tiles, M2), batch=9, nb0=9, sC=(1, 0), c_col_stride=9,
                         c_dtype=ops.F32, row_len=S.len2 if self.pad_tile_skip else None, rows_per_b=T2 * F2, rows_inner=F2,   # (K-tiles beyond an utterance: skipped)
                         gather=dict(operand=1, nI=T2, nJ=F2, SI=T1, SJ=F1, C=C_, si=2, sj=2,
                                     taps=[(kh - pe._pad, kw - pe._pad) for kh in range(3) for kw in range(3)]))
        else:
            col = S.col
            if S.col_gen != self._col_gen:  # another forward has reused the workspace since: rebuild the image
                ops.im2col(S.out1, col, B, T1, F1, C_, pad=pe._pad)
                self._col_gen += 1
            ops.gemm(dout2, col, pe.conv[2].weight.grad, C_, C_, M2, C_, 9 * C_, 9 * C_, transA=True, transB=True,
                     atomic=True, splitk=self._splitk(tiles, M2), batch=9, nb0=9, sB=(C_, 0), sC=(1, 0), c_col_stride=9,
                     c_dtype=ops.F32)
        dout1 = self._buf("dout1", (B, T1, F1, C_), cdt, dev)
        if self.conv2_implicit and self._conv2_implicit(cdt, C_, M2):
            # four implicit GEMMs (one per parity class of (t1, f1)) gather dout2 and write the class's rows of dout1 with
            # the ReLU gate of conv1's output in the epilogue: no 3 GB dcol image, no col2im pass
            for pt in (0, 1):
                for pf in (0, 1):
                    nI, nJ = (T1 - pt + 1) // 2, (F1 - pf + 1) // 2
                    slots = self._dgrad_slots(pt, pf, pe._pad)
                    name = f"pre.w2d{pt}{pf}"
                    # (row_len: conv1's output -- the ReLU gate -- is zero beyond an utterance's len1 <= 2 * len2 frames, so row tiles
                    #  with i >= len2 are zero-filled without a K loop: 40 % of the tiles of an unshaped 5-30 s batch)
                    ops.gemm(dout2, W[name], dout1, B * nI * nJ, C_, len(slots) * C_, C_, W.pitch(name), C_,
                             epi=ops.EPI_MUL_POS, aux_in=S.out1, ldaux=C_, row_len=S.len2 if self.pad_tile_skip else None,
                             rows_per_b=nI * nJ, rows_inner=nJ,
                             gather=dict(nI=nI, nJ=nJ, SI=T2, SJ=F2, C=C_, si=1, sj=1,
                                         # input position 2 i + pt is read by output i + (pt + pad - kh) / 2 through tap kh
                                         taps=[((pt + pe._pad - kh) // 2, (pf + pe._pad - kw) // 2) for kh, kw in slots]),
                             rowmap=dict(nI=nI, nJ=nJ, OI=T1, OJ=F1, si=2, sj=2, oi=pt, oj=pf))
        else:
            dcol = self._buf("dcol", (M2, 9 * C_), cdt, dev)
            ops.gemm(dout2, W["pre.w2t"], dcol, M2, 9 * C_, C_, C_, W.pitch("pre.w2t"), 9 * C_)
            ops.col2im_relu(dcol, S.out1, dout1, B, T1, F1, C_, pad=pe._pad)
        ops.conv1_bwd(dout1, S.mel, S.len0, pe.conv[0].weight.grad, pe.conv[0].bias.grad, C_, pad=pe._pad)
        self._wgrad_join()
        if self.grad_ready_hook is not None:
            self._hook(*fp.range_of("pre_encode."))

    def _ffn_bwd(self, pfx, ff, ln, saved, dr, W, M, d, dff, cdt, dev, df=None, next_cast=None, boundary=None):
        """`df` = the already cast / dropped / scaled residual-branch gradient when the previous LayerNorm backward produced
        it in its own pass; `next_cast` = (scale, Dropout) of the sub-block that follows in backward order: this block's
        LayerNorm backward then emits that operand too.  `boundary` = (norm_out module, r4, mean5, rstd5) of the layer BELOW when
        this is a layer's first feed-forward block: its LayerNorm backward and that norm_out's run as one kernel
        (mi355x_layernorm2_bwd), the returned dr is then already the gradient w.r.t. the lower layer's r4.
        Returns (dr, cast_for_next_or_None)."""
        x, y, mean, rstd, h, a, d_in, d_res = saved
        if df is None:
            df = self._new(M, d, dtype=cdt, device=dev)
            ops.drop_scale_cast(dr, df, M * d, 0.5, d_res)
        dh = self._new(M, dff, dtype=cdt, device=dev)
        dy = self._new(M, d, dtype=cdt, device=dev)
        if a is None:  # fused forward: the input-gradient chain in one launch, which also re-creates linear2's weight-gradient operand
            a = self._new(M, dff, dtype=cdt, device=dev)
            ops.ffn_bwd_dgrad(df, W[pfx + ".w2tp"], W[pfx + ".w1tp"], h, dh, a, dy, M, d, dff, d_in)
            self._wgrad(df, d, 0, a, dff, 0, ff.linear2.weight.grad, d, dff, M, bias_grad=ff.linear2.bias.grad)
            self._wgrad(dh, dff, 0, y, d, 0, ff.linear1.weight.grad, dff, d, M, bias_grad=ff.linear1.bias.grad)
        else:
            self._wgrad(df, d, 0, a, dff, 0, ff.linear2.weight.grad, d, dff, M, bias_grad=ff.linear2.bias.grad)
            if d_in is None:  # the forward stored swish'(h) * mask (swish_g): one multiply per element
                ops.gemm(df, W[pfx + ".w2t"], dh, M, dff, d, d, W.pitch(pfx + ".w2t"), dff, epi=ops.EPI_DSWISH_G, aux_in=h)
            else:
                ops.gemm(df, W[pfx + ".w2t"], dh, M, dff, d, d, W.pitch(pfx + ".w2t"), dff, epi=ops.EPI_DSWISH, aux_in=h, drop=d_in)
            self._wgrad(dh, dff, 0, y, d, 0, ff.linear1.weight.grad, dff, d, M, bias_grad=ff.linear1.bias.grad)
            ops.gemm(dh, W[pfx + ".w1t"], dy, M, d, dff, dff, W.pitch(pfx + ".w1t"), d)
        nxt = self._cast_buf(next_cast, M, d, cdt, dev)
        if boundary is not None:
            ln_lo, r4, mean5, rstd5 = boundary
            dr_lo = self._new(M, d, dtype=torch.float32, device=dev)
            ops.layernorm2_bwd(dy, x, ln.weight, mean, rstd, ln.weight.grad, ln.bias.grad, dr, r4, ln_lo.weight, mean5, rstd5,
                               ln_lo.weight.grad, ln_lo.bias.grad, dr_lo, M, d, cast_out=nxt,
                               cast_scale=next_cast[0] if nxt is not None else 1.0,
                               cast_drop=next_cast[1] if nxt is not None else None)
            return dr_lo, nxt
        ops.layernorm_bwd(dy, x, ln.weight, mean, rstd, dr, True, ln.weight.grad, ln.bias.grad, M, d, cast_out=nxt,
                          cast_scale=next_cast[0] if nxt is not None else 1.0,
                          cast_drop=next_cast[1] if nxt is not None else None)
        return dr, nxt

    def _cast_buf(self, next_cast, M, d, cdt, dev):
        """bf16 operand buffer for a fused LayerNorm-backward cast (bf16 compute only; d must be one the 8-wide kernel covers
        or the C side falls back to a separate pass -- still correct)"""
        if next_cast is None or cdt != torch.bfloat16 or (M * d) % 8 or not self.ln_cast_fuse:
            return None
        return self._new(M, d, dtype=cdt, device=dev)

    def _attn_block_bwd(self, i, a, saved, dao, S, W, M, B, T2, d, dk, scale, cdt, dev):
        """linear_out -> attention core -> q | k | v projections, backward (heads of width d_k % 8 == 0, or fp32)"""
        r1, y2, mean2, rstd2, qkv, p, qu, qv, s_, pd, ctx, d_att, d_ares, lse = saved
        self._wgrad(dao, d, 0, ctx, d, 0, a.linear_out.weight.grad, d, d, M, bias_grad=a.linear_out.bias.grad)
        dctx = self._new(M, d, dtype=cdt, device=dev)
        ops.gemm(dao, W[f"L{i}.att.wot"], dctx, M, d, d, d, W.pitch(f"L{i}.att.wot"), d)
        gu, gv_ = a.pos_bias_u.grad, a.pos_bias_v.grad
        adjacent = gv_.data_ptr() - gu.data_ptr() == 4 * d  # (the two bias gradients as one [2 * d] piece of the flat buffer)
        dqkv, dqu, dqv = self._attn_bwd((qu, qv, s_, pd, lse), qkv, p, a.pos_bias_u, a.pos_bias_v, ctx, dctx, S.len2, B, T2, d,
                                        dk, scale, d_att, cdt, dev, S.dpos_f32[i], S.dp_all[i],
                                        bias_grads=gu if (adjacent and lse is not None) else None, pk=getattr(S, "pk", None))
        if dqu is None:
            pass  # fused attention: dq and both bias gradients came out of the dQ kernel
        elif cdt == torch.bfloat16 and adjacent:
            ops.add2_colsum(dqu, dqv, dqkv, 3 * d, M, d, gu)  # dq = dqu + dqv and both bias gradients in one pass
        else:
            ops.colsum(dqu, gu, M, d)
            ops.colsum(dqv, gv_, M, d)
            ops.add2(dqu, dqv, dqkv, 3 * d, M, d)
        gq, gk, gv = a.linear_q.weight.grad, a.linear_k.weight.grad, a.linear_v.weight.grad
        sw = (gk.data_ptr() - gq.data_ptr()) // 4
        sb = (a.linear_k.bias.grad.data_ptr() - a.linear_q.bias.grad.data_ptr()) // 4
        if (cdt == torch.bfloat16 and sw > 0 and (gv.data_ptr() - gk.data_ptr()) // 4 == sw
                and (a.linear_v.bias.grad.data_ptr() - a.linear_k.bias.grad.data_ptr()) // 4 == sb):
            # q, k, v weight (and bias) gradients as ONE batched TN GEMM: the three gradients are equally spaced in the
            # flat gradient buffer, the three dY column blocks equally spaced in dqkv
            if self._wg_pending is not None:  # three more problems of the layer's grouped launch
                for j, lin in enumerate((a.linear_q, a.linear_k, a.linear_v)):
                    self._wgrad(dqkv, 3 * d, j * d, y2, d, 0, lin.weight.grad, d, d, M, bias_grad=lin.bias.grad)
            else:
                with self._wgrad_scope(dqkv, y2):
                    ops.gemm(dqkv, y2, gq, d, d, M, 3 * d, d, d, transA=True, transB=True, atomic=True,
                             splitk=self._splitk(3 * self._tiles(d, d, True), M), batch=3, nb0=3, sA=(d, 0), sC=(sw, 0),
                             c_dtype=ops.F32, colsum_out=a.linear_q.bias.grad, colsum_stride=sb)
        else:
            for j, lin in enumerate((a.linear_q, a.linear_k, a.linear_v)):
                self._wgrad(dqkv, 3 * d, j * d, y2, d, 0, lin.weight.grad, d, d, M, bias_grad=lin.bias.grad)
        dy2 = self._new(M, d, dtype=cdt, device=dev)
        ops.gemm(dqkv, W[f"L{i}.att.wqkvt"], dy2, M, d, 3 * d, 3 * d, W.pitch(f"L{i}.att.wqkvt"), d)
        return dy2

    def _attn_block_bwd_padded(self, i, a, saved, dao, S, W, Wf, M, B, T2, d, dk, dkp, dA, scale, cdt, dev):
        """the same with zero-padded heads (bf16, d_k % 8 != 0): the heads' weight gradients are TN GEMMs batched over the heads
        whose strides step over the pad lanes, bias-sized gradients are summed in the padded layout and added back without it"""
        r1, y2, mean2, rstd2, qkv, p, qu, qv, s_, pd, ctx, d_att, d_ares, lse = saved
        H = self.n_heads
        P = 2 * T2 - 1
        with self._wgrad_scope(dao, ctx):
            ops.gemm(dao, ctx, a.linear_out.weight.grad, d, dk, M, d, dA, d, transA=True, transB=True, atomic=True,
                     splitk=self._splitk(self._tiles(d, dk, True) * H, M), batch=H, nb0=H, sB=(dkp, 0), sC=(dk, 0),
                     c_dtype=ops.F32)
            ops.colsum(dao, a.linear_out.bias.grad, M, d)
        dctx = self._new(M, dA, dtype=cdt, device=dev)
        ops.gemm(dao, W[f"L{i}.att.wot"], dctx, M, dA, d, d, W.pitch(f"L{i}.att.wot"), dA)
        dqkv, dqu, dqv = self._attn_bwd((qu, qv, s_, pd, lse), qkv, p, Wf[f"L{i}.att.bu"], Wf[f"L{i}.att.bv"], ctx, dctx, S.len2, B,
                                        T2, dA, dkp, scale, d_att, cdt, dev, S.dpos_f32[i], S.dp_all[i], pk=getattr(S, "pk", None))
        sc = torch.zeros(2, dA, dtype=torch.float32, device=dev)
        ops.colsum(dqu, sc[0], M, dA)
        ops.colsum(dqv, sc[1], M, dA)
        self._unpad_add(a.pos_bias_u.grad.view(-1), sc[0], dkp)
        self._unpad_add(a.pos_bias_v.grad.view(-1), sc[1], dkp)
        ops.add2(dqu, dqv, dqkv, 3 * dA, M, dA)
        lins = (a.linear_q, a.linear_k, a.linear_v)
        gq, gk, gvw = (lin.weight.grad for lin in lins)
        sw = (gk.data_ptr() - gq.data_ptr()) // 4
        if sw > 0 and (gvw.data_ptr() - gk.data_ptr()) // 4 == sw:
            self._heads_wgrad(dqkv, 3 * dA, 0, y2, d, gq, M, 3, dA, sw)
        else:
            for j, lin in enumerate(lins):
                self._heads_wgrad(dqkv, 3 * dA, j * dA, y2, d, lin.weight.grad, M, 1, 0, 0)
        sb = torch.zeros(3 * dA, dtype=torch.float32, device=dev)
        with self._wgrad_scope(dqkv, sb):
            ops.colsum(dqkv, sb, M, 3 * dA)
            for j, lin in enumerate(lins):
                self._unpad_add(lin.bias.grad, sb[j * dA:(j + 1) * dA], dkp)
        dy2 = self._new(M, d, dtype=cdt, device=dev)
        ops.gemm(dqkv, W[f"L{i}.att.wqkvt"], dy2, M, d, 3 * dA, 3 * dA, W.pitch(f"L{i}.att.wqkvt"), d)
        return dy2

    def _layer_bwd(self, i, L, dxo, S, sl, W, Wf):
        B, F_, T, T1, F1, T2, F2, M, cdt, training, seed = S.dims
        dev = dxo.device
        d, H, dk, dff = self.d_model, self.n_heads, self.d_k, self.d_ff
        P = 2 * T2 - 1
        Tp, Pp = _pad8(T2), _pad8(P)
        scale = 1.0 / math.sqrt(dk)
        pk = getattr(S, "pk", None)
        Mg, cu = M, (pk.cu if pk is not None else None)   # Mg: rows of the padded grid (conv core); M: rows of the (packed) chain
        if pk is not None:
            M = pk.Mp
        # ---- norm_out: dr = dLN(dxo)
        r4, mean5, rstd5 = sl.out
        pre = getattr(S, "pre_bwd", None)
        if pre is not None:  # the layer above ran this LayerNorm backward together with its own norm_feed_forward1's
            dr, df2 = pre
            S.pre_bwd = None
        else:
            dr = self._new(M, d, dtype=torch.float32, device=dev)
            ln = L.norm_out
            # every LayerNorm backward also emits the bf16 (scaled, dropped) copy of the new residual gradient that the next
            # sub-block's output GEMMs consume -- one read of the fp32 gradient and one launch less per sub-block
            nc = (0.5, sl.ff2[7])
            df2 = self._cast_buf(nc, M, d, cdt, dev)
            ops.layernorm_bwd(dxo, r4, ln.weight, mean5, rstd5, dr, False, ln.weight.grad, ln.bias.grad, M, d, cast_out=df2,
                              cast_scale=0.5, cast_drop=nc[1] if df2 is not None else None)
        # ---- FFN 2
        dr, db_pre = self._ffn_bwd(f"L{i}.ff2", L.feed_forward2, L.norm_feed_forward2, sl.ff2, dr, W, M, d, dff, cdt, dev,
                                   df=df2, next_cast=(1.0, sl.conv[11]))
        # ---- convolution module
        c = L.conv
        bn = c.batch_norm
        k = self.conv_kernel_size
        r2, y3, mean3, rstd3, pw1, g, cc, bmean, brstd, count, z, d_cres = sl.conv
        db = db_pre
        if db is None:
            db = self._new(M, d, dtype=cdt, device=dev)
            ops.drop_scale_cast(dr, db, M * d, 1.0, d_cres)
        self._wgrad(db, d, 0, z, d, 0, c.pointwise_conv2.weight.grad, d, d, M, bias_grad=c.pointwise_conv2.bias.grad)
        dz = self._new(M, d, dtype=cdt, device=dev)
        ops.gemm(db, W[f"L{i}.conv.pw2t"], dz, M, d, d, d, W.pitch(f"L{i}.conv.pw2t"), d)
        if pk is not None:   # onto the padded grid of the conv core; frames beyond an utterance carry no gradient
            dzp = self._new(Mg, d, dtype=cdt, device=dev)
            ops.rows_unpack(dz, dzp, S.len2, cu, T2, Mg, d)
            dz = dzp
        sums = S.bn_sums[i]
        padl = self.conv_pad_left
        self._defer_point(1)
        ln_norm = self.conv_norm_type == "layer_norm"
        if not ln_norm:
            ops.bn_swish_bwd_reduce(dz, cc, bmean, brstd, bn.weight, bn.bias, sums, Mg, d, dgamma=bn.weight.grad, dbeta=bn.bias.grad)
            if training and S.bn_world > 1:
                self._sync_stats(sums)
        self._defer_point(5)   # (experimental entry points of the weight-gradient launch: behind the BatchNorm reduction ...)
        dpw1 = self._new(M, 2 * d, dtype=cdt, device=dev)
        if ln_norm:
            # Swish -> LayerNorm -> depthwise conv -> GLU, backwards (count / bmean / brstd carry yln / the per-frame mean / rstd)
            dyln = self._new(Mg, d, dtype=cdt, device=dev)
            ops.swish_mask_bwd(count, dz, dyln, None, T2, Mg, d)
            dcc32 = self._new(Mg, d, dtype=torch.float32, device=dev)
            if cdt == torch.bfloat16:
                dcc = self._new(Mg, d, dtype=cdt, device=dev)
                ops.layernorm_bwd(dyln, cc, bn.weight, bmean, brstd, dcc32, False, bn.weight.grad, bn.bias.grad, Mg, d, cast_out=dcc)
            else:
                ops.layernorm_bwd(dyln, cc, bn.weight, bmean, brstd, dcc32, False, bn.weight.grad, bn.bias.grad, Mg, d)
                dcc = dcc32
            dg = self._new(Mg, d, dtype=cdt, device=dev)
            ops.dwconv_bwd(dcc, g, c.depthwise_conv.weight, dg, c.depthwise_conv.weight.grad, c.depthwise_conv.bias.grad, B, T2, d, k,
                           pad_left=padl)
            self._defer_point(6)
            ops.glu_bwd(pw1, dg, dpw1, S.len2, T2, Mg, d, cu=cu)
        elif self.fuse_bn_dwconv_bwd and self.fuse_glu_dwconv_bwd and padl < 0:
            # BatchNorm + Swish backward applied while the depthwise backward stages its gradient tile, the GLU backward while it
            # writes its result: one launch for four, and neither the [B, T', d] gradient w.r.t. the BatchNorm input nor the one
            # w.r.t. the GLU output is written or read back
            side = self.tap_reduce_side and self.wgrad_side_stream
            sc = ops.dwconv_tap_scratch(i, B, d, k, dev) if side else None   # (its own slabs per layer: the reduction runs later)
            ops.dwconv_bwd_bnswish(dz, cc, bmean, brstd, bn.weight, bn.bias, sums, count, training, g, c.depthwise_conv.weight, None,
                                   c.depthwise_conv.weight.grad, c.depthwise_conv.bias.grad, B, T2, d, k, glu_in=pw1, glu_din=dpw1,
                                   glu_len=S.len2, glu_cu=cu, scratch=sc, defer_reduce=side)
            if side:
                # the second stage of the tap / bias gradient: nothing on the chain reads it -- it joins the weight-gradient stream
                with self._wgrad_scope(sc, c.depthwise_conv.weight.grad):
                    ops.dwconv_tap_reduce(sc, B, d, k, c.depthwise_conv.weight.grad, c.depthwise_conv.bias.grad)
            self._defer_point(6)
        else:
            dg = self._new(Mg, d, dtype=cdt, device=dev)
            if self.fuse_bn_dwconv_bwd and padl < 0:
                ops.dwconv_bwd_bnswish(dz, cc, bmean, brstd, bn.weight, bn.bias, sums, count, training, g, c.depthwise_conv.weight, dg,
                                       c.depthwise_conv.weight.grad, c.depthwise_conv.bias.grad, B, T2, d, k)
            else:
                dcc = self._new(Mg, d, dtype=cdt, device=dev)
                ops.bn_swish_bwd_apply(dz, cc, bmean, brstd, bn.weight, bn.bias, sums, count, training, dcc, Mg, d)
                ops.dwconv_bwd(dcc, g, c.depthwise_conv.weight, dg, c.depthwise_conv.weight.grad, c.depthwise_conv.bias.grad, B, T2, d, k,
                               pad_left=padl)
            self._defer_point(6)   # (... and behind the depthwise backward, beside the GLU backward and the pointwise dgrad GEMM)
            ops.glu_bwd(pw1, dg, dpw1, S.len2, T2, Mg, d, cu=cu)
        self._wgrad(dpw1, 2 * d, 0, y3, d, 0, c.pointwise_conv1.weight.grad, 2 * d, d, M, bias_grad=c.pointwise_conv1.bias.grad)
        dy3 = self._new(M, d, dtype=cdt, device=dev)
        ops.gemm(dpw1, W[f"L{i}.conv.pw1t"], dy3, M, d, 2 * d, 2 * d, W.pitch(f"L{i}.conv.pw1t"), d)
        ln = L.norm_conv
        self._defer_point(3)
        nc = (1.0, sl.att[12])
        dao_pre = self._cast_buf(nc, M, d, cdt, dev)
        ops.layernorm_bwd(dy3, r2, ln.weight, mean3, rstd3, dr, True, ln.weight.grad, ln.bias.grad, M, d, cast_out=dao_pre,
                          cast_scale=1.0, cast_drop=nc[1] if dao_pre is not None else None)
        # ---- self-attention
        a = L.self_attn
        r1, y2, mean2, rstd2, qkv, p, qu, qv, s_, pd, ctx, d_att, d_ares, lse = sl.att
        dao = dao_pre
        if dao is None:
            dao = self._new(M, d, dtype=cdt, device=dev)
            ops.drop_scale_cast(dr, dao, M * d, 1.0, d_ares)
        _, dkp, dA = self._geometry(cdt)
        self._defer_point(2)
        if dkp != dk:
            dy2 = self._attn_block_bwd_padded(i, a, sl.att, dao, S, W, Wf, M, B, T2, d, dk, dkp, dA, scale, cdt, dev)
        else:
            dy2 = self._attn_block_bwd(i, a, sl.att, dao, S, W, M, B, T2, d, dk, scale, cdt, dev)
        ln = L.norm_self_att
        nc = (0.5, sl.ff1[7])
        df1 = self._cast_buf(nc, M, d, cdt, dev)
        ops.layernorm_bwd(dy2, r1, ln.weight, mean2, rstd2, dr, True, ln.weight.grad, ln.bias.grad, M, d, cast_out=df1,
                          cast_scale=0.5, cast_drop=nc[1] if df1 is not None else None)
        # ---- FFN 1 (its LayerNorm backward together with the norm_out backward of the layer below, where the kernel covers it)
        lo = S.layers[i - 1] if i > 0 else None
        if lo is not None and d == 512 and self.fuse_layer_boundary_norms and self.fuse_boundary_bwd:
            r4_lo, mean5_lo, rstd5_lo = lo.out
            dr, df2_lo = self._ffn_bwd(f"L{i}.ff1", L.feed_forward1, L.norm_feed_forward1, sl.ff1, dr, W, M, d, dff, cdt, dev, df=df1,
                                       next_cast=(0.5, lo.ff2[7]),
                                       boundary=(self.layers[i - 1].norm_out, r4_lo, mean5_lo, rstd5_lo))
            S.pre_bwd = (dr, df2_lo)
            return dr
        dr, _ = self._ffn_bwd(f"L{i}.ff1", L.feed_forward1, L.norm_feed_forward1, sl.ff1, dr, W, M, d, dff, cdt, dev, df=df1)
        return dr
