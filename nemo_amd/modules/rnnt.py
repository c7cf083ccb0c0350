"""Drop-ins for `nemo.collections.asr.modules.RNNTDecoder` (prediction network, rnnt.py:552-900) and `RNNTJoint` (joint
network, rnnt.py:1280-1800) on MI355X -- the transducer head of FastConformer-Transducer (BASELINE.json configs[3]).

Same constructor dicts (`prednet`, `jointnet`), typed call signatures and state-dict keys (`prediction.embed.weight`,
`prediction.dec_rnn.lstm.{weight,bias}_{ih,hh}_l*`, `pred.*`, `enc.*`, `joint_net.N.*` with N = 2 when the joint has a
Dropout layer and 1 otherwise -- the reference's nn.Sequential indices).  The arithmetic is sequenced here over
libmi355x_asr.so: gate / projection / output-layer contractions are `mi355x_gemm`, the LSTM cell, the embedding with its
start-of-sequence frame, the broadcast-add + ReLU + dropout of the joint and their gradients are the kernels of
`csrc/transducer.hip`, the loss is `mi355x_rnnt_loss`.

`RNNTJoint(fuse_loss_wer=True, fused_batch_size=n)` is the training path of the recipe
(`examples/asr/conf/fastconformer/fast-conformer_transducer_bpe.yaml`, joint.fuse_loss_wer / fused_batch_size): the
[B, T, U+1, V+1] logits exist for n utterances at a time, the loss kernels produce their gradient in the same pass, and
the joint's own backward (output layer, ReLU/dropout gate, the two reductions over u and over t) runs right behind it, so
nothing of size T*U*V outlives its sub-batch.  Unsupported options (normalization_mode, random_state_sampling, tanh /
sigmoid joints, masking_prob, adapters, export) raise NotImplementedError.
"""
from __future__ import annotations

import os
from collections import OrderedDict
from typing import Any, Dict, List, Optional

import torch
from torch import nn

from .. import ops
from ..core import (AcousticEncodedRepresentation, ElementType, EmbeddedTextType, LabelsType, LengthsType, LogprobsType,
                    LossType, NeuralModule, NeuralType, typecheck)
from ..flat import FlatParams
from ..packing import PackPlan


def _pad8(n):
    return (n + 7) // 8 * 8


class _ModuleBase(NeuralModule):
    """flat parameters + packed GEMM operand images + compute dtype, shared by the two modules"""

    def _init_engine(self, compute_dtype):
        self.compute_dtype = compute_dtype
        self.grad_ready_hook = None
        self._flatp = FlatParams(self)
        self._plans = {}
        self._weights_version = -1
        self._token = None
        self._step_seed = 0

    def flat_parameters(self):
        self._flatp.ensure()
        return self._flatp

    def weights_updated(self):
        self._weights_version += 1

    def load_state_dict(self, *a, **k):
        r = super().load_state_dict(*a, **k)
        self.weights_updated()
        return r

    def _cdt(self):
        cdt = torch.float32
        if self.compute_dtype is not None:
            cdt = self.compute_dtype
        elif torch.is_autocast_enabled() and torch.get_autocast_gpu_dtype() == torch.bfloat16:
            cdt = torch.bfloat16
        if cdt == torch.bfloat16 and any(int(getattr(self, a, 8)) % 8 for a in ("pred_hidden", "joint_hidden", "encoder_hidden")):
            raise NotImplementedError("bf16 compute needs hidden sizes divisible by 8; use compute_dtype=torch.float32")
        return cdt

    def _plan(self, cdt, device):
        key = (cdt, str(device), self._flatp.generation)
        plan = self._plans.get(key)
        if plan is None:
            self._plans = {}
            p = PackPlan(cdt, device)
            self._declare_images(p)
            p.finalize()
            plan = [p, -2]
            self._plans[key] = plan
        if plan[1] != self._weights_version:
            plan[0].run()
            plan[1] = self._weights_version
        return plan[0]

    def _tok(self, device):
        if self._token is None or self._token.device != device:
            self._token = torch.zeros(1, device=device, requires_grad=True)
        return self._token

    @staticmethod
    def _wgrad(dY, ldy, X, ldx, dW, n_out, n_in, rows, bias_grad=None):
        """dW[n_out, n_in] += dY^T @ X (atomic split-K TN GEMM); bias_grad += column sums of dY"""
        bf16 = dY.dtype == torch.bfloat16
        tiles = (((n_out + 255) // 256) * ((n_in + 127) // 128)) if bf16 else (((n_out + 63) // 64) * ((n_in + 63) // 64))
        nk = (rows + 63) // 64
        sk = max(1, min(max(1, nk // 4), max(1, 256 // tiles), 16))
        if bias_grad is not None:
            ops.colsum(dY, bias_grad, rows, n_out, ld=ldy)
        ops.gemm(dY, X, dW, n_out, n_in, rows, ldy, ldx, n_in, transA=True, transB=True, atomic=True, splitk=sk,
                 c_dtype=ops.F32)


# ================================================================================================ prediction network
class _LSTMDropout(nn.Module):  # common/parts/rnn.py:151 (parameter container: `lstm` is never called)
    def __init__(self, input_size, hidden_size, num_layers, dropout, forget_gate_bias, t_max, weights_init_scale,
                 hidden_hidden_bias_scale):
        super().__init__()
        self.lstm = nn.LSTM(input_size=input_size, hidden_size=hidden_size, num_layers=num_layers, dropout=dropout or 0.0)
        if t_max is not None:
            raise NotImplementedError("chrono initialisation (t_max)")
        if forget_gate_bias is not None:  # rnn.py:210-218
            for name, v in self.lstm.named_parameters():
                if "bias_ih" in name:
                    v.data[hidden_size: 2 * hidden_size].fill_(forget_gate_bias)
                if "bias_hh" in name:
                    v.data[hidden_size: 2 * hidden_size] *= float(hidden_hidden_bias_scale)
        self.dropout = nn.Dropout(dropout) if dropout else None
        for name, v in self.named_parameters():
            if "weight" in name or "bias" in name:
                v.data *= float(weights_init_scale)


class _DecoderFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, token, module, targets):
        g, saved = module._forward_impl(targets, save=True)
        ctx.module, ctx.saved = module, saved
        return g

    @staticmethod
    def backward(ctx, dg):
        ctx.module._backward_impl(ctx.saved, dg)
        ctx.saved = None
        return None, None, None


class RNNTDecoder(_ModuleBase):
    @property
    def input_types(self):
        return OrderedDict({"targets": NeuralType(("B", "T"), LabelsType()),
                            "target_length": NeuralType(tuple("B"), LengthsType()),
                            "states": NeuralType(("D", "B", "D"), ElementType(), optional=True)})

    @property
    def output_types(self):
        return OrderedDict({"outputs": NeuralType(("B", "D", "T"), EmbeddedTextType()),
                            "prednet_lengths": NeuralType(tuple("B"), LengthsType()),
                            "states": NeuralType(("D", "B", "D"), ElementType(), optional=True)})

    def __init__(self, prednet: Dict[str, Any], vocab_size: int, normalization_mode: Optional[str] = None,
                 random_state_sampling: bool = False, blank_as_pad: bool = True, compute_dtype: Optional[torch.dtype] = None):
        super().__init__()
        bad = []
        if normalization_mode is not None: bad.append(f"normalization_mode={normalization_mode}")
        if random_state_sampling: bad.append("random_state_sampling")
        if not blank_as_pad: bad.append("blank_as_pad=False")
        if prednet.get("rnn_hidden_size", -1) > 0: bad.append("rnn_hidden_size (LSTM projection)")
        if bad:
            raise NotImplementedError("MI355X RNNTDecoder does not implement: " + ", ".join(bad))
        self.pred_hidden = prednet["pred_hidden"]
        self.pred_rnn_layers = prednet["pred_rnn_layers"]
        self.vocab_size, self.blank_idx, self.blank_as_pad = vocab_size, vocab_size, blank_as_pad
        self.dropout = float(prednet.get("dropout", 0.0) or 0.0)
        H = self.pred_hidden
        if H % 4:
            raise ValueError("pred_hidden must be a multiple of 4 (8 for bf16 compute): 16-byte vector accesses")
        self.prediction = nn.ModuleDict({
            "embed": nn.Embedding(vocab_size + 1, H, padding_idx=self.blank_idx),  # rnnt.py:880-883
            "dec_rnn": _LSTMDropout(H, H, self.pred_rnn_layers, self.dropout, prednet.get("forget_gate_bias", 1.0),
                                    prednet.get("t_max", None), prednet.get("weights_init_scale", 1.0),
                                    prednet.get("hidden_hidden_bias_scale", 0.0)),
        })
        self._init_engine(compute_dtype)

    def _declare_images(self, p):
        lstm = self.prediction["dec_rnn"].lstm
        for l in range(self.pred_rnn_layers):
            wih, whh = getattr(lstm, f"weight_ih_l{l}").data, getattr(lstm, f"weight_hh_l{l}").data
            p.add_matrix(f"l{l}.wih", wih); p.add_matrix(f"l{l}.wiht", wih, True)
            p.add_matrix(f"l{l}.whh", whh); p.add_matrix(f"l{l}.whht", whh, True)

    @typecheck()
    def forward(self, targets, target_length, states=None):
        if states is not None:
            raise NotImplementedError("stateful prediction (decoding) is not on the training hot path")
        self._flatp.ensure(targets.device)
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            g = _DecoderFn.apply(self._tok(targets.device), self, targets)
        else:
            g = self._forward_impl(targets)[0]
        return g, target_length, None

    def _forward_impl(self, targets, save=False):
        dev = targets.device
        cdt = self._cdt()
        W = self._plan(cdt, dev)
        B, U = targets.shape
        U1, H, L = U + 1, self.pred_hidden, self.pred_rnn_layers
        lstm = self.prediction["dec_rnn"].lstm
        training = self.training
        if training:
            self._step_seed = (self._step_seed + 1) & 0x3FFFFFFF
        tg = targets.to(torch.int64).contiguous()
        x0 = torch.empty(U1 * B, H, dtype=cdt, device=dev)
        ops.embed_sos_fwd(tg, self.prediction["embed"].weight, x0, B, U, H, self.blank_idx)
        inp, layers = x0, []
        for l in range(L):
            b_ih, b_hh = getattr(lstm, f"bias_ih_l{l}"), getattr(lstm, f"bias_hh_l{l}")
            z = torch.empty(U1 * B, 4 * H, dtype=torch.float32, device=dev)
            ops.gemm(inp, W[f"l{l}.wih"], z, U1 * B, 4 * H, H, H, W.pitch(f"l{l}.wih"), 4 * H, bias=b_ih)
            c_all = torch.empty(U1 * B, H, dtype=torch.float32, device=dev)
            h_all = torch.empty(U1 * B, H, dtype=torch.float32, device=dev)
            h_lp = torch.empty(U1 * B, H, dtype=cdt, device=dev)
            for t in range(U1):
                zt = z[t * B:(t + 1) * B]
                if t > 0:  # z_t += h_{t-1} W_hh^T, in place
                    if self._fwd_splitk > 1:
                        # [B, H] x [H, 4H] with B = 32 rows: 4H / 128 output tiles, each a chain of H / 64 K-tiles -- split-K with
                        # atomic accumulation into the f32 pre-activations that are already there shortens the chain
                        ops.gemm(h_lp[(t - 1) * B:t * B], W[f"l{l}.whh"], zt, B, 4 * H, H, H, W.pitch(f"l{l}.whh"), 4 * H,
                                 atomic=True, splitk=self._fwd_splitk, c_dtype=ops.F32)
                    else:  # residual epilogue with aux_in == C
                        ops.gemm(h_lp[(t - 1) * B:t * B], W[f"l{l}.whh"], zt, B, 4 * H, H, H, W.pitch(f"l{l}.whh"), 4 * H,
                                 epi=ops.EPI_RESID, aux_in=zt)
                ops.lstm_cell_fwd(zt, b_hh, c_all[(t - 1) * B:t * B] if t > 0 else None, c_all[t * B:(t + 1) * B],
                                  h_all[t * B:(t + 1) * B], h_lp[t * B:(t + 1) * B], B, H)
            # nn.LSTM drops the outputs of every layer but the last; LSTMDropout drops the last one (rnn.py:219,233-236)
            d = ops.Dropout(self.dropout if training else 0.0, self._step_seed, 200 + l)
            if d.threshold:
                out = torch.empty(U1 * B, H, dtype=cdt, device=dev)
                ops.drop_scale_cast(h_all, out, U1 * B * H, 1.0, d)
                if l == L - 1:
                    out_f32 = torch.empty(U1 * B, H, dtype=torch.float32, device=dev)
                    ops.drop_scale_cast(h_all, out_f32, U1 * B * H, 1.0, d)
            else:
                out, out_f32 = h_lp, h_all
            layers.append((inp, z, c_all, h_lp, d))
            inp = out
        g = out_f32.view(U1, B, H).permute(1, 2, 0)  # [B, H, U+1] (a view: the joint re-lays it out once)
        return g, ((tg, layers, B, U, cdt) if save else None)

    _bptt_splitk = int(os.environ.get("MI355X_LSTM_BPTT_SPLITK", "8"))
    _fwd_splitk = int(os.environ.get("MI355X_LSTM_FWD_SPLITK", "1"))

    def _backward_impl(self, saved, dg):
        tg, layers, B, U, cdt = saved
        dev = dg.device
        W = self._plan(cdt, dev)
        U1, H, L = U + 1, self.pred_hidden, self.pred_rnn_layers
        lstm = self.prediction["dec_rnn"].lstm
        d_out = dg.permute(2, 0, 1).contiguous().view(U1 * B, H).to(torch.float32)  # time-major
        for l in range(L - 1, -1, -1):
            inp, z, c_all, h_lp, d = layers[l]
            dh_all = torch.empty(U1 * B, H, dtype=torch.float32, device=dev)
            ops.drop_scale_cast(d_out, dh_all, U1 * B * H, 1.0, d)
            dc = torch.zeros(B, H, dtype=torch.float32, device=dev)
            dz = torch.empty(U1 * B, 4 * H, dtype=cdt, device=dev)
            for t in range(U1 - 1, -1, -1):
                ops.lstm_cell_bwd(dh_all[t * B:(t + 1) * B], dc, z[t * B:(t + 1) * B], c_all[t * B:(t + 1) * B],
                                  c_all[(t - 1) * B:t * B] if t > 0 else None, dz[t * B:(t + 1) * B], B, H)
                if t > 0:  # dh_{t-1} += dz_t W_hh
                    prev = dh_all[(t - 1) * B:t * B]
                    # [B, 4H] x [4H, H] with B = 32 rows: H / 128 output tiles only -- split-K (atomic accumulation into the
                    # f32 gradient that is already there) spreads the 4H-long reduction over 8x as many workgroups
                    ops.gemm(dz[t * B:(t + 1) * B], W[f"l{l}.whht"], prev, B, H, 4 * H, 4 * H, W.pitch(f"l{l}.whht"), H,
                             atomic=True, splitk=self._bptt_splitk, c_dtype=ops.F32)
            w_ih, w_hh = getattr(lstm, f"weight_ih_l{l}"), getattr(lstm, f"weight_hh_l{l}")
            b_ih, b_hh = getattr(lstm, f"bias_ih_l{l}"), getattr(lstm, f"bias_hh_l{l}")
            self._wgrad(dz, 4 * H, inp, H, w_ih.grad, 4 * H, H, U1 * B, bias_grad=b_ih.grad)
            ops.colsum(dz, b_hh.grad, U1 * B, 4 * H)
            if U1 > 1:  # rows of step t pair with h_{t-1}
                self._wgrad(dz[B:], 4 * H, h_lp[:(U1 - 1) * B], H, w_hh.grad, 4 * H, H, (U1 - 1) * B)
            d_inp = torch.empty(U1 * B, H, dtype=torch.float32, device=dev)
            ops.gemm(dz, W[f"l{l}.wiht"], d_inp, U1 * B, H, 4 * H, 4 * H, W.pitch(f"l{l}.wiht"), H)
            d_out = d_inp
        ops.embed_sos_bwd(tg, d_out, self.prediction["embed"].weight.grad, B, U, H, self.blank_idx)
        if self.grad_ready_hook is not None:
            self.grad_ready_hook(0, self._flatp.flat.numel())


# ================================================================================================ joint network
class _JointFn(torch.autograd.Function):
    """un-fused joint: logits [B, T, U+1, V+1] out (any loss on top)"""

    @staticmethod
    def forward(ctx, enc, dec, token, module):
        logits, saved = module._joint_fwd(enc, dec)
        ctx.module, ctx.saved = module, saved
        return logits

    @staticmethod
    def backward(ctx, dlogits):
        denc, ddec = ctx.module._joint_bwd(ctx.saved, dlogits)
        ctx.saved = None
        return denc, ddec, None, None


class _FusedJointLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, enc, dec, token, module, enc_len, transcripts, t_len):
        loss, denc, ddec, gtmp = module._fused_fwd_bwd(enc, dec, enc_len, transcripts, t_len)
        ctx.module = module
        ctx.save_for_backward(denc, ddec, gtmp)
        return loss

    @staticmethod
    def backward(ctx, go):
        denc, ddec, gtmp = ctx.saved_tensors
        fp = ctx.module._flatp
        fp.grad.add_(gtmp * go)  # (a scalar times the joint's gradient buffer: plumbing, once per step)
        if ctx.module.grad_ready_hook is not None:
            ctx.module.grad_ready_hook(0, fp.flat.numel())
        return denc * go, ddec * go, None, None, None, None, None


class RNNTJoint(_ModuleBase):
    @property
    def input_types(self):
        return OrderedDict({
            "encoder_outputs": NeuralType(("B", "D", "T"), AcousticEncodedRepresentation()),
            "decoder_outputs": NeuralType(("B", "D", "T"), EmbeddedTextType()),
            "encoder_lengths": NeuralType(tuple("B"), LengthsType(), optional=True),
            "transcripts": NeuralType(("B", "T"), LabelsType(), optional=True),
            "transcript_lengths": NeuralType(tuple("B"), LengthsType(), optional=True),
            "compute_wer": NeuralType(optional=True),
        })

    @property
    def output_types(self):
        if not self._fuse_loss_wer:
            return OrderedDict({"outputs": NeuralType(("B", "T", "T", "D"), LogprobsType())})
        return OrderedDict({"loss": NeuralType(elements_type=LossType(), optional=True),
                            "wer": NeuralType(elements_type=ElementType(), optional=True),
                            "wer_numer": NeuralType(elements_type=ElementType(), optional=True),
                            "wer_denom": NeuralType(elements_type=ElementType(), optional=True)})

    def __init__(self, jointnet: Dict[str, Any], num_classes: int, num_extra_outputs: int = 0, vocabulary: Optional[List] = None,
                 log_softmax: Optional[bool] = None, preserve_memory: bool = False, fuse_loss_wer: bool = False,
                 fused_batch_size: Optional[int] = None, experimental_fuse_loss_wer: Any = None, masking_prob: float = -1.0,
                 compute_dtype: Optional[torch.dtype] = None):
        super().__init__()
        if experimental_fuse_loss_wer is not None:
            fuse_loss_wer = experimental_fuse_loss_wer
        if fuse_loss_wer and fused_batch_size is None:
            raise ValueError("If `fuse_loss_wer` is set, then `fused_batch_size` cannot be None!")  # rnnt.py:1432-1433
        bad = []
        if jointnet.get("activation", "relu").lower() != "relu": bad.append(f"activation={jointnet.get('activation')}")
        if masking_prob > 0.0: bad.append("masking_prob")
        if num_extra_outputs: bad.append("num_extra_outputs")
        if log_softmax: bad.append("log_softmax=True (the MI355X loss fuses the log-softmax, as the reference's GPU loss does)")
        if bad:
            raise NotImplementedError("MI355X RNNTJoint does not implement: " + ", ".join(bad))
        self.vocabulary = vocabulary
        self._vocab_size = num_classes
        self._num_extra_outputs = 0
        self._num_classes = num_classes + 1
        self._fuse_loss_wer, self._fused_batch_size = fuse_loss_wer, fused_batch_size
        # fused joint + loss: cut every sub-batch to its own longest encoder / target length, as the reference does
        # (rnnt.py:1559-1600); costs one host read of the lengths per step.  MI355X_RNNT_TRUNCATE=0 keeps the padded grid.
        self.truncate_sub_batches = os.environ.get("MI355X_RNNT_TRUNCATE", "1") != "0"
        self._loss, self._wer = None, None
        self.log_softmax, self.preserve_memory, self.masking_prob = log_softmax, preserve_memory, masking_prob
        self.encoder_hidden, self.pred_hidden = jointnet["encoder_hidden"], jointnet["pred_hidden"]
        self.joint_hidden, self.activation = jointnet["joint_hidden"], "relu"
        self.dropout = float(jointnet.get("dropout", 0.0) or 0.0)
        if self.joint_hidden % 4 or self.encoder_hidden % 4 or self.pred_hidden % 4:
            raise ValueError("encoder_hidden, pred_hidden and joint_hidden must be multiples of 4 (8 for bf16 compute)")
        self.pred = nn.Linear(self.pred_hidden, self.joint_hidden)
        self.enc = nn.Linear(self.encoder_hidden, self.joint_hidden)
        layers = [nn.ReLU(inplace=True)] + ([nn.Dropout(p=self.dropout)] if self.dropout else []) + \
                 [nn.Linear(self.joint_hidden, self._num_classes)]  # rnnt.py:1790-1795: the Linear's index depends on dropout
        self.joint_net = nn.Sequential(*layers)
        self.temperature = 1.0
        self._init_engine(compute_dtype)

    # -------------------------------------------------------------------------------------------- reference API surface
    @property
    def num_classes_with_blank(self):
        return self._num_classes

    @property
    def fuse_loss_wer(self):
        return self._fuse_loss_wer

    def set_fuse_loss_wer(self, fuse_loss_wer, loss=None, metric=None):
        self._fuse_loss_wer = fuse_loss_wer
        self._loss, self._wer = loss, metric

    def set_loss(self, loss):
        if not self._fuse_loss_wer:
            raise ValueError("Attempting to set loss module even though `fuse_loss_wer` is not set!")  # rnnt.py:1866-1870
        self._loss = loss

    def set_wer(self, wer):
        if not self._fuse_loss_wer:
            raise ValueError("Attempting to set WER module even though `fuse_loss_wer` is not set!")
        self._wer = wer

    @property
    def loss(self):
        return self._loss

    @property
    def fused_batch_size(self):
        return self._fused_batch_size

    def set_fused_batch_size(self, fused_batch_size):
        self._fused_batch_size = fused_batch_size

    def _declare_images(self, p):
        out = self.joint_net[-1]
        p.add_matrix("enc.w", self.enc.weight.data); p.add_matrix("enc.wt", self.enc.weight.data, True)
        p.add_matrix("pred.w", self.pred.weight.data); p.add_matrix("pred.wt", self.pred.weight.data, True)
        p.add_matrix("out.w", out.weight.data); p.add_matrix("out.wt", out.weight.data, True)

    # -------------------------------------------------------------------------------------------- forward (typed)
    @typecheck()
    def forward(self, encoder_outputs, decoder_outputs, encoder_lengths=None, transcripts=None, transcript_lengths=None,
                compute_wer: bool = False):
        self._flatp.ensure(encoder_outputs.device)
        if decoder_outputs is None and not (self._fuse_loss_wer and compute_wer):
            raise ValueError("decoder_outputs can only be None for fused step!")   # rnnt.py:1485-1486
        if not self._fuse_loss_wer:
            if torch.is_grad_enabled() and (encoder_outputs.requires_grad or decoder_outputs.requires_grad or
                                            any(p.requires_grad for p in self.parameters())):
                return _JointFn.apply(encoder_outputs, decoder_outputs, self._tok(encoder_outputs.device), self)
            return self._joint_fwd(encoder_outputs, decoder_outputs)[0]
        if self._loss is None:
            raise ValueError("`fuse_loss_wer` flag is set, but `loss` and `wer` modules were not provided! ")  # rnnt.py:1505
        if encoder_lengths is None or transcript_lengths is None:
            raise ValueError("`fuse_loss_wer` is set, therefore encoder and target lengths must be provided as well!")
        wer = wer_num = wer_denom = None
        if compute_wer:
            # rnnt.py:1592-1632: greedy decoding + WER per sub-batch of `fused_batch_size` utterances (the metric's state is
            # updated and read back per sub-batch; its numerator / denominator are summed, the ratio averaged) -- the decode of a
            # sub-batch is one launch here (modules/rnnt_decoding.py), its encoder rows are narrowed to the sub-batch's own length
            if self._wer is None:
                raise ValueError("`fuse_loss_wer` flag is set, but `loss` and `wer` modules were not provided! ")
            wers, wer_num, wer_denom = [], 0, 0
            B = encoder_outputs.shape[0]
            fb = self._fused_batch_size or B
            enc_det = encoder_outputs.detach()
            el_host = encoder_lengths.detach().cpu()
            for b0 in range(0, B, fb):
                b1 = min(B, b0 + fb)
                tmax = int(el_host[b0:b1].max())
                self._wer.update(predictions=enc_det[b0:b1, :, :tmax], predictions_lengths=encoder_lengths[b0:b1],
                                 targets=transcripts[b0:b1].detach(), targets_lengths=transcript_lengths[b0:b1])
                w, n_, d_ = self._wer.compute()
                self._wer.reset()
                wers.append(w); wer_num += n_; wer_denom += d_
            wer = sum(wers) / len(wers)
        if decoder_outputs is None:   # WER-only pass (rnnt.py:1500-1503: the loss is skipped without decoder outputs)
            return None, wer, wer_num, wer_denom
        if not torch.is_grad_enabled():  # validation: the loss only, no backward GEMMs
            loss = self._fused_fwd_bwd(encoder_outputs, decoder_outputs, encoder_lengths, transcripts, transcript_lengths,
                                       need_grad=False)[0]
            return loss, wer, wer_num, wer_denom
        loss = _FusedJointLossFn.apply(encoder_outputs, decoder_outputs, self._tok(encoder_outputs.device), self,
                                       encoder_lengths, transcripts, transcript_lengths)
        return loss, wer, wer_num, wer_denom

    def joint(self, f, g):
        """rnnt_abstract.AbstractRNNTJoint.joint: f [B,T,D], g [B,U+1,H] -> [B,T,U+1,V+1]"""
        return self.forward(encoder_outputs=f.transpose(1, 2), decoder_outputs=g.transpose(1, 2)) if not self._fuse_loss_wer \
            else _JointFn.apply(f.transpose(1, 2), g.transpose(1, 2), self._tok(f.device), self)

    # -------------------------------------------------------------------------------------------- shared pieces
    def _project(self, enc, dec, W, cdt):
        """encoder [B,D,T] and prediction [B,H,U1] outputs -> operand copies and their projections f [B*T,J], g [B*U1,J]"""
        B, D, T = enc.shape
        U1, H, J = dec.shape[2], self.pred_hidden, self.joint_hidden
        dev = enc.device
        xe32 = enc.transpose(1, 2).contiguous().view(B * T, D)   # no copy when enc is the encoder's [B,T,d] view
        xd32 = dec.transpose(1, 2).contiguous().view(B * U1, H)
        xe = torch.empty(B * T, D, dtype=cdt, device=dev); xd = torch.empty(B * U1, H, dtype=cdt, device=dev)
        ops.drop_scale_cast(xe32.float(), xe, B * T * D, 1.0)
        ops.drop_scale_cast(xd32.float(), xd, B * U1 * H, 1.0)
        f = torch.empty(B * T, J, dtype=cdt, device=dev); g = torch.empty(B * U1, J, dtype=cdt, device=dev)
        ops.gemm(xe, W["enc.w"], f, B * T, J, D, D, W.pitch("enc.w"), J, bias=self.enc.bias)
        ops.gemm(xd, W["pred.w"], g, B * U1, J, H, H, W.pitch("pred.w"), J, bias=self.pred.bias)
        return xe, xd, f, g

    def _drop(self, sub):
        if not self.training or not self.dropout:
            return ops.NO_DROP
        return ops.Dropout(self.dropout, self._step_seed, 300 + sub)

    def _sub_fwd(self, f, g, b0, nb, T, U1, W, cdt, drop, ld=None):
        J, V1 = self.joint_hidden, self._num_classes
        dev = f.device
        n = nb * T * U1
        h = torch.empty(n, J, dtype=cdt, device=dev)
        ops.joint_combine_fwd(f[b0 * T:], g[b0 * U1:], h, nb, T, U1, J, drop)
        # `ld` = row pitch of the logits: V+1 (dense, the module's output contract) or, inside the fused joint + loss path,
        # roundup8(V+1) so that the GEMM stores 32-byte vectors although V+1 = 1025 is odd
        ld = ld or V1
        logits = torch.empty(nb * T * U1, ld, dtype=torch.float32, device=dev) if ld != V1 else torch.empty(
            nb, T, U1, V1, dtype=torch.float32, device=dev)
        out = self.joint_net[-1]
        ops.gemm(h, W["out.w"], logits, n, V1, J, J, W.pitch("out.w"), ld, bias=out.bias,
                 alpha=1.0 / self.temperature if self.temperature != 1.0 else 1.0)
        return h, logits

    def _sub_bwd(self, dlogits, h, b0, nb, T, U1, W, cdt, drop, df, dg32, gW, gb, dlog=None, df_rows=None, dg_pitch=None):
        """dlogits f32 [nb,T,U1,V1] (or `dlog`: the same gradient already as the pitched GEMM operand [n, roundup8(V1)], scaled
        by 1/temperature): output-layer gradients into (gW, gb), dpre reductions into df (rows b0*T..) and dg32.
        `df_rows` / `dg_pitch` (truncated sub-batches): the dense [nb*T, J] destination of the encoder-side reduction, and the
        number of prediction frames per utterance in dg32 when it is larger than this call's U1"""
        J, V1 = self.joint_hidden, self._num_classes
        V1p = _pad8(V1)
        dev = h.device
        n = nb * T * U1
        if dlog is None:
            dlog = torch.empty(n, V1p, dtype=cdt, device=dev)
            ops.cast_rows(dlogits, V1, dlog, V1p, n, V1, V1p, 1.0 / self.temperature if self.temperature != 1.0 else 1.0)
        self._wgrad(dlog, V1p, h, J, gW, V1, J, n, bias_grad=gb)
        dh = torch.empty(n, J, dtype=cdt, device=dev)
        ops.gemm(dlog, W["out.wt"], dh, n, J, V1p, V1p, W.pitch("out.wt"), J)
        ops.joint_combine_bwd(dh, h, df[b0 * T:] if df_rows is None else df_rows, nb, T, U1, J, drop.scale)
        gp = dg_pitch or U1
        for i in range(nb):  # dg[b,u,:] = sum_t dpre[b,t,u,:]: a column sum over the [T, U1*J] slab of each utterance
            ops.colsum(dh[i * T * U1:], dg32[(b0 + i) * gp:], T, U1 * J)

    def _proj_bwd(self, xe, xd, df, dg32, W, cdt, B, T, U1, gWe, gbe, gWp, gbp):
        D, H, J = self.encoder_hidden, self.pred_hidden, self.joint_hidden
        dev = xe.device
        dg = torch.empty(B * U1, J, dtype=cdt, device=dev)
        ops.drop_scale_cast(dg32, dg, B * U1 * J, 1.0)
        self._wgrad(df, J, xe, D, gWe, J, D, B * T, bias_grad=gbe)
        self._wgrad(dg, J, xd, H, gWp, J, H, B * U1, bias_grad=gbp)
        dxe = torch.empty(B, T, D, dtype=torch.float32, device=dev)
        dxd = torch.empty(B, U1, H, dtype=torch.float32, device=dev)
        ops.gemm(df, W["enc.wt"], dxe, B * T, D, J, J, W.pitch("enc.wt"), D)
        ops.gemm(dg, W["pred.wt"], dxd, B * U1, H, J, J, W.pitch("pred.wt"), H)
        return dxe.transpose(1, 2), dxd.transpose(1, 2)

    # -------------------------------------------------------------------------------------------- un-fused joint
    def _joint_fwd(self, enc, dec):
        cdt = self._cdt()
        W = self._plan(cdt, enc.device)
        B, D, T = enc.shape
        U1 = dec.shape[2]
        if self.training:
            self._step_seed = (self._step_seed + 1) & 0x3FFFFFFF
        xe, xd, f, g = self._project(enc, dec, W, cdt)
        drop = self._drop(0)
        h, logits = self._sub_fwd(f, g, 0, B, T, U1, W, cdt, drop)
        return logits, (xe, xd, h, drop, B, T, U1, cdt)

    def _joint_bwd(self, saved, dlogits):
        xe, xd, h, drop, B, T, U1, cdt = saved
        dev = xe.device
        W = self._plan(cdt, dev)
        J = self.joint_hidden
        out = self.joint_net[-1]
        df = torch.empty(B * T, J, dtype=cdt, device=dev)
        dg32 = torch.zeros(B * U1, J, dtype=torch.float32, device=dev)
        self._sub_bwd(dlogits.contiguous().float(), h, 0, B, T, U1, W, cdt, drop, df, dg32, out.weight.grad, out.bias.grad)
        denc, ddec = self._proj_bwd(xe, xd, df, dg32, W, cdt, B, T, U1, self.enc.weight.grad, self.enc.bias.grad,
                                    self.pred.weight.grad, self.pred.bias.grad)
        if self.grad_ready_hook is not None:
            self.grad_ready_hook(0, self._flatp.flat.numel())
        return denc, ddec

    # -------------------------------------------------------------------------------------------- fused joint + loss
    def _fused_fwd_bwd(self, enc, dec, enc_len, transcripts, t_len, need_grad=True):
        """rnnt.py:1518-1640: sub-batches of `fused_batch_size` utterances; per sub-batch joint -> loss -> joint backward.
        Returns (reduced loss, d loss / d enc, d loss / d dec, joint parameter gradients as a flat buffer) for an upstream
        gradient of 1 -- `_FusedJointLossFn.backward` scales them."""
        loss_mod = self._loss
        red = getattr(loss_mod, "reduction", "mean_batch")
        if red not in ("mean_batch", "sum"):
            raise NotImplementedError(f"fused joint: loss reduction '{red}' (implemented: mean_batch, sum)")
        cdt = self._cdt()
        dev = enc.device
        W = self._plan(cdt, dev)
        B, D, T = enc.shape
        U1, J = dec.shape[2], self.joint_hidden
        if self.training:
            self._step_seed = (self._step_seed + 1) & 0x3FFFFFFF
        fp = self._flatp
        gtmp = torch.zeros_like(fp.flat) if need_grad else None

        def gview(p):  # the slot of parameter p inside the temporary gradient buffer
            off = (p.data_ptr() - fp.flat.data_ptr()) // 4
            return gtmp[off: off + p.numel()].view(p.shape)

        out = self.joint_net[-1]
        xe, xd, f, g = self._project(enc, dec, W, cdt)
        df = torch.empty(B * T, J, dtype=cdt, device=dev) if need_grad else None
        dg32 = torch.zeros(B * U1, J, dtype=torch.float32, device=dev) if need_grad else None
        labels = transcripts.to(torch.int64).contiguous()
        el, tl = enc_len.to(torch.int64).contiguous(), t_len.to(torch.int64).contiguous()
        scale = 1.0 / B if red == "mean_batch" else 1.0
        costs = torch.empty(B, dtype=torch.float32, device=dev)
        fbs = int(self._fused_batch_size)
        # rnnt.py:1559-1600: every sub-batch is cut to ITS longest encoder / target length before joint and loss (cells beyond an
        # utterance's own lengths carry no loss and no gradient, so the result is the same; the joint GEMMs and the lattice of a
        # ragged batch shrink by the padding).  The cut needs the lengths on the host: one read per step, as in the reference.
        cuts = None
        if self.truncate_sub_batches:
            n_sub = (B + fbs - 1) // fbs
            pad = n_sub * fbs - B
            lens2 = torch.stack([torch.nn.functional.pad(el, (0, pad)).view(n_sub, fbs).amax(1),
                                 torch.nn.functional.pad(tl, (0, pad)).view(n_sub, fbs).amax(1)]).tolist()
            cuts = [(max(1, min(T, int(a))), max(1, min(U1, int(b) + 1))) for a, b in zip(*lens2)]
            if all(c == (T, U1) for c in cuts):
                cuts = None  # nothing to cut (fixed-length batches): the dense path below, no copies
            elif need_grad:
                df.zero_()   # rows beyond a sub-batch's cut are not written
        f3, g3 = f.view(B, T, J), g.view(B, U1, J)
        for si, b0 in enumerate(range(0, B, fbs)):
            nb = min(fbs, B - b0)
            drop = self._drop(si)
            fe = float(getattr(loss_mod, "fastemit_lambda", 0.0) or 0.0)
            cl = float(getattr(loss_mod, "clamp", 0.0) or 0.0)
            Ts, Us = cuts[si] if cuts is not None else (T, U1)
            cut = (Ts, Us) != (T, U1)
            if cut:  # dense copies of the sub-batch's [nb, Ts, J] / [nb, Us, J] corners (small next to the [nb, Ts, Us, V+1] logits)
                fs, gs = f3[b0:b0 + nb, :Ts].contiguous().view(nb * Ts, J), g3[b0:b0 + nb, :Us].contiguous().view(nb * Us, J)
                lab = labels[b0:b0 + nb, :max(Us - 1, 1)].contiguous() if labels.shape[1] else labels[b0:b0 + nb]
                fb0 = 0
                dfs = torch.empty(nb * Ts, J, dtype=cdt, device=dev) if need_grad else None
            else:
                fs, gs, lab, fb0, dfs = f, g, labels[b0:b0 + nb], b0, None
            sub_bwd_kw = dict(df_rows=dfs, dg_pitch=U1) if cut else {}
            if not need_grad:
                h, logits = self._sub_fwd(fs, gs, fb0, nb, Ts, Us, W, cdt, drop)
                costs[b0:b0 + nb] = ops.rnnt_loss(logits, lab, el[b0:b0 + nb], tl[b0:b0 + nb], self._vocab_size,
                                                  fastemit_lambda=fe, clamp=cl, grad_scale=scale)
                del h, logits
                continue
            if cdt == torch.bfloat16:
                # logits with row pitch roundup8(V+1) (vector stores from the GEMM); the loss kernel writes the logit gradient
                # directly as the bf16 operand of the backward GEMMs: the f32 gradient tensor and its cast pass do not exist
                V1 = self._num_classes
                V1p = _pad8(V1)
                h, logits = self._sub_fwd(fs, gs, fb0, nb, Ts, Us, W, cdt, drop, ld=V1p)
                dlog = torch.empty(nb * Ts * Us, V1p, dtype=cdt, device=dev)
                c = ops.rnnt_loss_pitched(logits, V1p, nb, Ts, Us, V1, lab, el[b0:b0 + nb], tl[b0:b0 + nb],
                                          self._vocab_size, dlog, V1p, fastemit_lambda=fe, clamp=cl,
                                          grad_scale=scale / self.temperature if self.temperature != 1.0 else scale)
                costs[b0:b0 + nb] = c
                self._sub_bwd(None, h, b0, nb, Ts, Us, W, cdt, drop, df, dg32, gview(out.weight), gview(out.bias), dlog=dlog,
                              **sub_bwd_kw)
            else:
                h, logits = self._sub_fwd(fs, gs, fb0, nb, Ts, Us, W, cdt, drop)
                grads = torch.empty_like(logits)
                c = ops.rnnt_loss(logits, lab, el[b0:b0 + nb], tl[b0:b0 + nb], self._vocab_size, grads=grads,
                                  fastemit_lambda=fe, clamp=cl, grad_scale=scale)
                costs[b0:b0 + nb] = c
                self._sub_bwd(grads, h, b0, nb, Ts, Us, W, cdt, drop, df, dg32, gview(out.weight), gview(out.bias), **sub_bwd_kw)
                del grads
            if cut:
                df.view(B, T, J)[b0:b0 + nb, :Ts].copy_(dfs.view(nb, Ts, J))
            del h, logits
        loss = costs.sum() * scale
        if not need_grad:
            return loss, None, None, None
        denc, ddec = self._proj_bwd(xe, xd, df, dg32, W, cdt, B, T, U1, gview(self.enc.weight), gview(self.enc.bias),
                                    gview(self.pred.weight), gview(self.pred.bias))
        return loss, denc.contiguous(), ddec.contiguous(), gtmp
