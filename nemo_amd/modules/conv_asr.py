"""Drop-in for `nemo.collections.asr.modules.ConvASRDecoder` (modules/conv_asr.py:407-510): Conv1d(d, V+1, k=1) +
log_softmax, same ctor kwargs, typed I/O, `vocabulary` / `num_classes_with_blank` / `_feat_in` / `temperature`
attributes and state-dict key `decoder_layers.0.{weight [V+1,d,1], bias}`; arithmetic = MFMA GEMM + wave-per-row
log-softmax kernels, one autograd node."""
from __future__ import annotations

from collections import OrderedDict

import torch
from torch import nn

from .. import ops
from ..core import AcousticEncodedRepresentation, LogprobsType, NeuralModule, NeuralType, typecheck
from ..flat import FlatParams
from ..packing import PackPlan


def _pad8(n):
    return (n + 7) // 8 * 8


class _DecoderFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, enc, token, module):
        logp, saved = module._forward_impl(enc)
        ctx.module, ctx.saved = module, saved
        return logp

    @staticmethod
    def backward(ctx, dlogp):
        denc = ctx.module._backward_impl(ctx.saved, dlogp)
        ctx.saved = None
        return denc, None, None


class ConvASRDecoder(NeuralModule):
    @property
    def input_types(self):
        return OrderedDict({"encoder_output": NeuralType(("B", "D", "T"), AcousticEncodedRepresentation())})

    @property
    def output_types(self):
        return OrderedDict({"logprobs": NeuralType(("B", "T", "D"), LogprobsType())})

    def __init__(self, feat_in, num_classes, init_mode="xavier_uniform", vocabulary=None, add_blank=True,
                 compute_dtype=None):
        super().__init__()
        if vocabulary is None and num_classes < 0:
            raise ValueError("Neither of the vocabulary and num_classes are set! At least one of them need to be set.")
        if num_classes <= 0:
            num_classes = len(vocabulary)
        if vocabulary is not None:
            if num_classes != len(vocabulary):
                raise ValueError(f"If vocabulary is specified, it's length should be equal to the num_classes. "
                                 f"Instead got: num_classes={num_classes} and len(vocabulary)={len(vocabulary)}")
            self.__vocabulary = vocabulary
        else:
            self.__vocabulary = None
        self._feat_in = feat_in
        self._num_classes = num_classes + 1 if add_blank else num_classes
        self.decoder_layers = nn.Sequential(nn.Conv1d(self._feat_in, self._num_classes, kernel_size=1, bias=True))
        if init_mode == "xavier_uniform":
            nn.init.xavier_uniform_(self.decoder_layers[0].weight, gain=1.0)  # init_weights(), conv_asr.py:448
        elif init_mode == "xavier_normal":
            nn.init.xavier_normal_(self.decoder_layers[0].weight, gain=1.0)
        elif init_mode == "kaiming_uniform":
            nn.init.kaiming_uniform_(self.decoder_layers[0].weight, nonlinearity="relu")
        elif init_mode == "kaiming_normal":
            nn.init.kaiming_normal_(self.decoder_layers[0].weight, nonlinearity="relu")
        else:
            raise ValueError(f"Unknown Initialization mode: {init_mode}")
        self.temperature = 1.0
        self.compute_dtype = compute_dtype
        self.grad_ready_hook = None
        self._flatp = FlatParams(self)
        self._plans = {}
        self._weights_version = -1
        self._token = None

    @property
    def vocabulary(self):
        return self.__vocabulary

    @property
    def num_classes_with_blank(self):
        return self._num_classes

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
        if self.compute_dtype is not None:
            return self.compute_dtype
        if torch.is_autocast_enabled() and torch.get_autocast_gpu_dtype() == torch.bfloat16:
            return torch.bfloat16
        return torch.float32

    def _plan(self, cdt, device):
        key = (cdt, str(device), self._flatp.generation)
        plan = self._plans.get(key)
        if plan is None:
            self._plans = {}
            p = PackPlan(cdt, device)
            w = self.decoder_layers[0].weight.data
            p.add_matrix("dec.w", w)
            p.add_matrix("dec.wt", w, transpose=True)
            p.finalize()
            plan = [p, -2]
            self._plans[key] = plan
        if plan[1] != self._weights_version:
            plan[0].run()
            plan[1] = self._weights_version
        return plan[0]

    @typecheck()
    def forward(self, encoder_output):
        self._flatp.ensure(encoder_output.device)
        if self._token is None or self._token.device != encoder_output.device:
            self._token = torch.zeros(1, device=encoder_output.device, requires_grad=True)
        if torch.is_grad_enabled() and (encoder_output.requires_grad or any(p.requires_grad for p in self.parameters())):
            return _DecoderFn.apply(encoder_output, self._token, self)
        return self._forward_impl(encoder_output)[0]

    def _forward_impl(self, enc):
        B, d, T = enc.shape
        dev = enc.device
        cdt = self._cdt()
        W = self._plan(cdt, dev)
        M, V1 = B * T, self._num_classes
        x = enc.transpose(1, 2).contiguous().view(M, d)  # no copy when enc is the encoder's [B,T,d] view
        # bf16 operand rows start on 16-byte boundaries: a d_model that is not a multiple of 8 (Squeezeformer-Medium: 324)
        # gets row pitch roundup8(d) with zero pad columns
        ldx = _pad8(d) if cdt == torch.bfloat16 else d
        if x.dtype != cdt or ldx != d:
            xc = torch.empty(M, ldx, dtype=cdt, device=dev)
            if ldx != d or (M * d) % 8:
                ops.cast_pitched(x.float() if x.dtype != torch.float32 else x, xc, M, d, ldx)
            else:
                ops.drop_scale_cast(x, xc, M * d, 1.0)
        else:
            xc = x
        logits = torch.empty(M, V1, dtype=torch.float32, device=dev)
        conv = self.decoder_layers[0]
        ops.gemm(xc, W["dec.w"], logits, M, V1, d, ldx, W.pitch("dec.w"), V1, bias=conv.bias,
                 alpha=1.0 / self.temperature if self.temperature != 1.0 else 1.0)
        logp = torch.empty(B, T, V1, dtype=torch.float32, device=dev)
        ops.log_softmax_fwd(logits, V1, logp, V1, M, V1)
        return logp, (xc, logp, B, T, d, cdt, enc.dtype)

    def _backward_impl(self, saved, dlogp):
        xc, logp, B, T, d, cdt, enc_dtype = saved
        dev = dlogp.device
        W = self._plan(cdt, dev)
        M, V1 = B * T, self._num_classes
        Vp = _pad8(V1)
        conv = self.decoder_layers[0]
        dlogits = torch.empty(M, Vp, dtype=cdt, device=dev)
        ops.log_softmax_bwd(dlogp.contiguous(), logp, V1, dlogits, Vp, M, V1,
                            1.0 / self.temperature if self.temperature != 1.0 else 1.0)
        ops.colsum(dlogits, conv.bias.grad, M, V1, ld=Vp)
        # d weight [V1, d] += dlogits^T @ x
        tiles = ((V1 + 255) // 256) * ((d + 127) // 128) if cdt == torch.bfloat16 else ((V1 + 63) // 64) * ((d + 63) // 64)
        nk = (M + 63) // 64
        ops.gemm(dlogits, xc, conv.weight.grad, V1, d, M, Vp, xc.shape[1], d, transA=True, transB=True, atomic=True,
                 splitk=max(1, min(max(1, nk // 4), 256 // tiles)), c_dtype=ops.F32)
        denc = torch.empty(B, T, d, dtype=torch.float32, device=dev)
        ops.gemm(dlogits, W["dec.wt"], denc, M, d, Vp, Vp, W.pitch("dec.wt"), d)
        if self.grad_ready_hook is not None:
            self.grad_ready_hook(0, self._flatp.flat.numel())
        return denc.transpose(1, 2).to(enc_dtype)

    def input_example(self, max_batch=1, max_dim=256):
        return (torch.randn(max_batch, self._feat_in, max_dim).to(next(self.parameters()).device),)
