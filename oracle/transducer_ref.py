"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the transducer head of FastConformer-Transducer,
plain torch on state-dicts with the reference's keys.  Prepared ahead of the HIP path (the loss kernels exist: csrc/rnnt.hip).

Follows `nemo/collections/asr/modules/rnnt.py`:
  * `RNNTDecoder.predict` (:552-760): embedding of the targets (`blank_as_pad`: the blank id = vocab_size is the padding
    row, all zeros), a zero start-of-sequence frame prepended (`add_sos`), LSTM stack (`common/parts/rnn.py` LSTMDropout:
    torch.nn.LSTM gate order i, f, g, o; h_t = o * tanh(c_t)), output transposed to [B, H, U+1];
  * `RNNTJoint.joint_after_projection` (:1280-1660): f = enc(encoder^T) [B,T,1,J], g = pred(decoder^T) [B,1,U+1,J],
    ReLU(f + g) -> Linear(J -> V+1): the LOGITS [B,T,U+1,V+1] (no log-softmax on the GPU path: the loss fuses it).
The loss on top is oracle/rnnt_ref.py.  Pinned against the reference classes (and the reference's pure-torch loss,
losses/rnnt_pytorch.py) by tests/golden/ref_transducer_tiny.npz."""
from __future__ import annotations

from typing import Dict

import torch
import torch.nn.functional as F
from torch import Tensor


def lstm_layer(x: Tensor, w_ih: Tensor, w_hh: Tensor, b_ih: Tensor, b_hh: Tensor) -> Tensor:
    """x [U, B, I] -> [U, B, H], zero initial state; written out gate by gate (an independent restatement of torch.nn.LSTM)"""
    U, B, _ = x.shape
    H = w_hh.shape[1]
    h = x.new_zeros(B, H)
    c = x.new_zeros(B, H)
    out = []
    for t in range(U):
        z = F.linear(x[t], w_ih, b_ih) + F.linear(h, w_hh, b_hh)
        i, f, g, o = z[:, :H], z[:, H:2 * H], z[:, 2 * H:3 * H], z[:, 3 * H:]
        c = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(g)
        h = torch.sigmoid(o) * torch.tanh(c)
        out.append(h)
    return torch.stack(out)


def prediction_network(P: Dict[str, Tensor], targets: Tensor, pfx: str = "prediction.") -> Tensor:
    """targets i64 [B, U] (padded) -> g [B, H, U+1]"""
    emb = P[pfx + "embed.weight"]
    y = emb[targets]                                             # [B, U, H]; the padding row (blank id) is zero
    B, U, H = y.shape
    y = torch.cat([y.new_zeros(B, 1, H), y], dim=1).transpose(0, 1)  # SOS frame, then time-major for the LSTM
    layer = 0
    while f"{pfx}dec_rnn.lstm.weight_ih_l{layer}" in P:
        q = f"{pfx}dec_rnn.lstm."
        y = lstm_layer(y, P[f"{q}weight_ih_l{layer}"], P[f"{q}weight_hh_l{layer}"], P[f"{q}bias_ih_l{layer}"],
                       P[f"{q}bias_hh_l{layer}"])
        layer += 1
    return y.transpose(0, 1).transpose(1, 2)


def joint_network(P: Dict[str, Tensor], enc: Tensor, dec: Tensor) -> Tensor:
    """enc [B, D, T], dec [B, H, U+1] -> logits [B, T, U+1, V+1]"""
    f = F.linear(enc.transpose(1, 2), P["enc.weight"], P["enc.bias"]).unsqueeze(2)
    g = F.linear(dec.transpose(1, 2), P["pred.weight"], P["pred.bias"]).unsqueeze(1)
    return F.linear(torch.relu(f + g), P["joint_net.1.weight"], P["joint_net.1.bias"])
