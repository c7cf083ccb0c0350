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

from . import conformer_ref as R


class _Emu:
    """`emulate_bf16` carrier for conformer_ref's rounding helpers (_q: value + gradient, _qw: weight image, _qg: gradient only)"""

    def __init__(self, on):
        self.emulate_bf16 = bool(on)


def lstm_layer(x: Tensor, w_ih: Tensor, w_hh: Tensor, b_ih: Tensor, b_hh: Tensor, emulate_bf16: bool = False) -> Tensor:
    """x [U, B, I] -> [U, B, H], zero initial state; written out gate by gate (an independent restatement of torch.nn.LSTM).
    emulate_bf16: the HIP bf16 path's storage points -- both weight images, the recurrent operand h_{t-1} (the f32 h stays the
    layer's output), and the gate pre-activations' GRADIENT (dz is the bf16 operand of the BPTT GEMMs); z, c, h are f32."""
    e = _Emu(emulate_bf16)
    w_ih, w_hh = R._qw(w_ih, e), R._qw(w_hh, e)
    U, B, _ = x.shape
    H = w_hh.shape[1]
    h = x.new_zeros(B, H)
    c = x.new_zeros(B, H)
    out = []
    for t in range(U):
        z = R._qg(F.linear(x[t], w_ih, b_ih) + F.linear(R._q(h, e) if t > 0 else h, w_hh, b_hh), e)
        i, f, g, o = z[:, :H], z[:, H:2 * H], z[:, 2 * H:3 * H], z[:, 3 * H:]
        c = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(g)
        h = torch.sigmoid(o) * torch.tanh(c)
        out.append(h)
    return torch.stack(out)


def prediction_network(P: Dict[str, Tensor], targets: Tensor, pfx: str = "prediction.", emulate_bf16: bool = False) -> Tensor:
    """targets i64 [B, U] (padded) -> g [B, H, U+1]"""
    e = _Emu(emulate_bf16)
    emb = P[pfx + "embed.weight"]
    y = R._q(emb[targets], e)                                    # [B, U, H]; the padding row (blank id) is zero
    B, U, H = y.shape
    y = torch.cat([y.new_zeros(B, 1, H), y], dim=1).transpose(0, 1)  # SOS frame, then time-major for the LSTM
    layer = 0
    while f"{pfx}dec_rnn.lstm.weight_ih_l{layer}" in P:
        q = f"{pfx}dec_rnn.lstm."
        if layer > 0:
            y = R._q(y, e)  # (a further layer reads the bf16 copy of the previous layer's output)
        y = lstm_layer(y, P[f"{q}weight_ih_l{layer}"], P[f"{q}weight_hh_l{layer}"], P[f"{q}bias_ih_l{layer}"],
                       P[f"{q}bias_hh_l{layer}"], emulate_bf16)
        layer += 1
    return y.transpose(0, 1).transpose(1, 2)


def joint_network(P: Dict[str, Tensor], enc: Tensor, dec: Tensor, emulate_bf16: bool = False) -> Tensor:
    """enc [B, D, T], dec [B, H, U+1] -> logits [B, T, U+1, V+1].
    emulate_bf16: operand copies of both inputs, the three weight images, the projections f / g and ReLU(f + g) are bf16; the
    logits stay f32 (the loss reads them) while their gradient is written as a bf16 GEMM operand."""
    e = _Emu(emulate_bf16)
    out = [k[:-len("weight")] for k in P if k.startswith("joint_net.") and k.endswith(".weight")][0]  # index 1, or 2 with dropout
    f = R._q(F.linear(R._q(enc.transpose(1, 2), e), R._qw(P["enc.weight"], e), P["enc.bias"]), e).unsqueeze(2)
    g = R._q(F.linear(R._q(dec.transpose(1, 2), e), R._qw(P["pred.weight"], e), P["pred.bias"]), e).unsqueeze(1)
    return R._qg(F.linear(R._q(torch.relu(f + g), e), R._qw(P[out + "weight"], e), P[out + "bias"]), e)


def greedy_decode(Pd: Dict[str, Tensor], Pj: Dict[str, Tensor], enc: Tensor, enc_len: Tensor, blank: int, max_symbols: int = 10):
    """Greedy transducer search, utterance by utterance (the textbook form of parts/submodules/rnnt_greedy_decoding.py: the batched
    frame loop :804-990 masks finished samples and restores their state, which per utterance is exactly this):
        state = 0, last = blank (zero embedding); for every frame t < T_b: up to `max_symbols` times
            g = LSTM(emb[last], state) -> logits = out(relu(enc_proj[t] + pred_proj(g))); k = argmax
            k == blank: next frame;  else emit (k, t), commit the LSTM state, last = k
    Pd / Pj: state-dicts of RNNTDecoder / RNNTJoint with the reference's keys; enc [B, D, T].  -> list of (tokens, frame indices)."""
    emb = Pd["prediction.embed.weight"]
    q = "prediction.dec_rnn.lstm."
    assert q + "weight_ih_l1" not in Pd, "one LSTM layer (the recipe's pred_rnn_layers: 1)"
    w_ih, w_hh, b_ih, b_hh = (Pd[q + n + "_l0"] for n in ("weight_ih", "weight_hh", "bias_ih", "bias_hh"))
    out = [k[:-len("weight")] for k in Pj if k.startswith("joint_net.") and k.endswith(".weight")][0]
    f_all = F.linear(enc.transpose(1, 2), Pj["enc.weight"], Pj["enc.bias"])   # [B, T, J]
    H = w_hh.shape[1]
    hyps = []
    for b in range(enc.shape[0]):
        h, c = torch.zeros(H), torch.zeros(H)
        last, toks, times = blank, [], []

        def pred(last, h, c):
            x = emb[last]  # the padding row (blank) is zero
            z = F.linear(x, w_ih, b_ih) + F.linear(h, w_hh, b_hh)
            i, f, g, o = z[:H], z[H:2 * H], z[2 * H:3 * H], z[3 * H:]
            c2 = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(g)
            h2 = torch.sigmoid(o) * torch.tanh(c2)
            return F.linear(h2, Pj["pred.weight"], Pj["pred.bias"]), h2, c2

        gp, hn, cn = pred(last, h, c)
        for t in range(int(enc_len[b])):
            for _ in range(max_symbols):
                logits = F.linear(torch.relu(f_all[b, t] + gp), Pj[out + "weight"], Pj[out + "bias"])
                k = int(torch.argmax(logits))
                if k == blank:
                    break
                toks.append(k); times.append(t)
                h, c, last = hn, cn, k
                gp, hn, cn = pred(last, h, c)
        hyps.append((toks, times))
    return hyps


def forced_decode_margins(Pd: Dict[str, Tensor], Pj: Dict[str, Tensor], enc: Tensor, enc_len: Tensor, blank: int, max_symbols: int,
                          hyps, f_all: Tensor = None):
    """Walk the greedy search of `greedy_decode` ALONG given hypotheses (list of (tokens, frame indices), e.g. what the device search
    returned) and report, for every decision, how far the followed label is below this restatement's own arg-max:
    -> list over utterances of lists of (frame, followed label, own arg-max, logit[arg-max] - logit[followed], max |logit|).
    A margin of 0 everywhere means the hypotheses ARE this search's; a positive margin is a decision the two sides took differently,
    and its size says whether that was a rounding-level near-tie or an error.  `f_all` (optional [B, T, J]): the encoder projection
    to use instead of enc @ W^T + b (a reduced-precision emulation passes its own rounding of it)."""
    emb = Pd["prediction.embed.weight"]
    q = "prediction.dec_rnn.lstm."
    w_ih, w_hh, b_ih, b_hh = (Pd[q + n + "_l0"] for n in ("weight_ih", "weight_hh", "bias_ih", "bias_hh"))
    out = [k[:-len("weight")] for k in Pj if k.startswith("joint_net.") and k.endswith(".weight")][0]
    if f_all is None:
        f_all = F.linear(enc.transpose(1, 2), Pj["enc.weight"], Pj["enc.bias"])
    H = w_hh.shape[1]
    report = []
    for b in range(enc.shape[0]):
        toks, times = list(hyps[b][0]), list(hyps[b][1])
        h, c = torch.zeros(H), torch.zeros(H)

        def pred(last, h, c):
            z = F.linear(emb[last], w_ih, b_ih) + F.linear(h, w_hh, b_hh)
            i, f, g, o = z[:H], z[H:2 * H], z[2 * H:3 * H], z[3 * H:]
            c2 = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(g)
            h2 = torch.sigmoid(o) * torch.tanh(c2)
            return F.linear(h2, Pj["pred.weight"], Pj["pred.bias"]), h2, c2

        gp, hn, cn = pred(blank, h, c)
        pos, rows = 0, []
        for t in range(int(enc_len[b])):
            for sym in range(max_symbols):
                logits = F.linear(torch.relu(f_all[b, t] + gp), Pj[out + "weight"], Pj[out + "bias"])
                own = int(torch.argmax(logits))
                follow = toks[pos] if (pos < len(toks) and times[pos] == t) else blank
                rows.append((t, follow, own, float(logits[own] - logits[follow]), float(logits.abs().max())))
                if follow == blank:
                    break
                pos += 1
                h, c = hn, cn
                gp, hn, cn = pred(follow, h, c)
            else:
                # the frame used up max_symbols emissions: the hypothesis must not hold more labels on it
                assert not (pos < len(toks) and times[pos] == t), (b, t)
        assert pos == len(toks), (b, pos, len(toks))   # every label of the hypothesis was consumed in frame order
        report.append(rows)
    return report

