"""conditioning of the conv.0 weight / bias gradient at BASELINE configs[0] (Small, B=2x10s): sum|terms| / |sum terms| per tap,
measured with the CPU oracle -- the evidence behind the looser tolerance of these two tensors in tests/test_baseline_configs_gpu.py"""
import sys; import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from oracle import conformer_ref as R
cfg = R.ConformerCfg.small(vocab=128, dropout=0.0, dropout_att=0.0, dropout_pre_encoder=0.0)
P = R.init_params(cfg, seed=0)
audio, alen, tok, tl = R.synthetic_batch(2, 10.0, vocab=128, seed=1234)
for k in R.trainable_keys(P): P[k].requires_grad_(True)
# capture grad wrt conv1 pre-activation by re-implementing the first conv with a hook
mel, mel_len = R.log_mel_features(audio, alen)
x = mel.transpose(1,2).unsqueeze(1)
w = P["encoder.pre_encode.conv.0.weight"]; b=P["encoder.pre_encode.conv.0.bias"]
orig = F.conv2d
store={}
def conv2d(inp, weight, bias=None, stride=1, padding=0, *a, **k):
    out = orig(inp, weight, bias, stride, padding, *a, **k)
    if weight is w:
        out.retain_grad(); store['out']=out; store['inp']=inp
    return out
F.conv2d = conv2d
out = R.model_forward(P, cfg, audio, alen, tok, tl, train=False, bn_training=True)
out["loss"].backward()
g = store['out'].grad  # [B,C,T1,F1]
inp = store['inp']
# dW[c,0,kh,kw] = sum_{b,t,f} g[b,c,t,f] * xpad[b,0,2t+kh,2f+kw]
xp = F.pad(inp,(1,1,1,1))
import itertools
cond=[]
for kh,kw in itertools.product(range(3),range(3)):
    patch = xp[:,0,kh:kh+2*g.shape[2]:2, kw:kw+2*g.shape[3]:2]  # [B,T1,F1]
    terms = g * patch.unsqueeze(1)
    s = terms.sum((0,2,3)); a = terms.abs().sum((0,2,3))
    cond.append((a/s.abs()).median().item())
print('median cond per tap', cond)
print('bias cond', (g.abs().sum((0,2,3))/g.sum((0,2,3)).abs()).median().item())
print('grad check', (w.grad - torch.stack([ (g*xp[:,0,kh:kh+2*g.shape[2]:2, kw:kw+2*g.shape[3]:2].unsqueeze(1)).sum((0,2,3)) for kh in range(3) for kw in range(3)],1).view(-1,1,3,3)).abs().max().item())
