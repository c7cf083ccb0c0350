"""Host-side time of the phases of one fit_step (forward issue, backward issue, optimizer issue) next to the GPU step time:
shows whether the Python side or the GPU bounds a model's step.  usage: python tools/host_phases.py [ctc|transducer|squeezeformer]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from nemo_amd.models import (EncDecCTCModel, EncDecRNNTModel, conformer_ctc_config, fastconformer_transducer_config,
                             squeezeformer_ctc_config)
from oracle import conformer_ref as R  # synthetic batch generator only

kind = sys.argv[1] if len(sys.argv) > 1 else "transducer"
dev = torch.device("cuda:0")
cdt = torch.bfloat16
torch.manual_seed(0)
if kind == "transducer":
    m = EncDecRNNTModel(fastconformer_transducer_config("large", vocab_size=1024, compute_dtype=cdt))
    m.decoder.compute_dtype = m.joint.compute_dtype = cdt
    vocab = 1024
elif kind == "squeezeformer":
    m = EncDecCTCModel(squeezeformer_ctc_config("medium", vocab_size=128, compute_dtype=cdt)); m.decoder.compute_dtype = cdt; vocab = 128
else:
    m = EncDecCTCModel(conformer_ctc_config("large", vocab_size=128, compute_dtype=cdt)); m.decoder.compute_dtype = cdt; vocab = 128
m = m.to(dev).train()
m.setup_optimization(dict(name="adamw", lr=1e-4, betas=[0.9, 0.98], weight_decay=1e-3))
audio, alen, tok, tl = R.synthetic_batch(32, 20.0, vocab=vocab, seed=0)
batch = [t.to(dev) for t in (audio, alen, tok, tl)]
for _ in range(3):
    m.fit_step(batch)
torch.cuda.synchronize()
acc = {"fwd": 0.0, "bwd": 0.0, "opt": 0.0}
N = 8
t_all = time.perf_counter()
for _ in range(N):
    m._optimizer.zero_grad()
    t0 = time.perf_counter()
    out = m.training_step(batch, 0)
    t1 = time.perf_counter()
    out["loss"].backward()
    m._after_backward()
    t2 = time.perf_counter()
    m._optimizer.step(lr=1e-4, grad_scale=1.0)
    for mod in m.trainable_modules():
        mod.weights_updated()
    t3 = time.perf_counter()
    acc["fwd"] += t1 - t0; acc["bwd"] += t2 - t1; acc["opt"] += t3 - t2
torch.cuda.synchronize()
wall = (time.perf_counter() - t_all) / N
print(kind, "wall ms/step", round(wall * 1e3, 2), "host issue ms: fwd", round(acc["fwd"] / N * 1e3, 2), "bwd", round(acc["bwd"] / N * 1e3, 2),
      "opt", round(acc["opt"] / N * 1e3, 2), "sum", round(sum(acc.values()) / N * 1e3, 2))
