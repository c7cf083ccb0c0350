"""In-process interleaved A/B of a run-time knob on the full training step (same box, same clocks, drift hits both arms):
    python tools/step_ab.py KEY=V0,V1 [rounds] [steps-per-arm]      e.g.  gemm6=0,1   or   env:MI355X_FOO=0,1 (read per launch only)
gemmN=a,b switches mi355x_gemm_config(N, .); arena=0,1 the step-scoped arena; env:NAME=a,b an environment knob that is read per call; enc.attr=a,b an encoder attribute.  The encoder runs on the eager sequencer (a recorded graph would freeze the arm)."""
import os
import sys
import time

GRAPHS = os.environ.get("AB_GRAPHS", "0") == "1"  # 1: every arm re-records the encoder's launch sequence (no host in the loop)
os.environ["MI355X_GRAPHS"] = "1" if GRAPHS else "0"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from nemo_amd import ops
from nemo_amd.models import EncDecCTCModel, conformer_ctc_config
from oracle import conformer_ref as R  # batch generator only

specs = sys.argv[1].split(";")   # several knobs in one process ("a=0,1;b=1,2"): one after the other, each back at its first arm afterwards
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 5
per = int(sys.argv[3]) if len(sys.argv) > 3 else 8
name, vals = specs[0].split("=")
def _num(v):
    if v.lstrip("-").isdigit():
        return int(v)
    try:
        return float(v)
    except ValueError:
        return v


arms = [_num(v) for v in vals.split(",")]


def set_arm(v):
    if name.startswith("gemm"):
        ops.gemm_config(int(name[4:]), v)
    elif name == "arena":
        m.encoder.use_arena = bool(v)
    elif name.startswith("cfg:"):   # cfg:ctc=0,1 -> mi355x_ctc_config(v) (dwconv, logmel, ctc: library-side variant switches)
        from nemo_amd._lib import lib
        getattr(lib, f"mi355x_{name[4:]}_config")(int(v))
    elif name.startswith("env:"):
        os.environ[name[4:]] = str(v)
    elif name.startswith("model."):
        setattr(m, name[6:], bool(v) if isinstance(getattr(m, name[6:]), bool) else v)
    elif name.startswith("enc."):
        setattr(m.encoder, name[4:], bool(v) if isinstance(getattr(m.encoder, name[4:]), bool) else v)
    else:
        raise SystemExit("unknown knob " + name)


dev = torch.device("cuda:0")
torch.manual_seed(0)
m = EncDecCTCModel(conformer_ctc_config("large", vocab_size=128, spec_augment=True, compute_dtype=torch.bfloat16))
m.decoder.compute_dtype = torch.bfloat16
m = m.to(dev).train()
m.setup_optimization()
audio, alen, tok, tl = R.synthetic_batch(32, 20.0, vocab=128, seed=1234)
batch = [t.to(dev) for t in (audio, alen, tok, tl)]
main_stream = torch.cuda.Stream(device=dev, priority=-1) if os.environ.get("AB_MAIN_PRIO") == "1" else None
if main_stream is not None:  # the whole step on a HIGH-priority stream (side streams keep the default priority)
    main_stream.wait_stream(torch.cuda.current_stream())
    torch.cuda.set_stream(main_stream)
for _ in range(4):
    m.fit_step(batch)
for spec in specs:
  name, vals = spec.split("=")
  arms = [_num(v) for v in vals.split(",")]
  res = {a: [] for a in arms}
  for r in range(rounds):
      for a in (arms if r % 2 == 0 else arms[::-1]):
          set_arm(a)
          if GRAPHS:
              m.encoder._graph_sets.clear()
              for _ in range(3):
                  m.fit_step(batch)
          m.fit_step(batch)
          torch.cuda.synchronize()
          t0 = time.perf_counter()
          for _ in range(per):
              m.fit_step(batch)
          torch.cuda.synchronize()
          res[a].append((time.perf_counter() - t0) / per * 1e3)
  for a in arms:
      v = sorted(res[a])
      print(f"{name}={a}: median {v[len(v)//2]:.2f} ms  min {v[0]:.2f}  all {[round(x, 2) for x in res[a]]}")
  set_arm(arms[0])
