import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nemo_amd import ops
dev="cuda"; B,T,d,k=32,501,512,31
ROT=int(os.environ.get("ROTATE","6"))  # rotating operand sets: cold operands, as inside the training step
g=torch.Generator(device=dev).manual_seed(0)
xs=[torch.randn(B,T,d,device=dev,generator=g).to(torch.bfloat16) for _ in range(ROT)]; dys=[torch.randn(B,T,d,device=dev,generator=g).to(torch.bfloat16) for _ in range(ROT)]
x=xs[0]; dy=dys[0]
w=torch.randn(d,1,k,device=dev,generator=g); bias=torch.randn(d,device=dev,generator=g)
ys=[torch.empty_like(x) for _ in range(ROT)]; dxs=[torch.empty_like(x) for _ in range(ROT)]; cnt=[0]
def nxt():
    cnt[0]+=1; return cnt[0]%ROT
y=torch.empty_like(x); dx=torch.empty_like(x); dw=torch.zeros(d,1,k,device=dev); db=torch.zeros(d,device=dev)
stats=torch.zeros(2,d,device=dev,dtype=torch.float64)
def t(name,fn):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): fn()
    e1.record(); torch.cuda.synchronize(); print(f"{name:10s} {e0.elapsed_time(e1)/20*1e3:8.1f} us", flush=True)
def f_fwd():
    i=nxt(); ops.dwconv_fwd(xs[i],w,bias,ys[i],stats,B,T,d,k)
def f_bwd():
    i=nxt(); ops.dwconv_bwd(dys[i],xs[i],w,dxs[i],dw,db,B,T,d,k)
def f_fwd_nostats():
    i=nxt(); ops.dwconv_fwd(xs[i],w,bias,ys[i],None,B,T,d,k)
t("dw_fwd", f_fwd)
t("dw_fwd_nostats", f_fwd_nostats)
t("dw_bwd", f_bwd)
