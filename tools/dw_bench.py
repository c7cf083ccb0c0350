import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nemo_amd import ops
dev="cuda"; B,T,d,k=32,501,512,31
g=torch.Generator(device=dev).manual_seed(0)
x=torch.randn(B,T,d,device=dev,generator=g).to(torch.bfloat16); dy=torch.randn(B,T,d,device=dev,generator=g).to(torch.bfloat16)
w=torch.randn(d,1,k,device=dev,generator=g); bias=torch.randn(d,device=dev,generator=g)
y=torch.empty_like(x); dx=torch.empty_like(x); dw=torch.zeros(d,1,k,device=dev); db=torch.zeros(d,device=dev)
stats=torch.zeros(2,d,device=dev,dtype=torch.float64)
def t(name,fn):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): fn()
    e1.record(); torch.cuda.synchronize(); print(f"{name:10s} {e0.elapsed_time(e1)/20*1e3:8.1f} us", flush=True)
t("dw_fwd", lambda: ops.dwconv_fwd(x,w,bias,y,stats,B,T,d,k))
t("dw_fwd_ns", lambda: ops.dwconv_fwd(x,w,bias,y,None,B,T,d,k))
t("dw_bwd", lambda: ops.dwconv_bwd(dy,x,w,dx,dw,db,B,T,d,k))
