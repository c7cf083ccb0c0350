"""raw HBM store / copy rates for reference (torch fill_/copy_), to put the GEMM epilogue numbers in context"""
import torch
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3
for mb in (16, 64, 128, 512):
    x = torch.empty(mb * 1024 * 1024 // 2, dtype=torch.bfloat16, device="cuda")
    y = torch.empty_like(x)
    a = t(lambda: x.fill_(1.0)); b = t(lambda: y.copy_(x))
    print(f"{mb:4d} MiB  fill {a*1e6:7.1f} us {mb*1.048576e6/a/1e12:6.2f} TB/s   copy {b*1e6:7.1f} us  {2*mb*1.048576e6/b/1e12:6.2f} TB/s (r+w)")
