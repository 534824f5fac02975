import time, torch, torch.nn.functional as F
dev="cuda:0"
W=torch.randn(256,256,device=dev); b=torch.randn(256,device=dev); x=torch.randn(256,device=dev)
W2=torch.randn(2056,256,device=dev); b2=torch.randn(2056,device=dev)
def t(f,n=2000):
    for _ in range(50): f()
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(n): f()
    host=(time.perf_counter()-t0)/n*1e6
    torch.cuda.synchronize(); tot=(time.perf_counter()-t0)/n*1e6
    return round(host,1), round(tot,1)
print("F.linear 1d", t(lambda: F.linear(x,W,b)))
print("addmv", t(lambda: torch.addmv(b,W,x)))
print("mv", t(lambda: torch.mv(W,x)))
print("matmul x@Wt", t(lambda: x@W.t()))
print("F.linear 2d [1,d]", t(lambda: F.linear(x[None],W,b)))
print("F.linear big 1d", t(lambda: F.linear(x,W2,b2)))
print("addmv big", t(lambda: torch.addmv(b2,W2,x)))
print("empty", t(lambda: torch.empty(256,device=dev)))
print("add", t(lambda: x+b))
import torch
print("rawstream", t(lambda: torch._C._cuda_getCurrentRawStream(0)))
print("current_stream", t(lambda: torch.cuda.current_stream().cuda_stream))
