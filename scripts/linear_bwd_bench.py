"""ic_linear_bwd per-call time (HIP events over a run of calls) at the shapes of train_step's loss heads."""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ironcortex_amd.hip import new, lib, stream_handle, check
dev = "cuda:0"
for B, n, rows, act in ((8, 256, 256, 1), (8, 512, 512, 1), (16, 516, 512, 2), (16, 512, 512, 2), (8, 512, 1, 0), (8, 256, 2, 0)):
    x, W, b, gy = torch.randn(B, n, device=dev), torch.randn(rows, n, device=dev), torch.randn(rows, device=dev), torch.randn(B, rows, device=dev)
    gx, gW, gb, slab = torch.empty_like(x), torch.empty_like(W), torch.empty_like(b), torch.empty(B, rows, device=dev)
    a = new("ic_linear_bwd_args", B=B, n=n, rows=rows, act=act, W=W, b=b, x=x, gy=gy, gx=gx, gW=gW, gb=gb, slab=slab)
    L, st = lib(), stream_handle()
    for _ in range(20):
        check(L.ic_linear_bwd(ctypes.byref(a), st))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    N = 200
    e0.record()
    for _ in range(N):
        L.ic_linear_bwd(ctypes.byref(a), st)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / N * 1e3
    alg = 4.0 * (rows * n * 3 + B * (2 * n + 2 * rows))
    print(f"B={B} n={n} rows={rows} act={act}: {us:.1f} us per call (two launches), {alg / us / 1e3:.0f} GB/s algorithmic")
