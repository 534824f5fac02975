"""ic_attn_fwd against the [T,T] PyTorch formulation at generation size (one sequence, T = 100, no gradient)."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ironcortex_amd.hip import ops
from ironcortex_amd import hip
dev = "cuda:0"
for (B, H, T, dqk, dv, afa) in ((1, 4, 100, 64, 64, True), (1, 4, 100, 128, 64, False), (16, 4, 276, 128, 64, False)):
    q, k, v = (torch.randn(B, H, T, w, device=dev) * 0.3 for w in (dqk, dqk, dv))
    alpha = torch.tensor(0.05, device=dev) if afa else None
    idx = torch.arange(T, device=dev, dtype=torch.float32)
    def hipf():
        return ops.attention(q, k, v, alpha=alpha, eps_log=1e-12)
    def torchf():
        s = q @ k.transpose(-2, -1)
        if afa:
            s = s * torch.exp(-alpha * (idx.view(T, 1) - idx.view(1, T)).abs())
        p = torch.softmax(s, -1)
        lg = (p + 1e-12).log()
        return p @ v, -lg.mean(dim=(1, 2, 3)), -(lg * p).sum(-1).mean(dim=(1, 2))
    row = {"B": B, "H": H, "T": T, "dqk": dqk}
    with torch.no_grad():
        for name, f in (("hip", hipf), ("torch", torchf)):
            for _ in range(20):
                f()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0 = time.perf_counter(); e0.record()
            for _ in range(200):
                f()
            e1.record(); t1 = time.perf_counter()
            torch.cuda.synchronize()
            row[name + "_gpu_us"] = round(e0.elapsed_time(e1) * 5, 1)
            row[name + "_host_us"] = round((t1 - t0) * 5e3, 1)
        hip.prof_enable(True); hip.prof_read()
        for _ in range(50):
            hipf()
        pr = hip.prof_read(); hip.prof_enable(False)
        row["kernel_us"] = round(1e3 * pr["attn_fwd"][1] / pr["attn_fwd"][0], 1)
    print(json.dumps(row), flush=True)
