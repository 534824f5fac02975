"""200 train_steps on a fixed random batch in both stream modes and both script configs: every metric must stay
finite (a coarse end-to-end stability check after kernel work; per-step correctness is pinned by the reference's
two-step fixtures in tests/).  Prints the mean of each loss term over the first and the last 10 steps: the terms
with a trained head behind them (rtd, denoise, critic, verify) fall; the FF term follows its moving thresholds."""
import json, os, random, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ironcortex_amd as ic
from bench import make_config
dev = torch.device("cuda:0")
for name in ("10m", "default"):
    for mode in ("strict", "multi"):
        cfg = make_config(name)
        torch.manual_seed(0); random.seed(0)
        m = ic.CortexReasoner(ic.hex_neighbors(cfg.R), ic.hex_axial_coords(cfg.R), {"sensor": 0, "motor": cfg.R - 1}, cfg).to(dev)
        m.stream_mode = mode
        opt = torch.optim.AdamW(m.parameters(), lr=1e-3, fused=True)
        toks = torch.randint(0, 255, (8, 256), device=dev)
        hist = []
        for i in range(200):
            out = ic.train_step(m, opt, toks, ic.LossWeights(), dev)
            hist.append(out)
            assert all(v == v and abs(v) < 1e9 for v in out.values()), (i, out)
        keys = ("total", "ff", "rtd", "denoise", "critic", "verify", "grad_norm")
        row = {"config": name, "mode": mode}
        for k in keys:
            row[k] = [round(sum(h[k] for h in hist[:10]) / 10, 4), round(sum(h[k] for h in hist[-10:]) / 10, 4)]
        print(json.dumps(row), flush=True)
print("train_sanity ok")
