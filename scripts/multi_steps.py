"""N train_steps of the 10M bench config in one stream mode (for rocprofv3 --kernel-trace --stats)."""
import os, random, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ironcortex_amd as ic
from bench import make_config
dev = torch.device("cuda:0")
cfg = make_config("10m")
torch.manual_seed(0); random.seed(0)
m = ic.CortexReasoner(ic.hex_neighbors(cfg.R), ic.hex_axial_coords(cfg.R), {"sensor": 0, "motor": cfg.R - 1}, cfg).to(dev)
m.stream_mode = sys.argv[1] if len(sys.argv) > 1 else "multi"
opt = torch.optim.AdamW(m.parameters(), lr=1e-3, fused=True)
toks = torch.randint(0, 255, (8, 256), device=dev)
for _ in range(int(sys.argv[2]) if len(sys.argv) > 2 else 6):
    ic.train_step(m, opt, toks, ic.LossWeights(), dev)
torch.cuda.synchronize()
