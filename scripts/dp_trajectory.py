import os, sys, random
sys.path.insert(0, "/root/repo")
import torch
import ironcortex_amd as ic
from bench import make_config
dev = torch.device("cuda:0")
dp = None
if "--dp" in sys.argv:
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29543")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    from ironcortex_amd.parallel import DataParallel
    dp = DataParallel()
cfg = make_config("10m")
torch.manual_seed(0); random.seed(0)
m = ic.CortexReasoner(ic.hex_neighbors(cfg.R), ic.hex_axial_coords(cfg.R), {"sensor": 0, "motor": cfg.R - 1}, cfg).to(dev)
opt = torch.optim.AdamW(m.parameters(), lr=1e-3, fused=True)
toks = torch.randint(0, 255, (8, 256), device=dev)
for i in range(9):
    out = ic.train_step(m, opt, toks, ic.LossWeights(), dev, dp=dp)
    print(i, repr(out["total"]), repr(out["grad_norm"]), m.counters["inner_steps"], flush=True)
