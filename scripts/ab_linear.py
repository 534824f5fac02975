"""train_step with the loss heads on the library's fused linear layers against the PyTorch modules, same box,
alternating (ops.linear_act_fits patched to False for the PyTorch leg)."""
import json, os, random, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ironcortex_amd as ic
from ironcortex_amd.hip import ops
from bench import make_config
dev = torch.device("cuda:0")
mode = sys.argv[1] if len(sys.argv) > 1 else "strict"
cfg = make_config("10m")
torch.manual_seed(0); random.seed(0)
m = ic.CortexReasoner(ic.hex_neighbors(cfg.R), ic.hex_axial_coords(cfg.R), {"sensor": 0, "motor": cfg.R - 1}, cfg).to(dev)
m.stream_mode = mode
opt = torch.optim.AdamW(m.parameters(), lr=1e-3, fused=True)
toks = torch.randint(0, 255, (8, 256), device=dev)
fits = ops.linear_act_fits
res = {"hip": [], "torch": []}
for rnd in range(4):
    for leg in ("hip", "torch"):
        ops.linear_act_fits = fits if leg == "hip" else (lambda x: False)
        for _ in range(3):
            ic.train_step(m, opt, toks, ic.LossWeights(), dev)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(15):
            ic.train_step(m, opt, toks, ic.LossWeights(), dev)
        torch.cuda.synchronize()
        res[leg].append(round((time.perf_counter() - t0) / 15 * 1e3, 3))
print(json.dumps({"mode": mode, "ms_per_step": res}))
