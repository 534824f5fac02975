import cProfile, pstats, random, sys, os, io
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ironcortex_amd as ic
from bench import make_config
dev = torch.device("cuda:0")
cfg = make_config(sys.argv[1] if len(sys.argv) > 1 else "10m")
torch.manual_seed(0); random.seed(0)
R = cfg.R
m = ic.CortexReasoner(ic.hex_neighbors(R), ic.hex_axial_coords(R), {"sensor": 0, "motor": R - 1}, cfg).to(dev)
if len(sys.argv) > 2:
    m.stream_mode = sys.argv[2]
opt = torch.optim.AdamW(m.parameters(), lr=1e-3)
toks = torch.randint(0, 255, (8, 256), device=dev)
ic.train_step(m, opt, toks, ic.LossWeights(), dev)
ic.train_step(m, opt, toks, ic.LossWeights(), dev)
pr = cProfile.Profile()
pr.enable()
ic.train_step(m, opt, toks, ic.LossWeights(), dev)
torch.cuda.synchronize()
pr.disable()
for key in ("tottime", "cumulative"):
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats(key).print_stats(45)
    print(s.getvalue()[:9000])
