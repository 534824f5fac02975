import cProfile, pstats, random, sys, os, io
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ironcortex_amd as ic
from bench import make_config
dev = torch.device("cuda:0")
cfg = make_config("10m")
torch.manual_seed(0); random.seed(0)
R = cfg.R
m = ic.CortexReasoner(ic.hex_neighbors(R), ic.hex_axial_coords(R), {"sensor": 0, "motor": R - 1}, cfg).to(dev)
opt = torch.optim.AdamW(m.parameters(), lr=1e-3)
toks = torch.randint(0, 255, (8, 256), device=dev)
for _ in range(2):
    ic.train_step(m, opt, toks, ic.LossWeights(), dev)
pr = cProfile.Profile()
pr.enable()
ic.train_step(m, opt, toks, ic.LossWeights(), dev)
torch.cuda.synchronize()
pr.disable()
s = io.StringIO()
st = pstats.Stats(pr, stream=s)
st.sort_stats("cumulative").print_callees("train_step")
print(s.getvalue()[:7000])
