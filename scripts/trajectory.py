"""Metrics of N consecutive train_steps of the 10M config (B=2, T=64) from one seed: `oracle` = oracle/model.py on the
CPU (reference-shaped), `hip` = the product in strict mode.  Same initial weights (the oracle's state_dict), same tokens,
same random streams, so the two trajectories coincide until fp32 rounding differences are amplified by the
training dynamics.  usage: trajectory.py oracle|hip N [state.pt]"""
import json, os, random, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import make_config
which, N = sys.argv[1], int(sys.argv[2])
state_path = sys.argv[3] if len(sys.argv) > 3 else os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "gpurun_out", "traj_state.pt")
cfg = make_config("10m")
B, T = 2, 64
from oracle import model as OM
torch.manual_seed(0); random.seed(0)
om = OM.build(cfg)
sd = {k: v.clone() for k, v in om.state_dict().items()}
tokens = torch.randint(0, 255, (B, T))
if which == "oracle":
    opt = torch.optim.AdamW(om.parameters(), lr=1e-3)
    step = lambda: OM.oracle_train_step(om, opt, tokens, OM.LossWeights())
else:
    import ironcortex_amd as ic
    dev = torch.device("cuda:0")
    m = ic.CortexReasoner(ic.hex_neighbors(cfg.R), ic.hex_axial_coords(cfg.R), {"sensor": 0, "motor": cfg.R - 1}, cfg)
    m.load_state_dict(sd)
    m = m.to(dev)
    opt = torch.optim.AdamW(m.parameters(), lr=1e-3)
    tk = tokens.to(dev)
    step = lambda: ic.train_step(m, opt, tk, ic.LossWeights(), dev)
torch.manual_seed(1); random.seed(1)  # (after both models exist: construction draws from the generator)
for i in range(N):
    out = step()
    print(json.dumps({"step": i, **{k: float(v) for k, v in out.items()}}), flush=True)
