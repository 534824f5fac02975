import sys, time, random, torch
sys.path.insert(0, "/root/repo")
import ironcortex_amd as ic
dev = torch.device("cuda:0")
def run(cfg, B, T, steps):
    torch.manual_seed(0); random.seed(0)
    R = cfg.R
    m = ic.CortexReasoner(ic.hex_neighbors(R), ic.hex_axial_coords(R), {"sensor": 0, "motor": R - 1}, cfg).to(dev)
    opt = torch.optim.AdamW(m.parameters(), lr=1e-3)
    toks = torch.randint(0, 255, (B, T), device=dev)
    for i in range(steps):
        torch.cuda.synchronize(); t0 = time.time()
        out = ic.train_step(m, opt, toks, ic.LossWeights(), dev)
        torch.cuda.synchronize()
        print(i, f"{time.time()-t0:.3f}s", {k: round(v, 4) for k, v in out.items()}, m.counters, flush=True)
    return m
run(ic.CortexConfig(R=7, d=32, V=256, K_inner=3, max_T=128), 2, 32, 2)
run(ic.CortexConfig(R=36, d=256, V=256, K_inner=8, B_br=2, k_active=8, max_T=512), 2, 256, 3)
run(ic.CortexConfig(R=10, d=256, V=256, K_inner=8, B_br=2, k_active=8, max_T=512, enable_adaptive_filter_dynamics=True,
    enable_precision_routed_messages=True, enable_radial_tangential_updates=True, enable_afa_attention=True,
    enable_ff_energy_alignment=True), 2, 256, 3)
