import sys, random, torch
sys.path.insert(0, "/root/repo")
import ironcortex_amd as ic
from bench import make_config
dev = torch.device("cuda:0")
for name in ("10m", "default"):
    cfg = make_config(name)
    torch.manual_seed(0); random.seed(0)
    m = ic.CortexReasoner(ic.hex_neighbors(cfg.R), ic.hex_axial_coords(cfg.R), {"sensor": 0, "motor": cfg.R - 1}, cfg).to(dev)
    opt = torch.optim.AdamW(m.parameters(), lr=1e-3)
    toks = torch.randint(0, 255, (8, 256), device=dev)
    E = m.engine
    orig = E._batch_uniform
    def dbg(bt):
        ss = bt["samples"]; K = bt["n_max"]; s0 = ss[0]
        for j, s in enumerate(ss):
            why = [s.chunk is bt["chunk"], s.i0 == bt["base"] + j * K, s.n == K, s.zero_in,
                   (s.flags, s.kwta_k, s.kwta_temp) == (s0.flags, s0.kwta_k, s0.kwta_temp),
                   s.ebias.data_ptr() == s0.ebias.data_ptr(), s.norm_in_w is s0.norm_in_w, s.keep[1] is s0.keep[1],
                   s.step_pos == s0.step_pos]
            if not all(why):
                print("sample", j, why, s.i0, bt["base"], s.n, K)
                break
        return orig(bt)
    E._batch_uniform = dbg
    for i in range(3):
        out = ic.train_step(m, opt, toks, ic.LossWeights(), dev)
    print(name, m.engine.counters, {k: round(v, 4) for k, v in out.items() if k in ("total", "grad_norm", "ff")})
    # why (not) uniform: the last recorded batch
