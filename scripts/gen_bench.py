#!/usr/bin/env python3
"""BASELINE.json configs[4]: mask-predict `generate` and `diffusion_generate` with energy
refinement, enable_afa_attention=True (10M config), forward only.  Reports inner steps/s and
generated tokens/s (one JSON line per generator)."""
import json
import os
import random
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ironcortex_amd as ic  # noqa: E402
from bench import make_config  # noqa: E402

dev = torch.device("cuda:0")
cfg = make_config("10m")
torch.manual_seed(0)
random.seed(0)
m = ic.CortexReasoner(ic.hex_neighbors(cfg.R), ic.hex_axial_coords(cfg.R), {"sensor": 0, "motor": cfg.R - 1}, cfg).to(dev)
m.eval()
prompt = torch.tensor(list(b"ROMEO:"), device=dev)
T_total = 100


def run(name, fn, reps=5):
    fn()
    torch.cuda.synchronize()
    m.counters["inner_steps"] = 0
    t0 = time.perf_counter()
    for _ in range(reps):
        out = fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    steps = m.counters["inner_steps"] / reps
    print(json.dumps({"generator": name, "T_total": T_total, "ms_per_call": dt * 1e3, "inner_steps_per_call": steps,
                      "inner_steps_per_s": steps / dt, "tokens_per_s": (T_total - prompt.numel()) / dt,
                      "engine": m.engine.enabled}), flush=True)
    return out


for engine in (True, False):
    m.engine.enabled = engine
    run("generate(max_outer_iters=8)", lambda: ic.generate(m, prompt, T_total=T_total))
    run("diffusion_generate(steps=4)", lambda: ic.diffusion_generate(m, prompt, T_total=T_total, diff_cfg=ic.DiffusionConfig(steps=4)))
