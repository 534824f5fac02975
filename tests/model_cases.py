"""Shared helpers for whole-model parity tests (fixtures: tests/golden/model_*.npz)."""
import os
import random

import numpy as np
import torch

from tests.golden import inputs as gi

GOLD = os.path.join(os.path.dirname(__file__), "golden")
MODEL_CASES = {
    "default": dict(R=7, d=32, V=32, K_inner=3, B_br=2, k_active=4, max_T=64),
    "det": dict(R=7, d=32, V=32, K_inner=4, B_br=2, k_active=4, max_T=64, train_deterministic_inner_loop=True),
    "allflags": dict(R=10, d=32, V=32, K_inner=3, B_br=2, k_active=5, max_T=64,
                     enable_adaptive_filter_dynamics=True, enable_precision_routed_messages=True,
                     enable_radial_tangential_updates=True, enable_afa_attention=True,
                     enable_ff_energy_alignment=True, surprise_lambda=0.05, tau_kappa=0.01),
}
MODEL_B, MODEL_T = 2, 24


def load_case(name):
    from ironcortex_amd.config import CortexConfig
    gold = np.load(os.path.join(GOLD, f"model_{name}.npz"))
    cfg = CortexConfig(**MODEL_CASES[name])
    sd = {k[3:]: torch.as_tensor(gold[k]) for k in gold.files if k.startswith("sd.")}
    tokens = (gi.det(31, MODEL_B, MODEL_T, kind="rand") * (cfg.V - 1)).long()
    focus = gi.det(32, MODEL_B, MODEL_T, kind="rand") < 0.3
    return gold, cfg, sd, tokens, focus


def seed_train():
    torch.manual_seed(77)
    random.seed(77)
