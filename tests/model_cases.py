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
# corners of the configuration space, same fixture layout (tests/golden/make_fixtures.py `corner`)
CORNER_CASES = {
    "kactive2": dict(R=7, d=32, V=32, K_inner=3, B_br=2, k_active=2, max_T=64),  # k_total = 0 (utils.py:118-120)
    "softkwta": dict(R=7, d=32, V=32, K_inner=3, B_br=2, k_active=4, max_T=64, kwta_soft_mode=True, kwta_k_start=12,
                     kwta_k_schedule=10, kwta_soft_temp=0.5),
    "nopred_nokwta": dict(R=7, d=32, V=32, K_inner=3, B_br=2, k_active=4, max_T=64, use_predictive_trace=False,
                          disable_kwta_during_gating=True),
    "afdvec": dict(R=10, d=32, V=32, K_inner=3, B_br=2, k_active=5, max_T=64, enable_adaptive_filter_dynamics=True,
                   afd_noise_mode="vector", enable_radial_tangential_updates=True, kwta_k=6, router_vectorized=False),
    "noff_surprise": dict(R=7, d=32, V=32, K_inner=3, B_br=2, k_active=4, max_T=64, enable_forward_forward=False,
                          enable_adaptive_filter_dynamics=True, enable_ff_energy_alignment=True, surprise_lambda=0.1,
                          surprise_lambda_schedule=4, init_decay=0.5),
}
# the shared-matrix router modes (reference gate.py:142-159,274-281), same fixture layout.  The reference's Hebbian
# block indexes router.W_edge, which only per_edge mode defines (training.py:240-243): with the forward-forward loss on,
# train_step raises AttributeError as soon as a positive-gain region and an active neighbour meet (SHARED_RAISE below),
# so the two-step fixtures of these modes run with enable_forward_forward=False.
SHARED_CASES = {
    "bydir": dict(R=10, d=32, V=32, K_inner=3, B_br=2, k_active=5, max_T=64, edge_transform_mode="by_direction",
                  enable_forward_forward=False),
    "bydir_prec": dict(R=10, d=32, V=32, K_inner=3, B_br=2, k_active=5, max_T=64, edge_transform_mode="by_direction",
                       enable_precision_routed_messages=True, enable_adaptive_filter_dynamics=True,
                       enable_forward_forward=False),
    "factorized": dict(R=10, d=32, V=32, K_inner=3, B_br=2, k_active=5, max_T=64, edge_transform_mode="factorized",
                       enable_forward_forward=False),
    "factorized_prec": dict(R=7, d=32, V=32, K_inner=3, B_br=2, k_active=4, max_T=64, edge_transform_mode="factorized",
                            enable_precision_routed_messages=True, enable_radial_tangential_updates=True,
                            enable_forward_forward=False),
}
# forward-forward loss ON in a shared-matrix mode: what the reference's train_step does (fixture shared_raise.npz)
SHARED_RAISE = {
    "bydir_ff": dict(R=10, d=32, V=32, K_inner=3, B_br=2, k_active=5, max_T=64, edge_transform_mode="by_direction"),
    "factorized_ff": dict(R=10, d=32, V=32, K_inner=3, B_br=2, k_active=5, max_T=64, edge_transform_mode="factorized"),
}
MODEL_B, MODEL_T = 2, 24

# Full-size cases: the exact bench model (reference train_tiny_shakespeare_10m.py:32-48)
# at the bench's T, and the default script config (train_tiny_shakespeare.py:59-67) with
# gate.gain_ema / homeo seeded tie-free (Gate.select_k's order on exact ties is torch.argsort's).
# name -> (CortexConfig kwargs, B, T)
BIG_CASES = {
    "bench10m": (dict(R=10, d=256, V=256, K_inner=8, B_br=2, k_active=8, max_T=512,
                      enable_adaptive_filter_dynamics=True, enable_precision_routed_messages=True,
                      enable_radial_tangential_updates=True, enable_afa_attention=True,
                      enable_ff_energy_alignment=True, enable_energy_verifier=True,
                      enable_forward_forward=True, debug_metrics_every_n_steps=2), 2, 256),
    "r36": (dict(R=36, d=256, V=256, K_inner=8, B_br=2, k_active=8, max_T=512), 2, 64),
}


def big_inputs(name):
    kw, B, T = BIG_CASES[name]
    tokens = (gi.det(31, B, T, kind="rand") * (kw["V"] - 1)).long()
    focus = gi.det(32, B, T, kind="rand") < 0.3
    return tokens, focus


def load_case(name):
    from ironcortex_amd.config import CortexConfig
    gold = np.load(os.path.join(GOLD, f"model_{name}.npz"))
    if name in BIG_CASES:
        cfg = CortexConfig(**BIG_CASES[name][0])
        order = gold["sd_order"].tolist()
        sd = {}
        for k in order:
            if "sd." + k in gold.files:
                sd[k] = torch.as_tensor(gold["sd." + k])
            elif "sdbits." + k in gold.files:
                shape = tuple(int(x) for x in gold["sdshape." + k])
                n = int(np.prod(shape))
                sd[k] = torch.as_tensor(np.unpackbits(gold["sdbits." + k])[:n].reshape(shape).astype(np.float32))
            else:
                spec = gold["sdgen." + k]  # [index, mean, std, *shape]
                sd[k] = gi.synth_tensor(int(spec[0]), [int(x) for x in spec[3:]], float(spec[1]), float(spec[2]))
        tokens, focus = big_inputs(name)
        return gold, cfg, sd, tokens, focus
    cfg = CortexConfig(**(MODEL_CASES[name] if name in MODEL_CASES else (CORNER_CASES[name] if name in CORNER_CASES else SHARED_CASES[name])))
    sd = {k[3:]: torch.as_tensor(gold[k]) for k in gold.files if k.startswith("sd.")}
    tokens = (gi.det(31, MODEL_B, MODEL_T, kind="rand") * (cfg.V - 1)).long()
    focus = gi.det(32, MODEL_B, MODEL_T, kind="rand") < 0.3
    return gold, cfg, sd, tokens, focus


def grad_probes(n_cols: int, n_rows: int):
    """Probe vectors for the stored projections of large matrix gradients (g @ p, q @ g)."""
    return gi.det(4101, n_cols), gi.det(4102, n_rows)


def seed_train():
    torch.manual_seed(77)
    random.seed(77)
