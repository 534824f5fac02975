"""Deterministic inputs shared by the fixture generator and the parity tests.

Fixtures store only OUTPUTS of the reference; the inputs are regenerated from these seeded
constructors (torch CPU generator, identical on every box with this image), which keeps the
committed files small.
"""
from __future__ import annotations

import math
from typing import Dict

import torch


def det(seed: int, *shape, kind: str = "randn", scale: float = 1.0, shift: float = 0.0) -> torch.Tensor:
    g = torch.Generator().manual_seed(int(seed))
    if kind == "randn":
        t = torch.randn(tuple(shape), generator=g)
    elif kind == "rand":
        t = torch.rand(tuple(shape), generator=g)
    elif kind == "xavier":  # utils.py:10-14 on a [d,d] Linear
        bound = math.sqrt(6.0 / (shape[-1] + shape[-2]))
        t = (torch.rand(tuple(shape), generator=g) * 2 - 1) * bound
    else:
        raise ValueError(kind)
    return t * scale + shift


# (name, d, m_time_pairs, dt, afd, rt, noise_mode, step_pos)
REGION_CASES = [
    ("d8_default", 8, 0, 0, False, False, "scalar", 0.0),
    ("d32_rot_dt3", 32, 16, 3, False, False, "scalar", 0.37),
    ("d128_rot", 128, 16, 0, False, False, "scalar", 0.5714286),
    ("d256_rot_dt2", 256, 16, 2, False, False, "scalar", 1.0),
    ("d32_afd_scalar", 32, 16, 1, True, False, "scalar", 0.25),
    ("d32_afd_vector", 32, 16, 0, True, False, "vector", 0.75),
    ("d32_rt", 32, 16, 2, False, True, "scalar", 0.6),
    ("d128_afd_rt", 128, 16, 1, True, True, "scalar", 0.4285714),
    ("d256_afd_rt_vec", 256, 16, 0, True, True, "vector", 0.1428571),
]


def region_inputs(seed: int, d: int, noise_mode: str = "scalar") -> Dict[str, torch.Tensor]:
    """Parameters, state and input vector of one region cell."""
    nshape = () if noise_mode == "scalar" else (d,)
    return {
        "Wr": det(seed + 1, d, d, kind="xavier"),
        "Wk": det(seed + 2, d, d, kind="xavier"),
        "Wv": det(seed + 3, d, d, kind="xavier"),
        "Wo": det(seed + 4, d, d, kind="xavier"),
        "norm_w": det(seed + 5, d, scale=0.1, shift=1.0),
        "raw_decay": det(seed + 6, d, scale=0.5),
        "pn_param": det(seed + 7, *nshape, scale=0.5, shift=-3.0),
        "on_param": det(seed + 8, *nshape, scale=0.5, shift=0.0),
        "rpn_param": det(seed + 9, scale=0.5, shift=-3.0).reshape(()),
        "ron_param": det(seed + 10, scale=0.5, shift=0.0).reshape(()),
        "num": det(seed + 11, d, scale=0.5),
        "den": det(seed + 12, d, kind="rand", shift=0.5),
        "var": det(seed + 13, d, kind="rand", scale=0.45, shift=0.05),
        "pred": det(seed + 14, d, scale=0.1),
        "radius": det(seed + 15, 1, kind="rand", shift=0.5),
        "radius_var": det(seed + 16, 1, kind="rand", scale=0.5, shift=0.1),
        "surprise": det(seed + 17, 1, kind="rand").reshape(()),
        "x_in": det(seed + 18, d),
        # cotangents for the scalar loss
        "g_h": det(seed + 20, d),
        "g_num": det(seed + 21, d, scale=0.3),
        "g_den": det(seed + 22, d, scale=0.3),
        "g_var": det(seed + 23, d, scale=0.3),
        "g_radius": det(seed + 24, 1, scale=0.3),
        "g_radius_var": det(seed + 25, 1, scale=0.3),
        "probe_p": det(seed + 26, d),
        "probe_q": det(seed + 27, d),
    }


# (name, R or explicit neighbours, d, mode, precision, mask seed)
ROUTER_CASES = [
    ("r3_per_edge", 3, 8, "per_edge", False),
    ("r7_per_edge", 7, 16, "per_edge", False),
    ("r10_per_edge_prec", 10, 16, "per_edge", True),
    ("r7_by_direction", 7, 16, "by_direction", False),
    ("r7_by_direction_prec", 7, 16, "by_direction", True),
    ("r7_factorized", 7, 16, "factorized", False),
    ("r7_factorized_prec", 7, 16, "factorized", True),
    ("r36_per_edge", 36, 32, "per_edge", False),
    ("r36_per_edge_prec", 36, 32, "per_edge", True),
]


def router_inputs(seed: int, R: int, d: int, E: int, mode: str) -> Dict[str, torch.Tensor]:
    nW = {"per_edge": E, "by_direction": 6, "factorized": E}[mode]
    out = {
        "H": det(seed + 1, R, d),
        "mask_prev": det(seed + 2, R, kind="rand") < 0.6,
        "key_W": det(seed + 4, nW, d, d, kind="xavier"),
        "raw_P": det(seed + 5, nW, d, scale=0.3, shift=-2.0),
        "query_W": det(seed + 6, R, d, d, kind="xavier"),
        "W_reg": det(seed + 7, 8, 2),
        "beta_cos": det(seed + 8, 8, scale=0.3),
        "beta_sin": det(seed + 9, 8, scale=0.3),
        "g_M": det(seed + 10, R, d),
    }
    if mode == "per_edge":
        out["W_edge"] = det(seed + 3, E, d, d, kind="xavier")
    elif mode == "by_direction":
        out["W_dir"] = det(seed + 3, 6, d, d, kind="xavier")
    else:
        out["U"] = det(seed + 3, d, d, kind="xavier")
        out["V"] = det(seed + 11, d, d, kind="xavier")
        out["w_edge"] = det(seed + 12, E, d, scale=0.2, shift=1.0)
    out["mask_prev"][0] = True
    return out


KWTA_CASES = [(32, 1), (32, 4), (32, 8), (32, 32), (128, 16), (128, 32), (256, 32), (256, 64), (256, 1)]


def kwta_rows(seed: int, d: int) -> torch.Tensor:
    """6 rows: 3 random, one all-zero, one with duplicate magnitudes around the threshold, one
    with +-pairs."""
    x = det(seed, 6, d)
    x[3] = 0.0
    x[4] = torch.round(x[4] * 2) / 2  # many exact ties
    x[5, 1::2] = -x[5, 0::2]
    return x


GATE_R = [4, 7, 10, 36, 64, 512]


def synth_tensor(idx: int, shape, mean: float, std: float) -> torch.Tensor:
    """A large parameter of a full-size model case: uniform with the recorded mean / standard
    deviation of the reference's own initialisation (the 10-40 M-parameter state_dicts of the
    full-size cases are regenerated from ~150 (index, mean, std) triples instead of being stored)."""
    g = torch.Generator().manual_seed(77000 + int(idx))
    u = torch.rand(tuple(int(x) for x in shape), generator=g) * 2 - 1
    return u * float(std * math.sqrt(3.0)) + float(mean)


class DeterministicDraws:
    """Makes a sampling loop (generate / diffusion_generate) reproducible across devices: inside the
    context torch.randn_like / rand / randint draw from ONE seeded CPU generator (results moved to
    the requested device), and torch.multinomial records the distribution it is given and returns
    either a CPU draw (fixture generation) or the recorded sample of the reference run (`replay`),
    so that the run under test follows the reference's trajectory exactly and only the
    distributions are compared."""

    def __init__(self, seed: int, replay=None):
        self.g = torch.Generator().manual_seed(int(seed))
        # multinomial draws from a generator of its own: replaying recorded samples must not shift the other draws
        self.g2 = torch.Generator().manual_seed(int(seed) + 1)
        self.replay = None if replay is None else [torch.as_tensor(r) for r in replay]
        self.probs, self.samples = [], []
        self._saved = {}

    def __enter__(self):
        import torch as T
        g = self.g
        self._saved = {k: getattr(T, k) for k in ("randn_like", "rand", "randint", "multinomial")}
        o = self._saved

        def randn_like(x, **k):
            return T.randn(tuple(x.shape), generator=g).to(device=x.device, dtype=x.dtype)

        def rand(*a, device=None, **k):
            return o["rand"](*a, generator=g, **k).to(device) if device is not None else o["rand"](*a, generator=g, **k)

        def randint(*a, device=None, **k):
            out = o["randint"](*a, generator=g, **k)
            return out.to(device) if device is not None else out

        def multinomial(probs, num_samples, replacement=False, **k):
            self.probs.append(probs.detach().float().cpu().clone())
            i = len(self.samples)
            if self.replay is not None:
                s = self.replay[i].to(torch.long)
            else:
                s = o["multinomial"](probs.detach().float().cpu(), num_samples, replacement, generator=self.g2)
            self.samples.append(s.clone())
            return s.to(probs.device)

        T.randn_like, T.rand, T.randint, T.multinomial = randn_like, rand, randint, multinomial
        return self

    def __exit__(self, *exc):
        import torch as T
        for k, v in self._saved.items():
            setattr(T, k, v)
        return False


GEN_PROMPT = [ord(c) for c in "ROMEO:"]  # train_tiny_shakespeare.py:36 gen_prompt
