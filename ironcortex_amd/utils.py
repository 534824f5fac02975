"""Primitives of the hot path with the reference's signatures (reference ironcortex/utils.py).

RMSNorm / KWTA / nms_topk / goodness run on the HIP library (ironcortex_amd.hip); the small
host-side helpers (burst schedule, halting, span extraction) are plain Python.
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F

from .constants import EPS_LOG
from .flat import ArenaParameter
from .hip import ops


def init_weights(module: nn.Module) -> None:
    """Xavier-uniform weights / zero bias for every Linear-like layer (utils.py:10-14)."""
    w = getattr(module, "weight", None)
    if isinstance(module, nn.Linear) or getattr(module, "_ic_linear", False):
        nn.init.xavier_uniform_(w)
        if getattr(module, "bias", None) is not None:
            nn.init.zeros_(module.bias)


class RMSNorm(nn.Module):
    """y = weight * x / sqrt(mean(x^2) + eps)   (utils.py:20-35), HIP kernel ic_rmsnorm_*."""

    def __init__(self, d: int, eps: float = 1e-5):
        super().__init__()
        self.eps = eps
        self.weight = ArenaParameter(torch.ones(d))

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return ops.RMSNormFn.apply(x, self.weight, self.eps)


def KWTA(x: torch.Tensor, k: int, *, soft: bool = False, temp: float = 1.0) -> torch.Tensor:
    """k-winners-take-all over the last dim (utils.py:38-60), HIP kernel ic_kwta_*."""
    if k <= 0:
        return torch.zeros_like(x)
    return ops.KWTAFn.apply(x, int(k), bool(soft), float(temp))


def neighbor_csr(neighbors: Sequence[Sequence[int]], device) -> Tuple[torch.Tensor, torch.Tensor]:
    off, flat = [0], []
    for nb in neighbors:
        flat.extend(int(j) for j in nb)
        off.append(len(flat))
    if not flat:
        flat = [0]
    return (torch.tensor(off, dtype=torch.int32, device=device),
            torch.tensor(flat, dtype=torch.int32, device=device))


def nms_topk(scores: torch.Tensor, k: int, neighbors: List[List[int]]) -> List[int]:
    """Greedy neighbour-suppressed top-k (utils.py:107-124) on the device.

    Total order: score descending, index ascending.  Returns the chosen indices in walk order."""
    R = scores.shape[0]
    off, nbr = neighbor_csr(neighbors, scores.device)
    sc = scores.detach().reshape(1, R).float().contiguous()
    dummy = torch.zeros(1, R, 4, device=scores.device)
    _, mask = ops.gate_scores_select(dummy, sc[0], sc[0], None, off, nbr, 0, 0, int(k), False, 0.0, 0.0,
                                     scores=sc)
    idx = mask[0].nonzero().flatten().tolist()
    vals = scores.detach().tolist()
    return sorted(idx, key=lambda i: (-vals[i], i))


def goodness(h_stack: torch.Tensor) -> torch.Tensor:
    """Scalar FF goodness: mean over everything of h^2 (utils.py:159-162)."""
    return ops.GoodnessFn.apply(h_stack.float()).mean()


def masked_mean(Y: torch.Tensor, M: torch.Tensor) -> torch.Tensor:
    """[B,T,d], bool [B,T] -> [B,d] (utils.py:63-68)."""
    w = M.to(Y.dtype)
    return (Y * w.unsqueeze(-1)).sum(dim=1) / w.sum(dim=1).clamp_min(1.0).unsqueeze(-1)


def extract_spans(mask_1d_bool: torch.Tensor) -> List[Tuple[int, int]]:
    """Half-open [start, end) runs of True (utils.py:71-81)."""
    bits = mask_1d_bool.tolist() if hasattr(mask_1d_bool, "tolist") else list(mask_1d_bool)
    spans, start = [], None
    for i, b in enumerate(bits + [False]):
        if b and start is None:
            start = i
        elif not b and start is not None:
            spans.append((start, i))
            start = None
    return spans


def uncertainty_from_logits(logits: torch.Tensor) -> float:
    """1 - mean max-probability (utils.py:127-130)."""
    return float((1.0 - F.softmax(logits, dim=-1).max(dim=-1).values.mean()).item())


def context_logprob(logits: torch.Tensor) -> torch.Tensor:
    """Negative entropy of the logits (utils.py:133-137)."""
    p = F.softmax(logits, dim=-1)
    return (p * (p + EPS_LOG).log()).sum(dim=-1).mean()


def schedule_burst(u_mean: float, v_hat: float, R: int) -> int:
    """utils.py:140-144."""
    return max(1, R // 8) if (u_mean > 0.30 or v_hat > 0.0) else 0


def should_halt(branch_scores, u_mean: float, k: int, K_inner: int) -> bool:
    """utils.py:147-151."""
    return u_mean < 0.05 and k >= max(1, K_inner // 2)


class RegionFFState(nn.Module):
    """Per-region FF threshold tau, an EMA (utils.py:165-186)."""

    def __init__(self, init_tau: float = 0.0):
        super().__init__()
        self.register_buffer("tau", torch.tensor(init_tau, dtype=torch.float32))

    def update_tau(self, g_pos_mean, mean_prec=None, *, alpha: float = 0.01, kappa: float = 0.0,
                   target_prec: float = 1.0) -> None:
        target = g_pos_mean.to(self.tau.dtype)
        if mean_prec is not None:
            target = target + kappa * (mean_prec.to(self.tau.dtype) - target_prec)
        self.tau = (1 - alpha) * self.tau + alpha * target


class FastAttrs:
    """Mix-in for nn.Module subclasses: nn.Module.__setattr__ walks the parameter / buffer / module
    registries on every assignment (1-2 us); the host-side bookkeeping attributes named in
    _plain_attrs (never tensors that must follow .to() or state_dict) and the already registered
    buffers named in _fast_buffers are set directly.  A reasoning loop assigns ~30 of them, while the
    GPU waits."""
    _plain_attrs: frozenset = frozenset()
    _fast_buffers: frozenset = frozenset()

    def __setattr__(self, name, value):
        if name in self._plain_attrs:
            object.__setattr__(self, name, value)
        elif name in self._fast_buffers and isinstance(value, torch.Tensor) and name in self.__dict__.get("_buffers", ()):
            self._buffers[name] = value
        else:
            super().__setattr__(name, value)
