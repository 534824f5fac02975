"""Workspace and the small heads around the inner step (reference heads.py:12-119).  The workspace sparsifier
is the HIP KWTA kernel; the MLP / RTD heads take the library's fused linear layers (forward ic_gemv_chain, backward
ic_linear_bwd) on the small row batches train_step differentiates through, plain PyTorch otherwise."""
from typing import Optional, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F

from .hip import ops
from .utils import KWTA, FastAttrs, RMSNorm


class VecLinear(nn.Linear):
    """nn.Linear whose 1-D inputs take the GEMV path (torch.addmv): on ROCm the GEMM path costs
    ~22 us of host time per call for a [d] vector, the GEMV path ~7 us.  Same parameters/keys."""

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if x.dim() == 1:
            return torch.addmv(self.bias, self.weight, x) if self.bias is not None else torch.mv(self.weight, x)
        return F.linear(x, self.weight, self.bias)


class Workspace(FastAttrs, nn.Module):
    """N slots of width d shared by all regions; writes are additive then KWTA'd (k = d//4)."""
    _fast_buffers = frozenset({"slots"})

    def __init__(self, d: int, N_slots: int = 8):
        super().__init__()
        self.N = N_slots
        self.norm = RMSNorm(d)
        self.register_buffer("slots", torch.zeros(N_slots, d), persistent=False)

    def read(self, keys: torch.Tensor) -> torch.Tensor:
        return F.softmax(keys @ self.slots.T, dim=-1) @ self.slots

    def write(self, delta_slots: torch.Tensor):
        assert delta_slots.shape == self.slots.shape
        self.slots = KWTA(self.slots + delta_slots, k=max(1, self.slots.shape[-1] // 4))


class MLP(nn.Module):
    def __init__(self, d_in: int, d_out: int, hidden: Optional[int] = None):
        super().__init__()
        hidden = hidden or max(d_in, d_out)
        self.net = nn.Sequential(VecLinear(d_in, hidden), nn.GELU(), VecLinear(hidden, d_out))

    def forward(self, x):
        if ops.linear_act_fits(x):  # train_step's loss rows: fused layers with the library's backward
            l0, l1 = self.net[0], self.net[2]
            return ops.linear_act(ops.linear_act(x, l0.weight, l0.bias, "gelu"), l1.weight, l1.bias)
        return self.net(x)


class PlannerHead(nn.Module):
    """(workspace, motor) -> (plan p, subgoal g)."""

    def __init__(self, d: int):
        super().__init__()
        self.mlp_p = MLP(d, d)
        self.mlp_g = MLP(2 * d, d)

    def forward(self, ws_state: torch.Tensor, motor_state: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        p = self.mlp_p(ws_state)
        return p, self.mlp_g(torch.cat([p, ws_state], dim=-1))


class CriticHead(nn.Module):
    """Expected goodness gain of spending more compute."""

    def __init__(self, d: int):
        super().__init__()
        self.mlp = MLP(2 * d, 1)

    def forward(self, ws_state: torch.Tensor, g: torch.Tensor) -> torch.Tensor:
        return self.mlp(torch.cat([ws_state, g], dim=-1)).squeeze(-1)


class VerifierHead(nn.Module):
    def __init__(self, d: int):
        super().__init__()
        self.mlp = MLP(d, 1)

    def forward(self, motor_state: torch.Tensor) -> torch.Tensor:
        return torch.sigmoid(self.mlp(motor_state)).squeeze(-1)


class RTDHead(nn.Module):
    """Replaced-token detection logits [*, 2]."""

    def __init__(self, d: int):
        super().__init__()
        self.clf = VecLinear(d, 2)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if ops.linear_act_fits(x):
            return ops.linear_act(x, self.clf.weight, self.clf.bias)
        return self.clf(x)


class TokenHead_MFS(nn.Module):
    """Mixture of Kf linear facets gated by a softmax; returns (logits [V], aux)."""

    def __init__(self, d: int, V: int, Kf: int = 8):
        super().__init__()
        self.V, self.Kf = V, Kf
        self.gate = VecLinear(d, Kf)
        self.facets = nn.ModuleList([VecLinear(d, V) for _ in range(Kf)])

    def stacked(self):
        """([Kf + Kf*V, d], [Kf + Kf*V]) gate and facet weights as one matrix (one GEMV per call)."""
        W = torch.cat([self.gate.weight] + [f.weight for f in self.facets], dim=0)
        b = torch.cat([self.gate.bias] + [f.bias for f in self.facets], dim=0)
        return W, b

    def forward_stacked(self, x: torch.Tensor, Wb) -> torch.Tensor:
        """Same result as forward(x)[0] from the stacked weights."""
        z = torch.addmv(Wb[1], Wb[0], x) if x.dim() == 1 else F.linear(x, Wb[0], Wb[1])
        g = F.softmax(z[: self.Kf], dim=-1)
        return (g.unsqueeze(-1) * z[self.Kf:].view(self.Kf, self.V)).sum(dim=0)

    def forward(self, x: torch.Tensor, targets: Optional[torch.Tensor] = None):
        g = F.softmax(self.gate(x), dim=-1)
        per_facet = torch.stack([f(x) for f in self.facets], dim=0)
        logits = (g.unsqueeze(-1) * per_facet).sum(dim=0)
        return logits, {"facet_balance": (g.mean() - 1.0 / self.Kf).pow(2)}

    def forward_rows(self, X: torch.Tensor):
        """forward() of every row of X [B,d] in one pass: (logits [B,V], facet_balance [B])."""
        W, b = self.stacked()
        z = F.linear(X, W, b)  # [B, Kf + Kf*V]
        g = F.softmax(z[:, : self.Kf], dim=-1)
        logits = (g.unsqueeze(-1) * z[:, self.Kf:].view(-1, self.Kf, self.V)).sum(dim=1)
        return logits, (g.mean(dim=-1) - 1.0 / self.Kf).pow(2)
