"""Energy verifier head + one descent step on the prediction (reference energy.py:9-67,
ff_energy.py:11-33).  Small dense MLP work, plain PyTorch."""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F

from .heads import VecLinear
from .hip import ops


class EnergyVerifierHead(nn.Module):
    def __init__(self, ctx_dim: int, y_dim: int, hidden: int, *, aux_dim: int = 0):
        super().__init__()
        self.aux_dim = aux_dim
        n_in = ctx_dim + y_dim + aux_dim
        self.net = nn.Sequential(nn.LayerNorm(n_in), VecLinear(n_in, hidden), nn.SiLU(),
                                 VecLinear(hidden, hidden), nn.SiLU(), VecLinear(hidden, 1))

    def forward(self, ctx: torch.Tensor, y_hat: torch.Tensor, aux: torch.Tensor | None = None) -> torch.Tensor:
        parts = [ctx, y_hat]
        if self.aux_dim > 0:
            if aux is None:
                aux = ctx.new_zeros((self.aux_dim,) if ctx.dim() == 1 else (ctx.shape[0], self.aux_dim))
            if ctx.dim() == 1:
                aux = aux.view(-1)
            elif aux.dim() == 1:
                aux = aux.unsqueeze(0).expand(ctx.shape[0], -1)
            parts.append(aux)
        z = torch.cat(parts, dim=-1)
        if ops.linear_act_fits(z):  # train_step's loss rows: LayerNorm in PyTorch, the three layers on the library
            ln, l0, l1, l2 = self.net[0], self.net[1], self.net[3], self.net[5]
            h = F.layer_norm(z, ln.normalized_shape, ln.weight, ln.bias, ln.eps)
            h = ops.linear_act(ops.linear_act(h, l0.weight, l0.bias, "silu"), l1.weight, l1.bias, "silu")
            return ops.linear_act(h, l2.weight, l2.bias).squeeze(-1)
        return self.net(z).squeeze(-1)


def energy_descent_step(verifier, ctx, y_hat, alpha: float, sigma: float = 0.0, clamp=None):
    y_hat.requires_grad_(True)
    energy = verifier(ctx, y_hat).sum()
    (grad,) = torch.autograd.grad(energy, y_hat, create_graph=False)
    y_next = y_hat - alpha * grad + (sigma * torch.randn_like(y_hat) if sigma > 0 else 0.0)
    if clamp is not None:
        y_next = y_next.clamp(*clamp)
    return y_next.detach(), energy.detach()


def ff_energy_loss(E_pos: torch.Tensor, E_neg: torch.Tensor, tau: float = 0.0) -> torch.Tensor:
    return F.softplus(E_pos - tau).mean() + F.softplus(-(E_neg - tau)).mean()


def ff_step(verifier, ctx_pos, y_pos, ctx_neg, y_neg, tau: float = 0.0) -> dict:
    E_pos = verifier(ctx_pos.detach(), y_pos.detach())
    E_neg = verifier(ctx_neg.detach(), y_neg.detach())
    loss = ff_energy_loss(E_pos, E_neg, tau=tau)
    loss.backward()
    return {"loss": float(loss.detach().item()), "E_pos": float(E_pos.detach().mean().item()),
            "E_neg": float(E_neg.detach().mean().item())}


class Thinker:
    """A few noisy energy-descent steps on the output distribution (reference thinking.py:10-44)."""

    def __init__(self, verifier, max_steps: int = 3, alpha=(2e-2, 5e-2), sigma: float = 0.01, restarts: int = 1):
        self.verifier, self.max_steps = verifier, max_steps
        self.alpha = alpha if isinstance(alpha, tuple) else (alpha, alpha)
        self.sigma, self.restarts = sigma, restarts

    def optimize(self, ctx: torch.Tensor, y0: torch.Tensor):
        best = None
        for _ in range(self.restarts):
            y_hat = y0 + 0.01 * torch.randn_like(y0)
            energies = []
            for _ in range(self.max_steps):
                a = torch.empty(1).uniform_(*self.alpha).item()
                y_hat, E = energy_descent_step(self.verifier, ctx, y_hat, alpha=a, sigma=self.sigma)
                energies.append(E)
                if len(energies) > 1 and abs(energies[-1] - energies[-2]) < 1e-4:
                    break
            final_E = self.verifier(ctx, y_hat).detach()
            if best is None or final_E.mean() < best[0].mean():
                best = (final_E, y_hat.detach(), energies)
        assert best is not None
        return best
