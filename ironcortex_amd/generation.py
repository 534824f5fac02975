"""Mask-predict and diffusion-style generators: callers of reasoning_loop, forward only
(reference generation.py:11-88, diffusion.py:11-77)."""
from __future__ import annotations

from dataclasses import dataclass
from typing import Callable, Iterable, Optional

import torch
import torch.nn.functional as F

from .energy import Thinker
from .model import CortexReasoner


@torch.no_grad()
def generate(model: CortexReasoner, prompt_tokens: torch.Tensor, T_total: int, max_outer_iters: int = 8,
             conf_threshold: float = 0.9, n_samples: int = 1) -> torch.Tensor:
    device = next(model.parameters()).device
    thinker = Thinker(model.verify, max_steps=3, alpha=(2e-2, 5e-2), sigma=0.01)
    MASK = model.V - 1
    best_tokens, best_energy = None, None
    for _ in range(n_samples):
        T0 = prompt_tokens.shape[0]
        tokens = torch.cat([prompt_tokens, torch.full((T_total - T0,), MASK, device=device, dtype=torch.long)], dim=0)
        conf = torch.zeros(T_total, device=device)
        H_prev, mask_prev = model.zeros_state(device)
        for _ in range(max_outer_iters):
            focus_map = (tokens == MASK) | (conf < conf_threshold)
            H_prev, mask_prev, logits, _traces = model.reasoning_loop(tokens, model.cfg.K_inner, focus_map, mask_prev, H_prev)
            motor_state = H_prev[model.io_idxs["motor"]]
            probs = F.softmax(logits, dim=-1)
            with torch.enable_grad():
                _, refined, _ = thinker.optimize(motor_state, probs)
            probs = refined.softmax(dim=-1)
            probs[-1] = 0  # never emit the padding token or NUL
            probs[0] = 0
            probs = probs / probs.sum(dim=-1, keepdim=True)
            pred = torch.multinomial(probs, num_samples=T_total, replacement=True)
            tokens = torch.where(focus_map, pred, tokens)
            conf[:] = float(probs.max().item())
            if bool((conf > conf_threshold).all()) and bool((tokens != MASK).all()):
                break
        energy = model.verify(motor_state, probs, aux=model._verify_aux(device))
        if best_energy is None or energy < best_energy:
            best_energy, best_tokens = energy, tokens.clone()
    return best_tokens


@dataclass
class DiffusionConfig:
    steps: int = 4
    noise_start: float = 1.0
    noise_end: float = 0.0


@torch.no_grad()
def diffusion_generate(model: CortexReasoner, prompt_tokens: torch.Tensor, T_total: int,
                       diff_cfg: Optional[DiffusionConfig] = None,
                       region_generators: Optional[Iterable[Callable[[torch.Tensor], torch.Tensor]]] = None) -> torch.Tensor:
    diff_cfg = diff_cfg or DiffusionConfig()
    device = next(model.parameters()).device
    T0 = prompt_tokens.shape[0]
    tokens = torch.randint(low=0, high=model.V - 1, size=(T_total,), device=device)
    tokens[:T0] = prompt_tokens.to(device)
    H_prev, mask_prev = model.zeros_state(device)
    for step in range(diff_cfg.steps, 0, -1):
        noise_p = diff_cfg.noise_end + (diff_cfg.noise_start - diff_cfg.noise_end) * step / diff_cfg.steps
        renoise = torch.rand(T_total, device=device) < noise_p
        noisy = torch.where(renoise, torch.randint(0, model.V - 1, (T_total,), device=device), tokens)
        focus_map = torch.ones(T_total, dtype=torch.bool, device=device)
        H_prev, mask_prev, logits, _traces = model.reasoning_loop(noisy, model.cfg.K_inner, focus_map, mask_prev, H_prev)
        logits[-1] = float("-inf")
        tokens = torch.multinomial(logits.softmax(dim=-1), num_samples=T_total, replacement=True)
        tokens[:T0] = prompt_tokens.to(device)
        for fn in region_generators or ():
            tokens = fn(tokens)
    return tokens
