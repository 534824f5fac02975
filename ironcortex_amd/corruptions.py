"""Negative-stream corruptions for forward-forward training (reference corruptions.py:8-69).
Host-side RNG bookkeeping: the random draws are made in the reference's order on the device
the tokens live on, so a CPU run consumes the CPU generator exactly as the reference does."""
from typing import Tuple

import torch


def corrupt(tokens: torch.Tensor, V: int, mode: str) -> Tuple[torch.Tensor, ...]:
    """-> (x_neg [T], is_real [T] in {0,1}, focus bool [T], denoise_targets [T], denoise_mask bool [T])"""
    T, dev = tokens.shape[0], tokens.device
    if mode == "RTD":
        hit = torch.rand(T, device=dev) < 0.15
        x_neg = torch.where(hit, torch.randint(0, V, (T,), device=dev), tokens)
        is_real = (x_neg == tokens).long()
    elif mode == "SPAN":
        start = torch.randint(0, T, (1,), device=dev, dtype=torch.float32)
        length = torch.poisson(torch.tensor([5.0], device=dev)).clamp(min=1)
        end = torch.clamp(start + length, max=float(T))
        pos = torch.arange(T, device=dev, dtype=torch.float32)
        hit = (pos >= start) & (pos < end)
        x_neg = torch.where(hit, torch.full_like(tokens, V - 1), tokens)
        is_real = (~hit).long()
    elif mode == "BLOCK":
        n = max(2, T // 8)
        start = torch.randint(0, T - n + 1, (1,), device=dev)
        pos = torch.arange(T, device=dev)
        hit = (pos >= start) & (pos < start + n)
        block = hit.nonzero(as_tuple=False).squeeze(-1)
        x_neg = tokens.clone()
        x_neg[block] = x_neg[block[torch.randperm(n, device=dev)]]
        is_real = (x_neg == tokens).long()
    else:
        raise ValueError("unknown mode")
    return x_neg, is_real, hit, tokens, hit
