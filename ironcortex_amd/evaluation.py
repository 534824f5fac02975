"""evaluate_perplexity with the reference's definition and return value (evaluation.py:14-52):
teacher forcing over every prefix length t = 1..T-1; the whole [B, t] batch of prefixes goes
through one reasoning_loop_batch call with an all-False focus map (so the sensor vector is zero and
the gate gets no focus boost), the next-token NLL is summed over the batch.  The region fast-weight
state carries over from call to call exactly as in the reference, which is why the
prefix-outer / batch-inner order is part of the definition."""
from __future__ import annotations

import math
from typing import Dict

import torch
import torch.nn.functional as F


@torch.no_grad()
def evaluate_perplexity(model, tokens: torch.Tensor, device: torch.device) -> Dict[str, float]:
    model.eval()
    tokens = tokens.to(device)
    B, T = tokens.shape
    if T < 2:
        raise ValueError("tokens must have length >= 2 for evaluation")
    ce_sum = torch.zeros((), dtype=torch.float64, device=device)  # one host read at the end, not T-1
    n_pred = 0
    for t in range(1, T):
        focus = torch.zeros(B, t, dtype=torch.bool, device=device)
        _, _, logits, _, _, _ = model.reasoning_loop_batch(tokens[:, :t], model.cfg.K_inner, focus)
        ce_sum += F.nll_loss(F.log_softmax(logits, dim=-1), tokens[:, t], reduction="sum").double()
        n_pred += B
    ce = float(ce_sum.item()) / max(1, n_pred)
    return {"cross_entropy": ce, "perplexity": float(math.exp(ce))}
