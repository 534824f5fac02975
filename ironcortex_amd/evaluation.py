"""evaluate_perplexity with the reference's definition (evaluation.py:14-52): for each prefix
length t the model reads tokens[:t] with the last position in focus and is scored on token t."""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F


@torch.no_grad()
def evaluate_perplexity(model, tokens: torch.Tensor, device: torch.device) -> float:
    model.eval()
    tokens = tokens.to(device)
    if tokens.dim() == 1:
        tokens = tokens.unsqueeze(0)
    nll, n = 0.0, 0
    for seq in tokens:
        for t in range(1, seq.shape[0]):
            prefix = seq[:t].unsqueeze(0)
            focus = torch.zeros_like(prefix, dtype=torch.bool)
            focus[:, -1] = True
            _H, _m, logits, *_ = model.reasoning_loop_batch(prefix, model.cfg.K_inner, focus)
            nll += float(F.cross_entropy(logits, seq[t].view(1)).item())
            n += 1
    return math.exp(nll / max(1, n))
