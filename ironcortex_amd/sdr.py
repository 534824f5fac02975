"""Fixed k-hot token codes (reference sdr.py:6-29): a gather from a [V,d] binary table."""
import torch
import torch.nn as nn


class SparseTokenEncoder(nn.Module):
    def __init__(self, V: int, d: int, k: int) -> None:
        super().__init__()
        if k > d:
            raise ValueError("k must be <= d")
        self.V, self.d, self.k = V, d, k
        table = torch.zeros(V, d)
        for token in range(V):  # same RNG consumption as the reference: one randperm(d) per token
            table[token, torch.randperm(d)[:k]] = 1.0
        self.register_buffer("codes", table)

    def forward(self, tokens: torch.Tensor) -> torch.Tensor:
        return self.codes[tokens]
