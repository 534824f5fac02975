"""Checkpoint save/load with the reference's signatures, file layout and key names
(checkpoint.py:13-53): {"model": state_dict, "optimizer": state_dict, "step": int}; loaded with
weights_only=True.  Call sites: train_tiny_shakespeare.py:95 `load_checkpoint(path, model,
optimizer, map_location=device)`, :160 `save_checkpoint(model, optimizer, step, path)`."""
from __future__ import annotations

import os
from typing import Optional, Union

import torch


def save_checkpoint(model, optimizer: Optional[torch.optim.Optimizer], step: int, path: Union[str, os.PathLike]) -> None:
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}  # arena views -> own storage
    torch.save({"model": sd, "optimizer": optimizer.state_dict() if optimizer is not None else None,
                "step": step}, path)


def load_checkpoint(path: Union[str, os.PathLike], model=None, optimizer: Optional[torch.optim.Optimizer] = None,
                    map_location: Union[str, torch.device, None] = None) -> int:
    ckpt = torch.load(path, map_location=map_location, weights_only=True)
    if model is not None:
        model.load_state_dict(ckpt["model"])
    if optimizer is not None and ckpt.get("optimizer") is not None:
        optimizer.load_state_dict(ckpt["optimizer"])
    return int(ckpt.get("step", 0))
