"""Checkpoint save/load with the reference's file layout and key names (checkpoint.py:13-53):
{"model": state_dict, "optimizer": state_dict, "step": int}; loaded with weights_only=True."""
from __future__ import annotations

import os
from typing import Optional

import torch


def save_checkpoint(model, optimizer, step: int, path: str) -> None:
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}  # views -> own storage
    torch.save({"model": sd, "optimizer": optimizer.state_dict() if optimizer is not None else None,
                "step": int(step)}, path)


def load_checkpoint(model, optimizer, path: str, device: Optional[torch.device] = None) -> int:
    ckpt = torch.load(path, map_location=device, weights_only=True)
    model.load_state_dict(ckpt["model"])
    if optimizer is not None and ckpt.get("optimizer") is not None:
        optimizer.load_state_dict(ckpt["optimizer"])
    return int(ckpt.get("step", 0))
