"""Flat parameter arenas.

The HIP kernels index region / edge weights as stacked tensors ([R,4,d,d], [E,d,d], ...) while
the reference API (and its checkpoints: regions.3.k_lin.weight, router.W_edge.1->0.weight, ...)
exposes one nn.Parameter per matrix.  Both hold: every Parameter is a VIEW into one contiguous
fp32 arena, laid out so each group of parameters forms a stacked tensor.  Gradients live in a
twin arena; a parameter's .grad is a view into it (or None when the parameter took no part in
the step, which is what AdamW must see to match the reference's skip behaviour).
"""
from __future__ import annotations

import math
from typing import Dict, List, Sequence, Tuple

import torch
import torch.nn as nn


class ArenaParameter(nn.Parameter):
    """A Parameter whose `.data = t` assignment COPIES into the arena when t has the same shape,
    dtype and device (so `lin.weight.data = torch.eye(d)` keeps the arena layout); anything
    else (Module._apply moving devices) re-points the storage as usual and the owner re-flattens."""

    @property
    def data(self):  # type: ignore[override]
        return torch.Tensor.data.__get__(self)

    @data.setter
    def data(self, value):
        cur = torch.Tensor.data.__get__(self)
        if (isinstance(value, torch.Tensor) and value.shape == cur.shape and value.dtype == cur.dtype
                and value.device == cur.device and getattr(self, "_ic_pinned", False)):
            if value.data_ptr() != cur.data_ptr():
                cur.copy_(value)
            return
        torch.Tensor.data.__set__(self, value)

    def __deepcopy__(self, memo):
        out = ArenaParameter(torch.Tensor.data.__get__(self).clone(), self.requires_grad)
        memo[id(self)] = out
        return out


# torch.optim picks its multi-tensor (foreach) kernels only when every parameter's exact type is in
# this list; without the entry AdamW falls back to ~8 launches PER PARAMETER (40 ms per train_step
# for the 10M config).  ArenaParameter is a plain dense fp32 tensor, so the foreach path applies.
for _mod in ("torch.optim.optimizer", "torch.utils._foreach_utils"):
    try:
        _types = __import__(_mod, fromlist=["_foreach_supported_types"])._foreach_supported_types
        if ArenaParameter not in _types:
            _types.append(ArenaParameter)
    except (ImportError, AttributeError):  # private list moved: the optimizer still works, just slower
        pass


def arena_param(shape: Sequence[int], fill: float = 0.0) -> ArenaParameter:
    return ArenaParameter(torch.full(tuple(shape), float(fill)))


class FlatParams:
    """Owns the arena of a set of parameter groups.

    groups: name -> list of Parameters with identical shape; group `name` is exposed as a stacked
    view of shape [len(list), *shape] via stacked(name) / stacked_grad(name).
    """

    def __init__(self, groups: Dict[str, List[nn.Parameter]]):
        self.groups = {k: list(v) for k, v in groups.items() if len(v) > 0}
        self.layout: Dict[str, Tuple[int, int, Tuple[int, ...]]] = {}
        off = 0
        for name, plist in self.groups.items():
            shape = tuple(plist[0].shape)
            numel = int(math.prod(shape))
            for p in plist:
                if tuple(p.shape) != shape:
                    raise ValueError(f"group {name}: mixed shapes")
            off = (off + 63) // 64 * 64  # 256-byte alignment of every group
            self.layout[name] = (off, numel, shape)
            off += numel * len(plist)
        self.total = off
        self.flat: torch.Tensor = None  # type: ignore
        self.gflat: torch.Tensor = None  # type: ignore
        self._gviews, self._gviews_of = {}, None
        self.reflat(force=True)

    # ------------------------------------------------------------------ storage
    def _all(self):
        for name, plist in self.groups.items():
            off, numel, shape = self.layout[name]
            for i, p in enumerate(plist):
                yield p, off + i * numel, numel, shape

    def reflat(self, force: bool = False) -> None:
        """(Re)build the arena on the device the parameters currently live on."""
        first = next(iter(self.groups.values()))[0] if self.groups else None
        if first is None:
            self.flat = torch.zeros(0)
            return
        dev = torch.Tensor.data.__get__(first).device
        ok = (not force) and self.flat is not None and self.flat.device == dev
        if ok:
            base = self.flat.data_ptr()
            for p, off, numel, shape in self._all():
                if torch.Tensor.data.__get__(p).data_ptr() != base + 4 * off:
                    ok = False
                    break
        if ok:
            return
        flat = torch.zeros(self.total, dtype=torch.float32, device=dev)
        with torch.no_grad():
            for p, off, numel, shape in self._all():
                src = torch.Tensor.data.__get__(p)
                if src.dtype != torch.float32:
                    raise TypeError("ironcortex_amd hot-path parameters are fp32 only")
                view = flat[off:off + numel].view(shape)
                view.copy_(src.to(dev))
                p._ic_pinned = False
                torch.Tensor.data.__set__(p, view)
                p._ic_pinned = True
                p.grad = None
        self.flat = flat
        self.gflat = None
        self._gviews, self._gviews_of = {}, None

    def stacked(self, name: str) -> torch.Tensor:
        off, numel, shape = self.layout[name]
        n = len(self.groups[name])
        return self.flat[off:off + n * numel].view((n,) + shape)

    def has(self, name: str) -> bool:
        return name in self.groups

    # ------------------------------------------------------------------ gradients
    def grad_arena(self) -> torch.Tensor:
        if self.gflat is None or self.gflat.device != self.flat.device:
            self.gflat = torch.zeros_like(self.flat)
        return self.gflat

    def stacked_grad(self, name: str) -> torch.Tensor:
        off, numel, shape = self.layout[name]
        n = len(self.groups[name])
        return self.grad_arena()[off:off + n * numel].view((n,) + shape)

    def begin_pass(self) -> None:
        """Called before the first gradient of a backward pass is accumulated: slices of
        parameters whose .grad is None must start from zero."""
        g = self.grad_arena()
        params = list(self._all())
        if all(p.grad is None for p, *_ in params):
            g.zero_()
            return
        for p, off, numel, shape in params:
            if p.grad is None:
                g[off:off + numel].zero_()

    def publish(self, name: str, touched: Sequence[bool]) -> None:
        """Give .grad views to the touched members of a group (untouched keep None)."""
        g = self.grad_arena()
        views = self._gviews.get(name) if self._gviews_of is g else None
        if views is None:  # the views of one arena never change: built once (slicing costs ~2.5 us each, every step)
            if self._gviews_of is not g:
                self._gviews, self._gviews_of = {}, g
            off, numel, shape = self.layout[name]
            views = self._gviews[name] = [g[off + i * numel: off + (i + 1) * numel].view(shape) for i in range(len(self.groups[name]))]
        for p, v, t in zip(self.groups[name], views, touched):
            if t and p.grad is None and p.requires_grad:
                p.grad = v
