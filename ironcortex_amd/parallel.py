"""Data parallelism for train_step: one process per GPU, torch.distributed over RCCL/xGMI.

The reference has no distributed code; the forward-forward step shards by batch
with ONE exchange per step:
  * gradients: every parameter's gradient is packed into flat fp32 buckets (the hot-path
    arenas are already flat) and all-reduced (sum / world); a parameter whose gradient is None
    on this rank but not on another receives the averaged gradient, one that is None everywhere
    stays None so AdamW skips it exactly as in a single-process run;
  * batch statistics that drive buffer updates (goodness gains, tau targets, verifier energy
    means) are averaged, the homeostasis mask is OR-ed;
  * the positive-stream states needed by the Hebbian edge nudges are all-gathered (B*R*d
    floats), so every rank applies the same update and replicas never diverge.
Region fast-weight state stays rank-local (it is not checkpointed state).
"""
from __future__ import annotations

import os
from typing import List, Optional

import torch
import torch.distributed as dist


class DataParallel:
    def __init__(self, group: Optional["dist.ProcessGroup"] = None, bucket_mb: int = 64, overlap: bool = True):
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed is not initialised")
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.bucket_elems = bucket_mb * (1 << 20) // 4
        # overlap: the arenas' all-reduces start (asynchronously, on the backend's own stream) at the end of the
        # backward pass, as soon as the deferred weight-gradient launches are queued -- they run beside the Hebbian /
        # homeostasis work, the flag exchange and the host side of publishing the gradients; sync_grads() only waits
        self.overlap = overlap and os.environ.get("IC_DP_OVERLAP", "1") != "0"
        self._inflight = None

    # ------------------------------------------------------------------ small helpers
    def mean(self, t: torch.Tensor) -> torch.Tensor:
        out = t.detach().clone().contiguous()
        dist.all_reduce(out, op=dist.ReduceOp.SUM, group=self.group)
        return out / self.world

    def mean_many(self, ts: List[torch.Tensor]) -> List[torch.Tensor]:
        """mean() of several float tensors in ONE all-reduce (a small collective costs its latency, not its bytes)."""
        flat = torch.cat([t.detach().reshape(-1) for t in ts])
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
        flat = flat / self.world
        out, off = [], 0
        for t in ts:
            out.append(flat[off:off + t.numel()].view(t.shape))
            off += t.numel()
        return out

    # asynchronous forms: the collective is queued on the backend's stream (behind what the current stream has queued so
    # far) and the caller goes on; result() waits for it (the current stream does, not the host) and finishes the value.
    # train_step issues them as soon as their inputs exist and picks the results up after the backward pass, so that
    # no small collective -- a latency and a meeting point of the ranks -- sits in the main stream's chain.
    def mean_many_async(self, ts: List[torch.Tensor]):
        flat = torch.cat([t.detach().reshape(-1) for t in ts])
        work = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        shapes = [(t.numel(), t.shape) for t in ts]

        def result() -> List[torch.Tensor]:
            work.wait()
            mean, out, off = flat / self.world, [], 0
            for n, shp in shapes:
                out.append(mean[off:off + n].view(shp))
                off += n
            return out
        return result

    def gather_async(self, t: torch.Tensor):
        t = t.detach().contiguous()
        parts = [torch.empty_like(t) for _ in range(self.world)]
        work = dist.all_gather(parts, t, group=self.group, async_op=True)

        def result() -> torch.Tensor:
            work.wait()
            return torch.cat(parts, dim=0)
        return result

    def any_async(self, t: torch.Tensor):
        out = t.to(torch.int32).contiguous()
        work = dist.all_reduce(out, op=dist.ReduceOp.MAX, group=self.group, async_op=True)

        def result() -> torch.Tensor:
            work.wait()
            return out.bool()
        return result

    def any(self, t: torch.Tensor) -> torch.Tensor:
        out = t.to(torch.int32).contiguous()
        dist.all_reduce(out, op=dist.ReduceOp.MAX, group=self.group)
        return out.bool()

    def gather(self, t: torch.Tensor) -> torch.Tensor:
        """Concatenate equal-shaped per-rank tensors along dim 0 (rank order)."""
        t = t.detach().contiguous()
        parts = [torch.empty_like(t) for _ in range(self.world)]
        dist.all_gather(parts, t, group=self.group)
        return torch.cat(parts, dim=0)

    def broadcast_module(self, module: torch.nn.Module, src: int = 0) -> None:
        """Make replicas identical at start (parameters and persistent buffers): one broadcast per
        dtype over a packed copy, not one per tensor (the reference-shaped model exposes every edge
        and region matrix as its own Parameter: ~1100 tensors at R=36)."""
        by_dtype = {}
        for t in list(module.parameters()) + list(module.buffers()):
            t = t.data if isinstance(t, torch.nn.Parameter) else t
            if t.numel() > 0:
                by_dtype.setdefault((t.dtype, t.device), []).append(t)
        for (dtype, _dev), ts in sorted(by_dtype.items(), key=lambda kv: str(kv[0])):
            wire = torch.uint8 if dtype == torch.bool else dtype
            flat = torch.cat([t.reshape(-1).to(wire) for t in ts])
            dist.broadcast(flat, src=src, group=self.group)
            views, off = [], 0
            for t in ts:
                views.append(flat[off:off + t.numel()].view(t.shape).to(dtype))
                off += t.numel()
            torch._foreach_copy_(ts, views)

    def any_flags(self, flags: torch.Tensor) -> torch.Tensor:
        """Element-wise OR of a uint8 flag vector across ranks (GradCoordinator: which matrices / parameters
        received a gradient on ANY rank), as one all-reduce; no host read here."""
        out = flags.to(torch.int32).contiguous()
        dist.all_reduce(out, op=dist.ReduceOp.MAX, group=self.group)
        return out

    # ------------------------------------------------------------------ gradients
    def attach(self, model) -> None:
        """Hook this exchange into the model's backward pass: the per-pass touched flags (and the
        has-a-gradient flags of the parameters outside the arenas) are OR-ed across ranks inside the
        GradCoordinator's single end-of-pass host read, so None-ness is rank-consistent before
        sync_grads() runs and needs no host read of its own."""
        coord = model._coord
        coord.dp = self
        self._inflight = None  # (a pass that raised leaves nothing behind for the next one)
        coord.extra_params = lambda: model.grad_layout()[1]
        coord.after_launch = (lambda: self.start_arena_reduce(model)) if (self.overlap and hasattr(model, "grad_layout")) else None

    @staticmethod
    def detach(model) -> None:
        model._coord.dp = None
        model._coord.after_launch = None

    def start_arena_reduce(self, model) -> None:
        """End of the backward pass (GradCoordinator, after every deferred launch and per-pass reduction has been
        queued: nothing writes the gradient arenas afterwards): start their all-reduces without waiting for them.
        Same collectives in the same order on every rank (the arenas exist on all ranks or on none)."""
        arenas, _ = model.grad_layout()
        self._inflight = [(fp.gflat, dist.all_reduce(fp.gflat, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
                          for fp in arenas if fp.gflat is not None and fp.gflat.numel() > 0]

    def sync_grads(self, model) -> None:
        """Average the gradients over ranks.  The hot-path gradients (region bank, router: > 90 % of the
        bytes) are all-reduced IN PLACE in their flat arenas -- no packing, no copy back; the remaining
        parameters (token mixer, heads, norms: a few MB) go through one packed bucket per bucket_mb.
        Needs attach(model) before the backward pass (train_step does both)."""
        if hasattr(model, "grad_layout"):
            arenas, extra = model.grad_layout()
            inflight, self._inflight = self._inflight, None
            if inflight is not None:  # started at the end of the backward pass (overlap)
                for g, work in inflight:
                    work.wait()
                    g.div_(self.world)
            else:
                for fp in arenas:
                    if fp.gflat is not None and fp.gflat.numel() > 0:
                        dist.all_reduce(fp.gflat, op=dist.ReduceOp.SUM, group=self.group)
                        fp.gflat.div_(self.world)
            live = [p for p in extra if p.grad is not None]  # rank-consistent (GradCoordinator._end)
            self._reduce_packed(live)
            return
        self._sync_grads_generic(model)

    def _reduce_packed(self, live: List[torch.nn.Parameter]) -> None:
        i = 0
        while i < len(live):
            j, n = i, 0
            while j < len(live) and (n == 0 or n + live[j].numel() <= self.bucket_elems):
                n += live[j].numel()
                j += 1
            flat = torch.cat([p.grad.reshape(-1) for p in live[i:j]])
            dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
            flat.div_(self.world)
            views, off = [], 0
            for p in live[i:j]:
                k = p.numel()
                views.append(flat[off:off + k].view_as(p.grad))
                off += k
            torch._foreach_copy_([p.grad for p in live[i:j]], views)  # one multi-tensor launch
            i = j

    def _sync_grads_generic(self, module: torch.nn.Module) -> None:
        """Any nn.Module (no arenas): None-ness made consistent with one flag exchange (a host read)."""
        params: List[torch.nn.Parameter] = [p for p in module.parameters() if p.requires_grad]
        if not params:
            return
        dev = params[0].device
        have = torch.tensor([0 if p.grad is None else 1 for p in params], dtype=torch.int32, device=dev)
        dist.all_reduce(have, op=dist.ReduceOp.MAX, group=self.group)
        have_l = have.tolist()
        live = [p for p, h in zip(params, have_l) if h]
        for p in live:
            if p.grad is None:
                p.grad = torch.zeros_like(p)
        self._reduce_packed(live)
