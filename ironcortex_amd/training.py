"""Forward-forward train_step with the reference's signature and 15-key metrics
(reference training.py:22-273).  The per-region Python loops with .item() syncs are replaced
by batched device ops; the update order is the reference's: backward -> grad norm -> Hebbian
edge nudges -> homeostasis -> optimizer.step."""
from __future__ import annotations

import os
import random
import weakref
from dataclasses import dataclass
from typing import Dict, List

import torch
import torch.nn.functional as F

from .constants import EPS_DIV
from .corruptions import corrupt
from .energy import ff_energy_loss
from .hip import ops
from .model import CortexReasoner


@dataclass
class LossWeights:
    ff: float = 1.0
    rtd: float = 0.5
    denoise: float = 1.0
    critic: float = 0.25
    verify: float = 0.25
    facet: float = 1e-3


def make_negatives(clean_tokens: torch.Tensor, V: int, host: torch.Tensor = None, keep_host: bool = False):
    """Corrupt every sequence (training.py:70-89).  Random draws happen on the CPU generator in
    the reference's order; only the results travel to the device (keep_host: they stay on the host for the caller to
    send; host: the caller's host copy of clean_tokens)."""
    dev = clean_tokens.device
    host = clean_tokens.cpu() if host is None else host
    neg, focus, targets, masks, real = [], [], [], [], []
    for b in range(host.shape[0]):
        mode = random.choices(["RTD", "SPAN", "BLOCK"], weights=[0.5, 0.3, 0.2])[0]
        x_neg, is_real, fm, d_t, d_m = corrupt(host[b], V, mode)
        neg.append(x_neg); focus.append(fm); targets.append(d_t); masks.append(d_m)
        real.append(bool(is_real.all()))
    if keep_host:
        return (torch.stack(neg), torch.stack(focus), torch.stack(targets), torch.stack(masks), real)
    return (torch.stack(neg).to(dev), torch.stack(focus).to(dev), torch.stack(targets), torch.stack(masks), real)


# IC_SPLIT=1: the corrupted half of a step's batch is drawn and mixed while the device runs the clean half's loops
# (reasoning_loop_batch(second=...)).  Off by default: measured on the 10m bench config (same box, 30 steps, twice each)
# 13.56 / 13.61 ms per step with it against 13.39 / 13.38 without -- the loops start 0.4 ms earlier, but the second
# token-mixer pass sits between the two halves of the loops (0.25 ms) and its backward is a second set of launches (0.3 ms).
SPLIT = os.environ.get("IC_SPLIT", "0") == "1"


_MODULES = weakref.WeakKeyDictionary()  # model -> tuple of its modules (the tree is fixed after construction)


_PARAMS = weakref.WeakKeyDictionary()  # model -> (number of registered parameters, tuple of them)


def _parameters(model):
    """tuple(model.parameters()) without walking the module tree on every step (0.2 ms); rebuilt
    when a module's parameter registry changed size (FlatParams re-binds keep the count)."""
    mods = _MODULES.get(model)
    if mods is None:
        mods = _MODULES[model] = tuple(model.modules())
    n = 0
    for mod in mods:
        n += len(mod._parameters)
    c = _PARAMS.get(model)
    if c is None or c[0] != n or any(q is not r for q, r in zip(c[1][:4], model.parameters())):
        c = _PARAMS[model] = (n, tuple(model.parameters()))
    return c[1]


def _ensure_training(model) -> None:
    """model.train(), skipped when every module is already in training mode: train() walks the
    tree and re-assigns the flag through nn.Module.__setattr__ on ~220 modules (0.3 ms per step)."""
    mods = _MODULES.get(model)
    if mods is None:
        mods = _MODULES[model] = tuple(model.modules())
    for mod in mods:
        if not mod.training:
            model.train()
            return


def train_step(model: CortexReasoner, optimizer: torch.optim.Optimizer, clean_tokens: torch.Tensor,
               λ: LossWeights, device: torch.device, dp=None) -> Dict[str, float]:
    """One FF training step.  dp (ironcortex_amd.parallel.DataParallel, optional) shards the batch
    over ranks: clean_tokens is then this rank's shard and the one exchange per step happens
    between backward and the updates."""
    _ensure_training(model)
    model.detach_state()
    cfg = model.cfg
    B, T = clean_tokens.shape
    step = getattr(model, "_train_step", 0) + 1
    model._train_step = step
    for reg in model.regions:
        reg.global_step = step
    lam_s = 0.0
    if cfg.enable_ff_energy_alignment:
        warm = min(1.0, step / cfg.surprise_lambda_schedule) if cfg.surprise_lambda_schedule > 0 else 1.0
        lam_s = cfg.surprise_lambda * warm
    model._lambda_s = lam_s
    zero = torch.zeros((), device=device)
    host_tokens = clean_tokens.cpu()  # (before anything is queued: the corruptions below are drawn on the host)
    neg = {}

    def negatives():
        """The corrupted sequences and every other host -> device copy of the step.  With the device-steered loops this
        runs AFTER the loops of the clean sequences have been queued (reasoning_loop_batch(second=...)), so the host
        work overlaps with them; the copies therefore go through pinned memory and do not wait for the stream (a copy
        from pageable memory does, and one issued behind the loops would hold the host until they finish)."""
        neg_tokens, focus_neg, d_targets, d_masks, is_real = make_negatives(clean_tokens, model.V, host_tokens,
                                                                            keep_host=clean_tokens.is_cuda)
        up = (lambda t: t.pin_memory().to(device, non_blocking=True)) if clean_tokens.is_cuda else (lambda t: t.to(device))
        neg["tokens"], neg["focus"] = (neg_tokens, focus_neg) if neg_tokens.device == clean_tokens.device else (up(neg_tokens), up(focus_neg))
        neg["rtd_target"] = up(~torch.tensor(is_real)).long()
        rows_ = [b for b in range(B) if bool(d_masks[b].any())]
        neg["d_rows"] = rows_
        if rows_:
            first = [int(d_masks[b].nonzero(as_tuple=False)[0].item()) for b in rows_]
            neg["d_pack"] = up(torch.stack([torch.stack([d_targets[b][i] for b, i in zip(rows_, first)]).long(), torch.tensor(rows_)]))
        return neg["tokens"], neg["focus"]

    # no loss below back-propagates through the inner-step logits / workspace (training.py:120-231):
    # the loops may run on the LoopEngine (one autograd node per loop)
    model.engine.new_step()
    model._logits_detached = True
    model.router.hold_bias = True  # every loop of this step feeds the one backward pass below
    try:
        if SPLIT:
            focus_pos = torch.zeros(B, T, dtype=torch.bool, device=clean_tokens.device)
            H_all, reg_all, logits_all, _tr, energy_all, entropy_all = model.reasoning_loop_batch(clean_tokens, cfg.K_inner, focus_pos, second=negatives)
        else:
            neg_tokens, focus_neg = negatives()
            tokens_all = torch.cat([clean_tokens, neg_tokens], dim=0)
            focus_all = torch.cat([torch.zeros_like(focus_neg, dtype=torch.bool), focus_neg], dim=0)
            H_all, reg_all, logits_all, _tr, energy_all, entropy_all = model.reasoning_loop_batch(tokens_all, cfg.K_inner, focus_all)
    finally:
        model._logits_detached = False
        model.router.hold_bias = False
        model.router._bias_cache = None
    # (reference training.py:53 clears the gradients first; nothing above reads a .grad, so the 0.13 ms of host work are
    # spent here, behind the loops' launches, instead of in front of them)
    optimizer.zero_grad()
    rtd_target, d_rows, d_pack = neg["rtd_target"], neg["d_rows"], neg.get("d_pack")
    H_pos, H_neg = H_all[:B], H_all[B:]
    mask_p, mask_n = reg_all[:B], reg_all[B:]
    logits_pos, logits_neg = logits_all[:B], logits_all[B:]
    ce_loss = F.cross_entropy(logits_pos, clean_tokens[:, -1])
    surprise = model.bank.surprise[0]  # [R]
    aux_pos = aux_neg = None
    if cfg.enable_ff_energy_alignment:
        ms = surprise.mean().expand(B)
        aux_pos = torch.stack([energy_all[:B], ms, entropy_all[:B]], dim=-1)
        aux_neg = torch.stack([energy_all[B:], ms, entropy_all[B:]], dim=-1)

    # ---- FF per-region losses on the final states (training.py:120-156), batched over regions
    ff_loss = zero
    gains = None
    ff_stats = e_stats = heb_gather = fired_any = None  # (data parallel: collectives in flight, see below)
    if cfg.enable_forward_forward:
        gpos = ops.GoodnessFn.apply(H_pos) - lam_s * surprise  # [B,R]
        gneg = ops.GoodnessFn.apply(H_neg) - lam_s * surprise
        tau = torch.stack([ff.tau for ff in model.reg_ff])
        ff_loss = ff_loss + (F.softplus(-(gpos - tau)) + F.softplus(gneg - tau)).mean(0).sum()
        with torch.no_grad():
            target = gpos.mean(0)
            gains = (gpos - gneg).mean(0)
            prec = (model.bank.var[0] + EPS_DIV).reciprocal().mean(-1) if cfg.tau_kappa != 0.0 else None

            def apply_ff_stats(target, gains, prec):
                if prec is not None:
                    target = target + cfg.tau_kappa * (prec - cfg.tau_target_prec)
                new_tau = (1 - 0.01) * tau + 0.01 * target
                for r, ff in enumerate(model.reg_ff):
                    ff.tau = new_tau[r]
                model.gate.gain_ema.mul_(0.9).add_((1.0 - 0.9) * gains)

            if dp is not None:  # one collective for the three (region state is rank-local, tau is replicated state: the
                # precision that moves it is the ranks' mean), started here and picked up after the backward pass:
                # tau and gain_ema are next used by the NEXT step (this step's losses hold their own copy of tau)
                ff_stats = dp.mean_many_async([target, gains] + ([prec] if prec is not None else []))
            else:
                apply_ff_stats(target, gains, prec)

    # ---- RTD on the negative stream's motor state
    motor = model.io_idxs["motor"]
    motor_pos, motor_neg = H_pos[:, motor, :], H_neg[:, motor, :]
    rtd_logits = model.rtdd(motor_neg)
    rtd_loss = F.cross_entropy(rtd_logits, rtd_target)
    rtd_acc = (rtd_logits.argmax(dim=-1) == rtd_target).float().mean()

    # ---- denoising (mask-predict): first masked position of each negative (training.py:170-186);
    # the token head runs once for all rows instead of once per sample
    denoise_loss = denoise_acc = zero
    if d_rows:
        target, sel = d_pack[0], d_pack[1]
        logits_d, facet = model.lm_head.forward_rows(motor_neg[sel])
        denoise_loss = (F.cross_entropy(logits_d, target, reduction="sum") + λ.facet * facet.sum()) / max(1, B)
        denoise_acc = (logits_d.argmax(dim=-1) == target).float().mean()

    # ---- critic regression
    critic_loss = zero
    if not cfg.train_deterministic_inner_loop:
        realized = (H_pos.pow(2).mean(dim=-1).mean(dim=-1) - H_neg.pow(2).mean(dim=-1).mean(dim=-1)).detach()
        on = mask_n.to(H_neg.dtype)
        ws_neg = (H_neg * on.unsqueeze(-1)).sum(1) / on.sum(1, keepdim=True).clamp_min(1.0)
        _p, g = model.plan(ws_neg, motor_neg)
        critic_loss = F.mse_loss(model.critic(ws_neg, g), realized)

    # ---- verifier energy FF loss
    if model.verify is not None and cfg.enable_energy_verifier:
        y_pos = F.one_hot(clean_tokens[:, -1], num_classes=model.V).float()
        y_neg = F.softmax(logits_neg.detach(), dim=-1)
        tau_v = model.verify_state.tau if (cfg.enable_ff_energy_alignment and model.verify_state is not None) else 0.0
        E_pos = model.verify(motor_pos.detach(), y_pos, aux=aux_pos)
        E_neg = model.verify(motor_neg.detach(), y_neg, aux=aux_neg)
        verifier_loss = ff_energy_loss(E_pos, E_neg, tau=tau_v)
        if cfg.enable_ff_energy_alignment and model.verify_state is not None:
            Ep, En = E_pos.mean().detach(), E_neg.mean().detach()

            def apply_e_stats(Ep, En):
                model.E_pos_mean = 0.99 * model.E_pos_mean + 0.01 * Ep
                model.E_neg_mean = 0.99 * model.E_neg_mean + 0.01 * En
                model.verify_state.update_tau(0.5 * (model.E_pos_mean + model.E_neg_mean))

            if dp is not None:
                e_stats = dp.mean_many_async([Ep, En])
            else:
                apply_e_stats(Ep, En)
    else:
        E_pos, E_neg = torch.zeros(B, device=device), torch.zeros(B, device=device)
        verifier_loss = zero

    total = λ.ff * ff_loss + λ.rtd * rtd_loss + λ.denoise * denoise_loss + λ.critic * critic_loss + λ.verify * verifier_loss
    coord = model._coord
    if dp is not None:
        dp.attach(model)  # None-ness of every gradient made rank-consistent inside the backward's one host read
        # what the updates after the backward pass need from the other ranks only depends on the forward: asked for now,
        # it travels while the backward runs
        with torch.no_grad():
            if cfg.enable_forward_forward and model.router.E > 0:
                heb_gather = dp.gather_async(torch.cat([H_pos.detach(), mask_p.to(H_pos.dtype).unsqueeze(-1)], dim=-1))
            fired_any = dp.any_async(mask_n.any(dim=0))
    coord.defer = True  # the backward's one host read waits until the updates below have been enqueued behind it
    try:
        total.backward()
        with torch.no_grad():
            if ff_stats is not None:  # (data parallel: the means asked for before the pass)
                got = ff_stats()
                gains = got[1]
                apply_ff_stats(got[0], got[1], got[2] if len(got) > 2 else None)
            if e_stats is not None:
                apply_e_stats(*e_stats())
            # Hebbian nudges on edges whose two ends were active in a positive sample (training.py:147-156,240-243)
            if gains is not None and model.router.E > 0:
                router = model.router
                t = router._edge_tables(model.reg_coords)
                Hh = H_pos.detach()
                mp = mask_p.to(H_pos.dtype)
                if dp is not None:  # every rank applies the whole global batch's nudges (states and masks in one gather)
                    both = heb_gather()
                    Hh, mp = both[..., :-1].contiguous(), both[..., -1].contiguous()
                coef = (1e-3 * gains.clamp_min(0.0))[router.dst_idx].unsqueeze(0) * mp[:, router.dst_idx] * mp[:, router.src_idx]
                if router.edge_transform_mode != "per_edge":
                    # the reference indexes router.W_edge here, which only per_edge mode defines (gate.py:136-137): it raises
                    # as soon as ONE update is due (positive gain, region and a neighbour active in a positive sample) --
                    # after the backward and the gain / tau updates, before homeostasis and the optimizer step.  Mirrored.
                    if bool(((gains > 0)[router.dst_idx].unsqueeze(0) & (mp[:, router.dst_idx] > 0) & (mp[:, router.src_idx] > 0)).any()):
                        raise AttributeError("'Router' object has no attribute 'W_edge' (edge_transform_mode="
                                             f"{router.edge_transform_mode!r}; reference training.py:240-243)")
                else:
                    ops.hebbian_rank1(router._flat.stacked("W"), t["dst_of_edge"], t["src"], t["widx"], coef, Hh)
                    if model._units is not None:  # bench.py's device-side unit counters: edges this launch updates
                        model._units[14:16] += torch.stack([torch.ones((), device=coef.device), (coef != 0).any(0).sum().float()])
            active_frac = mask_n.float().mean()
            fired = mask_n.any(dim=0)
            model.gate.update_homeo(fired_any() if dp is not None else fired)
            scalars = [t.detach().float().reshape(()) for t in (
                ff_loss, rtd_loss, denoise_loss, critic_loss, verifier_loss, E_pos.mean(), E_neg.mean(), rtd_acc,
                denoise_acc, active_frac, ce_loss, total)]
    finally:
        coord.defer = False
        coord.publish()  # (also on the way out of the mirrored AttributeError: the gradients exist by then, as in the reference)
        if dp is not None:
            dp.detach(model)
    model.finalize_grads()
    if dp is not None:
        dp.sync_grads(model)
    with torch.no_grad():
        grads = [p.grad for p in _parameters(model) if p.grad is not None]
        grad_norm = torch.stack(torch._foreach_norm(grads)).pow(2).sum().sqrt() if grads else zero
    optimizer.step()
    if cfg.profile and cfg.profile_every_n_steps > 0 and step % cfg.profile_every_n_steps == 0:
        model.report_profile()
    model.engine.prefetch_open()
    vals = torch.stack(scalars[:9] + [grad_norm.float().reshape(())] + scalars[9:]).tolist()  # one host sync for all metrics
    model.engine.resolve_open()  # (device-steered loops: their host-side record, now that the step has been waited for)
    keys = ("ff", "rtd", "denoise", "critic", "verify", "E_pos", "E_neg", "rtd_acc", "denoise_acc", "grad_norm",
            "active_frac", "ce_last", "total")
    out = dict(zip(keys, vals))
    out["E_gap"] = out["E_pos"] - out["E_neg"]
    out["lambda_s"] = float(lam_s)
    return out
