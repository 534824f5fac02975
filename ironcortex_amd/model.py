"""CortexReasoner: the reference's container and inner loop (model.py:33-503) with the
gate -> router -> region-cell phase executed by the HIP library.

Same constructor, same method signatures and return tuples, same state_dict keys.  Inside
forward_inner_step the three hot blocks are three fused calls (Gate.score_select,
Router.route, RegionBank.phase) instead of a Python loop over regions.
"""
from __future__ import annotations

import time
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F

from .config import CortexConfig
from .constants import EPS_DIV
from .energy import EnergyVerifierHead
from .engine import DeferredMask, LoopEngine
from .flat import FlatParams
from .gate import Gate, Router
from .heads import CriticHead, PlannerHead, RTDHead, TokenHead_MFS, Workspace
from .hip import ops
from .mixer import LocalTokenMixer
from .region import RegionBank, RWKVRegionCell
from .sdr import SparseTokenEncoder
from .utils import (FastAttrs, RegionFFState, RMSNorm, context_logprob, goodness, init_weights, schedule_burst,
                    should_halt, uncertainty_from_logits)


class CortexReasoner(FastAttrs, nn.Module):
    _plain_attrs = frozenset({"_sensor_override", "_focus_any_override", "_aux_override", "_sensor_cache", "_loop_cache",
                              "_in_loop", "_zero_H", "_heads_used"})

    def __init__(self, neighbor_indices: List[List[int]], reg_coords: torch.Tensor, io_idxs: Dict[str, int],
                 cfg: CortexConfig):
        super().__init__()
        self.cfg = cfg
        self.R, self.d, self.V = cfg.R, cfg.d, cfg.V
        self.io_idxs = io_idxs
        self.neighbors = neighbor_indices
        self.register_buffer("reg_coords", reg_coords.to(torch.float32))
        self.embed = SparseTokenEncoder(self.V, self.d, cfg.sdr_k)
        self._coord = ops.GradCoordinator()
        self.regions = nn.ModuleList([
            RWKVRegionCell(
                self.d, init_decay=cfg.init_decay,
                enable_adaptive_filter_dynamics=cfg.enable_adaptive_filter_dynamics,
                enable_radial_tangential_updates=cfg.enable_radial_tangential_updates,
                afd_noise_mode=cfg.afd_noise_mode, use_predictive_trace=cfg.use_predictive_trace,
                kwta_k=cfg.kwta_k, kwta_k_start=cfg.kwta_k_start, kwta_k_schedule=cfg.kwta_k_schedule,
                kwta_soft_mode=cfg.kwta_soft_mode, kwta_soft_temp=cfg.kwta_soft_temp,
                disable_kwta=cfg.disable_kwta_during_gating, _shared_bank=True)
            for _ in range(self.R)])
        # all cells are windows onto one stacked bank (plain attribute: not a submodule)
        object.__setattr__(self, "bank", RegionBank(list(self.regions), io_idxs["sensor"], io_idxs["motor"],
                                                    coord=self._coord))
        self.gate = Gate(self.R, self.neighbors, io_idxs)
        self.router = Router(self.neighbors, self.d, self.R,
                             enable_precision_routed_messages=cfg.enable_precision_routed_messages,
                             vectorized=cfg.router_vectorized, edge_transform_mode=cfg.edge_transform_mode,
                             _coord=self._coord)
        self.rtdd = RTDHead(self.d)
        self.lm_head = TokenHead_MFS(self.d, self.V, Kf=8)
        self.region_heads = nn.ModuleList([nn.Linear(self.d, self.V) for _ in range(self.R)])
        self.work = Workspace(self.d, N_slots=8)
        if cfg.train_deterministic_inner_loop:
            self.plan = None
            self.critic = None
        else:
            self.plan = PlannerHead(self.d)
            self.critic = CriticHead(self.d)
        self.verify = None
        self.verify_state = None
        if cfg.enable_energy_verifier:
            aux_dim = 3 if cfg.enable_ff_energy_alignment else 0
            self.verify = EnergyVerifierHead(self.d, self.V, hidden=self.d, aux_dim=aux_dim)
            if cfg.enable_ff_energy_alignment:
                self.verify_state = RegionFFState()
                self.register_buffer("E_pos_mean", torch.tensor(0.0), persistent=False)
                self.register_buffer("E_neg_mean", torch.tensor(0.0), persistent=False)
        self.local_mix = LocalTokenMixer(self.d, n_head=4, block_size=cfg.max_T, m_tok=64,
                                         use_afa=cfg.enable_afa_attention)
        self.norm_in = RMSNorm(self.d)
        self.reg_ff = nn.ModuleList([RegionFFState() for _ in range(self.R)])
        self.apply(init_weights)
        # facets / region heads as stacked arenas ([Kf,V,d], [R,V,d]) for the loop-heads kernel
        object.__setattr__(self, "_heads_flat", FlatParams({
            "facet_W": [f.weight for f in self.lm_head.facets], "facet_b": [f.bias for f in self.lm_head.facets],
            "rh_W": [h.weight for h in self.region_heads], "rh_b": [h.bias for h in self.region_heads]}))
        self._profile_times = {"attention": 0.0, "routing": 0.0, "region": 0.0, "total": 0.0}
        self._profile_mem = {"attention": 0, "routing": 0, "region": 0}
        self._sensor_cache = None
        self._sensor_override = None
        self._focus_any_override = self._aux_override = None
        self.counters = {"inner_steps": 0}
        self._units = None  # device counters for bench.py (off by default)
        self._loop_cache = None
        self._in_loop = False
        self._heads_used = None
        # True while a caller (train_step) promises not to back-propagate through the inner-step
        # logits / workspace: the loop then runs on the LoopEngine (engine.py)
        self._logits_detached = False
        # "strict": reasoning_loop_batch runs the samples one after the other, region state carried from
        # sample to sample -- the reference's schedule (model.py:462-503) and the mode every parity and
        # headline number uses.  "multi": the samples are independent loops run in lockstep, every one
        # from the state the bank holds when the call starts (SURVEY §7's multi-stream mode,
        # LoopEngine.run_lockstep); each equals a batch-1 reasoning_loop from that state.
        self.stream_mode = "strict"
        object.__setattr__(self, "engine", LoopEngine(self))

    # ------------------------------------------------------------------ plumbing
    def _apply(self, fn, recurse=True):
        super()._apply(fn, recurse)
        self.bank.to_device()
        self._heads_flat.reflat()
        return self

    def grad_layout(self):
        """(gradient arenas, parameters whose gradient lives outside them) for the data-parallel exchange:
        the region bank's and the router's gradients are already flat buffers (ironcortex_amd/flat.py)."""
        c = getattr(self, "_grad_layout", None)
        n = sum(1 for _ in self.parameters())
        if c is None or c[0] != n:
            arenas = [self.bank.flat, self.router._flat]
            inside = {id(p) for fp in arenas for plist in fp.groups.values() for p in plist}
            extra = [p for p in self.parameters() if p.requires_grad and id(p) not in inside]
            c = (n, arenas, extra)
            object.__setattr__(self, "_grad_layout", c)
        return c[1], c[2]

    def count_units(self, on: bool) -> None:
        """Count region steps / active edges on the device (no sync) for throughput reports."""
        dev = self.reg_coords.device
        self._units = torch.zeros(16, device=dev) if on else None
        self.counters["inner_steps"] = 0

    def read_units(self) -> Dict[str, float]:
        self.engine.flush_units()
        u = self._units.tolist() if self._units is not None else [0.0] * 16
        # region-phase launches and, per launch, the (stream, region) pairs served and the distinct
        # regions among them (their weights are read once per launch): forward / backward
        return {"inner_steps": self.counters["inner_steps"], "region_steps": u[0], "active_edges": u[1],
                "dst_regions": u[2], "fwd_region_launches": u[3], "fwd_unique_regions": u[4],
                "bwd_region_launches": u[5], "bwd_pairs": u[6], "bwd_unique_regions": u[7],
                # router backward launches: (stream, edge) pairs, distinct edges, distinct destinations
                "bwd_router_launches": u[8], "bwd_edge_pairs": u[9], "bwd_unique_edges": u[10], "bwd_dst_regions": u[11],
                # forward router launches (the branches of an outer iteration share one): distinct active edges /
                # destinations per launch (an edge's matrices are read once for all its streams)
                "fwd_unique_edges": u[12], "fwd_unique_dst": u[13],
                # Hebbian nudge launches and the edge matrices they update (coefficient non-zero)
                "hebbian_launches": u[14], "hebbian_edges": u[15]}

    def detach_state(self) -> None:
        if hasattr(self, "work") and self.work.slots.requires_grad:
            self.work.slots = self.work.slots.detach()
        self.bank.detach_state()

    def telemetry(self) -> dict:
        b = self.bank
        var = b.var[0]
        reg_stats = [{"state_var": v, "state_prec": p, "surprise_ema": s} for v, p, s in zip(
            var.mean(-1).tolist(), (var + EPS_DIV).reciprocal().mean(-1).tolist(), b.surprise[0].tolist())]
        mean = lambda k: float(sum(r[k] for r in reg_stats) / max(1, len(reg_stats)))
        attn = getattr(self.local_mix, "attn", None)
        afa = attn.telemetry() if attn is not None and hasattr(attn, "telemetry") else None
        if self.cfg.enable_precision_routed_messages:
            self.router._weight_stats()
        out = {"regions": reg_stats, "routing_weight_mean": self.router.last_weight_mean,
               "router_weight_entropy": self.router.last_weight_entropy, "state_var_mean": mean("state_var"),
               "state_prec_mean": mean("state_prec"), "surprise_ema": mean("surprise_ema"),
               "attn_entropy_mean": afa.get("attn_entropy_mean", 0.0) if afa else 0.0}
        if self.verify_state is not None:
            out.update(E_pos_mean=float(self.E_pos_mean.detach()), E_neg_mean=float(self.E_neg_mean.detach()),
                       verifier_tau=float(self.verify_state.tau.detach()))
        if afa is not None:
            out["afa"] = afa
        return out

    def report_profile(self) -> None:
        total = self._profile_times.get("total", 0.0)
        if total <= 0:
            return
        pct = {k: self._profile_times[k] / total * 100.0 for k in ("attention", "routing", "region")}
        mb = {k: v / 1e6 for k, v in self._profile_mem.items()}
        print(f"profile time%%: attn={pct['attention']:.1f} rout={pct['routing']:.1f} reg={pct['region']:.1f}; "
              f"memMB: attn={mb['attention']:.1f} rout={mb['routing']:.1f} reg={mb['region']:.1f}")
        for k in self._profile_times:
            self._profile_times[k] = 0.0
        for k in self._profile_mem:
            self._profile_mem[k] = 0

    def region_input(self, r: int, x_sensor: torch.Tensor, msg: torch.Tensor) -> torch.Tensor:
        x = msg + x_sensor if r in (self.io_idxs["sensor"], self.io_idxs["motor"]) else msg
        return self.norm_in(x)

    def _clock(self) -> float:
        """Profile clock; the stream is drained first so the timers measure run time, not
        launch time (the reference reads perf_counter without a sync)."""
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        return time.perf_counter()

    # ------------------------------------------------------------------ sensor vector
    def sensor_vector(self, tokens: torch.Tensor, focus_map: torch.Tensor) -> torch.Tensor:
        """model.py:231-247.  The mixer sees only (tokens, focus), never the region state, and
        its dropouts have p = 0, so within one reasoning loop the result is reused."""
        if self._sensor_override is not None:  # row of a batched mixer call (reasoning_loop_batch)
            return self._sensor_override
        key = (tokens.data_ptr(), tokens._version, focus_map.data_ptr(), focus_map._version,
               torch.is_grad_enabled(), next(self.local_mix.parameters())._version)
        if self._sensor_cache is not None and self._sensor_cache[0] == key:
            return self._sensor_cache[1]
        T, dev = tokens.shape[0], tokens.device
        tok_emb = self.embed(tokens).unsqueeze(0)
        idx = torch.arange(T, device=dev, dtype=torch.float32)
        pos = torch.stack([idx, idx / (T + EPS_DIV)], dim=-1).unsqueeze(0)
        vec = self.local_mix(tok_emb, pos, focus_map.unsqueeze(0), ws_slots=None, use_dropout=self.cfg.ff_dropout)[0]
        self._sensor_cache = (key, vec, tokens, focus_map)
        return vec

    def sensor_vectors_batch(self, tokens_batch: torch.Tensor, focus_batch: torch.Tensor):
        """All sequences of a batch through the token mixer in ONE call.  The
        mixer sees only (tokens, focus), never the region state, so row b equals
        sensor_vector(tokens_batch[b], focus_batch[b]); padding anchors are masked out
        (mixer.py valid_len).  Returns (vectors [N,d], mixer energy [N], attention entropy [N] or None)."""
        N, T = tokens_batch.shape
        dev = tokens_batch.device
        tok_emb = self.embed(tokens_batch)
        pc = getattr(self, "_pos_cache", None)
        if pc is None or pc[0] != (T, dev):  # (positions depend on T alone: three launches per call otherwise)
            idx = torch.arange(T, device=dev, dtype=torch.float32)
            pc = ((T, dev), torch.stack([idx, idx / (T + EPS_DIV)], dim=-1).unsqueeze(0))
            object.__setattr__(self, "_pos_cache", pc)
        pos = pc[1].expand(N, T, 2)
        vec = self.local_mix(tok_emb, pos, focus_batch, ws_slots=None, use_dropout=self.cfg.ff_dropout)
        ent = getattr(self.local_mix.attn, "last_attn_entropy_b", None)
        return vec, self.local_mix.last_energy_b.detach(), (ent.detach() if ent is not None else None)

    # ------------------------------------------------------------------ one inner micro-step
    def forward_inner_step(self, tokens: torch.Tensor, step_k: int, focus_map: torch.Tensor, H_prev: torch.Tensor,
                           reg_mask_prev: torch.Tensor, u_mean: float, burst_extra_k: int):
        """gate -> route -> active region updates -> heads (model.py:206-321)."""
        dev = H_prev.device
        prof = self.cfg.profile
        if prof:
            t_total = self._clock()
        sensor_vec = self.sensor_vector(tokens, focus_map)
        if prof:
            t0 = self._clock()
            self._profile_times["attention"] += t0 - t_total
        # 1) gate: scores + neighbour-suppressed top-k on the device
        focus_any = focus_map.any().to(torch.uint8).reshape(1) if focus_map.numel() > 0 else torch.ones(1, dtype=torch.uint8, device=dev)
        _, mask = self.gate.score_select(H_prev.unsqueeze(0), focus_any, u_mean, self.cfg.k_active, burst_extra_k)
        # 2) router messages from the previously active regions
        if prof:
            t0 = self._clock()
        M = self.router.route(H_prev.unsqueeze(0), reg_mask_prev.to(torch.uint8).reshape(1, -1), self.reg_coords)
        if prof:
            t1 = self._clock()
            self._profile_times["routing"] += t1 - t0
        # 3) all regions in one fused phase (active: step; inactive: trace + skip)
        step_pos = float(step_k) / float(max(1, self.cfg.K_inner - 1))
        H_cur = self.bank.phase(M, sensor_vec.unsqueeze(0), self.norm_in.weight, mask, step_pos)[0]
        if prof:
            t2 = self._clock()
            self._profile_times["region"] += t2 - t1
            self._profile_times["total"] += t2 - t_total
        self.counters["inner_steps"] += 1
        if self._units is not None and self.router.E > 0:
            ea = reg_mask_prev.to(torch.float32)[self.router.src_idx]
            into = torch.zeros(self.R, device=dev).index_add_(0, self.router.dst_idx, ea)
            n_act = mask.sum().float()
            self._units[:5] += torch.stack([n_act, ea.sum(), (into > 0).sum().float(), torch.ones_like(n_act), n_act])
            self._units[12:14] += torch.stack([ea.sum(), (into > 0).sum().float()])
        # 4) heads on motor & workspace -- no host sync: the active set stays on the device
        logits, rtd_logits, ws_state, motor_state = self._heads(H_cur, mask[0])
        return H_cur, mask[0].bool(), logits, rtd_logits, ws_state, motor_state

    def _heads(self, H_cur: torch.Tensor, mask_u8: torch.Tensor):
        """Workspace write + token / RTD heads of one inner step (model.py:296-321) for H_cur [R,d]
        and the uint8 active mask [R] -> (logits [V], rtd_logits [2], ws_state [d], motor_state [d])."""
        dev = H_cur.device
        reg_mask = mask_u8.bool()
        on = reg_mask.to(H_cur.dtype)
        motor = self.io_idxs["motor"]
        motor_state = H_cur[motor]
        # first N_slots active regions (index order) write the workspace (model.py:300-306)
        n_slots = self.work.slots.shape[0]
        rank = torch.cumsum(reg_mask.to(torch.int64), 0) - 1
        slot = torch.where(reg_mask & (rank < n_slots), rank, torch.full_like(rank, n_slots))
        delta = torch.zeros(n_slots + 1, self.d, device=dev).index_add_(0, slot, H_cur)[:n_slots]
        self.work.write(delta)  # io regions are forced on, so at least one region always writes
        ws_state = self.work.slots.mean(dim=0)
        # token logits: mean of the motor head and the heads of the other active regions (model.py:311-318)
        cache = self._head_cache()
        motor_logits = self.lm_head.forward_stacked(motor_state, cache["lm"])
        Wst, bst = cache["region_heads"]
        per_region = torch.baddbmm(bst.unsqueeze(-1), Wst, H_cur.unsqueeze(-1)).squeeze(-1)  # [R,V]
        w = on.clone()
        w[motor] = 0.0
        logits = (motor_logits + (w.unsqueeze(-1) * per_region).sum(0)) / (1.0 + w.sum())
        if torch.is_grad_enabled():
            if self._heads_used is None or self._heads_used.device != dev:
                self._heads_used = torch.zeros(self.R, dtype=torch.bool, device=dev)
            self._heads_used |= w.bool()
        rtd_logits = self.rtdd(motor_state)
        return logits, rtd_logits, ws_state, motor_state

    def _head_cache(self):
        """Stacked views of the token heads, rebuilt once per reasoning loop (the parameters do
        not change inside one); gradients flow back through the stack to the per-head Parameters."""
        c = self._loop_cache
        key = (torch.is_grad_enabled(), self.region_heads[0].weight._version, self.region_heads[0].weight.data_ptr())
        if c is None or c.get("key") != key or not self._in_loop:
            c = {"key": key, "lm": self.lm_head.stacked(),
                 "region_heads": (torch.stack([h.weight for h in self.region_heads]),
                                  torch.stack([h.bias for h in self.region_heads]))}
            self._loop_cache = c
        return c

    def finalize_grads(self) -> None:
        """After backward: region heads that never contributed keep grad None, as in the
        reference where only the heads of active regions enter the graph (one host sync)."""
        if self._heads_used is None:
            return
        used = self._heads_used.tolist()
        for r, h in enumerate(self.region_heads):
            if not used[r]:
                h.weight.grad = None
                h.bias.grad = None
        self._heads_used.zero_()

    @torch.no_grad()
    def zeros_state(self, device) -> Tuple[torch.Tensor, torch.Tensor]:
        return torch.zeros(self.R, self.d, device=device), torch.zeros(self.R, dtype=torch.bool, device=device)

    def _verify_aux(self, device, surprise: Optional[torch.Tensor] = None) -> Optional[torch.Tensor]:
        if not self.cfg.enable_ff_energy_alignment:
            return None
        ent = getattr(self.local_mix.attn, "last_attn_entropy", torch.tensor(0.0, device=device))
        surprise = self.bank.surprise[0] if surprise is None else surprise
        return torch.stack([self.local_mix.last_energy.detach().to(device),
                            surprise.mean().to(device), ent.detach().to(device)])

    # ------------------------------------------------------------------ reasoning loop
    def reasoning_loop(self, tokens: torch.Tensor, K_inner: int, focus_map: torch.Tensor,
                       reg_mask_prev: torch.Tensor, H_prev: torch.Tensor):
        """plan -> critic -> gate; branch; verify; halt (model.py:329-460)."""
        self._sensor_cache = None
        self._loop_cache = None
        self.router.begin_loop()
        try:
            if self.engine.usable(H_prev, not self._logits_detached):
                if (not torch.is_grad_enabled() and tokens.is_cuda and self.stream_mode == "strict" and not self.cfg.ff_dropout
                        and not self.cfg.profile and self.engine.steer_usable(H_prev.device, K_inner)):
                    return self._reasoning_loop_steered(tokens, K_inner, focus_map, reg_mask_prev, H_prev)
                return self.engine.run(tokens, K_inner, focus_map, reg_mask_prev, H_prev)
            self._in_loop = True
            return self._reasoning_loop(tokens, K_inner, focus_map, reg_mask_prev, H_prev)
        finally:
            self._in_loop = False
            self._loop_cache = None
            self.router.end_loop()

    def _reasoning_loop_steered(self, tokens, K_inner, focus_map, reg_mask_prev, H_prev):
        """A single no-grad loop (generation, evaluation) on the device-steered engine: the launches of its K_inner outer
        iterations queued back to back -- replayed as one captured graph from the second call on -- instead of one host
        wait per iteration; same kernels, same results (LoopEngine.run_steered with the caller's initial state)."""
        dev = H_prev.device
        eng = self.engine
        batched = self.sensor_vectors_batch(tokens.unsqueeze(0), focus_map.unsqueeze(0))
        fo_all, aux_all = self._loop_inputs(batched, focus_map.unsqueeze(0), 1, dev)
        self.detach_state()
        eng.reset()
        eng.begin_batch(1, K_inner * max(1, self.cfg.B_br), dev)
        H, mk, lg, tr = eng.run_steered(K_inner, batched[0], fo_all, aux_all, init=(H_prev, reg_mask_prev))
        self.local_mix.last_energy = batched[1][-1]
        if batched[2] is not None:
            self.local_mix.attn.last_attn_entropy = batched[2][-1]
        return H[0], mk[0].bool(), lg[0], tr[0]

    def _reasoning_loop(self, tokens, K_inner, focus_map, reg_mask_prev, H_prev):
        dev = H_prev.device
        last_logits = torch.zeros(self.V, device=dev)
        traces: List[torch.Tensor] = []
        if self.cfg.train_deterministic_inner_loop:
            for k in range(K_inner):
                u_mean = uncertainty_from_logits(last_logits) if k > 0 else 0.0
                self.gate.value_bias = 0.0
                H_cur, reg_mask, logits, _r, _w, _m = self.forward_inner_step(
                    tokens, k, focus_map, H_prev, reg_mask_prev, u_mean, 0)
                traces.append(H_cur.clone())
                H_prev, reg_mask_prev, last_logits = H_cur, reg_mask, logits
            return H_cur, reg_mask, last_logits, traces

        branches = [(self.work.slots.clone(), reg_mask_prev.clone(), H_prev.clone()) for _ in range(self.cfg.B_br)]
        scores = [torch.tensor(-1e9, device=dev) for _ in range(self.cfg.B_br)]
        motor = self.io_idxs["motor"]
        for k in range(K_inner):
            u_mean = uncertainty_from_logits(last_logits) if k > 0 else 0.0
            on = reg_mask_prev.to(H_prev.dtype)
            n_on = on.sum()
            # mean of the previously active rows (zeros when none): no host sync needed
            ws_prev = (H_prev * on.unsqueeze(-1)).sum(0) / n_on.clamp_min(1.0)
            with torch.no_grad():  # v_hat only steers the gate (model.py:381-386)
                _p, g = self.plan(ws_prev, H_prev[motor])
                v_hat = float(self.critic(ws_prev, g).item())
            self.gate.value_bias = v_hat
            burst = schedule_burst(u_mean, v_hat, self.R)
            for b in range(self.cfg.B_br):
                self.work.slots = branches[b][0].clone()
                H_b, m_b = branches[b][2].clone(), branches[b][1].clone()
                H_cur, reg_mask, logits, _rtd, _ws, motor_state = self.forward_inner_step(
                    tokens, k, focus_map, H_b, m_b, u_mean, burst)
                with torch.no_grad():  # branch scores only rank the branches (model.py:425-438)
                    ver_energy = self.verify(motor_state, F.softmax(logits, dim=-1), aux=self._verify_aux(dev))
                    delta_g = goodness(H_cur) - goodness(H_b)
                    scores[b] = 0.6 * delta_g + 0.2 * (-ver_energy) + 0.2 * context_logprob(logits)
                branches[b] = (self.work.slots.clone(), reg_mask.clone(), H_cur.clone())
                last_logits = logits
            top = torch.topk(torch.stack(scores), k=min(self.cfg.B_br, 2)).indices.tolist()
            branches = [branches[i] for i in top]
            scores = [scores[i] for i in top]
            traces.append(branches[0][2].clone())
            if should_halt(scores, u_mean, k, K_inner):
                break
            self.work.slots, reg_mask_prev, H_prev = branches[0]
        best = int(torch.stack(scores).argmax().item())
        self.work.slots, reg_mask, H_cur = branches[best]
        return H_cur, reg_mask, last_logits, traces

    def _reasoning_loop_batch_multi(self, K_inner: int, focus_batch: torch.Tensor, batched):
        """stream_mode == "multi": all samples as independent lockstep loops.  Launch groups hold at
        most 64 (loop, branch) streams; every loop of every group starts from the region state and
        workspace the bank holds NOW, and the bank ends with the last loop's final state."""
        eng = self.engine
        vec, en, ent = batched
        N, dev = vec.shape[0], vec.device
        C = 1 if self.cfg.train_deterministic_inner_loop else max(1, self.cfg.B_br)
        G = max(1, eng.MAX_LOCKSTEP_STREAMS // C)
        if focus_batch.numel() > 0:
            fo = focus_batch.reshape(N, -1).any(1).to(torch.uint8)
        else:
            fo = torch.ones(N, dtype=torch.uint8, device=dev)
        aux = None
        if self.verify is not None and getattr(self.verify, "aux_dim", 0) > 0:
            aux = torch.zeros(N, 4, device=dev)
            aux[:, 0] = en.detach()
            if ent is not None:
                aux[:, 2] = ent.detach().to(dev)
        self.detach_state()
        eng.resolve_pending()
        b = self.bank
        start = None
        if N > G:  # later groups restart from the same state: keep a copy
            start = {k: getattr(b, k).clone() for k in ("num", "den", "dt", "var", "surprise", "pred", "radius", "radius_var")}
            start_slots = self.work.slots.clone()
        self.router.begin_loop()
        H_l, M_l, L_l, T_l = [], [], [], []
        try:
            for g0 in range(0, N, G):
                g1 = min(N, g0 + G)
                if start is not None and g0 > 0:
                    for k, v in start.items():
                        setattr(b, k, v)
                    self.work.slots = start_slots
                H, mk, lg, tr = eng.run_lockstep(K_inner, vec[g0:g1], fo[g0:g1], aux[g0:g1] if aux is not None else None)
                H_l.append(H); M_l.append(mk); L_l.append(lg); T_l += tr
        finally:
            self.router.end_loop()
        self.local_mix.last_energy = en[-1]
        if ent is not None:
            self.local_mix.attn.last_attn_entropy = ent[-1]
        A = ent.detach().to(dev) if ent is not None else torch.zeros(N, device=dev)
        cat = lambda l: l[0] if len(l) == 1 else torch.cat(l, 0)
        return cat(H_l), cat(M_l).bool(), cat(L_l), T_l, en.detach(), A

    def _loop_inputs(self, batched, focus_batch: torch.Tensor, N: int, dev):
        """Per-loop inputs of the engine that do not depend on the loops' results: focus-any flags [N,C] and the
        verifier's auxiliary rows [N,C,4] (or None)."""
        nb_ = 1 if self.cfg.train_deterministic_inner_loop else max(1, self.cfg.B_br)
        if focus_batch.numel() > 0:
            fo_all = focus_batch.reshape(N, -1).any(1, keepdim=True).to(torch.uint8).repeat(1, nb_)
        else:
            fo_all = torch.ones(N, nb_, dtype=torch.uint8, device=dev)
        aux_all = None
        if self.verify is not None and getattr(self.verify, "aux_dim", 0) > 0:
            aux_all = torch.zeros(N, nb_, 4, device=dev)
            aux_all[:, :, 0] = batched[1].detach().reshape(N, 1)
            if batched[2] is not None:
                aux_all[:, :, 2] = batched[2].detach().reshape(N, 1).to(dev)
        return fo_all, aux_all

    def reasoning_loop_batch(self, tokens_batch: torch.Tensor, K_inner: int, focus_batch: torch.Tensor, second=None):
        """Sequential over the batch, region fast-weight state carried from one sample to the
        next (detached), exactly as the reference (model.py:462-503).

        second (optional, train_step): tokens_batch / focus_batch are only the FIRST part of the batch; second() returns
        (tokens, focus) of the rest.  Where the loops run device-steered (LoopEngine.run_steered) it is called after the
        first part's loops have been queued, so that the host work behind it overlaps with them; otherwise right away.
        Results are those of the one call on the concatenated batch."""
        dev = tokens_batch.device
        H_l, M_l, L_l, T_l, E_l, A_l = [], [], [], [], [], []
        eng = self.engine
        if second is not None:
            N1 = tokens_batch.shape[0]
            split = (N1 > 1 and tokens_batch.is_cuda and not self.cfg.ff_dropout and not self.cfg.profile
                     and self.stream_mode == "strict" and eng.steer_usable(dev, K_inner)
                     and ((torch.is_grad_enabled() and self._logits_detached) or not torch.is_grad_enabled())
                     and eng.usable(self.zeros_state(dev)[0], not self._logits_detached))
            if not split:
                t2, f2 = second()
                tokens_batch, focus_batch, second = torch.cat([tokens_batch, t2], 0), torch.cat([focus_batch, f2], 0), None
        N = tokens_batch.shape[0]
        batched = None
        if N > 1 and tokens_batch.is_cuda and not self.cfg.ff_dropout and not self.cfg.profile:
            batched = self.sensor_vectors_batch(tokens_batch, focus_batch)  # the loops only differ in region state
        if self.stream_mode not in ("strict", "multi"):
            raise ValueError(f"stream_mode={self.stream_mode!r} (need 'strict' or 'multi')")
        if self.stream_mode == "multi" and batched is not None and eng.lockstep_usable(dev, N):
            return self._reasoning_loop_batch_multi(K_inner, focus_batch, batched)
        try:
            H_zero, mask_zero = self.zeros_state(dev)  # never written to: shared by all loops
            self._zero_H = H_zero
            # train_step: the loops are independent in the backward (state detached between samples),
            # so the engine records them in one block and differentiates them as ONE node
            one_node = (batched is not None and torch.is_grad_enabled() and self._logits_detached
                        and eng.usable(H_zero, False))
            N_all = N if second is None else 2 * N  # (a split batch: the second part has as many loops as the first)
            if one_node:
                eng.begin_batch(N_all, K_inner * (1 if self.cfg.train_deterministic_inner_loop else max(1, self.cfg.B_br)), dev)
            recorded = []
            fo_all = aux_all = None
            if batched is not None and eng.usable(H_zero, not self._logits_detached):
                # per-loop inputs of the engine that do not depend on the loops' results: one batch
                # of ops for all N loops instead of a handful per loop (the GPU idles between loops)
                fo_all, aux_all = self._loop_inputs(batched, focus_batch, N, dev)
            # rows of the per-batch tensors: one unbind each instead of an indexing op per loop
            steer = fo_all is not None and eng.steer_usable(dev, K_inner) and (one_node or not torch.is_grad_enabled())
            if steer:
                # every loop of the batch queued back to back, the outer-iteration decisions taken on the device
                self.detach_state()
                if not one_node:  # no-grad call: nothing is kept from earlier loops
                    eng.reset()
                    eng.begin_batch(N_all, K_inner * max(1, self.cfg.B_br), dev)
                rest = []

                def second_inputs():
                    t2, f2 = second()
                    if t2.shape[0] != N:
                        raise ValueError(f"reasoning_loop_batch: second() returned {t2.shape[0]} sequences, the first part has {N}")
                    b2 = self.sensor_vectors_batch(t2, f2)
                    rest.append(b2)
                    return (b2[0],) + self._loop_inputs(b2, f2, N, dev)

                self.router.begin_loop()
                try:
                    H, mk, lg, tr = eng.run_steered(K_inner, batched[0], fo_all, aux_all,
                                                    second=second_inputs if second is not None else None)
                finally:
                    self.router.end_loop()
                en, ent = batched[1], batched[2]
                if rest:
                    en = torch.cat([en, rest[0][1]], 0)
                    ent = torch.cat([ent, rest[0][2]], 0) if ent is not None else None
                    N = N_all
                self.local_mix.last_energy = en[-1]
                if ent is not None:
                    self.local_mix.attn.last_attn_entropy = ent[-1]
                A = ent.detach().to(dev) if ent is not None else torch.zeros(N, device=dev)
                return H, mk.bool(), lg, tr, en.detach(), A
            if second is not None:
                raise RuntimeError("reasoning_loop_batch: a split batch reached the host-steered path (split predicate out of step)")
            rows = lambda t: t.unbind(0) if t is not None else None
            fo_r, aux_r = rows(fo_all), rows(aux_all)
            sv_r, en_r, ent_r = (rows(batched[0]), rows(batched[1]), rows(batched[2])) if batched is not None else (None,) * 3
            for b in range(N):
                self.detach_state()
                H_prev, mask_prev = H_zero, mask_zero
                self._focus_any_override = fo_r[b] if fo_r is not None else None
                self._aux_override = aux_r[b] if aux_r is not None else None
                if batched is not None:  # what a per-sequence mixer call would have left behind
                    self._sensor_override = sv_r[b]
                    self.local_mix.last_energy = en_r[b]
                    if ent_r is not None:
                        self.local_mix.attn.last_attn_entropy = ent_r[b]
                n_rec = len(eng._batch["samples"]) if one_node else 0
                H_cur, reg_mask, logits, traces = self.reasoning_loop(tokens_batch[b], K_inner, focus_batch[b], mask_prev, H_prev)
                if one_node and len(eng._batch["samples"]) > n_rec:
                    recorded.append(b)  # H_cur is a placeholder until end_batch()
                H_l.append(H_cur); M_l.append(reg_mask); L_l.append(logits); T_l.append(traces)
                E_l.append(self.local_mix.last_energy.detach())
                ent = getattr(getattr(self.local_mix, "attn", None), "last_attn_entropy", torch.tensor(0.0))
                A_l.append(ent.detach().to(dev))
            if one_node:
                H_eng = eng.end_batch(batched[0][recorded] if len(recorded) != N else batched[0])
                M_l = [mk.value if isinstance(mk, DeferredMask) else mk for mk in M_l]  # (all resolved by now)
                if len(recorded) == N:
                    H_l = H_eng  # [N,R,d]
                else:
                    for j, b in enumerate(recorded):
                        H_l[b] = H_eng[j]
        finally:
            self._sensor_override = None
            self._focus_any_override = self._aux_override = None
            self._zero_H = None
            eng._batch = None
            eng._pending = None
        H_all = H_l if torch.is_tensor(H_l) else torch.stack(H_l)
        M_all = torch.stack([mk.bool() for mk in M_l]) if len({mk.dtype for mk in M_l}) > 1 else torch.stack(M_l).bool()
        return (H_all, M_all, torch.stack(L_l), T_l, torch.stack(E_l), torch.stack(A_l))
