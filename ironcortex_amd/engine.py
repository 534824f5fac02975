"""LoopEngine: the reasoning loop of one sequence as a straight run of C-ABI calls.

The general path (CortexReasoner.forward_inner_step) wraps every gate/router/region call in its
own torch.autograd.Function and allocates every output with torch.empty; at one sequence per
loop that host work is ten times the GPU work.  The engine runs the same kernels on the same
inputs, but

  * all per-step tensors (outputs, saved activations, masks) live in a step arena: one pointer
    addition per field instead of an allocation, and a step's saved activations stay where the
    forward kernels wrote them until the backward reads them;
  * the C structs are built once per loop and only the moving pointers are patched per step;
  * autograd sees ONE node per loop (LoopFn).  Its backward walks the recorded steps in reverse
    with two C calls per step (ic_region_phase_bwd, ic_router_bwd), carrying the state
    gradients (num, den, var, pred, radius) from step to step and the H gradients from a
    step to the step that produced its input, exactly the chain torch autograd would follow;
  * the heads that only steer the loop (token logits, workspace, planner/critic, verifier) run
    without autograd.

It is used when the caller needs gradients only through the loop's final region states (the
forward-forward train_step: reference training.py:120-231 never back-propagates through the
inner-step logits) or no gradients at all (generation, evaluation).  Anything else takes the
general path.  Results are identical up to fp32 summation order of the small parameter
gradients (per-step rows are summed per loop instead of accumulated step by step).
"""
from __future__ import annotations

import ctypes
import warnings
import os
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch

from . import hip as hip_mod
from .hip import CLASSES, ENUMS, check, lib, new, ops, stream_handle
from .utils import schedule_burst, should_halt

_F32, _I32, _U8 = torch.float32, torch.int32, torch.uint8


# IC_FUSE_STEP=0: gate+router and the region phase as two C calls (ic_front_fwd, ic_region_phase_fwd)
# instead of ic_loop_step_fwd; same results, for A/B timing
FUSE_STEP = os.environ.get("IC_FUSE_STEP", "1") != "0"
# IC_DEFER_SYNC=0: every loop waits for its last iteration's scores before it returns (A/B timing)
DEFER_SYNC = os.environ.get("IC_DEFER_SYNC", "1") != "0"
# IC_POLL_SYNC=0: hipStreamSynchronize at every outer iteration instead of polling the pinned score block
POLL_SYNC = os.environ.get("IC_POLL_SYNC", "1") != "0"
# IC_STEER=0: the loops of a train_step are steered by the host (one wait per outer iteration) instead of
# on the device (LoopEngine.run_steered: no wait at all); same results, for A/B timing and the equality test
STEER = os.environ.get("IC_STEER", "1") != "0"
# IC_GRAPH=0: the steered loops are launched call by call every step instead of replayed as one captured HIP graph
GRAPH = os.environ.get("IC_GRAPH", "1") != "0"

def _copy_many(dst, src) -> None:
    """dst[i].copy_(src[i]) as one multi-tensor launch per dtype (torch._foreach_copy_ only takes its fused route when
    every tensor of the call has the same dtype; a mixed list silently becomes one copy kernel per tensor)."""
    groups = {}
    for d_, s_ in zip(dst, src):
        groups.setdefault((d_.dtype, s_.dtype), ([], []))
        groups[(d_.dtype, s_.dtype)][0].append(d_)
        groups[(d_.dtype, s_.dtype)][1].append(s_)
    for ds, ss in groups.values():
        if len(ds) == 1:
            ds[0].copy_(ss[0])
        else:
            torch._foreach_copy_(ds, ss)


class StructImage:
    """Per-step images of a C argument struct.  Most pointer fields of a step's arguments are
    `base + t * stride` (step arena, tape, gradient rows); instead of ~50 ctypes attribute
    writes per step, all steps' struct images are formed at once with numpy (int64 words) and a
    step costs one memmove plus the few fields that do not advance linearly."""

    def __init__(self, obj):
        self.obj = obj
        self.size = ctypes.sizeof(obj)
        assert self.size % 8 == 0
        self.addr = ctypes.addressof(obj)
        self.stride = np.zeros(self.size // 8, dtype=np.int64)
        self.lin = {}
        self.cols = {}

    @staticmethod
    def _offset(cls, path):
        off = 0
        for name in path:
            fld = getattr(cls, name)
            off += fld.offset
            cls = dict(cls._fields_)[name]
        return off

    def linear(self, path, base: int, stride: int) -> None:
        """Field `path` (tuple of names, nested structs allowed) = base + t * stride."""
        off = self._offset(type(self.obj), path)
        assert off % 8 == 0
        self.lin[off // 8] = (base, stride)

    def column(self, path, values) -> None:
        """Pointer-sized field `path` takes values[i] in image i (fields that do not advance linearly)."""
        off = self._offset(type(self.obj), path)
        assert off % 8 == 0
        self.cols[off // 8] = np.asarray(values, dtype=np.int64)

    def images(self, ts) -> "np.ndarray":
        """[len(ts), words] struct images for steps ts; call after every static field is set."""
        base = np.frombuffer((ctypes.c_char * self.size).from_address(self.addr), dtype=np.int64).copy()
        stride = np.zeros_like(base)
        for w, (b0, st_) in self.lin.items():
            base[w], stride[w] = b0, st_
        out = base[None, :] + np.asarray(ts, dtype=np.int64)[:, None] * stride[None, :]
        for w, vals in self.cols.items():
            out[:, w] = vals
        return out

    def load(self, row) -> None:
        ctypes.memmove(self.addr, row.ctypes.data, self.size)


class _Chunk:
    """cap steps of every field, field-major: fields[name] is [cap, *shape]."""

    def __init__(self, shapes: Dict[str, Tuple[tuple, torch.dtype]], cap: int, device):
        self.cap = cap
        self.fields = {k: torch.zeros((cap,) + tuple(s), dtype=dt, device=device) for k, (s, dt) in shapes.items()}
        self.ptr = {k: t.data_ptr() for k, t in self.fields.items()}
        self.stride = {k: t.stride(0) * t.element_size() for k, t in self.fields.items()}


class StepArena:
    """Chunks of step slots.  reserve(n) returns n contiguous slots; nothing moves on growth, so
    device pointers handed out earlier stay valid until reset()."""

    def __init__(self, shapes, chunk_steps: int = 128):
        self.shapes, self.chunk_steps = shapes, chunk_steps
        self.chunks: List[_Chunk] = []
        self.cur, self.pos = 0, 0

    def reserve(self, n: int, device) -> Tuple[_Chunk, int]:
        while True:
            if self.cur >= len(self.chunks):
                self.chunks.append(_Chunk(self.shapes, max(self.chunk_steps, n), device))
            c = self.chunks[self.cur]
            if c.cap - self.pos >= n:
                start = self.pos
                self.pos += n
                return c, start
            self.cur, self.pos = self.cur + 1, 0

    def reset(self) -> None:
        self.cur, self.pos = 0, 0


class DeferredMask:
    """Active-region mask of a recorded loop whose winner is read later (LoopEngine.resolve_pending):
    .value is the uint8 [R] view once resolved."""

    __slots__ = ("value",)

    def __init__(self):
        self.value = None


class _Sample:
    """Record of one loop: what the backward needs to replay the steps in reverse."""

    __slots__ = ("chunk", "i0", "n", "parent", "best", "step_pos", "init", "H0", "mask0", "sensor", "ebias",
                 "norm_in_w", "flags", "kwta_k", "kwta_temp", "keep", "ebias_grad", "zero_in", "ts")
    # step t of the loop sits in slot i0 + t * ts of its chunk (ts = 1: one loop after the other;
    # ts = N: N loops in lockstep, run_lockstep)


class LoopEngine:
    def __init__(self, model):
        self.model = model
        self.enabled = True
        self._arena: Optional[StepArena] = None
        self._arena_key = None
        self._ws = None
        self._state_in_arena = False
        self._unit_log = []
        self._in_cache = None
        self._pass = None  # accumulators of the backward pass in flight
        self._sess = None  # C structs shared by the loops between two reset()s
        self._vhat0 = None  # critic value of the all-zero input state (same for every loop of a session)
        self._hs_all = {}   # heads set-ups by number of streams per launch
        self._hs_ok = set()  # ... checked in full since the last reset()
        self._fa_buf = None
        self._batch = None  # loops of one reasoning_loop_batch call, differentiated as ONE node
        # A recorded loop of a batch does not wait for its last iteration's scores: the winner is
        # read (resolve_pending) after the NEXT loop's first launch is queued, so the host work
        # between two loops overlaps the GPU instead of following a stream sync.
        self._pending = None
        self._events = None
        self._wsb = None
        self._graph = None       # captured launch sequence of the steered loops (run_steered)
        self._graph_off = False  # (set when a capture failed: call-by-call launches from then on)
        self._cap_stream = None
        self.counters = {"batch_bwd": 0, "loop_bwd": 0, "lockstep": 0}  # backward passes taken all-loops-at-once / loop by loop
        model._coord.register(self._begin_pass, self._finish_pass)

    # ------------------------------------------------------------------ applicability
    def usable(self, H_prev: torch.Tensor, logits_need_grad: bool) -> bool:
        m = self.model
        if not self.enabled or not H_prev.is_cuda or H_prev.requires_grad:
            return False
        if torch.is_grad_enabled() and logits_need_grad:
            return False
        if m.cfg.profile or m.cfg.ff_dropout or (m.router.edge_transform_mode == "factorized" and not m.router.fact_lib):
            return False
        b = m.bank
        if b.S != 1 or b.num.device != H_prev.device:
            return True  # the first phase() call builds the state; done in run()
        return not any(getattr(b, k).requires_grad for k in ("num", "den", "var", "pred", "radius", "radius_var"))

    def resolve_pending(self) -> None:
        """Finish the bookkeeping of the last recorded loop whose final scores were not waited for:
        wait for the event recorded after its last launch, read the scores, pick the winner."""
        p, self._pending = self._pending, None
        if p is None:
            return
        p["event"].synchronize()
        vals = p["ls"].tolist()
        branches = p["branches"]
        top = sorted(range(len(branches)), key=lambda i: (-vals[i], i))[:min(p["B_br"], 2)]
        best = branches[top[0]][1]
        p["traces"].append(p["H_of"](best))
        p["finish"](best)

    def reset(self) -> None:
        """Start of a new forward/backward pass: every step slot may be reused.  The bank's state
        tensors are views of the last loop's final slot until now; copy them out first."""
        self.resolve_pending()
        self.resolve_open()
        self._rescue_state()
        self.flush_units()
        self._sess = None
        self._vhat0 = None
        self._vhat0_pending = None
        self._hs_ok = set()  # parameters may have moved since: _heads_setup checks every pointer again
        if self._arena is not None:
            self._arena.reset()

    def resolve_open(self) -> None:
        """Host-side bookkeeping of the last device-steered batch, if nobody asked for it yet (its backward pass runs
        from the device-side record): read here, when the step that ran it has long been waited for."""
        bt, self._steered_open = getattr(self, "_steered_open", None), None
        if bt is None:
            return
        sp = bt.get("steered")
        if sp is None or self.model._units is not None or sp.get("counted"):
            self.resolve_steered(bt)
            return
        # nobody asked for the loops' host-side samples: all the step still owes is the inner-step count and the critic
        # value the last iteration ran with (the full record stays available: resolve_steered(bt))
        self._steered_light(bt)

    def prefetch_open(self) -> None:
        """Start the copy of the open batch's device-side record into pinned host memory (train_step calls this before
        its one wait for the step, so that resolve_open finds the record on the host)."""
        bt = getattr(self, "_steered_open", None)
        sp = bt.get("steered") if bt is not None else None
        if sp is None or "host" in sp:
            return
        sb = sp["sb"]
        pin = getattr(self, "_pin_rec", None)
        if pin is None or pin[0].shape != sb["ctl"].shape or pin[1].shape != sb["rec"].shape:
            pin = self._pin_rec = (torch.empty(sb["ctl"].shape, dtype=sb["ctl"].dtype).pin_memory(),
                                   torch.empty(sb["rec"].shape, dtype=sb["rec"].dtype).pin_memory())
        pin[0].copy_(sb["ctl"], non_blocking=True)
        pin[1].copy_(sb["rec"], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        sp["host"] = (pin[0], pin[1], ev)

    def _steered_record(self, sp):
        """(ctl, rec) of a steered batch as numpy arrays: from the prefetched pinned copy when there is one."""
        host = sp.pop("host", None)
        if host is not None:
            host[2].synchronize()
            return host[0].numpy().copy(), host[1].numpy().copy()  # (the pinned buffers are reused by the next step)
        sb = sp["sb"]
        return sb["ctl"].cpu().numpy(), sb["rec"].cpu().numpy()  # (waits for the stream)

    def _steered_light(self, bt) -> None:
        m, sp = self.model, bt["steered"]
        K, N = sp["K"], bt["n_loops"]
        ctl, rec = self._steered_record(sp)
        sp["counted"] = True
        m.counters["inner_steps"] += int(rec[:N, 2].sum())
        last = ctl[(N - 1) * K + max(0, int(rec[N - 1, 3]) - 1)]
        m.gate.value_bias = float(np.array([last[ENUMS["IC_CTL_VHAT"]]], dtype=np.int32).view(np.float32)[0])

    def new_step(self) -> None:
        """train_step's first call.  A backward pass that raised (OOM, an ic_* error code, a user
        hook) never runs the coordinator's end-of-pass callback; tape slots booked by a
        grad-enabled forward that was never differentiated are never reclaimed.  A train_step owns
        its whole forward/backward, so at its start both are leftovers: clear them, then reset()."""
        m = self.model
        m._coord.abort()
        self._pass = None
        self._batch = None
        m.bank.tape.reset()
        if m.router._tape is not None:
            m.router._tape.reset()
        self.reset()
        self._prefetch_vhat0()

    def _prefetch_vhat0(self) -> None:
        """Every loop of a train_step starts from the all-zero state, so the critic value that steers its first
        iteration is the same for all of them and depends on the parameters alone: launch its chain NOW, while
        the stream is empty, and read it when the first loop needs it (an event, not a stream sync: by then the
        token mixer's kernels are queued behind it)."""
        m = self.model
        try:
            dev = m.reg_coords.device
            if (not self.enabled or dev.type != "cuda" or m.plan is None or m.cfg.train_deterministic_inner_loop
                    or m.cfg.profile or m.cfg.ff_dropout):
                return
            z = getattr(self, "_zero_state", None)
            if z is None or z[0].device != dev or z[0].shape != (m.R, m.d):
                z = self._zero_state = (torch.zeros(m.R, m.d, device=dev), torch.zeros(m.R, dtype=_U8, device=dev))
            hs = self._heads_setup(dev)
            sa, steer = hs["sa"], hs["steer"]
            L, st = lib(), stream_handle()
            sa.n_cand = 1
            sa.H_cand[0], sa.mask_cand[0] = z[0].data_ptr(), z[1].data_ptr()
            sa.scores = None
            for g in steer:
                g.nbatch = 1
            rc = L.ic_steer_select(ctypes.byref(sa), st) or L.ic_gemv_chain(steer, 6, st)
            if rc:
                check(rc, "ic_steer_select/ic_gemv_chain")
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(dev))
            self._vhat0_pending = (ev, hs["buf"]["ls"], hs["off"][1], dev)
        except Exception:
            self._vhat0_pending = None
            raise

    def _resolve_vhat0(self, dev) -> None:
        p, self._vhat0_pending = getattr(self, "_vhat0_pending", None), None
        if p is None or self._vhat0 is not None or p[3] != dev:
            return
        p[0].synchronize()
        self._vhat0 = float(p[1][p[2]])

    def _rescue_state(self) -> None:
        self.resolve_pending()
        if not self._state_in_arena:
            return
        m = self.model
        b = m.bank
        names = ("num", "den", "dt", "var", "surprise", "pred", "radius", "radius_var")
        src = [getattr(b, k) for k in names] + [m.work.slots] + ([m.router._last_wgt] if m.router._last_wgt is not None else [])
        with torch.no_grad():
            dst = [torch.empty_like(t, memory_format=torch.contiguous_format) for t in src]
            _copy_many(dst, src)  # (one multi-tensor launch per dtype instead of ten clones)
        for k, t in zip(names, dst):
            setattr(b, k, t)
        m.work.slots = dst[len(names)]
        if m.router._last_wgt is not None:
            m.router._last_wgt = dst[len(names) + 1]
        self._state_in_arena = False

    # ------------------------------------------------------------------ storage
    def _shapes(self):
        m = self.model
        b, rt = m.bank, m.router
        R, d, E = m.R, m.d, max(rt.E, 1)
        rd = ((R, d), _F32)
        sh = {"H": rd, "num_o": rd, "den_o": rd, "dt_o": ((R,), _I32), "M": rd, "scores": ((R,), _F32),
              "mask": ((R,), _U8), "inv_s": ((R, 2), _F32), "logits": ((m.V,), _F32),
              "slots": (tuple(m.work.slots.shape), _F32)}
        for k in ("x", "u1", "r_s", "k_s", "v_s", "y", "hpre", "gate"):
            sh[k] = rd
        if b.afd:
            sh.update(var_o=rd, surprise_o=((R,), _F32), sterm=rd)
        if b.use_pred:
            sh["pred_o"] = rd
        if b.rt:
            sh.update(radius_o=((R,), _F32), radius_var_o=((R,), _F32), ynorm=((R,), _F32))
        if rt.enable_precision_routed_messages:
            sh.update(msg=((E, d), _F32), keyv=((E, d), _F32), q=rd, wgt=((E,), _F32))
        if rt.fact_lib:  # factorized mode: V h of the senders, the mixed rows U applies to, rows with an active in-edge
            sh.update(fact_t=rd, fact_z=rd, fact_dact=((R,), _U8))
        return sh

    def _get_arena(self, device, steps: int) -> StepArena:
        m = self.model
        key = (device, m.R, m.d, m.router.E, m.bank.afd, m.bank.rt, m.bank.use_pred,
               m.router.enable_precision_routed_messages, m.router.fact_lib)
        if self._arena is None or self._arena_key != key:
            self._arena = StepArena(self._shapes(), chunk_steps=max(128, 4 * steps))
            self._arena_key = key
            self._ws = None
        return self._arena

    def _workspace(self, device):
        """Backward scratch shared by all steps (the stream serialises them)."""
        if self._ws is not None and self._ws["gM"].device == device and ("fact_gz" in self._ws) == self.model.router.fact_lib:
            return self._ws
        m = self.model
        R, d, E = m.R, m.d, max(m.router.E, 1)
        ns = ops.bwd_nslab(d)
        z = lambda *s: torch.zeros(*s, device=device)
        ws = {"gM": z(2, R, d), "ghpre": z(R, d), "slab2": z(ns, R, d), "slab": z(ns, R, d), "slab_rt": z(ns, R),
              "gH_tmp": z(2, R, d), "gHd": z(2, R, d), "rslab": z(ns, 2, E, d), "slab_q": z(ns, 2, R, d)}
        if m.router.fact_lib:
            ws.update(fact_gz=z(2, R, d), fact_gt=z(2, R, d), fact_slab=z(ns, 2, R, d), fact_gb=z(ns, 2, R))
        for side in ("a", "b"):
            for k in ("gnum", "gden", "gvar", "gpred"):
                ws[f"{k}_{side}"] = z(R, d)
            for k in ("gradius", "gradius_var"):
                ws[f"{k}_{side}"] = z(R)
        ws["_ptr"] = {k: t.data_ptr() for k, t in ws.items()}
        self._ws = ws
        return ws

    # ------------------------------------------------------------------ forward
    def run(self, tokens, K_inner, focus_map, reg_mask_prev, H_prev):
        """Same contract as CortexReasoner.reasoning_loop: (H_cur, reg_mask, last_logits, traces)."""
        m = self.model
        cfg, bank, router, gate = m.cfg, m.bank, m.router, m.gate
        dev = H_prev.device
        R, d, E = m.R, m.d, router.E
        grad = torch.is_grad_enabled()
        L = lib()
        st = stream_handle()
        if bank.S != 1 or bank.num.device != dev:
            bank.flat.reflat()
            bank.reset_state(dev, 1)
        n_max = K_inner * (1 if cfg.train_deterministic_inner_loop else cfg.B_br)
        arena = self._get_arena(dev, n_max)
        if not grad:
            self.reset()  # nothing is kept from a no-grad loop: outputs are cloned below
        elif H_prev is getattr(m, "_zero_H", None):
            self._resolve_vhat0(dev)
        chunk, i0 = arena.reserve(max(1, n_max), dev)
        CP, CS = chunk.ptr, chunk.stride

        sensor = m.sensor_vector(tokens, focus_map)
        sens_c = sensor.detach().contiguous()
        ebias = router._cached_bias(m.reg_coords)
        ebias_c = ebias.detach().contiguous()
        norm_in_w = m.norm_in.weight
        tables = router._edge_tables(m.reg_coords)
        kw_k, kw_soft, kw_temp = bank.kwta_now()
        flags = bank._flags(kw_soft, bank.cells[0].disable_kwta, True)
        # the branches of an outer iteration go through gate / router / heads as ONE launch each
        # (streams of the same call), so the loop's inputs are kept once per branch
        nbr_ = 1 if cfg.train_deterministic_inner_loop else max(1, cfg.B_br)
        fab = self._fa_buf
        if fab is None or fab.device != dev or fab.numel() != nbr_:
            fab = self._fa_buf = torch.ones(nbr_, dtype=_U8, device=dev)
        focus_any = fab  # stream-ordered reuse: earlier loops' launches have read it by the time it changes
        fo = getattr(m, "_focus_any_override", None)
        if fo is not None and fo.shape[0] == nbr_:
            focus_any = fo  # reasoning_loop_batch: any() of every sample's focus map in one op, a row per loop
        elif focus_map.numel() > 0:
            focus_any.copy_(focus_map.any())
        else:
            focus_any.fill_(1)
        ikey = (H_prev.data_ptr(), H_prev._version, reg_mask_prev.data_ptr(), reg_mask_prev._version, nbr_)
        if self._in_cache is not None and self._in_cache[0] == ikey:
            H0, mask0 = self._in_cache[1], self._in_cache[2]  # same (unmodified) tensors as the last loop
        else:
            H0 = H_prev.detach().float().reshape(1, R, d).repeat(nbr_, 1, 1)
            mask0 = reg_mask_prev.to(_U8).reshape(1, R).repeat(nbr_, 1)
            self._in_cache = (ikey, H0, mask0, H_prev, reg_mask_prev)
        # ---- structs: everything static is set once per session (the loops between two reset()s
        # that share parameters, tables and schedule), the per-loop pointers are patched below
        skey = (dev, bank.flat.flat.data_ptr(), router._flat.flat.data_ptr(), norm_in_w.data_ptr(),
                gate.gain_ema.data_ptr(), gate.homeo.data_ptr(), id(tables), kw_k, flags, kw_temp, nbr_)
        sess = self._sess
        if sess is None or sess[0] != skey:
            rP, bP = router._params(), bank.params()
            off, nbr = gate._tables(dev)
            ga = new("ic_gate_args", S=1, R=R, d=d, sensor_idx=m.io_idxs["sensor"], motor_idx=m.io_idxs["motor"],
                     force_io=1, w1=gate.w1, w2=gate.w2, w3=gate.w3, w4=gate.w4, gain_ema=gate.gain_ema,
                     homeo=gate.homeo, nbr_off=off, nbr=nbr)
            rspec = router.spec()
            ra = ops._router_args(rspec, tables, rP, 1)
            P = dict(bP)
            P["norm_in_w"] = norm_in_w
            fa = ops._region_args(bank.spec, P, 1, 0.0, kw_k, flags, kw_temp)
            sess = self._sess = (skey, rP, bP, off, nbr, ga, ra, fa, rspec, P)
        _, rP, bP, off, nbr, ga, ra, fa, rspec, P = sess
        ga.focus_any = focus_any.data_ptr()
        ra.ebias = ebias_c.data_ptr()
        fa.sensor = sens_c.data_ptr()
        fa.last_dir, fa.last_norm = bank.last_dir.data_ptr(), bank.last_norm.data_ptr()
        prec, afd, rt_, use_pred = rspec.precision, bank.afd, bank.rt, bank.use_pred
        g_fwd, f_fwd = L.ic_gate_fwd, L.ic_front_fwd
        r_fwd, p_fwd = L.ic_router_fwd, L.ic_region_phase_fwd
        ga_ref, ra_ref, fa_ref = ctypes.byref(ga), ctypes.byref(ra), ctypes.byref(fa)
        w5, w6 = gate.w5, gate.w6

        # state pointers before the first step: the bank's own tensors (kept alive in `init`)
        init = dict(num=bank.num, den=bank.den, var=bank.var, pred=bank.pred, radius=bank.radius,
                    radius_var=bank.radius_var, surprise=bank.surprise, dt=bank.dt)
        init = {k: (v if not v.requires_grad and v.is_contiguous() else v.detach().contiguous()) for k, v in init.items()}
        state = {k: v.data_ptr() for k, v in init.items()}
        out_names = [("num", "num_o"), ("den", "den_o"), ("dt", "dt_o")]
        if afd:
            out_names += [("var", "var_o"), ("surprise", "surprise_o")]
        if use_pred:
            out_names += [("pred", "pred_o")]
        if rt_:
            out_names += [("radius", "radius_o"), ("radius_var", "radius_var_o")]
        saved = ["x", "u1", "inv_s", "r_s", "k_s", "v_s", "y", "hpre", "gate"] + (["ynorm"] if rt_ else []) \
            + (["sterm"] if afd else [])
        rsaved = ["msg", "keyv", "q", "wgt"] if prec else (["fact_t", "fact_z", "fact_dact"] if router.fact_lib else [])
        denom = float(max(1, cfg.K_inner - 1))
        parent: List[int] = []
        step_pos_l: List[float] = []
        H0p, mask0p = H0.data_ptr(), mask0.data_ptr()

        def in_ptrs(par: int, b: int):
            """(H, mask) device addresses of the input of a step: slot of step `par`, or copy b of
            the loop's own H_prev / reg_mask_prev."""
            if par < 0:
                return H0p + b * R * d * 4, mask0p + b * R
            return CP["H"] + (i0 + par) * CS["H"], CP["mask"] + (i0 + par) * CS["mask"]

        # What the next front() + region() calls need that depends only on the slot numbers can be
        # set while the GPU is still busy with the current outer iteration (prepare_step), so that
        # after the iteration's host sync only the chosen parents and the gate's scalars remain.
        prepared = [-1]

        def slots_front(i: int, nb: int) -> None:
            a = i0 + i
            ga.S = nb
            ga.scores, ga.mask = CP["scores"] + a * CS["scores"], CP["mask"] + a * CS["mask"]
            if E > 0:
                ra.S, ra.M = nb, CP["M"] + a * CS["M"]
                for k in rsaved:
                    setattr(ra, k, CP[k] + a * CS[k])

        def slots_region(i: int, n: int, step_pos: float) -> None:
            a = i0 + i
            fa.S, fa.flags = n, (flags | chain_flag if n > 1 else flags)
            fa.step_pos, fa.M, fa.mask = step_pos, CP["M"] + a * CS["M"], CP["mask"] + a * CS["mask"]
            fa.num, fa.den, fa.dt = state["num"], state["den"], state["dt"]
            fa.var, fa.surprise = state["var"], state["surprise"]
            fa.pred, fa.radius, fa.radius_var = state["pred"], state["radius"], state["radius_var"]
            fa.H = CP["H"] + a * CS["H"]
            for k, ko in out_names:
                p = CP[ko] + a * CS[ko]
                setattr(fa, ko, p)
                state[k] = p + (n - 1) * CS[ko]
            for k in saved:
                setattr(fa, k, CP[k] + a * CS[k])

        def prepare_step(nb: int, step_k: int) -> None:
            """Slot-dependent arguments of the front() / region() pair that comes next (only call when
            that pair is certain to run: the region state pointers advance here)."""
            i = len(parent)
            slots_front(i, nb)
            slots_region(i, nb, float(step_k) / denom)
            prepared[0] = (i, nb)

        pending = [False]  # front() filled ga / ra and left the launch to region() (ic_loop_step_fwd)
        step_fwd = L.ic_loop_step_fwd

        def front(pars: List[int], step_k: int, u_mean: float, burst: int, stream, defer: bool = False) -> List[int]:
            """Gate + router of len(pars) new inner steps (consecutive slots) whose H inputs are the
            outputs of steps pars[b] (-1: the loop's H_prev), as ONE launch sequence with one
            stream per step.  Returns the step ids; region(id) must follow in order.  defer: the
            launch is left to the region() call that follows (one C call for the whole inner step,
            the router's combine inside the region kernel)."""
            nb = len(pars)
            i = len(parent)
            h_in, mk_in = in_ptrs(pars[0], 0)
            hs = ms = 0
            if nb > 1:
                h1, m1 = in_ptrs(pars[1], 1)
                hs, ms = (h1 - h_in) // 4, m1 - mk_in
                for b in range(2, nb):  # equally spaced inputs only (always true for two branches)
                    hb, mb = in_ptrs(pars[b], b)
                    if (hb - h_in) // 4 != b * hs or mb - mk_in != b * ms:
                        prepared[0] = -1
                        return [j for b2 in range(nb) for j in front([pars[b2]], step_k, u_mean, burst, stream)]
                if hs == R * d:
                    hs = 0
                if ms == R:
                    ms = 0
                if (hs == 0 and h1 - h_in != R * d * 4) or (ms == 0 and m1 - mk_in != R):
                    prepared[0] = -1
                    return [j for b2 in range(nb) for j in front([pars[b2]], step_k, u_mean, burst, stream)]
            if prepared[0] != (i, nb):
                slots_front(i, nb)
            ga.H, ga.H_sstride = h_in, hs
            ga.t5, ga.t6 = w5 * float(gate.value_bias), w6 * float(u_mean)
            ga.k_total = max(0, cfg.k_active + burst)
            if E > 0:  # gate and router read the same inputs: one launch sequence
                ra.H, ra.mask_prev = h_in, mk_in
                ra.H_sstride, ra.mask_sstride = hs, ms
                if defer and nb > 1:
                    pending[0] = True
                else:
                    rc = f_fwd(ga_ref, ra_ref, stream)
                    if rc:
                        check(rc, "ic_front_fwd")
            else:
                rc = g_fwd(ga_ref, stream)
                if rc:
                    check(rc, "ic_gate")
            # (E == 0: M stays at the zeros the arena was created with)
            for par in pars:
                parent.append(par)
                step_pos_l.append(float(step_k) / denom)
            m.counters["inner_steps"] += nb
            return list(range(i, i + nb))

        chain_flag = ENUMS["IC_F_CHAIN"]

        def region(i: int, stream, n: int = 1) -> None:
            """Region phase of the n consecutive steps i..i+n-1 (state chained from the previously
            executed step; n > 1: IC_F_CHAIN, the steps are the streams of one launch sequence)."""
            if prepared[0] == (i, n):
                prepared[0] = -1  # (used up)
            else:
                slots_region(i, n, step_pos_l[i])
            if pending[0]:
                pending[0] = False
                if ga.S != n:
                    raise RuntimeError("LoopEngine: deferred front launch and region phase disagree")
                rc = step_fwd(ga_ref, ra_ref, fa_ref, stream)
            else:
                rc = p_fwd(fa_ref, stream)
            if rc:
                check(rc, "ic_region_phase_fwd")

        def step(par: int, step_k: int, u_mean: float, burst: int) -> int:
            (i,) = front([par], step_k, u_mean, burst, st)
            region(i, st)
            return i

        Hf, Mf = chunk.fields["H"], chunk.fields["mask"]
        H_of = lambda i: Hf[i0 + i] if i >= 0 else H0[0]
        mask_of = lambda i: Mf[i0 + i] if i >= 0 else mask0[0]
        with torch.no_grad():
            best, last_step, traces = self._drive((step, front, region, prepare_step), in_ptrs, H_of, K_inner, dev, chunk, i0,
                                                  trace_views=bool(m._logits_detached and grad),
                                                  zero_in=H_prev is getattr(m, "_zero_H", None))
        n = len(parent)
        deferred = best is None  # the winner is not known yet (resolve_pending() fills it in)
        last_logits = (chunk.fields["logits"][i0 + last_step] if last_step >= 0 else torch.zeros(m.V, device=dev))
        if not (m._logits_detached and grad):  # train_step stacks the loops' logits before the arena is reset
            last_logits = last_logits.clone()
        if not deferred and best >= 0:
            m.work.slots = chunk.fields["slots"][i0 + best]

        # ---- hand the state back to the bank: views of the last slot, copied out by reset()
        # before the arena is recycled (15 of 16 loops of a train_step never need the copy)
        li = i0 + n - 1
        f = chunk.fields
        wgt_li = li
        if n > 0 and grad and self._batch is not None and n < n_max:
            # the loop halted early: the batched backward expects the state the NEXT loop starts from
            # in the last slot of this loop's block (the slot before the next block)
            lj = i0 + max(1, n_max) - 1
            for ko in ("num_o", "den_o", "dt_o", "var_o", "surprise_o", "pred_o", "radius_o", "radius_var_o"):
                if ko in f:
                    f[ko][lj].copy_(f[ko][li])
            li = lj
        if n > 0:
            nar = torch.narrow  # (cheaper than slicing: this runs once per loop while the GPU idles)
            bank.num, bank.den, bank.dt = nar(f["num_o"], 0, li, 1), nar(f["den_o"], 0, li, 1), nar(f["dt_o"], 0, li, 1)
            if afd:
                bank.var, bank.surprise = nar(f["var_o"], 0, li, 1), nar(f["surprise_o"], 0, li, 1)
            if use_pred:
                bank.pred = nar(f["pred_o"], 0, li, 1)
            if rt_:
                bank.radius, bank.radius_var = nar(f["radius_o"], 0, li, 1), nar(f["radius_var_o"], 0, li, 1)
            if prec:
                router._last_wgt = nar(nar(f["wgt"], 0, wgt_li, 1), 1, 0, E)
            self._state_in_arena = True
        if m._units is not None and E > 0 and n > 0:  # bench.py's device-side unit counters
            self._unit_log.append((chunk, i0, list(parent), mask0[0]))
            if not grad:
                self.flush_units()

        # (reasoning_loop_batch converts the stacked masks of its loops in one op)
        if deferred:
            reg_mask = DeferredMask()
        else:
            reg_mask = mask_of(best) if (self._batch is not None and grad and n > 0) else mask_of(best).bool()
        if grad and n > 0:
            s = _Sample()
            s.chunk, s.i0, s.n, s.parent, s.best, s.step_pos = chunk, i0, n, parent, best, step_pos_l
            s.init, s.H0, s.mask0, s.sensor, s.ebias, s.norm_in_w = init, H0, mask0, sens_c, ebias_c, norm_in_w
            s.flags, s.kwta_k, s.kwta_temp = flags, kw_k, kw_temp
            s.keep = (focus_any, tables, rP, bP)
            s.ebias_grad = bool(ebias.requires_grad)
            s.zero_in = H_prev is getattr(m, "_zero_H", None)
            s.ts = 1
            if deferred:
                slots_f = chunk.fields["slots"]

                def finish(best_: int, s=s, reg_mask=reg_mask) -> None:
                    s.best = best_
                    reg_mask.value = mask_of(best_)
                    m.work.slots = slots_f[i0 + best_]

                self._pending["finish"] = finish
                self._batch["samples"].append(s)
                H_cur = H_of(last_step)  # placeholder (end_batch() returns the real rows)
            elif self._batch is not None:  # differentiated by end_batch() together with the other loops
                self._batch["samples"].append(s)
                H_cur = H_of(best)
            else:
                H_cur = LoopFn.apply(sensor, bank.anchor(dev), self, s)
        else:
            H_cur = H_of(best).clone()
        return H_cur, reg_mask, last_logits, traces

    def flush_units(self) -> None:
        """Add the logged loops' region steps / active edges / destination regions to the model's
        counters: one batch of ops per arena chunk instead of one per loop."""
        m = self.model
        self.resolve_open()
        log, self._unit_log = self._unit_log, []
        if m._units is None or not log:
            return
        by_chunk = {}
        for chunk, i0, parent, mask0 in log:
            by_chunk.setdefault(id(chunk), (chunk, []))[1].append((i0, parent, mask0))
        for chunk, loops in by_chunk.values():
            field = chunk.fields["mask"]
            dev = field.device
            extra = torch.stack([mk for _, _, mk in loops]).float()  # the loops' own reg_mask_prev rows
            cap = field.shape[0]
            steps, prevs = [], []
            for li, (i0, parent, _) in enumerate(loops):
                for t, par in enumerate(parent):
                    steps.append(i0 + t)
                    prevs.append(i0 + par if par >= 0 else cap + li)
            allm = torch.cat([field.float(), extra], 0)
            masks = allm[torch.tensor(steps, device=dev)]
            ea = allm[torch.tensor(prevs, device=dev)][:, m.router.src_idx]  # [n,E]
            into = torch.zeros(len(steps), m.R, device=dev).index_add_(1, m.router.dst_idx, ea)
            # region-phase launches: the nb branches of an outer iteration are the streams of one launch
            nb = 1 if m.cfg.train_deterministic_inner_loop else max(1, m.cfg.B_br)
            grp = masks.view(-1, nb, m.R) if len(steps) % nb == 0 else masks.view(-1, 1, m.R)
            m._units[:5] += torch.stack([masks.sum(), ea.sum(), (into > 0).sum().float(),
                                         torch.tensor(float(grp.shape[0]), device=dev), grp.amax(1).sum()])
            eg = (ea.view(-1, nb, ea.shape[1]) if len(steps) % nb == 0 else ea.view(-1, 1, ea.shape[1])).amax(1)  # [launches,E]
            dg = torch.zeros(eg.shape[0], m.R, device=dev).index_add_(1, m.router.dst_idx, eg)
            m._units[12:14] += torch.stack([eg.sum(), (dg > 0).sum().float()])

    # ------------------------------------------------------------------ loop heads (HIP, no autograd)
    def _heads_setup(self, dev, nb: Optional[int] = None, lock: bool = False):
        """C structs + scratch for ic_heads_fwd / ic_gemv_chain / ic_steer_select for launches of nb
        streams (default: the branches of one loop), rebuilt when the parameters move (device change,
        deepcopy, load into new storage).  The pinned score block `ls` holds, per stream: the branch
        score at [0, n8), four statistics at [off_stats, +4*n8), the critic value of the stream as a
        candidate at [off_vhat, +n8), the verifier energy at [off_energy, +n8), n8 = max(8, nb)."""
        m = self.model
        hf = m._heads_flat
        if nb is None:
            nb = 1 if m.cfg.train_deterministic_inner_loop else max(1, m.cfg.B_br)
        ck = (nb, lock)  # (run_lockstep keeps its own structs: it leaves per-stream index arrays in them)
        hs = self._hs_all.get(ck)
        if hs is not None and ck in self._hs_ok and hs["key"][:3] == (dev, nb, hf.flat.data_ptr()):
            return hs  # checked in full since the last reset()
        hf.reflat()  # no-op while every facet / region-head Parameter still views the arena
        mods = [m.lm_head.gate]
        if m.plan is not None:
            mods += [m.plan.mlp_p.net[0], m.plan.mlp_p.net[2], m.plan.mlp_g.net[0], m.plan.mlp_g.net[2],
                     m.critic.mlp.net[0], m.critic.mlp.net[2]]
        if m.verify is not None:
            mods += [m.verify.net[0], m.verify.net[1], m.verify.net[3], m.verify.net[5]]
        key = (dev, nb, hf.flat.data_ptr()) + tuple(p.data_ptr() for l in mods for p in (l.weight, l.bias))
        self._hs_ok.add(ck)
        if hs is not None and hs["key"] == key:
            return hs
        R, d, V, Kf = m.R, m.d, m.V, m.lm_head.Kf
        n8 = max(8, nb)
        o_stats, o_vhat, o_energy = n8, 5 * n8, 6 * n8
        z = lambda *s: torch.zeros(*s, device=dev)
        # per-branch scratch blocks are contiguous: branch b of a batched call uses block b
        # planner / critic scratch: one block per candidate branch (the chain runs for both branches
        # of an outer iteration before their scores say which one steers the next iteration)
        # (steered loops: nobody on the host reads the score block while the loops run -- device memory)
        buf = dict(ls=(z(8 * n8) if lock == "steer" else torch.zeros(8 * n8).pin_memory()), ws_prev=z(nb, d), motor=z(d), p_h=z(nb, d), p=z(nb, d), g_h=z(nb, 2 * d), g=z(nb, d),
                   c_h=z(nb, 2 * d),
                   z=z(nb, Kf + Kf * V), per_region=z(nb, R, V), probs=z(nb, V), aux=z(nb, 4), gsq=z(nb, 2 * R))
        P = {k: t.data_ptr() for k, t in buf.items()}
        G = CLASSES["ic_gemv_desc"]

        def desc(g, lin_, xs, y, act, ln=None):
            g.W, g.b = lin_.weight.data_ptr(), lin_.bias.data_ptr()
            for i, (ptr_, n) in enumerate(xs):
                setattr(g, f"x{i}", ptr_)
                setattr(g, f"n{i}", n)
            g.y, g.rows, g.act = y, lin_.weight.shape[0], act
            if ln is not None:
                g.ln_g, g.ln_b, g.ln_eps = ln.weight.data_ptr(), ln.bias.data_ptr(), ln.eps
            assert lin_.weight.shape[1] == sum(n for _, n in xs) and lin_.weight.is_contiguous()

        ha = new("ic_heads_args", R=R, d=d, V=V, Kf=Kf, n_slots=m.work.slots.shape[0], motor_idx=m.io_idxs["motor"],
                 kwta_k=max(1, d // 4), lm_gate_W=m.lm_head.gate.weight, lm_gate_b=m.lm_head.gate.bias,
                 lm_W=hf.stacked("facet_W"), lm_b=hf.stacked("facet_b"), rh_W=hf.stacked("rh_W"),
                 rh_b=hf.stacked("rh_b"), z=buf["z"], per_region=buf["per_region"], probs=buf["probs"], gsq=buf["gsq"])
        ver = None
        if m.verify is not None:
            ver = (G * 3)()
            silu = ENUMS["IC_ACT_SILU"]
            ln, l1, l2, l3 = m.verify.net[0], m.verify.net[1], m.verify.net[3], m.verify.net[5]
            hid = l1.weight.shape[0]
            buf["h1"], buf["h2"] = z(nb, hid), z(nb, hid)
            P["h1"], P["h2"] = buf["h1"].data_ptr(), buf["h2"].data_ptr()
            desc(ver[0], l1, [(0, d), (P["probs"], V), (P["aux"], m.verify.aux_dim)], P["h1"], silu, ln)  # energy.py:17-30
            desc(ver[1], l2, [(P["h1"], hid)], P["h2"], silu)
            desc(ver[2], l3, [(P["h2"], hid)], 0, ENUMS["IC_ACT_NONE"])
            # batch strides (floats) used when the branches' verifier chains run as one launch per layer
            ver[0].x0_bs, ver[0].x1_bs, ver[0].x2_bs, ver[0].y_bs = R * d, V, 4, hid
            ver[1].x0_bs, ver[1].y_bs = hid, hid
            ver[2].x0_bs, ver[2].y_bs = hid, 1
        steer = None
        if m.plan is not None:
            steer = (G * 6)()
            gelu, none = ENUMS["IC_ACT_GELU"], ENUMS["IC_ACT_NONE"]
            desc(steer[0], mods[1], [(P["ws_prev"], d)], P["p_h"], gelu)             # heads.py:57-59
            desc(steer[1], mods[2], [(P["p_h"], d)], P["p"], none)
            desc(steer[2], mods[3], [(P["p"], d), (P["ws_prev"], d)], P["g_h"], gelu)
            desc(steer[3], mods[4], [(P["g_h"], 2 * d)], P["g"], none)
            desc(steer[4], mods[5], [(P["ws_prev"], d), (P["g"], d)], P["c_h"], gelu)  # heads.py:69-70
            desc(steer[5], mods[6], [(P["c_h"], 2 * d)], P["ls"] + o_vhat * 4, none)  # v_hat of candidate b at ls[o_vhat + b]
            for g_, (xb, yb) in zip(steer, (((d,), d), ((d,), d), ((d, d), 2 * d), ((2 * d,), d), ((d, d), 2 * d),
                                             ((2 * d,), 1))):
                g_.x0_bs, g_.y_bs = xb[0], yb
                if len(xb) > 1:
                    g_.x1_bs = xb[1]
        sa = new("ic_steer_args", R=R, d=d, motor_idx=m.io_idxs["motor"], ws_prev=buf["ws_prev"], motor=buf["motor"])
        # layer i of the verifier next to layer i of the planner chain, for ic_gemv_multi (refreshed per call)
        pairs = [(G * 2)() for _ in range(3)] if (ver is not None and steer is not None) else None
        tail = (G * 3)() if steer is not None else None
        hs = self._hs_all[ck] = dict(key=key, buf=buf, P=P, ha=ha, steer=steer, ver=ver, sa=sa, keep=mods, nb=nb,
                                     pairs=pairs, tail=tail, off=(o_stats, o_vhat, o_energy))
        return hs

    def _drive(self, steps, in_ptrs, H_of, K_inner, dev, chunk, i0, trace_views=False, zero_in=False):
        """The control flow of reference model.py:329-460 over step ids, heads on the HIP library.
        Within an outer iteration the branches are chained only through the region state: their
        gate, router, heads and verifier run as one launch sequence with one stream per branch;
        only the region phases run branch after branch."""
        step, front, region, prepare_step = steps
        m = self.model
        cfg, gate = m.cfg, m.gate
        L, st = lib(), stream_handle()
        # train_step never reads the traces after its backward: views of the arena (valid until reset())
        keep = (lambda t: t) if trace_views else (lambda t: t.clone())
        hs = self._heads_setup(dev)
        ha, ver, steer, sa, P, ls, nb = hs["ha"], hs["ver"], hs["steer"], hs["sa"], hs["P"], hs["buf"]["ls"], hs["nb"]
        ha_ref, sa_ref = ctypes.byref(ha), ctypes.byref(sa)
        R, d, motor = m.R, m.d, m.io_idxs["motor"]
        CP, CS = chunk.ptr, chunk.stride
        afd = "surprise_o" in CP
        aux_on = ver is not None and m.verify.aux_dim > 0
        aux_row = getattr(m, "_aux_override", None)
        if aux_on and aux_row is not None:  # reasoning_loop_batch: the rows of all samples built in one go
            hs["buf"]["aux"].copy_(aux_row)
        elif aux_on:  # model.py:396-404: [mixer energy, mean surprise (written per step), attention entropy]
            ent = getattr(m.local_mix.attn, "last_attn_entropy", None)
            a02 = torch.stack([m.local_mix.last_energy.detach().reshape(()),
                               (ent.detach().reshape(()).to(dev) if ent is not None else torch.zeros((), device=dev))])
            hs["buf"]["aux"][:, 0::2] = a02
        ha.aux_surprise = P["aux"] + 4 if aux_on else None
        n_slots = m.work.slots.shape[0]
        # the workspace the loop starts from, once per branch (stream-ordered reuse of one buffer:
        # the previous loop's first heads launch read it long before this copy runs)
        slots0 = hs["buf"].get("slots0")
        if slots0 is None or slots0.shape != (nb, n_slots, d):
            slots0 = hs["buf"]["slots0"] = torch.empty(nb, n_slots, d, device=dev)
        s0p = slots0.data_ptr()

        def settle() -> None:
            """Before this loop's first launch that writes the pinned score block (or its first
            sync): the previous loop's winner, then the workspace this loop starts from."""
            self.resolve_pending()
            slots0.copy_(m.work.slots.detach())

        slots_ptr = lambda i, b: CP["slots"] + (i0 + i) * CS["slots"] if i >= 0 else s0p + b * n_slots * d * 4
        lsp = P["ls"]
        heads_fwd, chain, vscore, steer_sel = L.ic_heads_fwd, L.ic_gemv_chain, L.ic_verify_score, L.ic_steer_select
        heads_gemv = L.ic_heads_gemv_fwd

        def heads(ids: List[int], pars: List[int], slots_from: List[int]) -> bool:
            """Workspace write + logits + statistics of the consecutive steps ids (one launch
            sequence).  False when the parents' blocks are not equally spaced (then: one by one)."""
            n = len(ids)
            a = i0 + ids[0]
            hp0, sp0 = in_ptrs(pars[0], 0)[0], slots_ptr(slots_from[0], 0)
            hps = sps = 0
            if n > 1:
                hps, sps = (in_ptrs(pars[1], 1)[0] - hp0) // 4, (slots_ptr(slots_from[1], 1) - sp0) // 4
                for b in range(2, n):
                    if (in_ptrs(pars[b], b)[0] - hp0) // 4 != b * hps or (slots_ptr(slots_from[b], b) - sp0) // 4 != b * sps:
                        return False
                if hps == 0 or sps == 0:
                    return False  # 0 means "contiguous" in the ABI
            ha.S, ha.Hpar_sstride, ha.slots_in_sstride = n, hps, sps
            ha.H, ha.mask = CP["H"] + a * CS["H"], CP["mask"] + a * CS["mask"]
            ha.H_par, ha.slots_in, ha.slots_out = hp0, sp0, CP["slots"] + a * CS["slots"]
            ha.surprise = CP["surprise_o"] + a * CS["surprise_o"] if afd else None
            ha.logits = CP["logits"] + a * CS["logits"]
            ha.stats = lsp + 8 * 4
            if ha.ws_mean:  # the first two planner layers ride in the heads' two launches (see verify())
                steer[0].nbatch = steer[1].nbatch = n
                rc = heads_gemv(ha_ref, ctypes.byref(steer[0]), ctypes.byref(steer[1]), st)
            else:
                rc = heads_fwd(ha_ref, st)
            if rc:
                check(rc, "ic_heads_fwd")
            return True

        pairs_, tail_ = hs["pairs"], hs["tail"]
        # ls is pinned host memory the kernels write directly (no device-to-host copy kernel in the
        # loop's critical path): wait for the stream, then read it on the host
        sync = torch.cuda.current_stream(dev).synchronize
        multi = L.ic_gemv_multi

        def verify(first: int, n: int, plan_too: bool) -> None:
            """Verifier energy + branch score of n consecutive steps (batched layers).  plan_too: the
            planner -> critic chain of the NEXT outer iteration runs for all n candidates in the same
            launches (its input, the mean of each step's active rows, was written by the heads): the
            first two layers rode in the heads' two launches, the next three ride with the verifier's, the
            last follows; the host picks the winner's v_hat once the scores are read."""
            ver[0].x0 = CP["H"] + (i0 + first) * CS["H"] + motor * d * 4
            ver[2].y = lsp + 48 * 4
            for g in ver:
                g.nbatch = n
            if not plan_too:
                rc = vscore(ver, 3, lsp + 8 * 4, lsp, st)
                if rc:
                    check(rc, "ic_verify_score")
                return
            for g in steer:
                g.nbatch = n
            for i_ in range(3):  # planner layers 0 and 1 already ran inside the heads' two launches
                pairs_[i_][0], pairs_[i_][1] = ver[i_], steer[i_ + 2]
                rc = multi(pairs_[i_], 2, lsp + 8 * 4, lsp if i_ == 2 else None, st)
                if rc:
                    check(rc, "ic_gemv_multi")
            tail_[0] = steer[5]
            rc = chain(tail_, 1, st)
            if rc:
                check(rc, "ic_gemv_chain")

        last = -1
        traces: List[torch.Tensor] = []
        if not cfg.train_deterministic_inner_loop and (ver is None or steer is None):
            # (TypeError: what the reference raises here -- it calls self.verify, which is None, model.py:425)
            raise TypeError("the branching reasoning loop needs the planner, critic and energy verifier "
                            "(reference model.py:376-438): 'NoneType' object is not callable")
        if cfg.train_deterministic_inner_loop:
            settle()
            cur, u_mean = -1, 0.0
            for k in range(K_inner):
                gate.value_bias = 0.0
                par = cur
                cur = step(par, k, u_mean, 0)
                heads([cur], [par], [par])
                traces.append(keep(H_of(cur)))
                if k + 1 < K_inner:
                    sync()
                    u_mean = float(ls[8])  # utils.py:127-130 of this step's logits (host sync)
            return cur, cur, traces

        B_br = cfg.B_br
        branches = [(-1, -1) for _ in range(B_br)]  # (step whose workspace slots, step whose H / mask)
        u_mean = 0.0
        planned = False  # True when the previous iteration already ran the planner for both candidates
        settled = False
        n_polled = 0
        lsn = hs["buf"].get("ls_np")
        if lsn is None:
            lsn = hs["buf"]["ls_np"] = ls.numpy()  # the same pinned words, read without a tensor op
        NAN = float("nan")

        def wait_vhat(n: int) -> bool:
            """Wait for the iteration's LAST kernel (the critic layer of the planner chain run for the n
            candidates, which writes v_hat into the pinned score block) by polling those words for
            the NaN the host left there, instead of a stream sync: the kernels before it finished
            earlier on the same stream and their words are in host memory before these are (measured:
            9.4 us launch-to-seen against 14.8 us with hipStreamSynchronize, scripts/poll_latency.py).
            False after ~10 ms without an answer (a v_hat that IS NaN): the caller syncs instead."""
            for _ in range(40000):
                for c in range(n):
                    v = lsn[40 + c]
                    if v != v:
                        break
                else:
                    return True
            return False

        for k in range(K_inner):
            if k == 0 and zero_in and self._vhat0 is not None:
                # every loop of a train_step starts from the shared all-zero state: the planner's
                # input (mean of the active rows: none) and so v_hat are the same as for the first loop
                vals = None
            elif not planned:
                if not settled:
                    settle()
                    settled = True
                # planner / critic on the best candidate so far, chosen on the device (model.py:371-386)
                sa.n_cand = len(branches)
                for c, (_, hstep) in enumerate(branches):
                    sa.H_cand[c], sa.mask_cand[c] = in_ptrs(hstep, c)
                sa.scores = lsp if k > 0 else None
                for g in steer:
                    g.nbatch = 1
                rc = steer_sel(sa_ref, st) or chain(steer, 6, st)
                if rc:
                    check(rc, "ic_steer_select/ic_gemv_chain")
            if k > 0 or not (zero_in and self._vhat0 is not None):
                if not (planned and POLL_SYNC and wait_vhat(n_polled)):
                    sync()
                vals = lsn.tolist()  # the one host wait of this outer iteration (numpy view: no tensor op)
                v_hat = vals[40]
                if k == 0 and zero_in:
                    self._vhat0 = v_hat
            else:
                v_hat = self._vhat0
            if k > 0:
                if len(branches) == 2:  # (score descending, index ascending)
                    top = [1, 0] if vals[1] > vals[0] else [0, 1]
                else:
                    top = sorted(range(len(branches)), key=lambda i: (-vals[i], i))[:min(B_br, 2)]
                if planned:
                    v_hat = vals[40 + top[0]]  # the winner's, from the chain run for every candidate
                branches = [branches[i] for i in top]
                trace_of = branches[0][1]
                u_mean = vals[8 + 4 * (B_br - 1)]
            gate.value_bias = v_hat
            burst = schedule_burst(u_mean, v_hat, m.R)
            pars = [branches[b][1] for b in range(B_br)]   # IndexError for B_br > 2, as in the reference
            sls = [branches[b][0] for b in range(B_br)]
            ids = front(pars, k, u_mean, burst, st, FUSE_STEP)  # gate + router of every branch ...
            region(ids[0], st, len(ids))                     # ... launched here: the branches share the state chain
            if k > 0:  # (host work that can wait goes after the launch)
                traces.append(keep(H_of(trace_of)))
            if not settled:  # the previous loop's winner is read while the GPU runs this loop's first step
                settle()
                settled = True
            # will there be a next iteration?  (should_halt only looks at this iteration's u_mean)
            planned = k + 1 < K_inner and not should_halt(None, u_mean, k, K_inner) and len(ids) <= 8
            ha.ws_mean = P["ws_prev"] if planned else None
            if planned and POLL_SYNC:
                n_polled = len(ids)
                lsn[40:40 + n_polled] = NAN  # replaced by the last kernel of this iteration's chain (see wait_vhat)
            if not heads(ids, pars, sls):
                raise RuntimeError("LoopEngine: branch inputs are not equally spaced (B_br > 2 is not supported)")
            verify(ids[0], len(ids), planned)
            branches = [(cur, cur) for cur in ids]
            last = ids[-1]
            if should_halt(None, u_mean, k, K_inner):
                break
            if k + 1 < K_inner:
                prepare_step(B_br, k + 1)  # while the GPU works through this iteration's launches
        if DEFER_SYNC and trace_views and self._batch is not None and last >= 0:
            # recorded loop of a batch: do not wait here (see resolve_pending)
            evs = self._events
            if evs is None:
                evs = self._events = [[torch.cuda.Event(), torch.cuda.Event()], 0]
            ev = evs[0][evs[1]]
            evs[1] ^= 1
            ev.record(torch.cuda.current_stream(dev))
            self._pending = {"event": ev, "ls": ls, "branches": branches, "B_br": B_br, "traces": traces, "H_of": H_of}
            return None, last, traces
        sync()
        vals = ls.tolist()
        top = sorted(range(len(branches)), key=lambda i: (-vals[i], i))[:min(B_br, 2)]
        branches = [branches[i] for i in top]
        traces.append(keep(H_of(branches[0][1])))
        return branches[0][1], last, traces



    # ------------------------------------------------------------------ device-steered strict loops
    def _drop_graph(self) -> None:
        gr, self._graph = self._graph, None
        if gr is not None:
            for ex in gr["exec"]:
                lib().ic_graph_destroy(ex)

    def steer_usable(self, dev, K_inner: int) -> bool:
        """Can the loops of a reasoning_loop_batch call run with the outer-iteration decisions taken on the
        device (run_steered)?  The branching loop with two branches, planner / critic / verifier present."""
        m = self.model
        cfg = m.cfg
        if not STEER or not self.enabled or dev.type != "cuda" or K_inner < 1:
            return False
        if cfg.train_deterministic_inner_loop or cfg.B_br != 2 or m.plan is None or m.verify is None:
            return False
        if cfg.profile or cfg.ff_dropout or (m.router.edge_transform_mode == "factorized" and not m.router.fact_lib) or m.router.E <= 0:
            return False
        return True

    def _steer_bufs(self, dev, N: int, K: int, C: int):
        """Static device buffers of the steered loops (fixed addresses from step to step: what a captured
        launch sequence needs): per-loop inputs, the control blocks, the per-loop records."""
        m = self.model
        key = (dev, N, K, C, m.d, max(m.router.E, 1))
        sb = getattr(self, "_sbufs", None)
        if sb is not None and sb["key"] == key:
            return sb
        W = ENUMS["IC_CTL_WORDS"]
        sb = dict(key=key, sens=torch.zeros(N, m.d, device=dev), fo=torch.ones(N, C, dtype=_U8, device=dev),
                  aux=torch.zeros(N, C, 4, device=dev), ebias=torch.zeros(max(m.router.E, 1), device=dev),
                  ctl=torch.zeros(N * K + 1, W, dtype=_I32, device=dev), rec=torch.zeros(N, 4, dtype=_I32, device=dev),
                  vhat0=torch.zeros(1, device=dev))
        self._sbufs = sb
        return sb

    def run_steered(self, K_inner: int, sensors: torch.Tensor, fo_all: torch.Tensor, aux_all, second=None, init=None):
        """Strict-sequential reasoning loops of a whole batch (reference model.py:462-503 over :329-460) with NO
        host wait inside: what the host decided between two outer iterations in _drive -- winner order, the
        winner's critic value, uncertainty, burst, halting (model.py:371-392,440-452) -- is decided by
        ic_loop_steer on the device and reaches the next iteration's launches through a control block
        (per-stream indirections of the kernels).  The host queues every launch of every loop back to back;
        an iteration behind a loop that halted is a pass-through (empty gate mask, state copied).  The loops
        run one after the other on one stream, sample j from the region state / workspace sample j-1 left
        behind: the reference's schedule, bit-identical to the host-steered path (tests/test_hip_steer.py).

        second (optional): the inputs above cover only the first N1 loops of the batch; the launches of those are
        queued, THEN second() is called for (sensors, fo_all, aux_all) of the other N - N1 loops and theirs are queued:
        the host work that produces them (train_step: the corrupted sequences and their token mixer pass) runs while
        the device is busy with the first loops.  Same launches in the same order either way.

        init (optional, no-grad calls): (H [R,d], mask [R]) the FIRST loop starts from instead of the all-zero state
        (generation: reasoning_loop hands each call the previous call's winner, reference generation.py:27-29); the
        critic value of its first iteration is then computed from that state.

        Call between begin_batch() and the use of the results.  sensors [N,d]; fo_all [N,C] uint8; aux_all
        [N,C,4] or None.  Returns (H [N,R,d] as ONE autograd node over `sensors` when grad is enabled,
        masks uint8 [N,R], last logits [N,V], traces: N lazy lists)."""
        m = self.model
        cfg, bank, router, gate = m.cfg, m.bank, m.router, m.gate
        dev = sensors.device
        R, d, E = m.R, m.d, router.E
        bt = self._batch
        chunk, base, KC, N = bt["chunk"], bt["base"], bt["n_max"], bt["n_loops"]
        C, K = 2, int(K_inner)
        N1 = int(sensors.shape[0])
        assert KC == K * C and (N == N1 if second is None else 0 < N1 < N)
        grad = torch.is_grad_enabled()
        Z = base - 1  # the slot before the first loop: the bank's state (begin_batch), H = 0, mask = 0, the workspace
        f, CP, CS = chunk.fields, chunk.ptr, chunk.stride
        L, st = lib(), stream_handle()
        self.counters["steered"] = self.counters.get("steered", 0) + 1
        sb = self._steer_bufs(dev, N, K, C)
        W = ENUMS["IC_CTL_WORDS"]
        ebias = router._cached_bias(m.reg_coords)  # (outside no_grad: requires_grad says whether the bias is trained)
        with torch.no_grad():  # (multi-tensor launches: one per dtype instead of a copy kernel per buffer)
            dst = [f["slots"][Z], sb["sens"][:N1], sb["fo"][:N1], sb["ebias"][:E]]
            src = [m.work.slots.detach(), sensors.detach(), fo_all, ebias.detach().reshape(-1)]
            if init is None:
                torch._foreach_zero_([f["H"][Z], f["mask"][Z]])
            else:
                assert not grad and N == 1, "run_steered: an initial state is a no-grad, single-loop feature"
                dst += [f["H"][Z], f["mask"][Z]]
                src += [init[0].detach().reshape(R, d), init[1].detach().to(_U8).reshape(R)]
            if aux_all is not None:
                dst.append(sb["aux"][:N1])
                src.append(aux_all)
            _copy_many(dst, src)
        hs = self._heads_setup(dev, C, lock="steer")
        ha, ver, steer, HP = hs["ha"], hs["ver"], hs["steer"], hs["P"]
        pairs_ = hs["pairs"]
        lsp = HP["ls"]

        # ---- critic value of the all-zero state (first iteration of every loop), left on the device
        pv, self._vhat0_pending = getattr(self, "_vhat0_pending", None), None
        if pv is not None and pv[3] == dev and init is None:
            # new_step() queued the chain; its result sits in the pinned score block of the host-steered set-up
            sb["vhat0"].copy_(pv[1][pv[2]:pv[2] + 1], non_blocking=True)
        else:
            sa = hs["sa"]
            sa.n_cand = 1
            sa.H_cand[0], sa.mask_cand[0] = CP["H"] + Z * CS["H"], CP["mask"] + Z * CS["mask"]
            sa.scores = None
            for g in steer:
                g.nbatch = 1
            y5 = steer[5].y
            steer[5].y = sb["vhat0"].data_ptr()
            rc = L.ic_steer_select(ctypes.byref(sa), st) or L.ic_gemv_chain(steer, 6, st)
            steer[5].y = y5
            if rc:
                check(rc, "ic_steer_select/ic_gemv_chain")

        # ---- everything the launches depend on: pointers and scalars that end up in kernel arguments
        tables = router._edge_tables(m.reg_coords)
        kw_k, kw_soft, kw_temp = bank.kwta_now()
        flags = bank._flags(kw_soft, bank.cells[0].disable_kwta, True)
        rP, bP = router._params(), bank.params()
        rspec = router.spec()
        prec, afd, rt_, use_pred = rspec.precision, bank.afd, bank.rt, bank.use_pred
        denom = float(max(1, cfg.K_inner - 1))
        dp_ = lambda ts: tuple(t.data_ptr() for t in ts if torch.is_tensor(t))
        key = (tuple(CP.values()), base, N, N1, K, R, d, E, flags, kw_k, kw_temp, cfg.k_active, cfg.K_inner, hs["key"],
               tuple(HP.values()), dp_(tables.values()), dp_(sb.values()), dp_(gate._tables(dev)), dp_(rP.values()),
               dp_(bP.values()), m.norm_in.weight.data_ptr(), gate.gain_ema.data_ptr(), gate.homeo.data_ptr(),
               bank.last_dir.data_ptr(), bank.last_norm.data_ptr(), (gate.w1, gate.w2, gate.w3, gate.w4, gate.w5, gate.w6),
               (rspec.R, rspec.d, rspec.E, rspec.nW, rspec.precision, rspec.shared_gemm), m.io_idxs["sensor"], m.io_idxs["motor"])

        def build():
            """C structs of the batch's launches: static fields once, the moving pointers as per-iteration images."""
            off, nbr = gate._tables(dev)
            ctl_p, rec_p = sb["ctl"].data_ptr(), sb["rec"].data_ptr()
            cw = W * 4
            o_in, o_ws, o_kt, o_t5, o_t6, o_dead = (4 * ENUMS[k_] for k_ in (
                "IC_CTL_IN_SLOT", "IC_CTL_SLOTS_SLOT", "IC_CTL_K_TOTAL", "IC_CTL_T5", "IC_CTL_T6", "IC_CTL_DEAD"))
            G = N * K
            gs = np.arange(G, dtype=np.int64)
            a0s = base + gs * C          # slot of branch 0 of global iteration g = j * K + k
            js, ks = gs // K, gs % K

            ga = new("ic_gate_args", S=C, R=R, d=d, sensor_idx=m.io_idxs["sensor"], motor_idx=m.io_idxs["motor"],
                     force_io=1, w1=gate.w1, w2=gate.w2, w3=gate.w3, w4=gate.w4, gain_ema=gate.gain_ema,
                     homeo=gate.homeo, nbr_off=off, nbr=nbr)
            ga.H = CP["H"]
            gimg = StructImage(ga)
            for k_ in ("scores", "mask"):
                gimg.linear((k_,), CP[k_] + base * CS[k_], C * CS[k_])
            gimg.linear(("in_slot",), ctl_p + o_in, cw)
            gimg.linear(("k_total_s",), ctl_p + o_kt, cw)
            gimg.linear(("t5_s",), ctl_p + o_t5, cw)
            gimg.linear(("t6_s",), ctl_p + o_t6, cw)
            gimg.column(("focus_any",), sb["fo"].data_ptr() + js * C)
            g_rows = gimg.images(gs)

            ra = ops._router_args(rspec, tables, rP, C)
            ra.ebias, ra.H, ra.mask_prev = sb["ebias"].data_ptr(), CP["H"], CP["mask"]
            rimg = StructImage(ra)
            for k_ in ["M"] + (["msg", "keyv", "q", "wgt"] if prec else (["fact_t", "fact_z", "fact_dact"] if router.fact_lib else [])):
                rimg.linear((k_,), CP[k_] + base * CS[k_], C * CS[k_])
            rimg.linear(("in_slot",), ctl_p + o_in, cw)
            r_rows = rimg.images(gs)

            P = dict(bP)
            P["norm_in_w"] = m.norm_in.weight
            fa = ops._region_args(bank.spec, P, C, 0.0, kw_k, flags | ENUMS["IC_F_CHAIN"], kw_temp)
            fa.last_dir, fa.last_norm = bank.last_dir.data_ptr(), bank.last_norm.data_ptr()
            fimg = StructImage(fa)
            state_pairs = [("num", "num_o"), ("den", "den_o"), ("dt", "dt_o"), ("var", "var_o"), ("surprise", "surprise_o"),
                           ("pred", "pred_o"), ("radius", "radius_o"), ("radius_var", "radius_var_o")]
            saved = ["x", "u1", "inv_s", "r_s", "k_s", "v_s", "y", "hpre", "gate"] + (["ynorm"] if rt_ else []) \
                + (["sterm"] if afd else [])
            for k_ in ["M", "mask", "H"] + saved:
                fimg.linear((k_,), CP[k_] + base * CS[k_], C * CS[k_])
            for k_, ko in state_pairs:
                if ko in CP:  # state in = what the slot before left; out = this iteration's slots
                    fimg.linear((k_,), CP[ko] + (base - 1) * CS[ko], C * CS[ko])
                    fimg.linear((ko,), CP[ko] + base * CS[ko], C * CS[ko])
            fimg.linear(("dead",), ctl_p + o_dead, cw)
            fimg.column(("sensor",), sb["sens"].data_ptr() + js * (d * 4))
            f_rows = fimg.images(gs)

            aux_on = ver is not None and m.verify.aux_dim > 0
            aux_p = sb["aux"].data_ptr()
            ha.S, ha.Hpar_sstride, ha.slots_in_sstride = C, 0, 0
            ha.H_par, ha.slots_in = CP["H"], CP["slots"]
            ha.stats = lsp + 8 * 4
            himg = StructImage(ha)
            for k_, fld in (("H", "H"), ("mask", "mask"), ("slots", "slots_out"), ("logits", "logits")):
                himg.linear((fld,), CP[k_] + base * CS[k_], C * CS[k_])
            if afd:
                himg.linear(("surprise",), CP["surprise_o"] + base * CS["surprise_o"], C * CS["surprise_o"])
            else:
                ha.surprise = None
            himg.linear(("par_slot",), ctl_p + o_in, cw)
            himg.linear(("slots_slot",), ctl_p + o_ws, cw)
            himg.column(("aux_surprise",), (aux_p + js * (C * 16) + 4) if aux_on else np.zeros(G, dtype=np.int64))
            himg.column(("ws_mean",), np.where(ks < K - 1, HP["ws_prev"], 0))  # the planner rides unless the loop ends here
            h_rows = himg.images(gs)

            motor = m.io_idxs["motor"]
            for g_ in ver:
                g_.nbatch = C
            for g_ in steer:
                g_.nbatch = C
            ver[2].y = lsp + 48 * 4
            for i_ in range(3):
                pairs_[i_][0], pairs_[i_][1] = ver[i_], steer[i_ + 2]
            crit = m.critic.mlp.net[2]
            la = new("ic_loop_steer_args", C=C, R=R, k_active=cfg.k_active, K_inner=K, zero_slot=Z, w5=float(gate.w5),
                     w6=float(gate.w6), vhat=lsp + 40 * 4, scores=lsp, stats=lsp + 8 * 4, vhat0=sb["vhat0"].data_ptr())
            cW, cb, cx, ccols = crit.weight.data_ptr(), crit.bias.data_ptr(), HP["c_h"], int(crit.weight.shape[1])
            la_ref = ctypes.byref(la)

            def launch_all(st, j0=0, j1=N) -> None:
                """Every launch of loops j0 .. j1-1 on stream st: (the first control block,) then loop after loop."""
                if j0 == 0:
                    la.crit_x, la.crit_cols = None, 0
                    la.mode, la.ctl_next = ENUMS["IC_STEER_INIT"], ctl_p
                    rc = L.ic_loop_steer(la_ref, st)
                    if rc:
                        check(rc, "ic_loop_steer")
                la.crit_x, la.crit_cols = cx, ccols
                STEP, LAST = ENUMS["IC_STEER_STEP"], ENUMS["IC_STEER_LAST"]
                ga_ref, ra_ref, fa_ref, ha_ref = ctypes.byref(ga), ctypes.byref(ra), ctypes.byref(fa), ctypes.byref(ha)
                s0_ref, s1_ref = ctypes.byref(steer[0]), ctypes.byref(steer[1])
                step_fwd, heads_gemv, heads_fwd, multi, vscore, steer_fn = (L.ic_loop_step_fwd, L.ic_heads_gemv_fwd, L.ic_heads_fwd,
                                                                             L.ic_gemv_multi, L.ic_verify_score, L.ic_loop_steer)
                Hm = CP["H"] + motor * d * 4
                HS = CS["H"]
                g = j0 * K
                for j in range(j0, j1):
                    ax = aux_p + j * (C * 16)
                    ver[0].x2 = ax
                    pairs_[0][0].x2 = ax
                    la.rec = rec_p + j * 16
                    for k in range(K):
                        a0 = base + g * C
                        gimg.load(g_rows[g])
                        rimg.load(r_rows[g])
                        fimg.load(f_rows[g])
                        fa.step_pos = float(k) / denom
                        rc = step_fwd(ga_ref, ra_ref, fa_ref, st)
                        if rc:
                            check(rc, "ic_loop_step_fwd")
                        himg.load(h_rows[g])
                        x0 = Hm + a0 * HS
                        la.k, la.slot0, la.ctl = k, a0, ctl_p + g * cw
                        if k < K - 1:
                            rc = heads_gemv(ha_ref, s0_ref, s1_ref, st)
                            if rc:
                                check(rc, "ic_heads_gemv_fwd")
                            pairs_[0][0].x0 = x0
                            rc = (multi(pairs_[0], 2, lsp + 8 * 4, None, st) or multi(pairs_[1], 2, lsp + 8 * 4, None, st)
                                  or multi(pairs_[2], 2, lsp + 8 * 4, lsp, st))
                            if rc:
                                check(rc, "ic_gemv_multi")
                            la.mode, la.crit_W, la.crit_b, la.ctl_next = STEP, cW, cb, ctl_p + (g + 1) * cw
                        else:
                            rc = heads_fwd(ha_ref, st)
                            if rc:
                                check(rc, "ic_heads_fwd")
                            ver[0].x0 = x0
                            rc = vscore(ver, 3, lsp + 8 * 4, lsp, st)
                            if rc:
                                check(rc, "ic_verify_score")
                            la.mode, la.crit_W, la.crit_b = LAST, None, None
                            la.ctl_next = ctl_p + (g + 1) * cw  # (the block after the last loop's is never read)
                        rc = steer_fn(la_ref, st)
                        if rc:
                            check(rc, "ic_loop_steer")
                        g += 1


            return launch_all

        def second_inputs():
            """The other loops' inputs (second()), into the static buffers behind the first loops' launches."""
            s2, fo2, aux2 = second()
            assert s2.shape[0] == N - N1
            with torch.no_grad():
                sb["sens"][N1:].copy_(s2.detach())
                sb["fo"][N1:].copy_(fo2)
                if aux2 is not None:
                    sb["aux"][N1:].copy_(aux2)
            return s2

        parts = [(0, N)] if second is None else [(0, N1), (N1, N)]
        use_graph = GRAPH and not hip_mod.PROF_ON and not self._graph_off
        gr = self._graph
        sensors2 = None
        if use_graph and gr is not None and gr["key"] == key:
            for i, ex in enumerate(gr["exec"]):  # the same launch sequence as last step: one call (per part)
                if i and second is not None:
                    sensors2 = second_inputs()
                rc = L.ic_graph_launch(ex, st)
                if rc:
                    check(rc, "ic_graph_launch")
        elif use_graph:
            # capture the sequence once (on a stream of its own: the default stream cannot be captured), then replay it
            self._drop_graph()
            launch_all = build()
            cs = self._cap_stream
            if cs is None or cs.device != dev:
                cs = self._cap_stream = torch.cuda.Stream(dev)
            execs, nodes = [], 0
            try:
                for j0, j1 in parts:
                    ex, nn = ctypes.c_void_p(), ctypes.c_int32(0)
                    with torch.cuda.stream(cs):
                        h = stream_handle()
                        rc = L.ic_graph_begin(h)
                        if rc:
                            check(rc, "ic_graph_begin")
                        try:
                            launch_all(h, j0, j1)
                        finally:
                            rc = L.ic_graph_end(h, ctypes.byref(ex), ctypes.byref(nn))
                        if rc:
                            check(rc, "ic_graph_end")
                    execs.append(ex)
                    nodes += nn.value
            except RuntimeError as exc:
                # a capture the runtime refuses (nothing has run: captured launches only execute on replay) must not
                # cost the step: from now on this engine launches call by call, as with IC_GRAPH=0
                for ex in execs:
                    L.ic_graph_destroy(ex)
                execs = None
                self._graph_off = True
                warnings.warn(f"LoopEngine: HIP graph capture failed ({exc}); launching the loops call by call from now on")
            if execs is not None:
                self._graph = {"key": key, "exec": execs, "nodes": nodes}
                self.counters["graph_captures"] = self.counters.get("graph_captures", 0) + 1
                for i, ex in enumerate(execs):
                    if i and second is not None:
                        sensors2 = second_inputs()
                    rc = L.ic_graph_launch(ex, st)
                    if rc:
                        check(rc, "ic_graph_launch")
            else:
                launch_all = build()
                for i, (j0, j1) in enumerate(parts):
                    if i and second is not None:
                        sensors2 = second_inputs()
                    launch_all(st, j0, j1)
        else:
            launch_all = build()
            for i, (j0, j1) in enumerate(parts):
                if i and second is not None:
                    sensors2 = second_inputs()
                launch_all(st, j0, j1)
        if sensors2 is not None:
            sensors = torch.cat([sensors, sensors2], 0)

        # ---- results, gathered on the device through the loops' records: nothing here waits for the GPU
        with torch.no_grad():
            recl = sb["rec"].long()
            bslot, lslot = recl[:, 0], recl[:, 1]
            H_out = f["H"].index_select(0, bslot)
            masks = f["mask"].index_select(0, bslot)
            logits = f["logits"].index_select(0, lslot)
            li = base + N * KC - 1  # (halted loops pass their state through to their block's last slot)
            nar = torch.narrow
            bank.num, bank.den, bank.dt = nar(f["num_o"], 0, li, 1), nar(f["den_o"], 0, li, 1), nar(f["dt_o"], 0, li, 1)
            if afd:
                bank.var, bank.surprise = nar(f["var_o"], 0, li, 1), nar(f["surprise_o"], 0, li, 1)
            if use_pred:
                bank.pred = nar(f["pred_o"], 0, li, 1)
            if rt_:
                bank.radius, bank.radius_var = nar(f["radius_o"], 0, li, 1), nar(f["radius_var_o"], 0, li, 1)
            if prec:
                router._last_wgt = f["wgt"].index_select(0, lslot[N - 1:N])[:, :E]
            m.work.slots = f["slots"].index_select(0, bslot[N - 1:N])[0]
        self._state_in_arena = True
        bt["steered"] = dict(sb=sb, K=K, C=C, Z=Z, flags=flags, kwta_k=kw_k, kwta_temp=kw_temp, keep=(sb["fo"], tables, rP, bP),
                             ebias_grad=bool(ebias.requires_grad), norm_in_w=m.norm_in.weight, denom=denom)
        traces = [SteeredTraces(self, bt, j) for j in range(N)]
        self._steered_open = bt
        if not grad:  # nothing may refer to the arena afterwards: the traces are copies
            traces = [[t.clone() for t in tr] for tr in traces]
            self._batch = None
            self._rescue_state()
            return H_out, masks, logits, traces
        bt["H_out"] = H_out
        self._batch = None
        H = LoopBatchFn.apply(sensors, bank.anchor(dev), self, bt)
        return H, masks, logits, traces

    def resolve_steered(self, bt) -> None:
        """Host-side record of the steered loops of a batch (the backward's parents and winners, traces, counters):
        ONE read of the control blocks the device wrote, after the loops ran."""
        sp = bt.pop("steered", None)
        if sp is None:
            return
        m = self.model
        sb, K, C, Z = sp["sb"], sp["K"], sp["C"], sp["Z"]
        chunk, base, KC, N = bt["chunk"], bt["base"], bt["n_max"], bt["n_loops"]
        f = chunk.fields
        ctl, rec = self._steered_record(sp)
        counted = sp.get("counted", False)  # (resolve_open's light path has already booked the counters)
        DEAD, TOP, VH = ENUMS["IC_CTL_DEAD"], ENUMS["IC_CTL_TOP"], ENUMS["IC_CTL_VHAT"]
        samples, n_inner = [], 0
        step_pos_all = [float(t // C) / sp["denom"] for t in range(KC)]
        for j in range(N):
            i0 = base + j * KC
            parent: List[int] = []
            n_it = 0
            for k in range(K):
                blk = ctl[j * K + k]
                if blk[DEAD]:
                    break
                if k == 0:
                    parent += [-1] * C
                else:
                    t0_ = int(blk[TOP])
                    parent += [(k - 1) * C + t0_, (k - 1) * C + (1 - t0_)]
                n_it += 1
            n = n_it * C
            assert n == int(rec[j, 2]) and n_it == int(rec[j, 3]), (j, n, rec[j])
            n_inner += n
            s = _Sample()
            s.chunk, s.i0, s.ts, s.n, s.parent, s.best = chunk, i0, 1, n, parent, int(rec[j, 0]) - i0
            s.step_pos = step_pos_all[:n]
            s.init = {k_: f[ko][i0 - 1] for k_, ko in (("num", "num_o"), ("den", "den_o"), ("var", "var_o"), ("pred", "pred_o"),
                                                     ("radius", "radius_o"), ("radius_var", "radius_var_o"),
                                                     ("surprise", "surprise_o"), ("dt", "dt_o")) if ko in f}
            s.H0, s.mask0 = f["H"][Z:Z + 1], f["mask"][Z:Z + 1]
            s.sensor, s.ebias, s.norm_in_w = sb["sens"][j], sb["ebias"], sp["norm_in_w"]
            s.flags, s.kwta_k, s.kwta_temp = sp["flags"], sp["kwta_k"], sp["kwta_temp"]
            s.keep, s.ebias_grad, s.zero_in = sp["keep"], sp["ebias_grad"], True
            samples.append(s)
            if m._units is not None:
                self._unit_log.append((chunk, i0, list(parent), f["mask"][Z]))
        bt["samples"] = samples
        if counted:
            return
        m.counters["inner_steps"] += n_inner
        # what the host-steered loop leaves in the gate: the critic value its last iteration ran with
        last = ctl[(N - 1) * K + max(0, int(rec[N - 1, 3]) - 1)]
        m.gate.value_bias = float(np.array([last[VH]], dtype=np.int32).view(np.float32)[0])

    # ------------------------------------------------------------------ multi-stream mode (lockstep loops)
    MAX_LOCKSTEP_STREAMS = 64  # ic_heads_fwd takes at most 64 streams per launch

    def lockstep_usable(self, dev, n_loops: int) -> bool:
        m = self.model
        C = 1 if m.cfg.train_deterministic_inner_loop else max(1, m.cfg.B_br)
        if not self.enabled or dev.type != "cuda" or n_loops < 1 or C > 2:
            return False
        if m.cfg.profile or m.cfg.ff_dropout or (m.router.edge_transform_mode == "factorized" and not m.router.fact_lib):
            return False
        if not m.cfg.train_deterministic_inner_loop and (m.plan is None or m.verify is None):
            return False
        if torch.is_grad_enabled() and not m._logits_detached:
            return False
        b = m.bank
        if b.S == 1 and b.num.device == dev and any(
                getattr(b, k).requires_grad for k in ("num", "den", "var", "pred", "radius", "radius_var")):
            return False
        return True

    def _vhat_zero_state(self, dev, H_ptr: int, mask_ptr: int) -> float:
        """Critic value of the all-zero input state (what every loop of a batch starts from)."""
        if self._vhat0 is not None:
            return self._vhat0
        hs = self._heads_setup(dev)
        sa, steer, ls = hs["sa"], hs["steer"], hs["buf"]["ls"]
        L, st = lib(), stream_handle()
        sa.n_cand = 1
        sa.H_cand[0], sa.mask_cand[0] = H_ptr, mask_ptr
        sa.scores = None
        for g in steer:
            g.nbatch = 1
        rc = L.ic_steer_select(ctypes.byref(sa), st) or L.ic_gemv_chain(steer, 6, st)
        if rc:
            check(rc, "ic_steer_select/ic_gemv_chain")
        torch.cuda.current_stream(dev).synchronize()
        self._vhat0 = float(ls[hs["off"][1]])
        return self._vhat0

    def run_lockstep(self, K_inner: int, sensors: torch.Tensor, focus_any: torch.Tensor, aux_rows):
        """Multi-stream mode: N INDEPENDENT reasoning loops advance in lockstep, each from
        the region state / workspace the bank holds now and from all-zero H_prev.  Per outer iteration
        the gate, router, heads, verifier and planner chain of all N * B_br (loop, branch) pairs are
        one launch sequence; the region phase runs once per branch over the N loops (the branches of
        one loop share its state chain, reference model.py:376-389, the loops do not).  The host reads
        one pinned score block per iteration and steers every loop separately (winner, critic value,
        uncertainty, burst, halting: model.py:371-392 per loop), handing the kernels per-stream parents
        and gate scalars (ic_gate_args.in_slot / t5_s / t6_s / k_total_s).

        Every loop computes what a batch-1 reasoning_loop from the same initial state computes; what
        differs from reasoning_loop_batch's strict schedule is only that sample j does NOT start from
        the state sample j-1 left behind.  Afterwards the bank holds the LAST loop's final state.

        sensors [N,d]; focus_any [N] uint8; aux_rows [N,4] or None (verifier aux: mixer energy, -, attention
        entropy, -).  Returns (H [N,R,d] -- one autograd node over `sensors` when grad is enabled --,
        masks uint8 [N,R], last logits [N,V], traces: list of N lists)."""
        m = self.model
        cfg, bank, router, gate = m.cfg, m.bank, m.router, m.gate
        dev = sensors.device
        R, d, E = m.R, m.d, router.E
        N = int(sensors.shape[0])
        det = bool(cfg.train_deterministic_inner_loop)
        C = 1 if det else max(1, cfg.B_br)
        S, K = C * N, int(K_inner)
        KC = K * C
        if S > self.MAX_LOCKSTEP_STREAMS:
            raise ValueError(f"run_lockstep: {N} loops x {C} branches > {self.MAX_LOCKSTEP_STREAMS} streams per launch")
        grad = torch.is_grad_enabled()
        L, st = lib(), stream_handle()
        self.counters["lockstep"] += 1
        self.resolve_pending()
        if bank.S != 1 or bank.num.device != dev:
            bank.flat.reflat()
            bank.reset_state(dev, 1)
        if not grad:
            self.reset()
        need = (KC + 1) * N
        arena = self._get_arena(dev, KC)
        if arena.chunk_steps < need:
            self._rescue_state()
            self._arena = arena = StepArena(self._shapes(), chunk_steps=need)
        chunk, ib = arena.reserve(need, dev)
        base = ib + N
        f, CP, CS = chunk.fields, chunk.ptr, chunk.stride
        assert CS["H"] == R * d * 4 and CS["mask"] == R
        n_slots = m.work.slots.shape[0]

        # ---- block of N copies of what every loop starts from (slots ib .. ib+N-1)
        state_pairs = [("num", "num_o"), ("den", "den_o"), ("dt", "dt_o"), ("var", "var_o"), ("surprise", "surprise_o"),
                       ("pred", "pred_o"), ("radius", "radius_o"), ("radius_var", "radius_var_o")]
        with torch.no_grad():
            for k_, ko in state_pairs:
                if ko in f:
                    f[ko][ib:ib + N] = getattr(bank, k_).detach().reshape((1,) + tuple(f[ko].shape[1:]))
            f["H"][ib:ib + N].zero_()
            f["mask"][ib:ib + N].zero_()
            f["slots"][ib:ib + N] = m.work.slots.detach()

        # ---- structs
        tables = router._edge_tables(m.reg_coords)
        ebias = router._cached_bias(m.reg_coords)
        ebias_c = ebias.detach().contiguous()
        norm_in_w = m.norm_in.weight
        kw_k, kw_soft, kw_temp = bank.kwta_now()
        flags = bank._flags(kw_soft, bank.cells[0].disable_kwta, True)
        rP, bP = router._params(), bank.params()
        off, nbr = gate._tables(dev)
        afd, rt_, use_pred = bank.afd, bank.rt, bank.use_pred
        rspec = router.spec()
        prec = rspec.precision
        fo = focus_any.to(_U8).reshape(1, N).expand(C, N).contiguous()  # stream s = b * N + j
        Hb, Mb, Sb = CP["H"] + ib * CS["H"], CP["mask"] + ib * CS["mask"], CP["slots"] + ib * CS["slots"]
        ga = new("ic_gate_args", S=S, R=R, d=d, sensor_idx=m.io_idxs["sensor"], motor_idx=m.io_idxs["motor"],
                 force_io=1, w1=gate.w1, w2=gate.w2, w3=gate.w3, w4=gate.w4, gain_ema=gate.gain_ema,
                 homeo=gate.homeo, nbr_off=off, nbr=nbr)
        ga.focus_any, ga.H = fo.data_ptr(), Hb
        ra = ops._router_args(rspec, tables, rP, S)
        ra.ebias, ra.H, ra.mask_prev = ebias_c.data_ptr(), Hb, Mb
        P = dict(bP)
        P["norm_in_w"] = norm_in_w
        fa = ops._region_args(bank.spec, P, N, 0.0, kw_k, flags, kw_temp)
        sens_c = sensors.detach().float().contiguous()
        fa.sensor = sens_c.data_ptr()
        last_dir, last_norm = torch.zeros(N, R, d, device=dev), torch.zeros(N, R, device=dev)
        fa.last_dir, fa.last_norm = last_dir.data_ptr(), last_norm.data_ptr()
        ga_ref, ra_ref, fa_ref = ctypes.byref(ga), ctypes.byref(ra), ctypes.byref(fa)

        hs = self._heads_setup(dev, S, lock=True)
        ha, ver, steer, HP, ls = hs["ha"], hs["ver"], hs["steer"], hs["P"], hs["buf"]["ls"]
        if "ctl_host" not in hs["buf"]:  # [in_slot | k_total | t5 | t6] per stream; reused by every call (each ends in a sync)
            hs["buf"]["ctl_host"] = torch.zeros(4 * S, dtype=_I32).pin_memory()
            hs["buf"]["ctl_dev"] = torch.zeros(4 * S, dtype=_I32, device=dev)
        ctl_host, ctl_dev = hs["buf"]["ctl_host"], hs["buf"]["ctl_dev"]
        ci = ctl_host.numpy()
        cf = ci.view(np.float32)
        cp = ctl_dev.data_ptr()
        ga.in_slot, ga.k_total_s, ga.t5_s, ga.t6_s = cp, cp + 4 * S, cp + 8 * S, cp + 12 * S
        ra.in_slot = cp
        pairs_, tail_ = hs["pairs"], hs["tail"]
        o_stats, o_vhat, o_energy = hs["off"]
        lsp = HP["ls"]
        lsn = hs["buf"].get("ls_np")
        if lsn is None:
            lsn = hs["buf"]["ls_np"] = ls.numpy()
        ha_ref = ctypes.byref(ha)
        ha.S, ha.Hpar_sstride, ha.slots_in_sstride = S, 0, 0
        ha.H_par, ha.par_slot = Hb, cp
        ha.slots_in, ha.slots_slot = Sb, cp
        ha.stats = lsp + o_stats * 4
        aux_on = ver is not None and m.verify.aux_dim > 0
        if aux_on:
            ab = hs["buf"]["aux"]
            ab.zero_()
            if aux_rows is not None:
                ab.view(C, N, 4).copy_(aux_rows.reshape(1, N, 4).to(dev))
        ha.aux_surprise = HP["aux"] + 4 if aux_on else None
        motor = m.io_idxs["motor"]

        out_names = [("num", "num_o"), ("den", "den_o"), ("dt", "dt_o")]
        if afd:
            out_names += [("var", "var_o"), ("surprise", "surprise_o")]
        if use_pred:
            out_names += [("pred", "pred_o")]
        if rt_:
            out_names += [("radius", "radius_o"), ("radius_var", "radius_var_o")]
        saved = ["x", "u1", "inv_s", "r_s", "k_s", "v_s", "y", "hpre", "gate"] + (["ynorm"] if rt_ else []) \
            + (["sterm"] if afd else [])
        rsaved = ["msg", "keyv", "q", "wgt"] if prec else (["fact_t", "fact_z", "fact_dact"] if router.fact_lib else [])
        denom = float(max(1, cfg.K_inner - 1))
        w5, w6 = gate.w5, gate.w6
        sync = torch.cuda.current_stream(dev).synchronize
        NAN = float("nan")

        # ---- host-side record of every loop
        pars = [[-1] * C for _ in range(N)]       # steps whose output feeds branch b of the coming iteration
        parent = [[] for _ in range(N)]           # parent step of every executed step (the backward's chain)
        u_mean, v_hat = [0.0] * N, [0.0] * N
        done = [False] * N                        # the loop has halted (reference should_halt) or finished
        halt_now = [False] * N
        best, n_steps = [-1] * N, [0] * N
        trace_steps: List[List[int]] = [[] for _ in range(N)]  # the winner of every executed iteration (tensors at the end)

        def settle(k_prev: int) -> None:
            """Scores of iteration k_prev are in the pinned block: winner, next parents, u_mean, v_hat per loop."""
            vals = lsn.tolist()
            t0_ = k_prev * C
            for j in range(N):
                if done[j]:
                    continue
                if C == 2:
                    top = [1, 0] if vals[N + j] > vals[j] else [0, 1]
                else:
                    top = [0]
                w = t0_ + top[0]
                best[j] = w
                trace_steps[j].append(w)
                if halt_now[j] or k_prev + 1 >= K:
                    done[j] = True
                    continue
                pars[j] = [t0_ + i for i in top]
                u_mean[j] = vals[o_stats + 4 * ((C - 1) * N + j)]
                v_hat[j] = 0.0 if det else vals[o_vhat + top[0] * N + j]

        if not det:
            if grad:
                self._resolve_vhat0(dev)
            v0 = self._vhat_zero_state(dev, Hb, Mb)
            v_hat = [v0] * N
        k_last, was_planned = -1, False
        with torch.no_grad():
            for k in range(K):
                if k > 0:
                    polled = False
                    if POLL_SYNC and was_planned:
                        for _ in range(40000):
                            w_ = lsn[o_vhat:o_vhat + S]
                            if not (w_ != w_).any():
                                polled = True
                                break
                    if not polled:
                        sync()
                    settle(k - 1)
                    if all(done):
                        break
                # per-stream inputs of this iteration's launches (stream s = b * N + j), written a slice at a time
                kt = [cfg.k_active] * N if det else [max(0, cfg.k_active + schedule_burst(u_mean[j], v_hat[j], R)) for j in range(N)]
                t5 = [0.0] * N if det else [w5 * v for v in v_hat]
                t6 = [w6 * u for u in u_mean]
                for b in range(C):
                    ci[b * N:(b + 1) * N] = [j if pars[j][b] < 0 else N + pars[j][b] * N + j for j in range(N)]
                    ci[S + b * N:S + (b + 1) * N] = kt
                    cf[2 * S + b * N:2 * S + (b + 1) * N] = t5
                    cf[3 * S + b * N:3 * S + (b + 1) * N] = t6
                for j in range(N):
                    if not done[j]:
                        parent[j].extend(pars[j])
                gate.value_bias = 0.0 if det else v_hat[N - 1]
                ctl_dev.copy_(ctl_host, non_blocking=True)
                a0 = base + k * C * N
                ga.scores, ga.mask = CP["scores"] + a0 * CS["scores"], CP["mask"] + a0 * CS["mask"]
                if E > 0:
                    ra.M = CP["M"] + a0 * CS["M"]
                    for k_ in rsaved:
                        setattr(ra, k_, CP[k_] + a0 * CS[k_])
                    rc = L.ic_front_fwd(ga_ref, ra_ref, st)
                else:
                    rc = L.ic_gate_fwd(ga_ref, st)
                if rc:
                    check(rc, "ic_front_fwd")
                fa.step_pos = float(k) / denom
                for b in range(C):  # branch b of every loop: state from the block before (branch b-1 / last iteration)
                    a = a0 + b * N
                    fa.M, fa.mask, fa.H = CP["M"] + a * CS["M"], CP["mask"] + a * CS["mask"], CP["H"] + a * CS["H"]
                    for k_, ko in state_pairs:
                        if ko in CP:
                            setattr(fa, k_, CP[ko] + (a - N) * CS[ko])
                    for k_, ko in out_names:
                        setattr(fa, ko, CP[ko] + a * CS[ko])
                    for k_ in saved:
                        setattr(fa, k_, CP[k_] + a * CS[k_])
                    rc = L.ic_region_phase_fwd(fa_ref, st)
                    if rc:
                        check(rc, "ic_region_phase_fwd")
                m.counters["inner_steps"] += sum(C for j in range(N) if not done[j])
                for j in range(N):
                    if not done[j]:
                        n_steps[j] = (k + 1) * C
                        halt_now[j] = (not det) and should_halt(None, u_mean[j], k, K)
                planned = (not det) and k + 1 < K and not all(done[j] or halt_now[j] for j in range(N))
                # heads of all S streams, the verifier and (planned) the planner chain of every candidate
                ha.H, ha.mask = CP["H"] + a0 * CS["H"], CP["mask"] + a0 * CS["mask"]
                ha.slots_out = CP["slots"] + a0 * CS["slots"]
                ha.surprise = CP["surprise_o"] + a0 * CS["surprise_o"] if afd else None
                ha.logits = CP["logits"] + a0 * CS["logits"]
                ha.ws_mean = HP["ws_prev"] if planned else None
                if planned:
                    if POLL_SYNC:
                        lsn[o_vhat:o_vhat + S] = NAN
                    steer[0].nbatch = steer[1].nbatch = S
                    rc = L.ic_heads_gemv_fwd(ha_ref, ctypes.byref(steer[0]), ctypes.byref(steer[1]), st)
                else:
                    rc = L.ic_heads_fwd(ha_ref, st)
                if rc:
                    check(rc, "ic_heads_fwd")
                if ver is not None and not det:
                    ver[0].x0 = CP["H"] + a0 * CS["H"] + motor * d * 4
                    ver[2].y = lsp + o_energy * 4
                    for g in ver:
                        g.nbatch = S
                    if not planned:
                        rc = L.ic_verify_score(ver, 3, lsp + o_stats * 4, lsp, st)
                        if rc:
                            check(rc, "ic_verify_score")
                    else:
                        for g in steer:
                            g.nbatch = S
                        for i_ in range(3):
                            pairs_[i_][0], pairs_[i_][1] = ver[i_], steer[i_ + 2]
                            rc = L.ic_gemv_multi(pairs_[i_], 2, lsp + o_stats * 4, lsp if i_ == 2 else None, st)
                            if rc:
                                check(rc, "ic_gemv_multi")
                        tail_[0] = steer[5]
                        rc = L.ic_gemv_chain(tail_, 1, st)
                        if rc:
                            check(rc, "ic_gemv_chain")
                k_last, was_planned = k, planned
            if not all(done):
                sync()
                for j in range(N):
                    halt_now[j] = True
                settle(k_last)

        # ---- results; the bank continues from the last loop's final state (views of its last slot)
        # one index tensor for everything gathered out of the arena: winners, last steps, every iteration's winner
        flat_tr = [base + w * N + j for j in range(N) for w in trace_steps[j]]
        idx = torch.tensor([base + best[j] * N + j for j in range(N)] + [base + (n_steps[j] - 1) * N + j for j in range(N)]
                           + flat_tr, device=dev)
        bslot, lslot = idx[:N], idx[N:2 * N]
        masks = f["mask"][bslot]
        logits = f["logits"][lslot]
        tr_rows = f["H"][idx[2 * N:]]  # (a copy: valid after the arena is recycled)
        traces, o_ = [], 0
        for j in range(N):
            traces.append(list(tr_rows[o_:o_ + len(trace_steps[j])].unbind(0)))
            o_ += len(trace_steps[j])
        if m._units is not None:  # bench / sweep unit counters: executed (loop, step) pairs' active regions
            ex = torch.tensor([base + t * N + j for j in range(N) for t in range(n_steps[j])], device=dev)
            m._units[0] += f["mask"][ex].sum().float()
        jl = N - 1
        li = base + (n_steps[jl] - 1) * N + jl
        nar = torch.narrow
        bank.num, bank.den, bank.dt = nar(f["num_o"], 0, li, 1), nar(f["den_o"], 0, li, 1), nar(f["dt_o"], 0, li, 1)
        if afd:
            bank.var, bank.surprise = nar(f["var_o"], 0, li, 1), nar(f["surprise_o"], 0, li, 1)
        if use_pred:
            bank.pred = nar(f["pred_o"], 0, li, 1)
        if rt_:
            bank.radius, bank.radius_var = nar(f["radius_o"], 0, li, 1), nar(f["radius_var_o"], 0, li, 1)
        if prec:
            router._last_wgt = nar(nar(f["wgt"], 0, li, 1), 1, 0, E)
        m.work.slots = f["slots"][base + best[jl] * N + jl]
        self._state_in_arena = True
        if not grad:
            H_out = f["H"][bslot]  # (index copy: nothing below refers to the arena)
            self._rescue_state()
            return H_out, masks, logits, traces
        samples = []
        step_pos = [float(t // C) / denom for t in range(KC)]
        zero_H = getattr(m, "_zero_H", None)
        for j in range(N):
            s = _Sample()
            s.chunk, s.i0, s.ts, s.n, s.parent, s.best = chunk, base + j, N, n_steps[j], parent[j][:n_steps[j]], best[j]
            s.step_pos = step_pos[:n_steps[j]]
            s.init, s.H0, s.mask0, s.sensor = None, None, None, sens_c[j]
            s.ebias, s.norm_in_w = ebias_c, norm_in_w
            s.flags, s.kwta_k, s.kwta_temp = flags, kw_k, kw_temp
            s.keep = (fo, tables, rP, bP, ctl_host, ctl_dev, last_dir, last_norm)
            s.ebias_grad = bool(ebias.requires_grad)
            s.zero_in = True
            samples.append(s)
        bt = {"samples": samples, "chunk": chunk, "base": base, "n_max": KC, "n_loops": N, "lockstep": True,
              "H_out": f["H"][bslot]}  # (one gather instead of a row index per loop)
        H_out = LoopBatchFn.apply(sensors, bank.anchor(dev), self, bt)
        return H_out, masks, logits, traces

    # ------------------------------------------------------------------ backward
    @staticmethod
    def _steered_stub(sp) -> "_Sample":
        """What the batched backward and the per-pass reductions read from a loop's host-side record that does not
        depend on the loop's results (device-steered batches, before their record is read back)."""
        s = _Sample()
        s.flags, s.kwta_k, s.kwta_temp = sp["flags"], sp["kwta_k"], sp["kwta_temp"]
        s.keep, s.ebias_grad, s.norm_in_w, s.ebias = sp["keep"], sp["ebias_grad"], sp["norm_in_w"], sp["sb"]["ebias"]
        s.zero_in, s.ts = True, 1
        return s

    def backward(self, s: _Sample, gH_final: torch.Tensor):
        m = self.model
        bank, router = m.bank, m.router
        dev = gH_final.device
        R, d, E = m.R, m.d, router.E
        Ez = max(E, 1)
        L = lib()
        st = stream_handle()
        bank.coord.begin()
        ws = self._workspace(dev)
        WP = ws["_ptr"]
        ns = ops.bwd_nslab(d)
        n_r = s.best + 1  # steps after the winning one feed nothing that reaches the loss
        flags = s.flags
        afd, rt_, use_pred = bool(flags & ops.IC_F_AFD), bool(flags & ops.IC_F_RT), bool(flags & ops.IC_F_PRED)
        prec = router.enable_precision_routed_messages
        chunk, i0 = s.chunk, s.i0
        CP, CS = chunk.ptr, chunk.stride

        # ---- per-loop zero block: everything the kernels only write for active rows
        zf = {"gH": R * d, "g_norm_w_rows": R * d, "g_raw_decay_rows": R * d, "g_norm_in_rows": R * d, "gsens": 2 * d,
              "g_ebias_part": ns * Ez}
        if afd:
            zf.update(g_pn_rows=R * d, g_on_rows=R * d)
        if rt_:
            zf["g_rnoise_rows"] = R * 2
        if prec:
            zf["g_rawP_rows"] = Ez * d
        if router.fact_lib:
            zf["g_fw_rows"] = Ez * d
        Z, zp = {}, {}
        pad = lambda n: (n + 63) // 64 * 64  # every field starts 256-byte aligned (float4 row stores)
        zall = torch.zeros(sum(pad(n_r * sz) for sz in zf.values()), device=dev)  # ONE fill for the whole loop
        zoff = 0
        for k, sz in zf.items():
            Z[k] = zall[zoff:zoff + n_r * sz].view(n_r, sz)
            zp[k] = Z[k].data_ptr()
            zoff += pad(n_r * sz)
        Z["gH"][s.best].copy_(gH_final.reshape(-1))
        has_grad = [False] * n_r
        has_grad[s.best] = True

        # ---- tape slots for the deferred weight gradients
        tape = bank.tape
        t0 = tape.alloc_n(n_r, dev)
        tape.buf["mask"][t0:t0 + n_r, 0].copy_(chunk.fields["mask"][i0:i0 + n_r])
        tape.visited[t0:t0 + n_r].fill_(1)
        TB = {k: (tape.buf[k].data_ptr(), tape.buf[k].stride(0) * 4) for k in ("g", "x", "o")}
        if E > 0:
            rtape = router._get_tape(1, dev)
            rt0 = rtape.alloc_n(n_r, dev)
            rtape.visited[rt0:rt0 + n_r].fill_(1)
            RB = {k: (rtape.buf[k].data_ptr(), rtape.buf[k].stride(0) * rtape.buf[k].element_size())
                  for k in rtape.buf}

        # ---- structs
        tables, rP, bP = s.keep[1], s.keep[2], s.keep[3]
        P = dict(bP)
        P["norm_in_w"] = s.norm_in_w
        b = new("ic_region_bwd_args")
        b.f = ops._region_args(bank.spec, P, 1, 0.0, s.kwta_k, flags, s.kwta_temp)
        b.f.sensor = s.sensor.data_ptr()
        b.gM, b.ghpre, b.slab2, b.slab, b.slab_rt = WP["gM"], WP["ghpre"], WP["slab2"], WP["slab"], WP["slab_rt"]
        rb = new("ic_router_bwd_args")
        rspec = router.spec()
        if E > 0:
            rb.f = ops._router_args(rspec, tables, rP, 1)
            rb.f.ebias = s.ebias.data_ptr()
            rb.f.M = WP["gM"]  # unused by the backward, must be non-NULL
            rb.src_off, rb.src_edges = tables["src_off"].data_ptr(), tables["src_edges"].data_ptr()
            rb.slab, rb.gHd, rb.slab_q = WP["rslab"], WP["gHd"], WP["slab_q"]
            if router.fact_lib:
                rb.fact_gz, rb.fact_gt, rb.fact_slab, rb.fact_gb = WP["fact_gz"], WP["fact_gt"], WP["fact_slab"], WP["fact_gb"]
        b_ref, rb_ref = ctypes.byref(b), ctypes.byref(rb)
        p_bwd, r_bwd = L.ic_region_phase_bwd, L.ic_router_bwd
        saved = ["x", "u1", "inv_s", "r_s", "k_s", "v_s", "y", "hpre", "gate"] + (["ynorm"] if rt_ else []) \
            + (["sterm"] if afd else [])
        rsaved = ["msg", "keyv", "q", "wgt"] if prec else (["fact_t", "fact_z", "fact_dact"] if router.fact_lib else [])
        init = {k: v.data_ptr() for k, v in s.init.items()}
        state_in = [("num", "num_o"), ("den", "den_o"), ("dt", "dt_o")] + ([("var", "var_o")] if afd else []) \
            + ([("pred", "pred_o")] if use_pred else []) \
            + ([("radius", "radius_o"), ("radius_var", "radius_var_o")] if rt_ else [])
        gnames = ["gnum", "gden"] + (["gvar"] if afd else []) + (["gpred"] if use_pred else []) \
            + (["gradius", "gradius_var"] if rt_ else [])
        rows = ["g_norm_w_rows", "g_raw_decay_rows", "g_norm_in_rows"] + (["g_pn_rows", "g_on_rows"] if afd else []) \
            + (["g_rnoise_rows"] if rt_ else [])
        side, have_state_grad = "a", False
        H0p, mask0p = s.H0.data_ptr(), s.mask0.data_ptr()
        nb0 = s.H0.shape[0]

        def in_ptrs(par: int, b_: int):
            if par < 0:
                b_ = min(b_, nb0 - 1)
                return H0p + b_ * R * d * 4, mask0p + b_ * R
            return CP["H"] + (i0 + par) * CS["H"], CP["mask"] + (i0 + par) * CS["mask"]

        # every linearly advancing field of the region backward, for all steps at once
        bimg = StructImage(b)
        sl0 = i0 * 1  # slot of step 0
        for k in ("M", "mask"):
            bimg.linear(("f", k), CP[k] + sl0 * CS[k], CS[k])
        for k, ko in state_in:  # the step's input state = the previous step's output (t == 0: patched below)
            bimg.linear(("f", k), CP[ko] + (sl0 - 1) * CS[ko], CS[ko])
        if rt_:
            bimg.linear(("f", "radius_o"), CP["radius_o"] + sl0 * CS["radius_o"], CS["radius_o"])
        for k in saved:
            bimg.linear(("f", k), CP[k] + sl0 * CS[k], CS[k])
        bimg.linear(("gH",), zp["gH"], R * d * 4)
        bimg.linear(("gsens",), zp["gsens"], 2 * d * 4)
        for k in rows:
            bimg.linear((k,), zp[k], zf[k] * 4)
        for k, fld in (("g", "tape_g"), ("x", "tape_x"), ("o", "tape_o")):
            bimg.linear((fld,), TB[k][0] + t0 * TB[k][1], TB[k][1])
        b_rows = bimg.images(range(n_r))

        def region_bwd(t: int, gM_ptr: int) -> None:
            nonlocal side, have_state_grad
            bimg.load(b_rows[t])
            f = b.f
            f.step_pos = s.step_pos[t]
            if t == 0:
                for k, _ in state_in:
                    setattr(f, k, init[k])
            b.gM = gM_ptr
            other = "b" if side == "a" else "a"
            for g in gnames:
                setattr(b, g + "_o", WP[f"{g}_{side}"] if have_state_grad else None)
                setattr(b, g, WP[f"{g}_{other}"])
            rc = p_bwd(b_ref, st)
            if rc:
                check(rc, "ic_region_phase_bwd")
            side, have_state_grad = other, True

        if E > 0:
            rimg = StructImage(rb)
            for k in rsaved:
                rimg.linear(("f", k), CP[k] + i0 * CS[k], CS[k])
            rimg.linear(("g_ebias_part",), zp["g_ebias_part"], ns * Ez * 4)
            if prec:
                rimg.linear(("g_rawP_rows",), zp["g_rawP_rows"], Ez * d * 4)
            if router.fact_lib:
                rimg.linear(("g_fw_rows",), zp["g_fw_rows"], Ez * d * 4)
            for k, fld in (("gmsg", "tape_gmsg"), ("H", "tape_H"), ("eact", "tape_eact"), ("gkey", "tape_gkey"),
                           ("gq", "tape_gq"), ("qact", "tape_qact"), ("gU", "tape_gU"), ("zU", "tape_zU"), ("aU", "tape_aU"),
                           ("gV", "tape_gV"), ("aV", "tape_aV")):
                if k in RB:
                    rimg.linear((fld,), RB[k][0] + rt0 * RB[k][1], RB[k][1])
            rb.gM = WP["gM"]
            r_rows = rimg.images(range(n_r))

        def router_bwd(ts: List[int]) -> None:
            """Router backward of the consecutive steps ts (1 or 2: the branches of one outer
            iteration go through as two streams of one launch sequence)."""
            n = len(ts)
            tA = ts[0]
            rimg.load(r_rows[tA])
            pars = [s.parent[t] for t in ts]
            rf = rb.f
            rf.S = n
            rf.H, rf.mask_prev = in_ptrs(pars[0], 0)
            rf.H_sstride = rf.mask_sstride = 0
            direct = all(p_ >= 0 and not has_grad[p_] for p_ in pars) and len(set(pars)) == n
            rb.gH_sstride = 0
            if n == 2:
                h1, m1 = in_ptrs(pars[1], 1)
                rf.H_sstride, rf.mask_sstride = (h1 - rf.H) // 4, m1 - rf.mask_prev
                if direct:
                    rb.gH_sstride = (pars[1] - pars[0]) * R * d
            rb.gH = zp["gH"] + pars[0] * R * d * 4 if direct else WP["gH_tmp"]
            rc = r_bwd(rb_ref, st)
            if rc:
                check(rc, "ic_router_bwd")
            for j_, p_ in enumerate(pars):
                if p_ >= 0:
                    if not direct:
                        Z["gH"][p_].add_(ws["gH_tmp"][j_].reshape(-1))
                    has_grad[p_] = True

        # steps 2k, 2k+1 are the two branches of outer iteration k (branching loop, B_br == 2)
        pairs = (not m.cfg.train_deterministic_inner_loop) and m.cfg.B_br == 2 and E > 0
        gM0, gM1 = WP["gM"], WP["gM"] + R * d * 4
        t = n_r - 1
        while t >= 0:
            if pairs and t % 2 == 1 and s.parent[t] != s.parent[t - 1] \
                    and (s.parent[t] < 0) == (s.parent[t - 1] < 0):
                region_bwd(t, gM1)
                region_bwd(t - 1, gM0)
                router_bwd([t - 1, t])
                t -= 2
            else:
                region_bwd(t, gM0)
                if E > 0:
                    router_bwd([t])
                t -= 1

        # ---- the small parameter gradients are reduced ONCE per backward pass (_finish_pass): the
        # per-step rows of this loop are only queued here
        if m._units is not None:
            n_on = chunk.fields["mask"][i0:i0 + n_r].sum().float()
            m._units[5:8] += torch.stack([torch.tensor(float(n_r), device=dev), n_on, n_on])
        self._pass["blocks"].append((Z, n_r, s))
        return Z["gsens"].view(n_r * 2, d).sum(0)

    # ------------------------------------------------------------------ loops of a batch as one node
    def begin_batch(self, n_loops: int, n_max: int, dev) -> None:
        """The next n_loops differentiable loops (reasoning_loop_batch) are recorded in ONE block of
        step slots, preceded by a slot that holds the state the first loop starts from, and are
        differentiated together by end_batch(): the reference detaches the region state between the
        samples of a batch (model.py:482-487), so their backward passes are independent and the
        same step of every loop can run as the streams of one launch."""
        m = self.model
        bank = m.bank
        if bank.S != 1 or bank.num.device != dev:
            bank.flat.reflat()
            bank.reset_state(dev, 1)
        need = n_loops * n_max + 1
        arena = self._get_arena(dev, n_max)
        if arena.chunk_steps < need:
            self._rescue_state()
            self._arena = arena = StepArena(self._shapes(), chunk_steps=need)
        c, pos = arena.cur, arena.pos
        if c < len(arena.chunks) and arena.chunks[c].cap - pos < need:
            arena.cur, arena.pos = c + 1, 0  # the whole batch in one chunk
        chunk, i = arena.reserve(1, dev)
        f = chunk.fields
        pairs = [("num", "num_o"), ("den", "den_o"), ("dt", "dt_o"), ("var", "var_o"), ("surprise", "surprise_o"),
                 ("pred", "pred_o"), ("radius", "radius_o"), ("radius_var", "radius_var_o")]
        dst, src = [], []
        for k, ko in pairs:  # the bank's state moves into the slot before the first loop
            if ko in f:
                dst.append(f[ko][i])
                src.append(getattr(bank, k).detach().reshape(f[ko][i].shape))
                setattr(bank, k, f[ko][i:i + 1])
        with torch.no_grad():
            _copy_many(dst, src)  # (one multi-tensor launch per dtype instead of eight copies)
        self._state_in_arena = True
        self._batch = {"samples": [], "chunk": chunk, "base": i + 1, "n_max": n_max, "n_loops": n_loops}

    def end_batch(self, sensors: torch.Tensor) -> Optional[torch.Tensor]:
        """[n_recorded, R, d] final states of the recorded loops, as one autograd node whose input
        is the batch of sensor vectors (row j of `sensors` fed recorded loop j)."""
        self.resolve_pending()
        bt, self._batch = self._batch, None
        if bt is None or not bt["samples"]:
            return None
        return LoopBatchFn.apply(sensors, self.model.bank.anchor(sensors.device), self, bt)

    def _batch_uniform(self, bt) -> bool:
        """True when the recorded loops sit n_max slots apart in one chunk, started from the shared
        all-zero input state, and share parameters, flags and schedule."""
        if bt.get("steered") is not None:
            return True  # device-steered loops (run_steered) are laid out that way by construction
        ss = bt["samples"]
        s0, K = ss[0], bt["n_max"]
        for j, s in enumerate(ss):
            if s.chunk is not bt["chunk"] or s.i0 != bt["base"] + j * K or s.n > K or not s.zero_in:
                return False  # (s.n < K: halted early; run() moved its final state to the block's last slot)
            if (s.flags, s.kwta_k, s.kwta_temp) != (s0.flags, s0.kwta_k, s0.kwta_temp):
                return False
            if s.ebias.data_ptr() != s0.ebias.data_ptr() or s.norm_in_w is not s0.norm_in_w or s.keep[1] is not s0.keep[1]:
                return False
            m_ = min(len(s.step_pos), len(s0.step_pos))
            if s.step_pos[:m_] != s0.step_pos[:m_]:
                return False
        return True

    def _workspace_batch(self, device, V: int):
        m = self.model
        R, d, E = m.R, m.d, max(m.router.E, 1)
        key = (device, V, R, d, E, m.router.fact_lib)
        if self._wsb is not None and self._wsb["_key"] == key:
            return self._wsb
        ns = ops.bwd_nslab(d)
        e = lambda *s: torch.empty(*s, device=device)
        ws = {"gM": e(V, R, d), "ghpre": e(V, R, d), "slab2": e(ns, V, R, d), "slab": e(ns, V, R, d),
              "slab_rt": e(ns, V, R), "gHd": e(V, R, d), "rslab": e(ns, V, E, d), "slab_q": e(ns, V, R, d)}
        if m.router.fact_lib:
            ws.update(fact_gz=e(V, R, d), fact_gt=e(V, R, d), fact_slab=e(ns, V, R, d), fact_gb=e(ns, V, R))
        for side in ("a", "b"):
            for k in ("gnum", "gden", "gvar", "gpred"):
                ws[f"{k}_{side}"] = e(V, R, d)
            for k in ("gradius", "gradius_var"):
                ws[f"{k}_{side}"] = e(V, R)
        ws["_ptr"] = {k: t.data_ptr() for k, t in ws.items()}
        ws["_key"] = key
        self._wsb = ws
        return ws

    def backward_batch(self, bt, gH_all: torch.Tensor) -> torch.Tensor:
        """Backward of every recorded loop at once: step t of all S loops is one launch sequence
        with S streams, so a pass costs n_max launch sequences instead of S * n_max and every weight
        row is read once per step for all loops.  Two slot layouts (arena, tapes, gradient buffers):
        loops recorded one after the other sit ss = n_max slots apart with consecutive steps in
        consecutive slots (ts = 1); loops run in lockstep (run_lockstep) sit in consecutive slots
        (ss = 1) with consecutive steps S slots apart (ts = S).  Returns d/d sensors [S, d]."""
        m = self.model
        bank, router = m.bank, m.router
        dev = gH_all.device
        R, d, E = m.R, m.d, router.E
        Ez = max(E, 1)
        sp = bt.get("steered")  # device-steered loops whose record (winners, parents) is still on the device:
        if sp is not None:      # everything below that depends on it stays a device tensor -- no host wait
            ss, S, K = None, bt["n_loops"], bt["n_max"]
            s0 = self._steered_stub(sp)
        else:
            ss = bt["samples"]
            S, K = len(ss), bt["n_max"]
            s0 = ss[0]
        V = S * K
        lock = bool(bt.get("lockstep"))
        sS, tS = (1, S) if lock else (K, 1)  # slot of (loop s, step t) = base + s * sS + t * tS
        # [S,K,...] <-> rows in slot order
        to_v = (lambda x: x.transpose(0, 1).reshape((V,) + tuple(x.shape[2:]))) if lock else \
            (lambda x: x.reshape((V,) + tuple(x.shape[2:])))
        from_v = (lambda y: y.view((K, S) + tuple(y.shape[1:])).transpose(0, 1)) if lock else \
            (lambda y: y.view((S, K) + tuple(y.shape[1:])))
        L, st = lib(), stream_handle()
        bank.coord.begin()
        ws = self._workspace_batch(dev, V)
        WP = ws["_ptr"]
        ns = ops.bwd_nslab(d)
        flags = s0.flags
        afd, rt_, use_pred = bool(flags & ops.IC_F_AFD), bool(flags & ops.IC_F_RT), bool(flags & ops.IC_F_PRED)
        prec = router.enable_precision_routed_messages
        chunk, base = bt["chunk"], bt["base"]
        CP, CS = chunk.ptr, chunk.stride

        # ---- zero block: everything the kernels only write for active rows, [V, size] per field
        zf = {"gH": R * d, "g_norm_w_rows": R * d, "g_raw_decay_rows": R * d, "g_norm_in_rows": R * d, "gsens": 2 * d,
              "g_ebias_part": ns * Ez}
        if afd:
            zf.update(g_pn_rows=R * d, g_on_rows=R * d)
        if rt_:
            zf["g_rnoise_rows"] = R * 2
        if prec:
            zf["g_rawP_rows"] = Ez * d
        if router.fact_lib:
            zf["g_fw_rows"] = Ez * d
        pad = lambda n: (n + 63) // 64 * 64
        zall = torch.zeros(sum(pad(V * sz) for sz in zf.values()), device=dev)
        Z, zp, zoff = {}, {}, 0
        for k, sz in zf.items():
            Z[k] = zall[zoff:zoff + V * sz].view(V, sz)
            zp[k] = Z[k].data_ptr()
            zoff += pad(V * sz)
        lanes = torch.arange(S, device=dev)
        if sp is not None:  # slot of every loop's winner (ic_loop_steer's record) -> step index inside its block
            best = sp["sb"]["rec"][:, 0].long() - (base + lanes * K)
        else:
            best = torch.tensor([s.best for s in ss], device=dev)
        from_v(Z["gH"])[lanes, best] = gH_all.reshape(S, R * d)

        # ---- tapes: one slot per (loop, step), laid out like the arena block
        tape = bank.tape
        t0 = tape.alloc_n(V, dev)
        steps = torch.arange(K, device=dev)
        live = steps[None, :] <= best[:, None]                      # [S,K]: steps that reach the loss
        mk = from_v(chunk.fields["mask"][base:base + V]) * live[:, :, None].to(_U8)           # [S,K,R]
        tape.buf["mask"][t0:t0 + V, 0].copy_(to_v(mk))               # the backward's own (masked) copy
        tape.visited[t0:t0 + V].fill_(1)
        mask_ptr, mask_stride = tape.buf["mask"].data_ptr() + t0 * tape.buf["mask"].stride(0), tape.buf["mask"].stride(0)
        TB = {k: (tape.buf[k].data_ptr(), tape.buf[k].stride(0) * 4) for k in ("g", "x", "o")}
        if E > 0:
            rtape = router._get_tape(1, dev)
            rt0 = rtape.alloc_n(V, dev)
            RB = {k: (rtape.buf[k].data_ptr(), rtape.buf[k].stride(0) * rtape.buf[k].element_size())
                  for k in rtape.buf}
            # where each stream's step reads its input states: the parent's slot (relative to base)
            if sp is not None:
                # the control block of every iteration holds the slots its branches started from (the zero-state slot
                # sits right before the block: relative slot -1 = "no parent", as the host-side record says it)
                Cb, Kit = sp["C"], sp["K"]
                i_in = ENUMS["IC_CTL_IN_SLOT"]
                par_abs = sp["sb"]["ctl"][:S * Kit, i_in:i_in + Cb].long().reshape(S, K)
                slot = torch.where(live, par_abs - base, torch.full_like(par_abs, -1))
                r_live = [t >= Cb for t in range(K)]                                          # (iteration 0 starts from the zero state)
            else:
                pars = [list(s.parent) + [-1] * (K - s.n) for s in ss]                       # halted loops: padded
                par = torch.tensor(pars, dtype=torch.int64, device=dev)                       # [S,K]
                slot = torch.where((par >= 0) & live, lanes[:, None] * sS + par * tS, torch.full_like(par, -1))
                r_live = [any(p >= 0 for p in col) for col in zip(*pars)]                     # host: per step
            in_slot = slot.t().contiguous().to(torch.int32)                                   # [K,S]
            rtape.visited[rt0:rt0 + V].copy_(to_v((slot >= 0).to(_U8)))
        if sp is not None and t0 == 0 and (E == 0 or (rt0 == 0 and not router.fact_lib)):
            # Which matrices this pass will touch follows from the forward's masks: the flags the deferred weight-gradient
            # launches raise at the END of the pass (region r active in a step that reaches the loss; edge e with an
            # active sender in the parent of such a step) go to the host NOW, behind the loops and ahead of the backward
            # kernels, so that the end of the pass need not wait for the device before .grad views are handed out.
            early = {bank._launch_wgrad: mk.amax((0, 1))}
            if E > 0:
                allm = chunk.fields["mask"][base:base + V]                                        # [V,R] by slot
                pa = (allm[slot.clamp_min(0).reshape(-1)] * (slot >= 0).reshape(-1, 1).to(_U8)).amax(0)   # [R]
                te = pa[router.src_idx].int()                                                     # [E]
                wl = s0.keep[1]["widx_long"]
                tw = torch.zeros(router.nW, dtype=torch.int32, device=dev).index_add_(0, wl, te).gt(0).to(_U8)
                if prec:
                    tq = torch.zeros(R, dtype=torch.int32, device=dev).index_add_(0, router.dst_idx, te).gt(0).to(_U8)
                    tw = torch.cat([tw, tw, tq])
                early[router._launch_wgrad] = tw
            rtp = router._tape if E > 0 else None
            m._coord.offer_early(early, lambda: tape.pos == V and (rtp is None or (router._tape is rtp and rtp.pos == V)))

        # ---- structs
        tables, rP, bP = s0.keep[1], s0.keep[2], s0.keep[3]
        P = dict(bP)
        P["norm_in_w"] = s0.norm_in_w
        sens = sp["sb"]["sens"] if sp is not None else torch.stack([s.sensor.reshape(-1) for s in ss]).contiguous()  # [S,d]
        b = new("ic_region_bwd_args")
        b.f = ops._region_args(bank.spec, P, S, 0.0, s0.kwta_k, flags, s0.kwta_temp)
        b.f.sensor = sens.data_ptr()
        b.sk = sS
        b.gM, b.ghpre, b.slab2, b.slab, b.slab_rt = WP["gM"], WP["ghpre"], WP["slab2"], WP["slab"], WP["slab_rt"]
        saved = ["x", "u1", "inv_s", "r_s", "k_s", "v_s", "y", "hpre", "gate"] + (["ynorm"] if rt_ else []) \
            + (["sterm"] if afd else [])
        state_in = [("num", "num_o"), ("den", "den_o"), ("dt", "dt_o")] + ([("var", "var_o")] if afd else []) \
            + ([("pred", "pred_o")] if use_pred else []) \
            + ([("radius", "radius_o"), ("radius_var", "radius_var_o")] if rt_ else [])
        gnames = ["gnum", "gden"] + (["gvar"] if afd else []) + (["gpred"] if use_pred else []) \
            + (["gradius", "gradius_var"] if rt_ else [])
        rows = ["g_norm_w_rows", "g_raw_decay_rows", "g_norm_in_rows"] + (["g_pn_rows", "g_on_rows"] if afd else []) \
            + (["g_rnoise_rows"] if rt_ else [])
        bimg = StructImage(b)
        bimg.linear(("f", "M"), CP["M"] + base * CS["M"], tS * CS["M"])
        bimg.linear(("f", "mask"), mask_ptr, tS * mask_stride)
        for k, ko in state_in:  # input state of step t = output of step t - 1 (t == 0: the slot(s) before the block)
            bimg.linear(("f", k), CP[ko] + (base - tS) * CS[ko], tS * CS[ko])
        if rt_:
            bimg.linear(("f", "radius_o"), CP["radius_o"] + base * CS["radius_o"], tS * CS["radius_o"])
        for k in saved:
            bimg.linear(("f", k), CP[k] + base * CS[k], tS * CS[k])
        bimg.linear(("gH",), zp["gH"], tS * R * d * 4)
        bimg.linear(("gsens",), zp["gsens"], tS * 2 * d * 4)
        for k in rows:
            bimg.linear((k,), zp[k], tS * zf[k] * 4)
        for k, fld in (("g", "tape_g"), ("x", "tape_x"), ("o", "tape_o")):
            bimg.linear((fld,), TB[k][0] + t0 * TB[k][1], tS * TB[k][1])
        b_rows = bimg.images(range(K))
        b_ref = ctypes.byref(b)
        p_bwd, r_bwd = L.ic_region_phase_bwd, L.ic_router_bwd

        if E > 0:
            rb = new("ic_router_bwd_args")
            rspec = router.spec()
            rb.f = ops._router_args(rspec, tables, rP, S)
            rb.f.ebias = s0.ebias.data_ptr()
            rb.f.M = WP["gM"]  # unused by the backward, must be non-NULL
            rb.f.H, rb.f.mask_prev = CP["H"] + base * CS["H"], CP["mask"] + base * CS["mask"]
            rb.src_off, rb.src_edges = tables["src_off"].data_ptr(), tables["src_edges"].data_ptr()
            rb.slab, rb.gHd, rb.slab_q = WP["rslab"], WP["gHd"], WP["slab_q"]
            if router.fact_lib:
                rb.fact_gz, rb.fact_gt, rb.fact_slab, rb.fact_gb = WP["fact_gz"], WP["fact_gt"], WP["fact_slab"], WP["fact_gb"]
            rb.gM, rb.gH, rb.sk = WP["gM"], zp["gH"], sS
            rimg = StructImage(rb)
            for k in (["msg", "keyv", "q", "wgt"] if prec else (["fact_t", "fact_z", "fact_dact"] if router.fact_lib else [])):
                rimg.linear(("f", k), CP[k] + base * CS[k], tS * CS[k])
            # the kernel writes [ns, S*sk, Ez] partials per launch: interleaved over the steps when the loops sit
            # K slots apart ([ns, V, Ez] in all), one [ns, S, Ez] block per step when they run in lockstep
            rimg.linear(("g_ebias_part",), zp["g_ebias_part"], (ns * S * Ez if lock else Ez) * 4)
            if prec:
                rimg.linear(("g_rawP_rows",), zp["g_rawP_rows"], tS * Ez * d * 4)
            if router.fact_lib:
                rimg.linear(("g_fw_rows",), zp["g_fw_rows"], tS * Ez * d * 4)
            for k, fld in (("gmsg", "tape_gmsg"), ("H", "tape_H"), ("eact", "tape_eact"), ("gkey", "tape_gkey"),
                           ("gq", "tape_gq"), ("qact", "tape_qact"), ("gU", "tape_gU"), ("zU", "tape_zU"), ("aU", "tape_aU"),
                           ("gV", "tape_gV"), ("aV", "tape_aV")):
                if k in RB:
                    rimg.linear((fld,), RB[k][0] + rt0 * RB[k][1], tS * RB[k][1])
            rimg.linear(("in_slot",), in_slot.data_ptr(), S * 4)
            r_rows = rimg.images(range(K))
            rb_ref = ctypes.byref(rb)

        side, have = "a", False
        if sp is not None:  # every step slot is replayed: the ones after a loop's winner are masked out (all inactive)
            step_pos, t_last = [float(t // sp["C"]) / sp["denom"] for t in range(K)], K - 1
        else:
            step_pos, t_last = max((s.step_pos for s in ss), key=len), max(s.best for s in ss)
        for t in range(t_last, -1, -1):  # steps after every loop's winner reach no loss
            bimg.load(b_rows[t])
            b.f.step_pos = step_pos[t]
            other = "b" if side == "a" else "a"
            for g in gnames:
                setattr(b, g + "_o", WP[f"{g}_{side}"] if have else None)
                setattr(b, g, WP[f"{g}_{other}"])
            rc = p_bwd(b_ref, st)
            if rc:
                check(rc, "ic_region_phase_bwd")
            side, have = other, True
            if E > 0 and r_live[t]:
                rimg.load(r_rows[t])
                rc = r_bwd(rb_ref, st)
                if rc:
                    check(rc, "ic_router_bwd")

        if m._units is not None:
            m._units[5:8] += torch.stack([torch.tensor(float(K), device=dev), mk.sum().float(), mk.amax(0).sum().float()])
            if E > 0:  # per router-backward launch (step t, all loops): active (stream, edge) pairs, distinct edges / destinations
                allm = chunk.fields["mask"][base:base + V]                                        # [V,R] by slot
                pm = allm[slot.clamp_min(0).reshape(-1)].view(S, K, R) * (slot >= 0)[:, :, None].to(_U8)
                ea = pm[:, :, router.src_idx]                                                     # [S,K,E]
                eany = ea.amax(0).float()                                                         # [K,E]
                dany = torch.zeros(K, R, device=dev).index_add_(1, router.dst_idx, eany).gt(0)
                m._units[8:12] += torch.stack([torch.tensor(float(sum(r_live)), device=dev), ea.sum().float(), eany.sum(), dany.sum().float()])
        self._pass["blocks"].append((Z, V, s0))
        self._pass.setdefault("keep", []).append((sens, best, in_slot if E > 0 else None, mk))
        return from_v(Z["gsens"]).reshape(S, K * 2, d).sum(1)

    # ------------------------------------------------------------------ per-pass reductions
    def _begin_pass(self) -> None:
        self._pass = {"blocks": []}

    def _finish_pass(self) -> None:
        """End of a backward pass: sum the queued per-step gradient rows of every loop and add
        them to the parameters' gradients (region norms / decay / noise levels, raw_P, norm_in,
        and the Fourier edge-bias coefficients through the closed form of gate.py:284-290)."""
        ps, self._pass = self._pass, None
        if not ps or not ps["blocks"]:
            return
        m = self.model
        bank, router = m.bank, m.router
        R, d, E = m.R, m.d, router.E
        Ez = max(E, 1)
        blocks = ps["blocks"]
        s0 = blocks[0][2]
        flags = s0.flags
        afd, rt_ = bool(flags & ops.IC_F_AFD), bool(flags & ops.IC_F_RT)
        prec = router.enable_precision_routed_messages
        tot = lambda k: torch.cat([Z[k] for Z, _, _ in blocks], 0)  # [sum n_r, size]
        sink = bank._grad_sink(0)
        sink("norm_w", tot("g_norm_w_rows").sum(0).view(R, d))
        sink("raw_decay", tot("g_raw_decay_rows").sum(0).view(R, d))
        if afd:
            if flags & ops.IC_F_NOISE_VEC:
                sink("pn_param", tot("g_pn_rows").sum(0).view(R, d))
                sink("on_param", tot("g_on_rows").sum(0).view(R, d))
            else:
                sink("pn_param", tot("g_pn_rows").view(-1, R, d).sum((0, 2)))
                sink("on_param", tot("g_on_rows").view(-1, R, d).sum((0, 2)))
        if rt_:
            rn = tot("g_rnoise_rows").sum(0).view(R, 2)
            sink("rpn_param", rn[:, 0].contiguous())
            sink("ron_param", rn[:, 1].contiguous())

        def accumulate(p, g):
            if p.requires_grad:
                p.grad = g if p.grad is None else p.grad + g

        accumulate(s0.norm_in_w, tot("g_norm_in_rows").view(-1, d).sum(0))
        if E > 0:
            if prec:
                router._grad_sink("raw_P", tot("g_rawP_rows").sum(0).view(Ez, d)[:E], s0.keep[1]["widx_long"])
            if router.fact_lib:
                router._grad_sink("w_e", tot("g_fw_rows").sum(0).view(Ez, d)[:E], s0.keep[1]["widx_long"])
            if s0.ebias_grad:  # ebias = 1 + fb_alpha * fb_scale * sum_m(cos S * beta_cos + sin S * beta_sin)
                g_eb = tot("g_ebias_part").view(-1, Ez).sum(0)[:E] * (router.fb_alpha * router.fb_scale)
                _, cosS, sinS = router._geom
                accumulate(router.beta_cos, g_eb @ cosS)
                accumulate(router.beta_sin, g_eb @ sinS)


class SteeredTraces:
    """Traces of one device-steered loop (the winner's states of every executed outer iteration, reference
    model.py:402): a list that is filled on first use, when the loops' record is read from the device."""

    def __init__(self, engine, bt, j):
        self._src = (engine, bt, j)
        self._items = None

    def _get(self):
        if self._items is None:
            eng, bt, j = self._src
            eng.resolve_steered(bt)
            s = bt["samples"][j]
            C = 2
            wins = [s.parent[k * C] for k in range(1, s.n // C)] + [s.best]
            Hf = s.chunk.fields["H"]
            self._items = [Hf[s.i0 + w] for w in wins]
        return self._items

    def __len__(self):
        return len(self._get())

    def __getitem__(self, i):
        return self._get()[i]

    def __iter__(self):
        return iter(self._get())


class LoopFn(torch.autograd.Function):
    """One autograd node per reasoning loop: forward hands out the loop's final region states
    (already computed by LoopEngine.run), backward replays the recorded steps in reverse."""

    @staticmethod
    def forward(ctx, sensor, anchor, engine, sample):
        ctx.engine, ctx.sample = engine, sample
        ctx.sshape = sensor.shape
        return sample.chunk.fields["H"][sample.i0 + sample.best].clone()

    @staticmethod
    def backward(ctx, gH):
        g_sensor = ctx.engine.backward(ctx.sample, gH.contiguous())
        return g_sensor.view(ctx.sshape), None, None, None


class LoopBatchFn(torch.autograd.Function):
    """One autograd node for all the loops of a reasoning_loop_batch call (LoopEngine.begin_batch /
    end_batch): forward stacks the loops' final region states, backward replays every loop's steps
    in reverse -- all loops at once when they are laid out uniformly, else one after another."""

    @staticmethod
    def forward(ctx, sensors, anchor, engine, bt):
        ctx.engine, ctx.bt = engine, bt
        ctx.sshape = sensors.shape
        if bt.get("H_out") is not None:
            return bt.pop("H_out")
        return torch.stack([s.chunk.fields["H"][s.i0 + s.best * s.ts] for s in bt["samples"]])

    @staticmethod
    def backward(ctx, gH):
        eng, bt = ctx.engine, ctx.bt
        gH = gH.contiguous()
        if eng.model._units is not None:
            eng.resolve_steered(bt)  # (bench.py's unit counters want the loops' record on the host)
        if bt.get("lockstep") or eng._batch_uniform(bt):
            eng.counters["batch_bwd"] += 1
            g = eng.backward_batch(bt, gH)
        else:
            eng.resolve_steered(bt)
            eng.counters["loop_bwd"] += len(bt["samples"])
            g = torch.stack([eng.backward(s, gH[j]) for j, s in enumerate(bt["samples"])])
        return g.view(ctx.sshape), None, None, None
