"""torch.autograd.Function wrappers over the C ABI (include/ironcortex_hip.h).

Each Function's forward/backward is one C call (a handful of kernel launches); weight
gradients are deferred to one ic_*_wgrad launch per backward pass through a tape
(see GradCoordinator).  No op here has a CPU path.
"""
from __future__ import annotations

import ctypes
import os
from typing import Callable, Dict, List, Optional

import torch

from . import ENUMS, check, lib, new, ptr, stream_handle

IC_F_AFD = ENUMS["IC_F_AFD"]
IC_F_RT = ENUMS["IC_F_RT"]
IC_F_PRED = ENUMS["IC_F_PRED"]
IC_F_NOKWTA = ENUMS["IC_F_NOKWTA"]
IC_F_KWTA_SOFT = ENUMS["IC_F_KWTA_SOFT"]
IC_F_NOISE_VEC = ENUMS["IC_F_NOISE_VEC"]
IC_F_NORM_IN = ENUMS["IC_F_NORM_IN"]
IC_RF_PRECISION = ENUMS["IC_RF_PRECISION"]


def _vp(t: Optional[torch.Tensor]) -> ctypes.c_void_p:
    """Device address of a tensor for a scalar-argument entry point.  Raises for a CPU tensor: a host address
    handed to a kernel is a GPU memory fault, not an error code."""
    return ctypes.c_void_p(ptr(t))


def _call(name: str, args) -> None:
    check(getattr(lib(), name)(ctypes.byref(args), stream_handle()), name)


def bwd_nslab(d: int) -> int:
    return int(lib().ic_bwd_nslab(ctypes.c_int32(d)))


# ============================================================================ coordination
class GradCoordinator:
    """Runs the deferred weight-gradient launches once per backward pass.

    Function.backward calls begin(); the first call of a pass zeroes the gradient slices that
    start from None and queues an end-of-pass callback on the autograd engine.  At the end of the
    pass every participant first LAUNCHES its deferred work and hands back its device-side
    "touched" flags (which matrices received a gradient); all flags are then read in ONE host
    sync and each participant publishes .grad views for its touched parameters (untouched ones
    keep None: what AdamW must see to skip them as the reference does).

    Data parallel (dp set by train_step for the duration of a step): the flags -- together with
    the has-a-gradient flags of the parameters outside the arenas (extra_params) -- are OR-ed
    across ranks in the same exchange, so that None-ness is identical on every rank without a
    host read of its own; a parameter another rank has a gradient for gets a zero gradient here."""

    check_early = False  # (tests set it on the class: the device flags are read as well and compared with the early ones)

    def __init__(self):
        self._pending = False
        self._begin: List[Callable[[], None]] = []
        self._finish: List[Callable[[], None]] = []
        self._flagged: List[tuple] = []  # (launch() -> Optional[uint8 tensor], publish(list of bool))
        self.dp = None
        self.extra_params: Optional[Callable[[], list]] = None
        self.after_launch: Optional[Callable[[], None]] = None  # (data parallel: start the arenas' all-reduces)
        # defer (train_step sets it around its backward): the end of the pass launches the deferred work but leaves the
        # one host read of the flags to an explicit publish(), so that the caller can enqueue more work in between
        self.defer = False
        self._held = None
        # early flags (offer_early): a caller that knows which matrices a pass will touch before the pass has run hands
        # the flags over at that point; they travel to pinned host memory behind it, and the end of the pass finds them
        # there instead of waiting for the device
        self._early = None
        self._pinned: Optional[torch.Tensor] = None
        self.early_used = 0

    def register(self, begin: Callable[[], None], finish: Callable[[], None]) -> None:
        self._begin.append(begin)
        self._finish.append(finish)

    def register_flagged(self, begin: Callable[[], None], launch: Callable[[], Optional[torch.Tensor]],
                         publish: Callable[[list], None]) -> None:
        self._begin.append(begin)
        self._flagged.append((launch, publish))

    def begin(self) -> None:
        if self._pending:
            return
        self._pending = True
        for f in self._begin:
            f()
        torch.autograd.Variable._execution_engine.queue_callback(self._end)

    def offer_early(self, flags: Dict[Callable, Optional[torch.Tensor]], valid: Callable[[], bool]) -> None:
        """flags: launch function of a flagged participant -> the uint8 flags its launch will return at the end of this
        pass (None where it will return None); valid() is asked at the end of the pass, before the launches, whether
        that still holds (nothing else joined the pass).  Ignored unless every flagged participant is covered.  Under
        data parallelism the flags are OR-ed across ranks right here (one small all-reduce behind the loops, ahead of
        the backward kernels), so that what reaches pinned memory is already rank-consistent."""
        if any(launch not in flags for launch, _ in self._flagged):
            return
        parts = [flags[launch] for launch, _ in self._flagged]
        live = [t.reshape(-1) for t in parts if t is not None]
        if not live:
            return
        allf = torch.cat(live)
        if self.dp is not None:
            allf = self.dp.any_flags(allf).to(torch.uint8)
        n = allf.numel()
        if self._pinned is None or self._pinned.numel() < n:
            self._pinned = torch.empty(max(n, 1024), dtype=torch.uint8).pin_memory()
        self._pinned[:n].copy_(allf, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self._early = ([None if t is None else t.numel() for t in parts], n, ev, valid)

    def _end(self) -> None:
        try:
            early, self._early = self._early, None
            if early is not None and not early[3]():
                early = None
            flags = [launch() for launch, _ in self._flagged]
            for f in self._finish:
                f()
            if early is not None and early[0] != [None if t is None else t.numel() for t in flags]:
                early = None
            if self.dp is not None:
                self._exchange_host_flags(early is None)  # (before the arenas' all-reduces join the backend's queue)
            if self.after_launch is not None:
                self.after_launch()
            if self.defer:
                self._held = (flags, early)
            else:
                self._publish(flags, early)
        finally:
            self._pending = False

    def _exchange_host_flags(self, no_early: bool) -> None:
        """Data parallel, end of the pass: what only the HOST knows -- which of the parameters outside the arenas hold a
        gradient on this rank, and whether this rank can use its early flags -- OR-ed across ranks on a side stream
        that waits for nothing of the backward pass (the host is milliseconds ahead of the device here), into pinned
        memory; _publish reads it without waiting for the pass's kernels."""
        self._dpx = None
        extra = self.extra_params() if self.extra_params is not None else []
        if not extra or extra[0].device.type != "cuda":
            return  # (nothing the host alone knows, or a CPU group: _publish takes the device path)
        device = extra[0].device
        host = [0 if p.grad is None else 1 for p in extra] + [1 if no_early else 0]
        if getattr(self, "_side", None) is None or self._side.device != device:
            self._side = torch.cuda.Stream(device)
        n = len(host)
        if getattr(self, "_pinned_x", None) is None or self._pinned_x.numel() < n:
            self._pinned_x = torch.empty(max(n, 256), dtype=torch.int32).pin_memory()
            self._pinned_x_in = torch.empty(max(n, 256), dtype=torch.int32).pin_memory()
        self._pinned_x_in[:n].copy_(torch.tensor(host, dtype=torch.int32))
        with torch.cuda.stream(self._side):
            t = self._pinned_x_in[:n].to(device, non_blocking=True)
            t = self.dp.any_flags(t)
            self._pinned_x[:n].copy_(t, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(self._side)
        self._dpx = (extra, n, ev, t)

    def publish(self) -> None:
        """The host read a deferred pass left open (no-op otherwise)."""
        held, self._held = self._held, None
        if held is not None:
            self._publish(*held)

    def _publish(self, flags, early=None) -> None:
        dpx, self._dpx = getattr(self, "_dpx", None), None
        extra_host = None
        if self.dp is not None:
            if dpx is None:
                early = None
            else:
                dpx[2].synchronize()  # (the side stream's exchange: waits for the other ranks' hosts, not for kernels)
                xh = self._pinned_x[:dpx[1]].tolist()
                if xh[-1]:  # some rank has no early flags: every rank takes the device path below
                    early = None
                else:
                    extra_host = (dpx[0], xh[:-1])
        if early is not None:
            early[2].synchronize()  # (long past: the copy was enqueued before the backward kernels)
            host = [bool(v) for v in self._pinned[:early[1]].tolist()]
            if self.check_early:
                dev_f = torch.cat([t.reshape(-1) for t in flags if t is not None])
                if self.dp is not None:
                    dev_f = self.dp.any_flags(dev_f)
                dev_host = dev_f.bool().tolist()
                if dev_host != host:
                    raise RuntimeError(f"GradCoordinator: early flags differ from the launches' flags: {host} vs {dev_host}")
            self.early_used += 1
            off = 0
            for (_, publish), t in zip(self._flagged, flags):
                if t is not None:
                    publish(host[off:off + t.numel()])
                    off += t.numel()
            if extra_host is not None:
                for p, h in zip(*extra_host):
                    if h and p.grad is None:
                        p.grad = torch.zeros_like(p)
            return
        extra = self.extra_params() if (self.dp is not None and self.extra_params is not None) else []
        parts = [t.reshape(-1) for t in flags if t is not None]
        if extra:
            parts.append(torch.tensor([0 if p.grad is None else 1 for p in extra], dtype=torch.uint8,
                                      device=extra[0].device))
        if not parts:
            return
        allf = torch.cat(parts)
        if self.dp is not None:
            allf = self.dp.any_flags(allf)
        host = allf.bool().tolist()  # the one host sync of the backward pass
        off = 0
        for (_, publish), t in zip(self._flagged, flags):
            if t is not None:
                publish(host[off:off + t.numel()])
                off += t.numel()
        for p, h in zip(extra, host[off:]):
            if h and p.grad is None:
                p.grad = torch.zeros_like(p)

    def abort(self) -> bool:
        """A backward pass that raised never reaches its end-of-pass callback, which would leave
        _pending set for good (every later pass would skip begin_pass() and never publish its
        gradients).  Called at a known boundary -- the start of a train_step / LoopEngine.reset():
        no pass can be in flight there, so a set flag is a stale one.  Returns True if it was."""
        stale, self._pending = self._pending, False
        self._held = self._early = None
        self._dpx = None
        return stale


class Tape:
    """Per-step gradient vectors kept until the end of the backward pass.

    fields: name -> per-slot shape; storage grows geometrically.  A slot is allocated in the
    forward of every differentiable call and marked visited by its backward."""

    def __init__(self, fields: Dict[str, tuple], dtypes: Optional[Dict[str, torch.dtype]] = None):
        self.fields = fields
        self.dtypes = dtypes or {}
        self.cap = 0
        self.pos = 0
        self.buf: Dict[str, torch.Tensor] = {}
        self.visited: torch.Tensor = None  # type: ignore

    def _grow(self, device, need: int) -> None:
        cap = max(64, self.cap)
        while cap < need:
            cap *= 2
        for name, shape in self.fields.items():
            dt = self.dtypes.get(name, torch.float32)
            nb = torch.zeros((cap,) + tuple(shape), dtype=dt, device=device)
            if self.cap and name in self.buf and self.buf[name].device == nb.device:
                nb[: self.cap].copy_(self.buf[name])
            self.buf[name] = nb
        nv = torch.zeros(cap, dtype=torch.uint8, device=device)
        if self.cap and self.visited is not None and self.visited.device == nv.device:
            nv[: self.cap].copy_(self.visited)
        self.visited = nv
        self.cap = cap

    def alloc(self, device) -> int:
        if self.pos + 1 > self.cap or (self.visited is not None and self.visited.device != device):
            if self.visited is not None and self.visited.device != device:
                self.cap, self.pos = 0, 0
            self._grow(device, self.pos + 1)
        t = self.pos
        self.pos += 1
        return t

    def alloc_n(self, n: int, device) -> int:
        """n consecutive slots (the LoopEngine books a whole loop at once)."""
        if self.visited is not None and self.visited.device != device:
            self.cap, self.pos = 0, 0
        if self.pos + n > self.cap:
            self._grow(device, self.pos + n)
        t = self.pos
        self.pos += n
        return t

    def slot(self, name: str, t: int) -> torch.Tensor:
        return self.buf[name][t]

    def reset(self) -> None:
        if self.pos:
            self.visited[: self.pos].zero_()
        self.pos = 0


# IC_SHARED_SORT=0: the shared-matrix (by_direction) forward GEMM walks its columns in (stream, region) order instead of
# sorted by activity pattern (A/B timing; same results bit for bit)
SHARED_SORT = os.environ.get("IC_SHARED_SORT", "1") != "0"
SHARED_SORT_MIN = int(os.environ.get("IC_SHARED_SORT_MIN", "8192"))  # columns (streams x regions) from which the sort is used
# IC_SHARED_GEMM_ALWAYS=1: the by_direction forward as a GEMM at any size (IC_RF_SHARED_GEMM); default: the library takes
# the per-edge kernels below 8192 columns (tests set this to cover the GEMM kernels at oracle-sized inputs)
SHARED_GEMM_ALWAYS = os.environ.get("IC_SHARED_GEMM_ALWAYS", "0") == "1"
_SHARED_WS: Dict[tuple, torch.Tensor] = {}

# IC_WGRAD_GEMM=0: the deferred per-edge / per-region weight gradients always through the row-slice kernels (A/B timing)
WGRAD_GEMM = os.environ.get("IC_WGRAD_GEMM", "1") != "0"
WGRAD_GEMM_MIN_TS = 64  # (stream, slot) pairs on the tapes from which the pair-list GEMM is used
WGRAD_GEMM_CALLS = [0]  # (tests: how many weight-gradient calls took the GEMM)

# rows of W per workgroup in the forward GEMVs (0 = library default); a tuning knob for the sweeps
ROWS_FWD = int(os.environ.get("IC_ROWS_FWD", "0"))


def _z(like: torch.Tensor, *shape, dtype=None) -> torch.Tensor:
    return torch.zeros(shape, dtype=dtype or torch.float32, device=like.device)


def _e(like: torch.Tensor, *shape, dtype=None) -> torch.Tensor:
    return torch.empty(shape, dtype=dtype or torch.float32, device=like.device)


# ============================================================================ region phase
class RegionPhaseSpec:
    """Static description of a bank of region cells + per-call scalars."""

    def __init__(self, R: int, d: int, m_time: int, flags: int, sensor_idx: int = -1, motor_idx: int = -1):
        self.R, self.d, self.m_time, self.flags = R, d, m_time, flags
        self.sensor_idx, self.motor_idx = sensor_idx, motor_idx
        self.pred_alpha = 0.95


def _region_args(spec: RegionPhaseSpec, P: Dict[str, torch.Tensor], S: int, step_pos: float, kwta_k: int,
                 flags: int, kwta_temp: float):
    a = new("ic_region_args", S=S, R=spec.R, d=spec.d, m_time=spec.m_time, kwta_k=kwta_k, flags=flags,
            sensor_idx=spec.sensor_idx, motor_idx=spec.motor_idx, rows_fwd=ROWS_FWD, step_pos=float(step_pos),
            kwta_temp=float(kwta_temp), pred_alpha=float(spec.pred_alpha))
    for k in ("W", "norm_w", "raw_decay", "norm_in_w", "W_time", "pn_param", "on_param", "rpn_param", "ron_param"):
        t = P.get(k)
        if t is not None and t.numel() > 0:
            setattr(a, k, ptr(t))
    return a


class RegionPhaseFn(torch.autograd.Function):
    """ic_region_phase_fwd / ic_region_phase_bwd.

    forward(anchor, M, sensor, norm_in_w, num, den, var, pred, radius, radius_var, box)
      -> (H, num', den', var', pred', radius', radius_var')     (None where a feature is off)
    box carries the non-differentiable context (spec, parameter tensors, mask, dt, ...).
    """

    @staticmethod
    def forward(ctx, anchor, M, sensor, norm_in_w, num, den, var, pred, radius, radius_var, box):
        spec: RegionPhaseSpec = box["spec"]
        P = dict(box["P"])
        if norm_in_w is not None:
            P["norm_in_w"] = norm_in_w
        S = M.shape[0]
        R, d = spec.R, spec.d
        flags = box["flags"]
        afd, rt, use_pred = bool(flags & IC_F_AFD), bool(flags & IC_F_RT), bool(flags & IC_F_PRED)
        mask, dt, surprise = box["mask"], box["dt"], box["surprise"]
        cont = lambda t: None if t is None else t.contiguous()
        M, sensor, num, den, var, pred = cont(M), cont(sensor), cont(num), cont(den), cont(var), cont(pred)
        radius, radius_var = cont(radius), cont(radius_var)
        a = _region_args(spec, P, S, box["step_pos"], box["kwta_k"], flags, box["kwta_temp"])
        H = _e(M, S, R, d)
        num_o, den_o = _e(M, S, R, d), _e(M, S, R, d)
        var_o = _e(M, S, R, d) if afd else None
        pred_o = _e(M, S, R, d) if use_pred else None
        radius_o = _e(M, S, R) if rt else None
        radius_var_o = _e(M, S, R) if rt else None
        surprise_o = _e(M, S, R) if afd else None
        last_dir = _z(M, S, R, d) if rt else None
        last_norm = _z(M, S, R) if rt else None
        dt_o = torch.empty_like(dt)
        sv = {k: _e(M, S, R, d) for k in ("x", "u1", "r_s", "k_s", "v_s", "y", "hpre", "gate")}
        sv["inv_s"] = _e(M, S, R, 2)
        if rt:
            sv["ynorm"] = _e(M, S, R)
        if afd:
            sv["sterm"] = _e(M, S, R, d)
        tensors = dict(M=M, sensor=sensor, mask=mask, num=num, den=den, var=var, pred=pred, radius=radius,
                       radius_var=radius_var, surprise=surprise, dt=dt, H=H, num_o=num_o, den_o=den_o,
                       var_o=var_o, pred_o=pred_o, radius_o=radius_o, radius_var_o=radius_var_o,
                       surprise_o=surprise_o, last_dir=last_dir, last_norm=last_norm, dt_o=dt_o, **sv)
        for k, t in tensors.items():
            if t is not None:
                setattr(a, k, ptr(t))
        _call("ic_region_phase_fwd", a)
        box["dt_o"], box["surprise_o"] = dt_o, surprise_o
        box["last_dir"], box["last_norm"] = last_dir, last_norm
        box["H_detached"] = H
        needs = any(t is not None and t.requires_grad for t in (anchor, M, sensor, num, den, var, pred, radius, radius_var))
        if needs:
            tape: Tape = box["tape"]
            slot = tape.alloc(M.device)
            tape.slot("mask", slot).copy_(mask)
            ctx.slot = slot
            ctx.box = box
            ctx.sv = sv
            ctx.P = P
            ctx.flags = flags
            ctx.has = (sensor is not None, norm_in_w is not None)
            ctx.saved_in = (M, sensor, num, den, var, pred, radius, radius_var, radius_o)
        outs = (H, num_o, den_o, var_o, pred_o, radius_o, radius_var_o)
        ctx.set_materialize_grads(False)
        return outs

    @staticmethod
    def backward(ctx, gH, gnum_o, gden_o, gvar_o, gpred_o, gradius_o, gradius_var_o):
        box = ctx.box
        spec: RegionPhaseSpec = box["spec"]
        coord: GradCoordinator = box["coord"]
        coord.begin()
        M, sensor, num, den, var, pred, radius, radius_var, radius_o = ctx.saved_in
        S, R, d = M.shape
        flags = ctx.flags
        afd, rt, use_pred = bool(flags & IC_F_AFD), bool(flags & IC_F_RT), bool(flags & IC_F_PRED)
        norm_in = bool(flags & IC_F_NORM_IN)
        tape: Tape = box["tape"]
        slot = ctx.slot
        tape.visited[slot:slot + 1].fill_(1)
        b = new("ic_region_bwd_args")
        b.f = _region_args(spec, ctx.P, S, box["step_pos"], box["kwta_k"], flags, box["kwta_temp"])
        fin = dict(M=M, sensor=sensor, mask=box["mask"], num=num, den=den, var=var, pred=pred, radius=radius,
                   radius_var=radius_var, dt=box["dt"], radius_o=radius_o, **ctx.sv)
        for k, t in fin.items():
            if t is not None:
                setattr(b.f, k, ptr(t))
        cont = lambda t: None if t is None else t.contiguous()
        gin = dict(gH=cont(gH), gnum_o=cont(gnum_o), gden_o=cont(gden_o), gvar_o=cont(gvar_o),
                   gpred_o=cont(gpred_o), gradius_o=cont(gradius_o), gradius_var_o=cont(gradius_var_o))
        nslab = bwd_nslab(d)
        out = dict(gM=_e(M, S, R, d), gnum=_e(M, S, R, d), gden=_e(M, S, R, d),
                   g_norm_w_rows=_z(M, S, R, d), g_raw_decay_rows=_z(M, S, R, d),
                   ghpre=_e(M, S, R, d), slab2=_e(M, nslab, S, R, d), slab=_e(M, nslab, S, R, d))
        if use_pred:
            out["gpred"] = _e(M, S, R, d)
        if norm_in:
            out["g_norm_in_rows"] = _z(M, S, R, d)
            if ctx.has[0]:
                out["gsens"] = _z(M, S, 2, d)
        if afd:
            out.update(gvar=_e(M, S, R, d), g_pn_rows=_z(M, S, R, d), g_on_rows=_z(M, S, R, d))
        if rt:
            out.update(gradius=_e(M, S, R), gradius_var=_e(M, S, R), g_rnoise_rows=_z(M, S, R, 2),
                       slab_rt=_e(M, nslab, S, R))
        keep = []
        for k, t in {**gin, **out}.items():
            if t is not None:
                keep.append(t)
                setattr(b, k, ptr(t))
        b.tape_g = tape.slot("g", slot).data_ptr()
        b.tape_x = tape.slot("x", slot).data_ptr()
        b.tape_o = tape.slot("o", slot).data_ptr()
        _call("ic_region_phase_bwd", b)
        # small per-region parameter gradients go straight into the gradient arena
        sink = box["grad_sink"]
        sink("norm_w", out["g_norm_w_rows"].sum(0))
        sink("raw_decay", out["g_raw_decay_rows"].sum(0))
        if afd:
            if flags & IC_F_NOISE_VEC:
                sink("pn_param", out["g_pn_rows"].sum(0))
                sink("on_param", out["g_on_rows"].sum(0))
            else:
                sink("pn_param", out["g_pn_rows"].sum((0, 2)))
                sink("on_param", out["g_on_rows"].sum((0, 2)))
        if rt:
            rn = out["g_rnoise_rows"].sum(0)
            sink("rpn_param", rn[:, 0].contiguous())
            sink("ron_param", rn[:, 1].contiguous())
        g_sensor = out["gsens"].sum(1) if (norm_in and ctx.has[0]) else None
        g_norm_in = out["g_norm_in_rows"].sum((0, 1)) if (norm_in and ctx.has[1]) else None
        return (None, out["gM"], g_sensor, g_norm_in, out["gnum"], out["gden"], out.get("gvar"),
                out.get("gpred"), out.get("gradius"), out.get("gradius_var"), None)


_REGION_WGRAD_TABLES: Dict[tuple, Dict[str, torch.Tensor]] = {}
_IOTA: Dict[tuple, torch.Tensor] = {}


def region_wgrad(tape: Tape, S: int, R: int, d: int, gW: torch.Tensor, touched: torch.Tensor) -> None:
    if tape.pos == 0:
        return
    TS = tape.pos * S
    if WGRAD_GEMM and WGRAD_GEMM_MIN_TS <= TS <= 4096 and d % 32 == 0 and 4 * R <= 4096:
        # Many tape slots (the LoopEngine's pass over all loops of a train_step): dW[r,j] = sum g (x) op is a
        # [d,TS] x [TS,d] product per matrix -- the router's pair-list GEMM on the matrix cores with the region tapes
        # read as 4R "edges" (edge 4r+j: gradient rows g[.,r,j,:], operand row of region r, matrix 4r+j).  The r/k/v
        # matrices take the x tape, the output projection the o tape: two launches, the other matrices' lists empty.
        dev = gW.device
        key = (R, dev)
        t = _REGION_WGRAD_TABLES.get(key)
        if t is None:
            e = torch.arange(4 * R, dtype=torch.int32, device=dev)
            neg = torch.full_like(e, -1)
            t = _REGION_WGRAD_TABLES[key] = {"src": (e // 4).contiguous(), "rkv": torch.where(e % 4 < 3, e, neg).contiguous(),
                                             "o": torch.where(e % 4 == 3, e, neg).contiguous()}
        act4 = tape.buf["mask"][:tape.pos].reshape(TS, R, 1).expand(TS, R, 4).contiguous()  # activity per (pair, pseudo edge)
        t4 = torch.zeros(4 * R, dtype=torch.uint8, device=dev)
        n = (2 * 4 * R + TS * 4 * R + 63) // 64 * 64
        ws = torch.empty(n, dtype=torch.int32, device=dev)
        WGRAD_GEMM_CALLS[0] += 1
        for widx, op in ((t["rkv"], tape.buf["x"]), (t["o"], tape.buf["o"])):
            a = new("ic_wgrad_edge_args", T=tape.pos, S=S, R=R, d=d, E=4 * R, nW=4 * R, which=0, dst_off=t["src"], src=t["src"],
                    widx=widx, tape_g=tape.buf["g"], tape_H=op, tape_eact=act4, visited=tape.visited, gW=gW, touched=t4)
            a.pair_ws, a.pair_ws_len = ptr(ws), n
            _call("ic_router_wgrad", a)
        touched.copy_(torch.maximum(touched, t4.view(R, 4).amax(1)))
        return
    a = new("ic_wgrad_region_args", T=tape.pos, S=S, R=R, d=d, tape_g=tape.buf["g"], tape_x=tape.buf["x"],
            tape_o=tape.buf["o"], mask=tape.buf["mask"], visited=tape.visited, gW=gW, touched=touched)
    _call("ic_region_wgrad", a)


# ============================================================================ router
class RouterSpec:
    def __init__(self, R: int, d: int, E: int, nW: int, precision: bool, shared_gemm: bool = False, factorized: bool = False):
        self.R, self.d, self.E, self.nW, self.precision = R, d, E, nW, precision
        self.shared_gemm = shared_gemm  # by_direction on a map where the GEMM formulation applies (IC_RF_SHARED)
        self.factorized = factorized    # edge_transform_mode "factorized" without precision routing (IC_RF_FACTORIZED)


def _router_args(spec: RouterSpec, T: Dict[str, torch.Tensor], P: Dict[str, torch.Tensor], S: int):
    flags = (IC_RF_PRECISION if spec.precision else 0) | (ENUMS["IC_RF_SHARED"] if spec.shared_gemm else 0)
    if spec.shared_gemm and SHARED_GEMM_ALWAYS:
        flags |= ENUMS["IC_RF_SHARED_GEMM"]
    if spec.factorized:
        flags |= ENUMS["IC_RF_FACTORIZED"]
    a = new("ic_router_args", S=S, R=spec.R, d=spec.d, E=spec.E, flags=flags, nW=spec.nW)
    a.dst_off, a.src, a.widx = T["dst_off"].data_ptr(), T["src"].data_ptr(), T["widx"].data_ptr()
    if spec.factorized:  # msg_e = ebias_e U (w_e * V h): two shared matrices + per-edge scales, identity tables of the row GEMMs
        a.fact_U, a.fact_V, a.fact_w = P["U"].data_ptr(), P["V"].data_ptr(), P["w_e"].data_ptr()
        a.fact_iota, a.fact_zero, a.fact_ones = T["iota"].data_ptr(), T["zero"].data_ptr(), T["ones"].data_ptr()
    else:
        a.W = P["W"].data_ptr()
    if spec.precision:
        a.key_W, a.query_W, a.raw_P = P["key_W"].data_ptr(), P["query_W"].data_ptr(), P["raw_P"].data_ptr()
    # the GEMM's columns: (stream, region) pairs; (stream, edge) pairs for the messages / keys of precision routing
    cols = S * (spec.E if spec.precision else spec.R)
    if (spec.shared_gemm or (spec.factorized and not spec.precision)) and SHARED_SORT and cols >= SHARED_SORT_MIN:
        # workspace of the column sort (ic_router_args.shared_ws): one per (columns, device, stream) -- calls that share
        # one are ordered on that stream -- reused by every call (forward and data gradient)
        dev = T["dst_off"].device
        key = (cols, dev, torch.cuda.current_stream(dev).cuda_stream)
        ws = _SHARED_WS.get(key)
        if ws is None:
            ws = _SHARED_WS[key] = torch.zeros(512 + 2 * cols, dtype=torch.int32, device=dev)
        a.shared_ws, a.shared_ws_len = ws.data_ptr(), ws.numel()
    return a


class RouterFn(torch.autograd.Function):
    """ic_router_fwd / ic_router_bwd.   forward(anchor, H, ebias, box) -> M"""

    @staticmethod
    def forward(ctx, anchor, H, ebias, box):
        spec: RouterSpec = box["spec"]
        S = H.shape[0]
        R, d, E = spec.R, spec.d, spec.E
        H = H.contiguous()
        ebias = ebias.contiguous()
        a = _router_args(spec, box["tables"], box["P"], S)
        M = _e(H, S, R, d)
        a.H, a.mask_prev, a.M, a.ebias = ptr(H), ptr(box["mask_prev"]), ptr(M), ptr(ebias)
        sv = {}
        if spec.precision:
            sv = dict(msg=_e(H, S, E, d), keyv=_e(H, S, E, d), q=_e(H, S, R, d), wgt=_z(H, S, E))
        elif spec.factorized:
            sv = dict(fact_t=_e(H, S, R, d), fact_z=_e(H, S, R, d), fact_dact=_e(H, S, R, dtype=torch.uint8))
        for k, t in sv.items():
            setattr(a, k, ptr(t))
        if E > 0:
            _call("ic_router_fwd", a)
        else:
            M.zero_()
        box["wgt"] = sv.get("wgt")
        if H.requires_grad or ebias.requires_grad or (anchor is not None and anchor.requires_grad):
            tape: Tape = box["tape"]
            ctx.slot = tape.alloc(H.device)
            ctx.box, ctx.sv, ctx.saved_in = box, sv, (H, ebias)
        return M

    @staticmethod
    def backward(ctx, gM):
        box = ctx.box
        spec: RouterSpec = box["spec"]
        box["coord"].begin()
        H, ebias = ctx.saved_in
        S, R, d, E = H.shape[0], spec.R, spec.d, spec.E
        tape: Tape = box["tape"]
        slot = ctx.slot
        if E == 0:
            return None, torch.zeros_like(H), torch.zeros_like(ebias), None
        tape.visited[slot:slot + 1].fill_(1)
        b = new("ic_router_bwd_args")
        b.f = _router_args(spec, box["tables"], box["P"], S)
        b.f.H, b.f.mask_prev, b.f.ebias = H.data_ptr(), box["mask_prev"].data_ptr(), ebias.data_ptr()
        b.f.M = gM.data_ptr()  # unused by the backward, must be non-NULL
        for k, t in ctx.sv.items():
            setattr(b.f, k, ptr(t))
        nslab = bwd_nslab(d)
        gM = gM.contiguous()
        out = dict(gH=_e(H, S, R, d), g_ebias_part=_z(H, nslab, S, E), slab=_e(H, nslab, S, E, d))
        if spec.precision:
            out.update(g_rawP_rows=_z(H, S, E, d), gHd=_e(H, S, R, d), slab_q=_e(H, nslab, S, R, d))
        b.gM = gM.data_ptr()
        for k, t in out.items():
            setattr(b, k, ptr(t))
        b.src_off, b.src_edges = box["tables"]["src_off"].data_ptr(), box["tables"]["src_edges"].data_ptr()
        if spec.factorized:
            fx = dict(g_fw_rows=_z(H, S, E, d), fact_gz=_e(H, S, R, d), fact_gt=_e(H, S, R, d),
                      fact_slab=_e(H, nslab, S, R, d), fact_gb=_e(H, nslab, S, R))
            for k, t in fx.items():
                setattr(b, k, ptr(t))
            for k in ("gU", "zU", "aU", "gV", "aV", "H", "eact"):
                setattr(b, "tape_" + k, tape.slot(k, slot).data_ptr())
            _call("ic_router_bwd", b)
            box["grad_sink"]("w_e", fx["g_fw_rows"].sum(0), box["tables"]["widx_long"])
            return None, out["gH"], out["g_ebias_part"].sum((0, 1)), None
        b.tape_gmsg = tape.slot("gmsg", slot).data_ptr()
        b.tape_H = tape.slot("H", slot).data_ptr()
        b.tape_eact = tape.slot("eact", slot).data_ptr()
        if spec.precision:
            b.tape_gkey = tape.slot("gkey", slot).data_ptr()
            b.tape_gq = tape.slot("gq", slot).data_ptr()
            b.tape_qact = tape.slot("qact", slot).data_ptr()
        _call("ic_router_bwd", b)
        if spec.precision:
            box["grad_sink"]("raw_P", out["g_rawP_rows"].sum(0), box["tables"]["widx_long"])
        g_ebias = out["g_ebias_part"].sum((0, 1))
        return None, out["gH"], g_ebias, None


def router_wgrad(tape: Tape, spec: RouterSpec, tables: Dict[str, torch.Tensor], S: int, which: int,
                 gW: torch.Tensor, touched: torch.Tensor) -> None:
    if tape.pos == 0 or spec.E == 0:
        return
    gname = ("gmsg", "gkey", "gq")[which]
    TS = tape.pos * S
    a = new("ic_wgrad_edge_args", T=tape.pos, S=S, R=spec.R, d=spec.d, E=spec.E, nW=spec.nW, which=which,
            dst_off=tables["dst_off"], src=tables["src"], widx=tables["widx"], tape_g=tape.buf[gname],
            tape_H=tape.buf["H"], tape_eact=tape.buf["qact" if which == 2 else "eact"], visited=tape.visited,
            gW=gW, touched=touched)
    if which != 2 and spec.nW != spec.E:  # shared matrices: workspace for the compacted list of contributing pairs
        n = (2 * spec.nW + TS * spec.E + 63) // 64 * 64 + 8 * spec.nW * spec.d * spec.d  # list + split-K partial tiles
        ws = torch.empty(n, dtype=torch.int32, device=gW.device)
        a.pair_ws, a.pair_ws_len = ptr(ws), n
    elif WGRAD_GEMM and WGRAD_GEMM_MIN_TS <= TS <= 4096 and spec.d % 32 == 0:
        # one matrix per edge (or per region: the queries), many tape slots: the same pair-list GEMM, a list per matrix.
        # The queries' tape is indexed by region: identity edge tables turn it into the edge form.
        if which == 2:
            it = _IOTA.get((spec.R, gW.device))  # (not kept in `tables`: the graph replay keys on that dict's pointers)
            if it is None:
                it = _IOTA[(spec.R, gW.device)] = torch.arange(spec.R + 1, dtype=torch.int32, device=gW.device)
            a.which, a.E, a.nW, a.dst_off, a.src, a.widx = 0, spec.R, spec.R, ptr(it), ptr(it), ptr(it)
        n_m, n_e = (spec.R, spec.R) if which == 2 else (spec.nW, spec.E)
        if n_e <= 4096 and n_m <= 8192:
            n = (2 * n_m + TS * n_e + 63) // 64 * 64
            ws = torch.empty(n, dtype=torch.int32, device=gW.device)
            a.pair_ws, a.pair_ws_len = ptr(ws), n
            WGRAD_GEMM_CALLS[0] += 1
        elif which == 2:  # (too large for the list kernels: back to the region-indexed row-slice kernel)
            a.which, a.E, a.nW = 2, spec.E, spec.nW
            a.dst_off, a.src, a.widx = ptr(tables["dst_off"]), ptr(tables["src"]), ptr(tables["widx"])
    _call("ic_router_wgrad", a)


def router_fact_wgrad(tape: Tape, spec: RouterSpec, tables: Dict[str, torch.Tensor], S: int, gU: torch.Tensor,
                      gV: torch.Tensor, touched: torch.Tensor) -> None:
    """Deferred dU = sum gM (x) z and dV = sum gt (x) H of the factorized mode: the shared-matrix weight-gradient GEMM
    over the (slot, stream, region) rows of the two tapes (identity edge tables).  touched: uint8 [2] (U, V)."""
    if tape.pos == 0 or spec.E == 0:
        return
    R, d = spec.R, spec.d
    for i, (g, op, act, gW) in enumerate((("gU", "zU", "aU", gU), ("gV", "H", "aV", gV))):
        a = new("ic_wgrad_edge_args", T=tape.pos, S=S, R=R, d=d, E=R, nW=1, which=0, dst_off=tables["iota"], src=tables["iota"],
                widx=tables["zero"], tape_g=tape.buf[g], tape_H=tape.buf[op], tape_eact=tape.buf[act], visited=tape.visited,
                gW=gW, touched=touched[i:i + 1])
        if R != 1:
            n = (2 + tape.pos * S * R + 63) // 64 * 64 + 8 * d * d
            ws = torch.empty(n, dtype=torch.int32, device=gW.device)
            a.pair_ws, a.pair_ws_len = ptr(ws), n
        _call("ic_router_wgrad", a)


# ============================================================================ gate
def gate_scores_select(H: torch.Tensor, gain_ema: torch.Tensor, homeo: torch.Tensor,
                       focus_any: Optional[torch.Tensor], nbr_off: torch.Tensor, nbr: torch.Tensor,
                       sensor_idx: int, motor_idx: int, k_total: int, force_io: bool, t5: float, t6: float,
                       weights=(1.0, 0.3, 0.5, 0.2), select: bool = True, scores: Optional[torch.Tensor] = None):
    """Gate.score_regions (+ select_k).  H [S,R,d] -> (scores [S,R], mask uint8 [S,R])."""
    S, R, d = H.shape
    H = H.detach().contiguous()
    mask = _e(H, S, R, dtype=torch.uint8)
    a = new("ic_gate_args", S=S, R=R, d=d, sensor_idx=sensor_idx, motor_idx=motor_idx, k_total=int(k_total),
            force_io=int(bool(force_io)), w1=weights[0], w2=weights[1], w3=weights[2], w4=weights[3],
            t5=float(t5), t6=float(t6), H=H, gain_ema=gain_ema, homeo=homeo, focus_any=focus_any,
            nbr_off=nbr_off, nbr=nbr, mask=mask)
    if scores is None:
        scores = _e(H, S, R)
        a.scores = scores.data_ptr()
        _call("ic_gate_fwd" if select else "ic_gate_scores", a)
        select = False
    else:
        scores = scores.contiguous()
        a.scores = scores.data_ptr()
    if select:
        _call("ic_gate_select", a)
    return scores, mask


# ============================================================================ primitives
class KWTAFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, k, soft, temp):
        shape = x.shape
        d = shape[-1]
        x2 = x.contiguous().view(-1, d)
        y, gate = torch.empty_like(x2), torch.empty_like(x2)
        check(lib().ic_kwta_fwd(_vp(x2), _vp(y),
                                _vp(gate), ctypes.c_int32(x2.shape[0]),
                                ctypes.c_int32(d), ctypes.c_int32(int(k)), ctypes.c_int32(int(bool(soft))),
                                ctypes.c_float(float(temp)), stream_handle()), "ic_kwta_fwd")
        ctx.save_for_backward(gate)
        ctx.soft = bool(soft)
        return y.view(shape)

    @staticmethod
    def backward(ctx, gy):
        (gate,) = ctx.saved_tensors
        if ctx.soft:
            raise RuntimeError("soft KWTA is differentiable only inside the fused region step")
        d = gate.shape[-1]
        g2 = gy.contiguous().view(-1, d)
        gx = torch.empty_like(g2)
        check(lib().ic_kwta_bwd(None, _vp(gate), _vp(g2),
                                _vp(gx), ctypes.c_int32(g2.shape[0]), ctypes.c_int32(d),
                                ctypes.c_int32(0), ctypes.c_int32(0), ctypes.c_float(1.0), stream_handle()),
              "ic_kwta_bwd")
        return gx.view(gy.shape), None, None, None


class RMSNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, eps):
        shape = x.shape
        d = shape[-1]
        x2 = x.contiguous().view(-1, d)
        w = w.contiguous()
        y = torch.empty_like(x2)
        inv = torch.empty(x2.shape[0], dtype=torch.float32, device=x.device)
        check(lib().ic_rmsnorm_fwd(_vp(x2), _vp(w),
                                   _vp(y), _vp(inv),
                                   ctypes.c_int32(x2.shape[0]), ctypes.c_int32(d), ctypes.c_float(float(eps)),
                                   stream_handle()), "ic_rmsnorm_fwd")
        ctx.save_for_backward(x2, w, inv)
        return y.view(shape)

    @staticmethod
    def backward(ctx, gy):
        x2, w, inv = ctx.saved_tensors
        d = x2.shape[-1]
        g2 = gy.contiguous().view(-1, d)
        gx, gw_rows = torch.empty_like(x2), torch.empty_like(x2)
        check(lib().ic_rmsnorm_bwd(_vp(x2), _vp(w),
                                   _vp(inv), _vp(g2),
                                   _vp(gx), _vp(gw_rows),
                                   ctypes.c_int32(x2.shape[0]), ctypes.c_int32(d), stream_handle()),
              "ic_rmsnorm_bwd")
        return gx.view(gy.shape), gw_rows.sum(0), None


class GoodnessFn(torch.autograd.Function):
    """mean_d H^2 per row (training.py:127-128)."""

    @staticmethod
    def forward(ctx, H):
        d = H.shape[-1]
        h2 = H.contiguous().view(-1, d)
        g = torch.empty(h2.shape[0], dtype=torch.float32, device=H.device)
        check(lib().ic_goodness_fwd(_vp(h2), _vp(g),
                                    ctypes.c_int32(h2.shape[0]), ctypes.c_int32(d), stream_handle()),
              "ic_goodness_fwd")
        ctx.save_for_backward(h2)
        ctx.shape = H.shape
        return g.view(H.shape[:-1])

    @staticmethod
    def backward(ctx, gg):
        (h2,) = ctx.saved_tensors
        d = h2.shape[-1]
        gg = gg.contiguous().view(-1)
        gH = torch.empty_like(h2)
        check(lib().ic_goodness_bwd(_vp(h2), _vp(gg),
                                    _vp(gH), ctypes.c_int32(h2.shape[0]),
                                    ctypes.c_int32(d), stream_handle()), "ic_goodness_bwd")
        return gH.view(ctx.shape)


LINEAR_MAX_ROWS = 16  # batch rows ic_linear_bwd takes
ACTS = {"none": ENUMS["IC_ACT_NONE"], "gelu": ENUMS["IC_ACT_GELU"], "silu": ENUMS["IC_ACT_SILU"]}


class LinearActFn(torch.autograd.Function):
    """y = act(x W^T + b) for x [B <= 16, n] (heads.py:39-91, energy.py:12-30 on train_step's loss rows): forward
    ic_gemv_chain (one launch), backward ic_linear_bwd (pre-activation recomputed; gW / gb / gx without atomics)."""

    @staticmethod
    def forward(ctx, x, W, b, act: int):
        B, n = x.shape
        rows = W.shape[0]
        x, W = x.contiguous(), W.contiguous()
        y = torch.empty(B, rows, device=x.device)
        g = new("ic_gemv_desc", W=W, b=b, x0=x, y=y, n0=n, rows=rows, act=act, nbatch=B, x0_bs=n, y_bs=rows)
        check(lib().ic_gemv_chain(ctypes.byref(g), ctypes.c_int32(1), stream_handle()), "ic_gemv_chain")
        ctx.save_for_backward(x, W, b)
        ctx.act = act
        return y

    @staticmethod
    def backward(ctx, gy):
        x, W, b = ctx.saved_tensors
        B, n = x.shape
        rows = W.shape[0]
        need_x, need_W, need_b = ctx.needs_input_grad[0], ctx.needs_input_grad[1], b is not None and ctx.needs_input_grad[2]
        gx = torch.empty_like(x) if need_x else None
        gW = torch.empty_like(W) if need_W else None
        gb = torch.empty_like(b) if need_b else None
        slab = torch.empty(B, rows, device=x.device) if need_x else None
        a = new("ic_linear_bwd_args", B=B, n=n, rows=rows, act=ctx.act, W=W, b=b, x=x, gy=gy.contiguous(), gx=gx, gW=gW, gb=gb, slab=slab)
        _call("ic_linear_bwd", a)
        return gx, gW, gb, None


LINEAR_MAX_OUT = 2048  # output rows ic_linear_bwd's input-gradient kernel stages per workgroup


def linear_act(x: torch.Tensor, W: torch.Tensor, b: Optional[torch.Tensor], act: str = "none") -> torch.Tensor:
    if W.shape[0] > LINEAR_MAX_OUT:  # wider than any head of the model: plain PyTorch
        y = torch.nn.functional.linear(x, W, b)
        return torch.nn.functional.gelu(y) if act == "gelu" else (torch.nn.functional.silu(y) if act == "silu" else y)
    return LinearActFn.apply(x, W, b, ACTS[act])


def linear_act_fits(x: torch.Tensor) -> bool:
    """A small fp32 batch of rows on the device: the shape of every loss head of train_step."""
    return x.is_cuda and x.dim() == 2 and 1 <= x.shape[0] <= LINEAR_MAX_ROWS and x.dtype == torch.float32


ATTN_SHAPES = ((32, 32), (64, 64), (128, 128), (96, 32), (128, 64), (192, 128))  # (dqk, dv) the library has kernels for


def _pad_cols(x: torch.Tensor, w: int) -> torch.Tensor:
    return x if x.shape[-1] == w else torch.nn.functional.pad(x, (0, w - x.shape[-1]))


class AttnFn(torch.autograd.Function):
    """softmax(q k^T [* exp(-alpha |i-j| dt)]) v for [B,H,T,*] rows on the matrix cores (ic_attn_fwd / ic_attn_bwd;
    iron_rope.py:201-235, adaptive_filter_attention.py:60-88): no [T,T] tensor.  Returns (out, stats) with
    stats[b,h,q] = (sum_k log(p+eps), sum_k p log(p+eps)) over the valid keys (no gradient, as in the reference)."""

    @staticmethod
    def forward(ctx, q, k, v, valid_len, alpha, lag_dt, eps_log):
        B, H, T, dqk = q.shape
        dv = v.shape[-1]
        q, k, v = q.contiguous(), k.contiguous(), v.contiguous()
        out = torch.empty(B, H, T, dv, device=q.device)
        lse = torch.empty(B, H, T, device=q.device)
        stats = torch.empty(B, H, T, 2, device=q.device)
        al = None if alpha is None else alpha.detach().reshape(1).contiguous()
        a = new("ic_attn_args", B=B, H=H, T=T, dqk=dqk, dv=dv, q=q, k=k, v=v, valid_len=valid_len, alpha=al,
                lag_dt=float(lag_dt), eps_log=float(eps_log), out=out, lse=lse, stats=stats)
        _call("ic_attn_fwd", a)
        ctx.save_for_backward(q, k, v, out, lse, valid_len, al)
        ctx.cfg = (float(lag_dt), float(eps_log), alpha is not None and alpha.requires_grad)
        ctx.mark_non_differentiable(stats)
        return out, stats

    @staticmethod
    def backward(ctx, gout, _gstats):
        q, k, v, out, lse, valid_len, al = ctx.saved_tensors
        lag_dt, eps_log, want_alpha = ctx.cfg
        B, H, T, dqk = q.shape
        dv = v.shape[-1]
        gout = gout.contiguous()
        drow = (gout * out).sum(-1)
        dq, dk, dvv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        part = torch.empty(B * H * ((T + 127) // 128), device=q.device) if al is not None else None
        f = new("ic_attn_args", B=B, H=H, T=T, dqk=dqk, dv=dv, q=q, k=k, v=v, valid_len=valid_len, alpha=al,
                lag_dt=lag_dt, eps_log=eps_log, out=out, lse=lse, stats=None)
        b = new("ic_attn_bwd_args", f=f, dout=gout, drow=drow, dq=dq, dk=dk, dv=dvv, dalpha_part=part)
        _call("ic_attn_bwd", b)
        galpha = part.sum().reshape(()) if want_alpha else None
        return dq, dk, dvv, None, galpha, None, None


def attention_fits(dqk: int, dv: int) -> bool:
    """True when the library has an attention kernel for q/k rows of dqk and v rows of dv columns (after zero-padding)."""
    return any(a >= dqk and b >= dv for a, b in ATTN_SHAPES)


def attention(q, k, v, valid_len=None, alpha=None, lag_dt: float = 1.0, eps_log: float = 1e-9):
    """AttnFn with the rows zero-padded to a (dqk, dv) pair the library has kernels for (zero columns change neither
    the scores nor the kept output columns)."""
    dqk, dv = q.shape[-1], v.shape[-1]
    if q.dtype != torch.float32 or k.dtype != torch.float32 or v.dtype != torch.float32:
        raise RuntimeError("ic_attn: fp32 tensors only")
    fit = [(a, b) for a, b in ATTN_SHAPES if a >= dqk and b >= dv]
    if not fit:
        raise RuntimeError(f"ic_attn: no kernel for q/k rows of {dqk} and v rows of {dv} columns")
    pa, pb = min(fit, key=lambda s: s[0] * 2 + s[1])
    vl = None if valid_len is None else valid_len.to(torch.int32).contiguous()
    out, stats = AttnFn.apply(_pad_cols(q, pa), _pad_cols(k, pa), _pad_cols(v, pb), vl, alpha, lag_dt, eps_log)
    return (out if pb == dv else out[..., :dv]), stats


def hebbian_rank1(W: torch.Tensor, dst_of_edge: torch.Tensor, src: torch.Tensor, widx: torch.Tensor,
                  coef: torch.Tensor, H: torch.Tensor) -> None:
    """In-place W[widx[e]] += sum_b coef[b,e] * outer(H[b,dst_e], H[b,src_e])."""
    B, R, d = H.shape
    E = src.shape[0]
    check(lib().ic_hebbian_rank1(_vp(W), _vp(dst_of_edge),
                                 _vp(src), _vp(widx),
                                 _vp(coef.contiguous()),
                                 _vp(H.contiguous()), ctypes.c_int32(B),
                                 ctypes.c_int32(R), ctypes.c_int32(d), ctypes.c_int32(E), stream_handle()),
          "ic_hebbian_rank1")
