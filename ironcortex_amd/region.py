"""RWKV region cells behind the reference API, executed by the fused HIP region phase.

`RegionBank` stacks the parameters/state of R cells ([R,4,d,d] weights, [S,R,d] state) and runs
one inner step's region work for all of them in 4 kernel launches (ic_region_phase_fwd).
`RWKVRegionCell` keeps the reference's constructor, attributes and methods (region.py:14-249);
inside a CortexReasoner every cell is a window onto the shared bank, on its own it owns a
one-region bank.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

from .constants import EPS_DIV, MAX_EXP, MAX_NOISE
from .flat import ArenaParameter, FlatParams
from .hip import ops
from .utils import RMSNorm


class _Lin(nn.Module):
    """Bias-free d x d projection whose weight lives in a flat arena."""

    _ic_linear = True

    def __init__(self, d_in: int, d_out: int):
        super().__init__()
        self.weight = ArenaParameter(torch.empty(d_out, d_in))
        self.bias = None
        nn.init.xavier_uniform_(self.weight)

    def forward(self, x: torch.Tensor) -> torch.Tensor:  # API compatibility only (not the hot path)
        return F.linear(x, self.weight)


def log_freq_bank(m: int, base: float = 10000.0) -> torch.Tensor:
    """[m,1] log-spaced frequencies (iron_rope.py:26-33)."""
    return torch.logspace(math.log(1.0 / base), 0.0, steps=m, base=math.e).reshape(m, 1)


class RegionBank:
    """Stacked storage + fused execution for a list of region cells (not an nn.Module: the
    Parameters stay registered on the cells, under the reference's names)."""

    def __init__(self, cells: List["RWKVRegionCell"], sensor_idx: int = -1, motor_idx: int = -1,
                 coord: Optional[ops.GradCoordinator] = None):
        c0 = cells[0]
        self.cells = cells
        self.R, self.d = len(cells), c0.d
        self.afd = c0.enable_adaptive_filter_dynamics
        self.rt = c0.enable_radial_tangential_updates
        self.noise_vec = c0.afd_noise_mode != "scalar"
        self.use_pred = c0.use_predictive_trace
        self.m_time = c0.m_time
        groups: Dict[str, List[nn.Parameter]] = {
            "W": [w for c in cells for w in (c.r_lin.weight, c.k_lin.weight, c.v_lin.weight, c.o_lin.weight)],
            "norm_w": [c.norm.weight for c in cells],
            "raw_decay": [c.raw_decay for c in cells],
            "pn_param": [c.process_noise_param for c in cells],
            "on_param": [c.obs_noise_param for c in cells],
            "rpn_param": [c.radius_process_noise_param for c in cells],
            "ron_param": [c.radius_obs_noise_param for c in cells],
        }
        self.flat = FlatParams(groups)
        self.spec = ops.RegionPhaseSpec(self.R, self.d, self.m_time, 0, sensor_idx, motor_idx)
        self.coord = coord or ops.GradCoordinator()
        self.coord.register_flagged(self.flat.begin_pass, self._launch_wgrad, self._publish)
        self.S = 1
        self.tape: ops.Tape = None  # type: ignore
        self._anchor: torch.Tensor = None  # type: ignore
        self._touched: torch.Tensor = None  # type: ignore
        self.reset_state(torch.device("cpu"))
        for i, c in enumerate(cells):
            c._bank, c._index = self, i

    # ------------------------------------------------------------------ state
    def reset_state(self, device, S: int = 1) -> None:
        R, d = self.R, self.d
        z = lambda *s: torch.zeros(*s, device=device)
        self.S = S
        self.num, self.den, self.var, self.pred = z(S, R, d), z(S, R, d), z(S, R, d), z(S, R, d)
        self.radius, self.radius_var, self.surprise = z(S, R), z(S, R), z(S, R)
        self.last_dir, self.last_norm = z(S, R, d), z(S, R)
        self.dt = torch.zeros(S, R, dtype=torch.int32, device=device)
        self.tape = ops.Tape({"g": (S, R, 4, d), "x": (S, R, d), "o": (S, R, d), "mask": (S, R)},
                             {"mask": torch.uint8})

    def to_device(self) -> None:
        """Follow the parameters after Module._apply (device moves)."""
        self.flat.reflat()
        dev = self.flat.flat.device
        if self.num.device != dev:
            for k in ("num", "den", "var", "pred", "radius", "radius_var", "surprise", "last_dir",
                      "last_norm", "dt"):
                setattr(self, k, getattr(self, k).detach().to(dev))
            self.tape = ops.Tape(self.tape.fields, self.tape.dtypes)
            self._anchor = None
            self._touched = None

    def detach_state(self) -> None:
        """region.py:148-157 for every cell."""
        for k in ("num", "den", "var", "pred", "radius", "radius_var", "surprise"):
            t = getattr(self, k)
            if t.requires_grad:  # (views of the loop engine's arena never do)
                setattr(self, k, t.detach())

    # ------------------------------------------------------------------ parameters
    def params(self, lo: int = 0, hi: Optional[int] = None) -> Dict[str, torch.Tensor]:
        hi = self.R if hi is None else hi
        f = self.flat
        P = {"W": f.stacked("W").view(self.R, 4, self.d, self.d)[lo:hi],
             "norm_w": f.stacked("norm_w")[lo:hi], "raw_decay": f.stacked("raw_decay")[lo:hi]}
        for k in ("pn_param", "on_param", "rpn_param", "ron_param"):
            P[k] = f.stacked(k)[lo:hi]
        if self.m_time > 0:
            P["W_time"] = self.cells[0].W_time.reshape(-1)
        return P

    def _flags(self, soft: bool, disable_kwta: bool, norm_in: bool) -> int:
        fl = 0
        fl |= ops.IC_F_AFD if self.afd else 0
        fl |= ops.IC_F_RT if self.rt else 0
        fl |= ops.IC_F_PRED if self.use_pred else 0
        fl |= ops.IC_F_NOISE_VEC if (self.afd and self.noise_vec) else 0
        fl |= ops.IC_F_NOKWTA if disable_kwta else 0
        fl |= ops.IC_F_KWTA_SOFT if soft else 0
        fl |= ops.IC_F_NORM_IN if norm_in else 0
        return fl

    def _grad_sink(self, lo: int):
        def sink(name: str, rows: torch.Tensor) -> None:
            g = self.flat.stacked_grad(name)
            n = rows.shape[0]
            g[lo:lo + n].add_(rows.view(g[lo:lo + n].shape))
        return sink

    def _launch_wgrad(self):
        """End of a backward pass, phase 1: the deferred dW launch; returns the device-side touched flags [R]
        (read by the GradCoordinator together with every other participant's, one host sync)."""
        if self.tape.pos == 0:
            return None
        dev = self.flat.flat.device
        if self._touched is None or self._touched.device != dev:
            self._touched = torch.zeros(self.R, dtype=torch.uint8, device=dev)
        self._touched.zero_()
        gW = self.flat.stacked_grad("W")
        ops.region_wgrad(self.tape, self.S, self.R, self.d, gW, self._touched)
        self.tape.reset()
        return self._touched

    def _publish(self, touched) -> None:
        """Phase 2: .grad views for the regions that took part (on any rank, under data parallelism)."""
        self.flat.publish("W", [t for t in touched for _ in range(4)])
        self.flat.publish("norm_w", touched)
        self.flat.publish("raw_decay", touched)
        if self.afd:
            self.flat.publish("pn_param", touched)
            self.flat.publish("on_param", touched)
        if self.rt:
            self.flat.publish("rpn_param", touched)
            self.flat.publish("ron_param", touched)

    def anchor(self, device) -> torch.Tensor:
        if self._anchor is None or self._anchor.device != device:
            self._anchor = torch.zeros(1, device=device, requires_grad=True)
        return self._anchor

    # ------------------------------------------------------------------ execution
    def kwta_now(self):
        """(k, soft, temp) from the cells' shared schedule (region.py:238-248)."""
        c = self.cells[0]
        warm = min(1.0, c.global_step / c.kwta_k_schedule) if c.kwta_k_schedule > 0 else 1.0
        k = int(round(c.kwta_k_start + (c.kwta_k - c.kwta_k_start) * warm))
        if c.kwta_soft_mode and warm < 1.0:
            return max(1, k), True, c.kwta_soft_temp * (1.0 - warm) + 1.0 * warm
        return max(1, k), False, 1.0

    def phase(self, M: torch.Tensor, sensor: Optional[torch.Tensor], norm_in_w: Optional[torch.Tensor],
              mask: torch.Tensor, step_pos: float) -> torch.Tensor:
        """One inner step for all regions.  M [S,R,d], sensor [S,d], mask uint8 [S,R] -> H [S,R,d]."""
        if M.shape[0] != self.S or self.num.device != M.device:
            self.flat.reflat()
            self.reset_state(M.device, M.shape[0])
        k, soft, temp = self.kwta_now()
        box = dict(spec=self.spec, P=self.params(), flags=self._flags(soft, self.cells[0].disable_kwta, norm_in_w is not None),
                   mask=mask, dt=self.dt, surprise=self.surprise, step_pos=step_pos, kwta_k=k,
                   kwta_temp=temp, tape=self.tape, coord=self.coord, grad_sink=self._grad_sink(0))
        anchor = self.anchor(M.device) if torch.is_grad_enabled() else None
        H, self.num, self.den, var, pred, radius, radius_var = ops.RegionPhaseFn.apply(
            anchor, M, sensor, norm_in_w, self.num, self.den, self.var if self.afd else None,
            self.pred if self.use_pred else None, self.radius if self.rt else None,
            self.radius_var if self.rt else None, box)
        self.dt = box["dt_o"]
        if self.afd:
            self.var, self.surprise = var, box["surprise_o"]
        if self.use_pred:
            self.pred = pred
        if self.rt:
            self.radius, self.radius_var = radius, radius_var
            m3 = mask.bool().unsqueeze(-1)
            self.last_dir = torch.where(m3, box["last_dir"], self.last_dir)
            self.last_norm = torch.where(mask.bool(), box["last_norm"], self.last_norm)
        return H

    def step_single(self, r: int, x_in: torch.Tensor, step_pos: float) -> torch.Tensor:
        """RWKVRegionCell.step for one cell: a one-region launch on the cell's slices."""
        if self.S != 1 or self.num.device != x_in.device:
            self.flat.reflat()
            self.reset_state(x_in.device, 1)
        c = self.cells[r]
        k, soft, temp = c._kwta_now()
        d = self.d
        spec = ops.RegionPhaseSpec(1, d, self.m_time, 0, -1, -1)
        sl = lambda t: t[:, r:r + 1].contiguous()
        tape = _SliceTape(self.tape, r)
        box = dict(spec=spec, P=self.params(r, r + 1), flags=self._flags(soft, c.disable_kwta, False),
                   mask=torch.ones(1, 1, dtype=torch.uint8, device=x_in.device), dt=sl(self.dt),
                   surprise=sl(self.surprise), step_pos=step_pos, kwta_k=k, kwta_temp=temp, tape=tape,
                   coord=self.coord, grad_sink=self._grad_sink(r))
        anchor = self.anchor(x_in.device) if torch.is_grad_enabled() else None
        H, num, den, var, pred, radius, radius_var = ops.RegionPhaseFn.apply(
            anchor, x_in.reshape(1, 1, d), None, None, sl(self.num), sl(self.den),
            sl(self.var) if self.afd else None, sl(self.pred) if self.use_pred else None,
            sl(self.radius) if self.rt else None, sl(self.radius_var) if self.rt else None, box)
        put = lambda full, new: torch.cat([full[:, :r], new, full[:, r + 1:]], dim=1)
        self.num, self.den = put(self.num, num), put(self.den, den)
        self.dt = put(self.dt, box["dt_o"])
        if self.afd:
            self.var, self.surprise = put(self.var, var), put(self.surprise, box["surprise_o"])
        if self.use_pred:
            self.pred = put(self.pred, pred)
        if self.rt:
            self.radius, self.radius_var = put(self.radius, radius), put(self.radius_var, radius_var)
            self.last_dir, self.last_norm = put(self.last_dir, box["last_dir"]), put(self.last_norm, box["last_norm"])
        return H.reshape(d)


class _SliceTape:
    """Tape facade that lands a one-region launch in region r's rows of the bank tape (S == 1)."""

    def __init__(self, tape: ops.Tape, r: int):
        self.tape, self.r = tape, r

    @property
    def visited(self):
        return self.tape.visited

    def alloc(self, device) -> int:
        t = self.tape.alloc(device)
        self.tape.slot("mask", t).zero_()
        return t

    def slot(self, name: str, t: int) -> torch.Tensor:
        return self.tape.slot(name, t)[:, self.r:self.r + 1]


class RWKVRegionCell(nn.Module):
    """RWKV-like region processor with per-channel decay and delta-t skip handling.

    Constructor, attributes and methods follow reference region.py:25-249; the arithmetic of
    step() runs in the HIP region phase."""

    def __init__(self, d: int, m_time_pairs: int = 16, init_decay: float = 0.25, *,
                 enable_adaptive_filter_dynamics: bool = False,
                 enable_radial_tangential_updates: bool = False, afd_noise_mode: str = "scalar",
                 use_predictive_trace: bool = True, kwta_k: int = 0, kwta_k_start: int = 0,
                 kwta_k_schedule: int = 0, kwta_soft_mode: bool = False, kwta_soft_temp: float = 1.0,
                 disable_kwta: bool = False, _shared_bank: bool = False):
        super().__init__()
        self.d = d
        self.enable_adaptive_filter_dynamics = enable_adaptive_filter_dynamics
        self.enable_radial_tangential_updates = enable_radial_tangential_updates
        self.afd_noise_mode = afd_noise_mode
        self.use_predictive_trace = use_predictive_trace
        self.kwta_k = kwta_k if kwta_k > 0 else max(1, d // 8)
        self.kwta_k_start = kwta_k_start if kwta_k_start > 0 else self.kwta_k
        self.kwta_k_schedule = kwta_k_schedule
        self.kwta_soft_mode = kwta_soft_mode
        self.kwta_soft_temp = kwta_soft_temp
        self.disable_kwta = disable_kwta
        self.global_step = 0
        self.surprise_beta = 0.99
        self.radius_momentum = 0.2
        self.norm = RMSNorm(d)
        self.r_lin, self.k_lin = _Lin(d, d), _Lin(d, d)
        self.v_lin, self.o_lin = _Lin(d, d), _Lin(d, d)
        raw_init = math.log((1 / math.sqrt(init_decay)) - 1)
        self.raw_decay = ArenaParameter(torch.full((d,), raw_init))
        nshape = () if afd_noise_mode == "scalar" else (d,)
        self.process_noise_param = ArenaParameter(torch.full(nshape, -4.0))
        self.obs_noise_param = ArenaParameter(torch.zeros(nshape))
        self.radius_process_noise_param = ArenaParameter(torch.tensor(-4.0))
        self.radius_obs_noise_param = ArenaParameter(torch.tensor(0.0))
        self.m_time = max(0, min(d // 8 // 2, m_time_pairs))
        if self.m_time > 0:
            self.register_buffer("W_time", log_freq_bank(self.m_time))
        self._bank: RegionBank = None  # type: ignore
        self._index = 0
        if not _shared_bank:
            RegionBank([self])

    # ------------------------------------------------------------------ plumbing
    def _apply(self, fn, recurse=True):
        super()._apply(fn, recurse)
        if self._bank is not None and self._bank.cells[0] is self and len(self._bank.cells) == 1:
            self._bank.to_device()
        return self

    def _kwta_now(self):
        warm = min(1.0, self.global_step / self.kwta_k_schedule) if self.kwta_k_schedule > 0 else 1.0
        k = int(round(self.kwta_k_start + (self.kwta_k - self.kwta_k_start) * warm))
        if self.kwta_soft_mode and warm < 1.0:
            return max(1, k), True, self.kwta_soft_temp * (1.0 - warm) + 1.0 * warm
        return max(1, k), False, 1.0

    def _get(self, name: str) -> torch.Tensor:
        return getattr(self._bank, name)[0, self._index]

    def _set(self, name: str, value) -> None:
        full = getattr(self._bank, name)
        value = torch.as_tensor(value, device=full.device).to(full.dtype).reshape(full[0, self._index].shape)
        new = full.clone()
        new[0, self._index] = value
        setattr(self._bank, name, new)

    state_num = property(lambda s: s._get("num"), lambda s, v: s._set("num", v))
    state_den = property(lambda s: s._get("den"), lambda s, v: s._set("den", v))
    state_var = property(lambda s: s._get("var"), lambda s, v: s._set("var", v))
    pred = property(lambda s: s._get("pred"), lambda s, v: s._set("pred", v))
    surprise_ema = property(lambda s: s._get("surprise"), lambda s, v: s._set("surprise", v))
    radius = property(lambda s: s._get("radius").reshape(1), lambda s, v: s._set("radius", v))
    radius_var = property(lambda s: s._get("radius_var").reshape(1), lambda s, v: s._set("radius_var", v))
    last_dir = property(lambda s: s._get("last_dir"))
    last_norm = property(lambda s: s._get("last_norm").reshape(1))

    @property
    def dt(self) -> int:
        return int(self._bank.dt[0, self._index])

    @dt.setter
    def dt(self, v: int) -> None:
        self._bank.dt[0, self._index] = int(v)

    # ------------------------------------------------------------------ reference methods
    def decay_rate(self) -> torch.Tensor:
        return -F.softplus(self.raw_decay).clamp_min(-MAX_EXP)

    def decay(self, dt: int = 1) -> torch.Tensor:
        return torch.exp(self.decay_rate() * dt).clamp_min(EPS_DIV)

    def _noise(self, p: torch.Tensor, expand: bool) -> torch.Tensor:
        n = F.softplus(p).clamp_min(EPS_DIV).clamp_max(MAX_NOISE)
        return n.expand(self.d) if (expand and n.numel() == 1) else n

    def process_noise(self) -> torch.Tensor:
        return self._noise(self.process_noise_param, True)

    def obs_noise(self) -> torch.Tensor:
        return self._noise(self.obs_noise_param, True)

    def radius_process_noise(self) -> torch.Tensor:
        return self._noise(self.radius_process_noise_param, False)

    def radius_obs_noise(self) -> torch.Tensor:
        return self._noise(self.radius_obs_noise_param, False)

    def fast_forward(self, dt: int):
        """Exact decay over dt skipped micro-steps (region.py:128-135)."""
        if dt <= 0:
            return
        lam = self.decay(dt=dt)
        self.state_num = self.state_num * lam
        self.state_den = self.state_den * lam
        if self.enable_adaptive_filter_dynamics:
            self.state_var = self.state_var * lam.pow(2) + self.process_noise() * dt

    def skip(self):
        self._bank.dt[0, self._index] += 1

    def predict(self, msg: torch.Tensor, alpha: float = 0.95):
        if not self.use_predictive_trace:
            return
        self.pred = alpha * self.pred + (1.0 - alpha) * msg

    def detach_state(self):
        self._bank.detach_state()

    def telemetry(self) -> dict:
        prec = (self.state_var + EPS_DIV).reciprocal()
        return {"state_var": float(self.state_var.mean().detach()),
                "state_prec": float(prec.mean().detach()),
                "surprise_ema": float(self.surprise_ema.detach())}

    def step(self, x_in: torch.Tensor, step_pos_scalar: float) -> torch.Tensor:
        """One region update on a [d] vector (region.py:168-249), fused on the GPU."""
        if self._bank.flat.flat.device != x_in.device:
            raise RuntimeError("RWKVRegionCell parameters and input are on different devices")
        return self._bank.step_single(self._index, x_in, float(step_pos_scalar))
