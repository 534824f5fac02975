"""CPU oracle for the IronCortex inner-step hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ironcortex_amd/ imports this package; only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may use it, and only as the checker.

A restatement, in plain torch fp32 on the CPU, of the reference algorithm (citations are
file:line into the reference tree).  Every function is differentiable by ordinary autograd so
the same code checks the HIP backward.  Pinned against fixtures generated from the real
reference by tests/golden/make_fixtures.py (tests/test_oracle_vs_reference.py).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

EPS_DIV = 1e-8   # constants.py:1
EPS_LOG = 1e-12  # constants.py:2
MAX_EXP = 40.0   # constants.py:3
MAX_NOISE = 1.0  # constants.py:4


# --------------------------------------------------------------------------- primitives
def rmsnorm(x: torch.Tensor, weight: torch.Tensor, eps: float = 1e-5) -> torch.Tensor:
    """utils.py:33-35."""
    s = x.pow(2).mean(dim=-1, keepdim=True).add(eps).sqrt()
    return weight * (x / s)


def kwta(x: torch.Tensor, k: int, soft: bool = False, temp: float = 1.0) -> torch.Tensor:
    """utils.py:38-60: keep units with |x| >= k-th largest |x| (ties keep extras)."""
    d = x.shape[-1]
    if k <= 0:
        return torch.zeros_like(x)
    k = max(1, min(k, d))
    mag = x.abs()
    thr = torch.topk(mag, k, dim=-1).values[..., -1:].expand_as(mag)
    keep = mag >= thr
    if soft:
        g = torch.sigmoid((mag - thr) / temp)
        return x * (keep.to(x.dtype) + (~keep).to(x.dtype) * g)
    return x * keep


def rotate_pairs(v: torch.Tensor, theta: torch.Tensor) -> torch.Tensor:
    """iron_rope.py:49-66: rotate channel pairs (2j,2j+1), j < len(theta), by theta[j]."""
    m = theta.shape[0]
    if m == 0:
        return v
    c, s = torch.cos(theta), torch.sin(theta)
    head = v[..., : 2 * m].reshape(*v.shape[:-1], m, 2)
    a, b = head[..., 0], head[..., 1]
    rot = torch.stack((a * c - b * s, a * s + b * c), dim=-1).reshape(*v.shape[:-1], 2 * m)
    return torch.cat([rot, v[..., 2 * m:]], dim=-1)


def time_freqs(d: int, m_time_pairs: int = 16) -> torch.Tensor:
    """region.py:87-91 + iron_rope.py:26-33: exp(linspace(ln 1e-4, 0, m))."""
    m = max(0, min(d // 8 // 2, m_time_pairs))
    if m == 0:
        return torch.zeros(0)
    return torch.logspace(math.log(1.0 / 10000.0), 0.0, steps=m, base=math.e)


def noise_level(param: torch.Tensor) -> torch.Tensor:
    """region.py:100-126."""
    return F.softplus(param).clamp_min(EPS_DIV).clamp_max(MAX_NOISE)


# --------------------------------------------------------------------------- region cell
def region_step(
    p: Dict[str, torch.Tensor],
    st: Dict[str, torch.Tensor],
    x_in: torch.Tensor,
    step_pos: float,
    *,
    dt: int,
    afd: bool = False,
    rt: bool = False,
    use_pred: bool = True,
    kwta_k: int = 0,
    kwta_soft: bool = False,
    kwta_temp: float = 1.0,
    disable_kwta: bool = False,
) -> Tuple[torch.Tensor, Dict[str, torch.Tensor]]:
    """One RWKVRegionCell.step (region.py:168-249) on a single [d] vector.

    p: Wr,Wk,Wv,Wo [d,d]; norm_w, raw_decay [d]; W_time [m]; pn_param/on_param (scalar or [d]);
       rpn_param/ron_param scalars.
    st: num, den, var, pred [d]; radius, radius_var [1]; surprise scalar.
    Returns h and the new state (functional; st is not modified).
    """
    d = x_in.shape[-1]
    new = dict(st)
    if use_pred:
        x = rmsnorm(x_in + st["pred"], p["norm_w"])  # region.py:174
        new["pred"] = torch.zeros_like(st["pred"])   # region.py:175
    else:
        x = rmsnorm(x_in, p["norm_w"])
    # fast_forward(dt + 1), region.py:128-135,178
    steps = dt + 1
    rate = -F.softplus(p["raw_decay"]).clamp_min(-MAX_EXP)  # the clamp is a no-op (region.py:95)
    lam = torch.exp(rate * steps).clamp_min(EPS_DIV)
    num = st["num"] * lam
    den = st["den"] * lam
    var = st["var"]
    if afd:
        pn = noise_level(p["pn_param"])
        pn = pn.expand(d) if pn.numel() == 1 else pn
        var = var * lam.pow(2) + pn * steps
    r = torch.sigmoid(F.linear(x, p["Wr"]))
    k = F.linear(x, p["Wk"])
    v = F.linear(x, p["Wv"])
    if p["W_time"].numel() > 0:
        v = rotate_pairs(v, step_pos * p["W_time"])  # region.py:185-188
    w = torch.exp(k.clamp_max(MAX_EXP))
    if afd:  # region.py:192-212
        on = noise_level(p["on_param"])
        on = on.expand(d) if on.numel() == 1 else on
        prior = num / (den + EPS_DIV)
        resid = w * v - prior
        obs_var = torch.exp((-k).clamp_max(MAX_EXP)) * on + EPS_DIV
        gain = (var / (var + obs_var)).clamp(EPS_DIV, 1 - EPS_DIV)
        post = prior + gain * resid
        new["num"] = post * (den + w)
        new["den"] = den + w
        new["var"] = (1 - gain) * var
        surprise = (resid.pow(2) * (var + obs_var).reciprocal()).sum()
        new["surprise"] = 0.99 * st["surprise"] + (1 - 0.99) * surprise.detach()
        y = r * post
    else:  # region.py:214-216
        new["num"] = num + w * v
        new["den"] = den + w
        y = r * (new["num"] / (new["den"] + EPS_DIV))
    if rt:  # region.py:218-235
        nrm = y.norm(2, dim=-1, keepdim=True).clamp_min(EPS_DIV)
        direction = y / nrm
        new["last_dir"] = direction.detach()
        new["last_norm"] = nrm.detach()
        h_dir = F.linear(direction, p["Wo"])
        prior_var = st["radius_var"] + noise_level(p["rpn_param"])
        resid = nrm - st["radius"]
        gain = prior_var / (prior_var + noise_level(p["ron_param"]) + EPS_DIV)
        upd = st["radius"] + gain * resid
        new["radius_var"] = (1 - gain) * prior_var
        mu = 0.2
        new["radius"] = (1 - mu) * st["radius"] + mu * upd
        h = x + new["radius"] * h_dir
    else:
        h = x + F.linear(y, p["Wo"])
    if not disable_kwta:
        h = kwta(h, kwta_k, soft=kwta_soft, temp=kwta_temp)
    return h, new


def kwta_schedule(kwta_k: int, kwta_k_start: int, schedule: int, global_step: int,
                  soft_mode: bool, soft_temp: float) -> Tuple[int, bool, float]:
    """region.py:239-248 -> (k, soft, temp)."""
    warm = min(1.0, global_step / schedule) if schedule > 0 else 1.0
    k = int(round(kwta_k_start + (kwta_k - kwta_k_start) * warm))
    if soft_mode and warm < 1.0:
        return max(1, k), True, soft_temp * (1.0 - warm) + 1.0 * warm
    return max(1, k), False, 1.0


def predict_trace(pred: torch.Tensor, msg: torch.Tensor, alpha: float = 0.95) -> torch.Tensor:
    """region.py:140-146."""
    return alpha * pred + (1.0 - alpha) * msg


def region_phase(
    P: Dict[str, torch.Tensor],
    state: Dict[str, torch.Tensor],
    M: torch.Tensor,
    sensor: Optional[torch.Tensor],
    mask: torch.Tensor,
    step_pos: float,
    *,
    norm_in_w: Optional[torch.Tensor],
    sensor_idx: int,
    motor_idx: int,
    afd: bool = False,
    rt: bool = False,
    use_pred: bool = True,
    kwta_k: int = 0,
    kwta_soft: bool = False,
    kwta_temp: float = 1.0,
    disable_kwta: bool = False,
) -> Tuple[torch.Tensor, Dict[str, torch.Tensor]]:
    """The region loop of forward_inner_step (model.py:271-294) for S streams.

    P: W [R,4,d,d] (r,k,v,o), norm_w, raw_decay [R,d], W_time [m], pn_param/on_param [R] or [R,d],
       rpn_param/ron_param [R].
    state: num, den, var, pred [S,R,d]; radius, radius_var, surprise [S,R]; dt int [S,R].
    M [S,R,d], sensor [S,d] or None, mask bool [S,R].
    """
    S, R, d = M.shape
    out = {k: v.clone() for k, v in state.items()}
    rows: Dict[str, List[List[torch.Tensor]]] = {
        k: [[None] * R for _ in range(S)] for k in ("H", "num", "den", "var", "pred", "radius", "radius_var")
    }
    new_dt = state["dt"].clone()
    new_sur = state["surprise"].clone()
    for s in range(S):
        for r in range(R):
            cur = {
                "num": state["num"][s, r], "den": state["den"][s, r], "var": state["var"][s, r],
                "pred": state["pred"][s, r], "radius": state["radius"][s, r].reshape(1),
                "radius_var": state["radius_var"][s, r].reshape(1), "surprise": state["surprise"][s, r],
            }
            if not bool(mask[s, r]):  # model.py:277-282
                nxt = dict(cur)
                if use_pred:
                    nxt["pred"] = predict_trace(cur["pred"], M[s, r])
                new_dt[s, r] = state["dt"][s, r] + 1
                h = torch.zeros(d)
            else:
                x = M[s, r]
                if norm_in_w is not None:  # model.py:198-204
                    if sensor is not None and r in (sensor_idx, motor_idx):
                        x = x + sensor[s]
                    x = rmsnorm(x, norm_in_w)
                p = {
                    "Wr": P["W"][r, 0], "Wk": P["W"][r, 1], "Wv": P["W"][r, 2], "Wo": P["W"][r, 3],
                    "norm_w": P["norm_w"][r], "raw_decay": P["raw_decay"][r], "W_time": P["W_time"],
                }
                if afd:
                    p["pn_param"], p["on_param"] = P["pn_param"][r], P["on_param"][r]
                if rt:
                    p["rpn_param"], p["ron_param"] = P["rpn_param"][r], P["ron_param"][r]
                h, nxt = region_step(
                    p, cur, x, step_pos, dt=int(state["dt"][s, r]), afd=afd, rt=rt, use_pred=use_pred,
                    kwta_k=kwta_k, kwta_soft=kwta_soft, kwta_temp=kwta_temp, disable_kwta=disable_kwta,
                )
                new_dt[s, r] = 0
                new_sur[s, r] = nxt["surprise"].detach()
            rows["H"][s][r] = h
            for k in ("num", "den", "var", "pred"):
                rows[k][s][r] = nxt[k]
            rows["radius"][s][r] = nxt["radius"].reshape(())
            rows["radius_var"][s][r] = nxt["radius_var"].reshape(())
    stack = lambda key: torch.stack([torch.stack(rows[key][s]) for s in range(S)])
    H = stack("H")
    for k in ("num", "den", "var", "pred", "radius", "radius_var"):
        out[k] = stack(k)
    out["dt"] = new_dt
    out["surprise"] = new_sur
    return H, out


# --------------------------------------------------------------------------- wiring
def hex_axial_coords(R: int) -> torch.Tensor:
    """wiring.py:14-43: clockwise spiral of axial (q, r) cells, ring by ring from the east."""
    cells: List[Tuple[int, int]] = []
    ring = 0
    while len(cells) < R:
        if ring == 0:
            cells.append((0, 0))
        else:
            q, r = ring, 0
            for dq, dr in ((0, -1), (-1, 0), (-1, 1), (0, 1), (1, 0), (1, -1)):
                for _ in range(ring):
                    if len(cells) < R:
                        cells.append((q, r))
                    q, r = q + dq, r + dr
        ring += 1
    return torch.tensor(cells[:R], dtype=torch.float32).reshape(-1, 2)


def hex_neighbors(R: int) -> List[List[int]]:
    """wiring.py:46-67: neighbours in E, NE, NW, W, SW, SE order; missing ones omitted."""
    coords = [(int(q), int(r)) for q, r in hex_axial_coords(R).tolist()]
    index = {c: i for i, c in enumerate(coords)}
    out: List[List[int]] = []
    for q, r in coords:
        nb = []
        for dq, dr in ((1, 0), (1, -1), (0, -1), (-1, 0), (-1, 1), (0, 1)):
            j = index.get((q + dq, r + dr))
            if j is not None:
                nb.append(j)
        out.append(nb)
    return out


def edge_list(neighbors: Sequence[Sequence[int]]) -> Tuple[List[int], List[int]]:
    """gate.py:124-133: dst-major edge order; returns (src, dst)."""
    src, dst = [], []
    for r, nb in enumerate(neighbors):
        for s in nb:
            src.append(s)
            dst.append(r)
    return src, dst


def edge_directions(src_xy: torch.Tensor, dst_xy: torch.Tensor) -> torch.Tensor:
    """gate.py:344-350."""
    diff = dst_xy - src_xy
    dirs = diff.new_tensor([[1, 0], [1, -1], [0, -1], [-1, 0], [-1, 1], [0, 1]])
    return (diff.unsqueeze(1) == dirs.unsqueeze(0)).all(-1).float().argmax(-1).long()


# --------------------------------------------------------------------------- gate
def gate_scores(H: torch.Tensor, gain_ema: torch.Tensor, homeo: torch.Tensor, focus_any: bool,
                value_bias: float, u_mean: float, sensor_idx: int, motor_idx: int) -> torch.Tensor:
    """gate.py:42-68 for one stream: H [R,d] -> scores [R]."""
    R = H.shape[0]
    focus = torch.zeros(R)
    if focus_any:
        focus[sensor_idx] = 1.0
        focus[motor_idx] = 1.0
    return (1.0 * H.abs().mean(dim=-1) + 0.3 * gain_ema + 0.5 * focus - 0.2 * homeo
            + 0.5 * float(value_bias) + 0.3 * float(u_mean))


def nms_topk(scores: torch.Tensor, k: int, neighbors: Sequence[Sequence[int]],
             stable: bool = True) -> List[int]:
    """utils.py:107-124.  stable=True is the build's declared total order (score desc, index
    asc); stable=False is torch's default unstable argsort as the reference calls it."""
    order = torch.argsort(scores, descending=True, stable=stable).tolist()
    chosen: List[int] = []
    blocked = set()
    for i in order:
        if i in blocked:
            continue
        chosen.append(i)
        if len(chosen) >= k:
            break
        blocked.add(i)
        blocked.update(neighbors[i])
    return chosen


def gate_select(scores: torch.Tensor, k_active: int, burst: int, neighbors, sensor_idx: int,
                motor_idx: int, io_force_on: bool = True, stable: bool = True) -> torch.Tensor:
    """gate.py:70-87."""
    k_total = max(0, k_active + burst)
    mask = torch.zeros(scores.shape[0], dtype=torch.bool)
    if io_force_on:
        mask[[sensor_idx, motor_idx]] = True
        k_total = max(0, k_total - 2)
    mask[nms_topk(scores, k_total, neighbors, stable=stable)] = True
    return mask


# --------------------------------------------------------------------------- router
def fourier_edge_bias(coords: torch.Tensor, src: torch.Tensor, dst: torch.Tensor, W_reg: torch.Tensor,
                      beta_cos: torch.Tensor, beta_sin: torch.Tensor, fb_alpha: float = 0.1) -> torch.Tensor:
    """gate.py:284-290: per-edge multiplier 1 + alpha * b(delta)."""
    m = W_reg.shape[0]
    S = (coords[dst] - coords[src]) @ W_reg.T
    b = (torch.cos(S) * beta_cos + torch.sin(S) * beta_sin).sum(-1) * (1.0 / math.sqrt(m))
    return 1.0 + fb_alpha * b


def router_messages(
    P: Dict[str, torch.Tensor],
    H: torch.Tensor,
    mask_prev: torch.Tensor,
    coords: torch.Tensor,
    neighbors,
    *,
    mode: str = "per_edge",
    precision: bool = False,
    fb_alpha: float = 0.1,
) -> Tuple[torch.Tensor, torch.Tensor]:
    """Router.aggregate (gate.py:251-335) for one stream, written as the explicit loop over
    destinations (the reference's own loop form is gate.py:183-249).

    P: W_edge [E,d,d] (per_edge) | W_dir [6,d,d] (by_direction) | U,V [d,d] + w_edge [E,d] (factorized);
       key_W, raw_P (indexed like the message transform; per edge for factorized), query_W [R,d,d];
       W_reg [8,2], beta_cos, beta_sin [8].
    Returns (M [R,d], weights [E] with 0 on inactive edges).
    """
    R, d = H.shape
    src_l, dst_l = edge_list(neighbors)
    src, dst = torch.tensor(src_l, dtype=torch.long), torch.tensor(dst_l, dtype=torch.long)
    E = len(src_l)
    weights = torch.zeros(E)
    if E == 0:
        return torch.zeros(R, d), weights
    if fb_alpha != 0.0:
        bias = fourier_edge_bias(coords, src, dst, P["W_reg"], P["beta_cos"], P["beta_sin"], fb_alpha)
    else:
        bias = torch.ones(E)
    dirs = edge_directions(coords[src], coords[dst]) if mode == "by_direction" else None
    rows = []
    wlist: List[torch.Tensor] = [torch.zeros(()) for _ in range(E)]
    if not bool(mask_prev[src].any()):  # gate.py:259-260
        return torch.zeros(R, d), weights
    # gate.py:293: queries of ALL regions are formed (and so receive a gradient, possibly zero)
    Q = torch.stack([F.linear(H[r], P["query_W"][r]) for r in range(R)]) if precision else None
    e = 0
    for r in range(R):
        msgs, logws, ids = [], [], []
        q = Q[r] if precision else None
        for s in neighbors[r]:
            eid = e
            e += 1
            if not bool(mask_prev[s]):
                continue
            if mode == "per_edge":
                m = F.linear(H[s], P["W_edge"][eid])
                widx = eid
            elif mode == "by_direction":
                widx = int(dirs[eid])
                m = F.linear(H[s], P["W_dir"][widx])
            else:
                m = F.linear(F.linear(H[s], P["V"]) * P["w_edge"][eid], P["U"])
                widx = eid
            m = m * bias[eid]
            msgs.append(m)
            ids.append(eid)
            if precision:  # gate.py:292-311
                key = F.linear(H[s], P["key_W"][widx])
                score = (q * key).sum() / math.sqrt(d)
                mah = ((m - H[r]).pow(2) * F.softplus(P["raw_P"][widx])).sum().clamp_max(MAX_EXP / 0.5)
                logws.append(score - 0.5 * mah)
        if not msgs:
            rows.append(torch.zeros(d))
            continue
        if precision:  # gate.py:312-317 (segment softmax over incoming edges)
            lw = torch.stack(logws)
            mx = lw.max()
            logZ = mx + torch.exp(lw - mx).sum().clamp_min(EPS_DIV).log()
            w = torch.exp(lw - logZ)
        else:
            w = torch.ones(len(msgs))
        acc = torch.zeros(d)
        for wi, mi, eid in zip(w, msgs, ids):
            acc = acc + wi * mi
            wlist[eid] = wi
        rows.append(acc)
    return torch.stack(rows), torch.stack(wlist)


# --------------------------------------------------------------------------- training-side pieces
def ff_goodness(H: torch.Tensor) -> torch.Tensor:
    """training.py:127-128: mean of squares over the last dim."""
    return H.pow(2).mean(dim=-1)


def hebbian_updates(W_edge: torch.Tensor, neighbors, gains: Sequence[float], H_pos: torch.Tensor,
                    mask_pos: torch.Tensor) -> torch.Tensor:
    """training.py:147-156,240-243 applied to a stacked [E,d,d] copy of the edge weights."""
    W = W_edge.clone()
    eid_of = {}
    e = 0
    for r, nb in enumerate(neighbors):
        for s in nb:
            eid_of[(s, r)] = e
            e += 1
    B = H_pos.shape[0]
    for r, nb in enumerate(neighbors):
        g = float(gains[r])
        if not g > 0:
            continue
        for b in range(B):
            if not bool(mask_pos[b, r]):
                continue
            for s in nb:
                if not bool(mask_pos[b, s]):
                    continue
                W[eid_of[(s, r)]].add_(1e-3 * g * torch.outer(H_pos[b, r], H_pos[b, s]))
    return W
