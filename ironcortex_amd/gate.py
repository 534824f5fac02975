"""Gate (compute allocation) and Router (hex-neighbour message passing) with the reference's
API (gate.py:16-350), executed by the HIP library.

Router weights are Parameters under the reference's names (W_edge["s->r"].weight, key_lin, ...)
that are views into stacked [E,d,d] arenas the kernels index in place.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional

import torch
import torch.nn as nn

from .constants import EPS_DIV, EPS_LOG
from .flat import ArenaParameter, FlatParams
from .hip import ops
from .region import _Lin
from .utils import FastAttrs, neighbor_csr

_HEX_DIRS = ((1, 0), (1, -1), (0, -1), (-1, 0), (-1, 1), (0, 1))


class Gate(FastAttrs, nn.Module):
    """Discrete compute allocator over regions (gate.py:16-96)."""
    _plain_attrs = frozenset({"value_bias"})

    def __init__(self, R: int, neighbor_indices: List[List[int]], io_idxs: Dict[str, int]):
        super().__init__()
        self.R = R
        self.neighbors = neighbor_indices
        self.io_idxs = io_idxs
        self.register_buffer("gain_ema", torch.zeros(R))
        self.register_buffer("homeo", torch.zeros(R))
        self.w1, self.w2, self.w3, self.w4, self.w5, self.w6 = 1.0, 0.3, 0.5, 0.2, 0.5, 0.3
        self.value_bias = 0.0
        self._csr = None

    def _tables(self, device):
        if self._csr is None or self._csr[0].device != device:
            self._csr = neighbor_csr(self.neighbors, device)
        return self._csr

    def overlap_with_focus(self, focus_mask: Optional[torch.Tensor]) -> torch.Tensor:
        s = torch.zeros(self.R, device=self.gain_ema.device)
        if focus_mask is not None and not (focus_mask.numel() > 0 and not bool(focus_mask.any())):
            s[self.io_idxs["sensor"]] = 1.0
            s[self.io_idxs["motor"]] = 1.0
        return s

    def _focus_flag(self, focus_mask: Optional[torch.Tensor], device) -> Optional[torch.Tensor]:
        """Device-side `any focus bit set` (no host sync); None mask -> no boost (gate.py:44-47)."""
        if focus_mask is None:
            return None
        if focus_mask.numel() == 0:
            return torch.ones(1, dtype=torch.uint8, device=device)
        return focus_mask.any().to(torch.uint8).reshape(1).to(device)

    def score_select(self, H: torch.Tensor, focus_any: Optional[torch.Tensor], u_mean: float,
                     k_active: int, burst_extra_k: int, io_force_on: bool = True):
        """Fused score_regions + select_k for H [S,R,d] -> (scores [S,R], mask uint8 [S,R])."""
        off, nbr = self._tables(H.device)
        return ops.gate_scores_select(
            H, self.gain_ema, self.homeo, focus_any, off, nbr, self.io_idxs["sensor"], self.io_idxs["motor"],
            max(0, k_active + burst_extra_k), io_force_on, self.w5 * float(self.value_bias),
            self.w6 * float(u_mean), (self.w1, self.w2, self.w3, self.w4))

    def score_regions(self, H_hint: torch.Tensor, focus_mask: Optional[torch.Tensor], u_mean: float) -> torch.Tensor:
        """H_hint [R,d] -> scores [R] (gate.py:54-68)."""
        off, nbr = self._tables(H_hint.device)
        sc, _ = ops.gate_scores_select(
            H_hint.unsqueeze(0), self.gain_ema, self.homeo, self._focus_flag(focus_mask, H_hint.device), off, nbr,
            self.io_idxs["sensor"], self.io_idxs["motor"], 0, False, self.w5 * float(self.value_bias),
            self.w6 * float(u_mean), (self.w1, self.w2, self.w3, self.w4), select=False)
        return sc[0]

    def select_k(self, scores: torch.Tensor, k_active: int, burst_extra_k: int, io_force_on: bool = True) -> torch.Tensor:
        """Neighbour-suppressed top-k + forced sensor/motor (gate.py:70-87) -> bool [R]."""
        off, nbr = self._tables(scores.device)
        dummy = torch.zeros(1, self.R, 4, device=scores.device)
        _, mask = ops.gate_scores_select(
            dummy, self.gain_ema, self.homeo, None, off, nbr, self.io_idxs["sensor"], self.io_idxs["motor"],
            max(0, k_active + burst_extra_k), io_force_on, 0.0, 0.0, scores=scores.detach().reshape(1, -1).float())
        return mask[0].bool()

    def update_gain(self, r: int, goodness_gain: float, beta: float = 0.9):
        self.gain_ema[r] = beta * self.gain_ema[r] + (1.0 - beta) * float(goodness_gain)

    def update_homeo(self, reg_mask: torch.Tensor, eta: float = 1e-3, target: float = 0.1):
        self.homeo += eta * (reg_mask.float() - target)


class Router(FastAttrs, nn.Module):
    """Hex-graph message router with learned edge transforms and a Fourier relative bias."""
    _plain_attrs = frozenset({"_bias_cache", "_last_wgt", "hold_bias", "_tables", "_tables_key", "_geom"})

    def __init__(self, neighbor_indices: List[List[int]], d: int, R: int, *,
                 enable_precision_routed_messages: bool = False, vectorized: bool = True,
                 edge_transform_mode: str = "per_edge", _coord: Optional[ops.GradCoordinator] = None):
        super().__init__()
        self.R, self.d = R, d
        self.neighbors = neighbor_indices
        self.enable_precision_routed_messages = enable_precision_routed_messages
        self.vectorized = vectorized
        self.edge_transform_mode = edge_transform_mode
        src, dst, keys = [], [], []
        for r in range(R):  # dst-major edge order (gate.py:124-133)
            for s in neighbor_indices[r]:
                src.append(s)
                dst.append(r)
                keys.append(f"{s}->{r}")
        self.edge_keys = keys
        self.E = len(keys)
        self.register_buffer("src_idx", torch.tensor(src, dtype=torch.long))
        self.register_buffer("dst_idx", torch.tensor(dst, dtype=torch.long))
        mk = lambda: _Lin(d, d)
        if edge_transform_mode == "per_edge":
            self.W_edge = nn.ModuleDict({k: mk() for k in keys})
            self.key_lin = nn.ModuleDict({k: mk() for k in keys})
            self.raw_P_edge = nn.ParameterDict({k: ArenaParameter(torch.zeros(d)) for k in keys})
            groups = {"W": [self.W_edge[k].weight for k in keys], "key_W": [self.key_lin[k].weight for k in keys],
                      "raw_P": [self.raw_P_edge[k] for k in keys]}
            self.nW = self.E
        elif edge_transform_mode == "by_direction":
            self.W_dir = nn.ModuleList([mk() for _ in range(6)])
            self.key_lin_dir = nn.ModuleList([mk() for _ in range(6)])
            self.raw_P_dir = nn.ParameterList([ArenaParameter(torch.zeros(d)) for _ in range(6)])
            groups = {"W": [l.weight for l in self.W_dir], "key_W": [l.weight for l in self.key_lin_dir],
                      "raw_P": list(self.raw_P_dir)}
            self.nW = 6
        elif edge_transform_mode == "factorized":
            # msg_e = U (w_e * V h_src)   (gate.py:150-159,278-281): two SHARED d x d matrices, i.e.
            # plain [E_act,d] x [d,d] GEMMs -- run through rocBLAS (MFMA) from PyTorch, see _route_factorized
            self.U, self.V = mk(), mk()
            self.w_edge = nn.ParameterDict({k: ArenaParameter(torch.ones(d)) for k in keys})
            self.key_lin = nn.ModuleDict({k: mk() for k in keys})
            self.raw_P_edge = nn.ParameterDict({k: ArenaParameter(torch.zeros(d)) for k in keys})
            groups = {"U": [self.U.weight], "V": [self.V.weight], "w_e": [self.w_edge[k] for k in keys],
                      "key_W": [self.key_lin[k].weight for k in keys], "raw_P": [self.raw_P_edge[k] for k in keys]}
            self.nW = self.E
        else:
            raise ValueError(f"unknown edge_transform_mode {edge_transform_mode}")
        self.query_lin = nn.ModuleList([mk() for _ in range(R)])
        groups["query_W"] = [l.weight for l in self.query_lin]
        self._flat = FlatParams(groups)
        self.fb_alpha = 0.1
        m_reg = 8
        self.register_buffer("W_reg", torch.randn(m_reg, 2) * 1.0)  # gaussian bank (iron_rope.py:34-35)
        self.beta_cos = nn.Parameter(torch.zeros(m_reg))
        self.beta_sin = nn.Parameter(torch.zeros(m_reg))
        nn.init.normal_(self.beta_cos, std=0.01)
        nn.init.normal_(self.beta_sin, std=0.01)
        self.fb_scale = 1.0 / math.sqrt(m_reg)
        self.last_weight_mean: float = 0.0
        self.last_weight_entropy: float = 0.0
        self._last_wgt: Optional[torch.Tensor] = None
        self._tables: Optional[Dict[str, torch.Tensor]] = None
        self._tables_key = None
        self._geom = None
        self._coord = _coord or ops.GradCoordinator()
        self._coord.register_flagged(self._flat.begin_pass, self._launch_wgrad, self._publish)
        self._tape: Optional[ops.Tape] = None
        self._tape_S = 0
        self._anchor = None
        self._own_coord = _coord is None

    # ------------------------------------------------------------------ plumbing
    def _apply(self, fn, recurse=True):
        super()._apply(fn, recurse)
        self._flat.reflat()
        self._tables = None
        self._geom = None
        self._tape = None
        self._anchor = None
        return self

    def _edge_tables(self, reg_coords: torch.Tensor) -> Dict[str, torch.Tensor]:
        dev = self.src_idx.device
        key = (dev, reg_coords.data_ptr(), reg_coords._version)
        if self._tables is not None and self._tables_key == key:
            return self._tables
        R, E = self.R, self.E
        src, dst = self.src_idx.tolist(), self.dst_idx.tolist()
        dst_off = [0] * (R + 1)
        for r in dst:
            dst_off[r + 1] += 1
        for r in range(R):
            dst_off[r + 1] += dst_off[r]
        out_edges: List[List[int]] = [[] for _ in range(R)]
        for e, s in enumerate(src):
            out_edges[s].append(e)
        src_off, src_edges = [0], []
        for r in range(R):
            src_edges.extend(out_edges[r])
            src_off.append(len(src_edges))
        if self.edge_transform_mode == "by_direction":
            rc = reg_coords.to(dev)
            widx = self._edge_directions(rc[self.src_idx], rc[self.dst_idx]).tolist()
        else:
            widx = list(range(E))
        deg = max([dst_off[r + 1] - dst_off[r] for r in range(R)] + [0])
        if deg > 8:
            raise ValueError("Router: at most 8 incoming edges per region are supported")
        i32 = lambda x: torch.tensor(x if len(x) else [0], dtype=torch.int32, device=dev)
        # shared matrices as ONE GEMM (IC_RF_SHARED): needs every destination's incoming edges -- and, for the input
        # gradient's GEMM over the senders, every sender's outgoing edges -- to use distinct matrices (true on a hex
        # map: one neighbour per direction); otherwise the per-edge GEMV kernels
        self._shared_gemm = (self.edge_transform_mode == "by_direction" and E > 0 and all(
            len({widx[e] for e in range(dst_off[r], dst_off[r + 1])}) == dst_off[r + 1] - dst_off[r] for r in range(R))
            and all(len({widx[e] for e in out_edges[r]}) == len(out_edges[r]) for r in range(R)))
        self._tables = {"dst_off": i32(dst_off), "src": i32(src), "widx": i32(widx), "src_off": i32(src_off),
                        "src_edges": i32(src_edges), "dst_of_edge": i32(dst),
                        "widx_long": torch.tensor(widx if E else [0], dtype=torch.long, device=dev)}
        if E > 0 and len(set(widx)) != E:  # shared parameters: the edges of every parameter, in edge order, padded with E
            groups = [[e for e in range(E) if widx[e] == k] for k in range(self.nW)]
            width = max(len(g_) for g_ in groups)
            self._tables["edges_by_w"] = torch.tensor([g_ + [E] * (width - len(g_)) for g_ in groups], dtype=torch.long, device=dev)
        if self.edge_transform_mode == "factorized":  # identity edge tables of the two shared-matrix row GEMMs
            self._tables.update(iota=i32(list(range(R + 1))), zero=i32([0] * R), ones=torch.ones(R, device=dev))
        self._tables_key = key
        return self._tables

    def spec(self) -> ops.RouterSpec:
        """Static description for the C calls (the edge tables must have been built: _edge_tables)."""
        return ops.RouterSpec(self.R, self.d, self.E, self.nW, self.enable_precision_routed_messages,
                              shared_gemm=bool(getattr(self, "_shared_gemm", False)), factorized=self.fact_lib)

    @property
    def fact_lib(self) -> bool:
        """Factorized mode on the library (IC_RF_FACTORIZED): without precision routing, where the messages of a
        destination are summed and U applies once per destination.  With precision routing every edge's message is
        needed on its own: that combination stays on _route_factorized (general path only)."""
        return self.edge_transform_mode == "factorized" and not self.enable_precision_routed_messages and self.E > 0

    def _edge_bias(self, reg_coords: torch.Tensor) -> torch.Tensor:
        """1 + fb_alpha * b(delta) per edge (gate.py:284-290); trig of the static geometry is cached."""
        dev = self.src_idx.device
        if self.fb_alpha == 0.0 or self.E == 0:
            return torch.ones(max(self.E, 1), device=dev)[: self.E]
        key = (dev, reg_coords.data_ptr(), reg_coords._version, self.W_reg._version)
        if self._geom is None or self._geom[0] != key:
            coords = reg_coords.to(dev)
            S = (coords[self.dst_idx] - coords[self.src_idx]) @ self.W_reg.T
            self._geom = (key, torch.cos(S), torch.sin(S))
        _, c, s = self._geom
        b = (c * self.beta_cos + s * self.beta_sin).sum(-1) * self.fb_scale
        return 1.0 + self.fb_alpha * b

    def begin_loop(self) -> None:
        """Inside one reasoning loop the bias parameters are fixed: compute 1 + alpha*b once.
        With hold_bias set (train_step: all loops of a step feed ONE backward pass) the value is
        kept from loop to loop until a parameter version changes."""
        if not (getattr(self, "hold_bias", False) and isinstance(getattr(self, "_bias_cache", None), tuple)):
            self._bias_cache = "armed"

    def end_loop(self) -> None:
        if not getattr(self, "hold_bias", False):
            self._bias_cache = None

    def _cached_bias(self, reg_coords: torch.Tensor) -> torch.Tensor:
        c = getattr(self, "_bias_cache", None)
        if c is None:
            return self._edge_bias(reg_coords)
        key = (torch.is_grad_enabled(), self.fb_alpha, self.beta_cos._version, self.beta_sin._version)
        if c == "armed" or c[0] != key:
            c = (key, self._edge_bias(reg_coords))
            self._bias_cache = c
        return c[1]

    def _params(self) -> Dict[str, torch.Tensor]:
        f = self._flat
        if self.edge_transform_mode == "factorized":
            return {"U": f.stacked("U"), "V": f.stacked("V"), "w_e": f.stacked("w_e"), "key_W": f.stacked("key_W"),
                    "raw_P": f.stacked("raw_P"), "query_W": f.stacked("query_W")}
        return {"W": f.stacked("W"), "key_W": f.stacked("key_W"), "raw_P": f.stacked("raw_P"),
                "query_W": f.stacked("query_W")}

    def _grad_sink(self, name: str, rows: torch.Tensor, widx: torch.Tensor) -> None:
        """Per-edge gradient rows [E, d] into the stacked gradient of the parameters they index through widx.  One
        parameter per edge: a scatter without collisions.  Shared parameters (by_direction: raw_P of a direction is
        used by every edge of that direction): a gather of each parameter's edges in edge order and a sum over them --
        index_add_ would add colliding rows with atomics, in an order that changes from run to run."""
        by_w = self._tables.get("edges_by_w") if self._tables else None
        if by_w is None or rows.shape[0] != self.E:
            self._flat.stacked_grad(name).index_add_(0, widx, rows)
            return
        padded = torch.cat([rows, rows.new_zeros(1, rows.shape[1])])  # (index E: the empty places of the table)
        self._flat.stacked_grad(name).add_(padded[by_w].sum(1))

    def _launch_wgrad(self):
        """End of a backward pass, phase 1: the deferred dW / dKey / dQuery launches; returns the device-side
        touched flags [nW (+ nW + R with precision routing)] for the GradCoordinator's single host read."""
        tape = self._tape
        if tape is None or tape.pos == 0:
            return None
        dev = self._flat.flat.device
        spec = self.spec()
        tables = self._tables
        if spec.factorized:  # dU, dV (shared-matrix GEMMs over the tapes' rows); w_e: the edges active in a visited slot
            tuv = torch.zeros(2, dtype=torch.uint8, device=dev)
            ops.router_fact_wgrad(tape, spec, tables, self._tape_S, self._flat.stacked_grad("U"), self._flat.stacked_grad("V"), tuv)
            n = tape.pos
            te = (tape.buf["eact"][:n] * tape.visited[:n].view(n, 1, 1)).amax((0, 1)).to(torch.uint8)
            tape.reset()
            return torch.cat([tuv, te])
        tw = torch.zeros(self.nW, dtype=torch.uint8, device=dev)
        ops.router_wgrad(tape, spec, tables, self._tape_S, 0, self._flat.stacked_grad("W"), tw)
        if spec.precision:
            tk = torch.zeros(self.nW, dtype=torch.uint8, device=dev)
            tq = torch.zeros(self.R, dtype=torch.uint8, device=dev)
            ops.router_wgrad(tape, spec, tables, self._tape_S, 1, self._flat.stacked_grad("key_W"), tk)
            ops.router_wgrad(tape, spec, tables, self._tape_S, 2, self._flat.stacked_grad("query_W"), tq)
            tw = torch.cat([tw, tk, tq])
        tape.reset()
        return tw

    def _publish(self, flags) -> None:
        """Phase 2: .grad views for the touched matrices."""
        if self.fact_lib:
            self._flat.publish("U", flags[0:1])
            self._flat.publish("V", flags[1:2])
            self._flat.publish("w_e", flags[2:])
            return
        tw_l = flags[: self.nW]
        if self.enable_precision_routed_messages:
            tk_l, tq_l = flags[self.nW: 2 * self.nW], flags[2 * self.nW:]
            if any(tw_l):  # the reference stacks the queries of ALL regions (gate.py:293): zero grads, not None
                tq_l = [True] * self.R
            self._flat.publish("key_W", tk_l)
            self._flat.publish("raw_P", tk_l)
            self._flat.publish("query_W", tq_l)
        self._flat.publish("W", tw_l)

    def _get_tape(self, S: int, device) -> ops.Tape:
        if self._tape is None or self._tape_S != S:
            R, E, d = self.R, max(self.E, 1), self.d
            fields = {"gmsg": (S, E, d), "H": (S, R, d), "eact": (S, E)}
            dtypes = {"eact": torch.uint8}
            if self.fact_lib:  # rows and operands of dU / dV instead of per-edge message gradients
                fields = {"gU": (S, R, d), "zU": (S, R, d), "aU": (S, R), "gV": (S, R, d), "aV": (S, R), "H": (S, R, d), "eact": (S, E)}
                dtypes = {"eact": torch.uint8, "aU": torch.uint8, "aV": torch.uint8}
            if self.enable_precision_routed_messages:
                fields.update(gkey=(S, E, d), gq=(S, R, d), qact=(S, R))
                dtypes["qact"] = torch.uint8
            self._tape, self._tape_S = ops.Tape(fields, dtypes), S
        return self._tape

    # ------------------------------------------------------------------ execution
    def route(self, H: torch.Tensor, mask_prev: torch.Tensor, reg_coords: torch.Tensor) -> torch.Tensor:
        """Batched entry: H [S,R,d], mask_prev uint8 [S,R] -> M [S,R,d]."""
        S = H.shape[0]
        dev = H.device
        if self._flat.flat.device != dev:
            raise RuntimeError("Router parameters and input are on different devices")
        if self.edge_transform_mode == "factorized" and not self.fact_lib:
            return torch.stack([self._route_factorized(H[s], mask_prev[s].bool(), reg_coords) for s in range(S)])
        tables = self._edge_tables(reg_coords)
        ebias = self._cached_bias(reg_coords)
        spec = self.spec()
        box = dict(spec=spec, tables=tables, P=self._params(), mask_prev=mask_prev.contiguous(),
                   tape=self._get_tape(S, dev), coord=self._coord, grad_sink=self._grad_sink)
        if torch.is_grad_enabled():
            if self._anchor is None or self._anchor.device != dev:
                self._anchor = torch.zeros(1, device=dev, requires_grad=True)
            anchor = self._anchor
        else:
            anchor = None
        M = ops.RouterFn.apply(anchor, H, ebias, box)
        self._last_wgt = box.get("wgt")
        return M

    def _route_factorized(self, H: torch.Tensor, mask_prev: torch.Tensor, reg_coords: torch.Tensor) -> torch.Tensor:
        """Factorised edge transforms for one stream (gate.py:251-335 with :278-281).  The shared
        U and V make this the one GEMM-shaped mode: [E_act,d] x [d,d] twice, on rocBLAS.  Only the
        ACTIVE edges' parameters enter the graph (one host read of the active set, as the
        reference does), so inactive edges keep grad None."""
        R, d, dev = self.R, self.d, H.device
        active = mask_prev[self.src_idx].nonzero().flatten().tolist()
        self._last_wgt = torch.zeros(1, max(self.E, 1), device=dev)[:, : self.E]
        if not active:
            return torch.zeros(R, d, device=dev)
        act = torch.tensor(active, device=dev)
        src, dst = self.src_idx[act], self.dst_idx[act]
        keys = [self.edge_keys[e] for e in active]
        scale = torch.stack([self.w_edge[k] for k in keys])
        msg = self.U(self.V(H[src]) * scale)
        if self.fb_alpha != 0.0:
            msg = msg * self._cached_bias(reg_coords)[act].unsqueeze(-1)
        weight = None
        if self.enable_precision_routed_messages:
            q = torch.stack([self.query_lin[r](H[r]) for r in range(R)])
            Wk = torch.stack([self.key_lin[k].weight for k in keys])
            kv = torch.bmm(Wk, H[src].unsqueeze(-1)).squeeze(-1)
            prec = torch.nn.functional.softplus(torch.stack([self.raw_P_edge[k] for k in keys]))
            logw = (q[dst] * kv).sum(-1) / math.sqrt(d) - 0.5 * ((msg - H[dst]).pow(2) * prec).sum(-1).clamp_max(80.0)
            top = torch.full((R,), float("-inf"), device=dev).scatter_reduce(0, dst, logw, reduce="amax")
            z = torch.zeros(R, device=dev).scatter_add(0, dst, torch.exp(logw - top[dst]))
            weight = torch.exp(logw - (top + z.clamp_min(EPS_DIV).log())[dst])
            self._last_wgt = torch.zeros(1, self.E, device=dev).index_put((torch.zeros_like(act), act), weight.detach())
            msg = weight.unsqueeze(-1) * msg
        return torch.zeros(R, d, device=dev).index_add(0, dst, msg)

    def messages(self, H: torch.Tensor, reg_mask_prev: torch.Tensor, reg_coords: torch.Tensor) -> torch.Tensor:
        """H [R,d], bool [R] -> M [R,d] (gate.py:337-342); both the vectorised and the loop
        form of the reference map to the same kernels."""
        M = self.route(H.unsqueeze(0), reg_mask_prev.to(torch.uint8).reshape(1, -1), reg_coords)[0]
        if self.enable_precision_routed_messages:
            self._weight_stats()
        else:
            self.last_weight_mean, self.last_weight_entropy = 0.0, 0.0
        return M

    aggregate = messages
    _messages_loop = messages

    @property
    def last_weights(self) -> Dict[str, float]:
        """edge key -> routing weight of the last call (gate.py:318); materialised on demand."""
        if self._last_wgt is None or not self.enable_precision_routed_messages:
            return {}
        w = self._last_wgt[0].tolist()
        return {k: w[e] for e, k in enumerate(self.edge_keys) if w[e] != 0.0}

    def _weight_stats(self) -> None:
        """last_weight_mean / entropy over the active edges (gate.py:326-334)."""
        w = self._last_wgt[0] if self._last_wgt is not None else None
        if w is None or w.numel() == 0:
            self.last_weight_mean, self.last_weight_entropy = 0.0, 0.0
            return
        act = w > 0
        n = act.sum().clamp_min(1)
        mean = (w * act).sum() / n
        p = w / w.sum().clamp_min(EPS_DIV)
        ent = (-(p + EPS_LOG).log() * p * act).sum()
        both = torch.stack([mean, ent]).tolist()
        self.last_weight_mean, self.last_weight_entropy = float(both[0]), float(both[1])

    @staticmethod
    def _edge_directions(src: torch.Tensor, dst: torch.Tensor) -> torch.Tensor:
        """Axial coordinate difference -> hex direction 0..5 (gate.py:344-350)."""
        diff = dst - src
        dirs = diff.new_tensor(_HEX_DIRS)
        return (diff.unsqueeze(1) == dirs.unsqueeze(0)).all(-1).float().argmax(-1).long()
