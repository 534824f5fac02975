#!/usr/bin/env python3
"""Kernel roofline sweep (BASELINE.json configs[2]): region phase + router forward/backward at
R in {64..512}, k_active in {4..32}, S streams, d=256, synthetic inputs.
Prints one JSON line per point: per-kernel average duration (HIP events on the launch stream),
algorithmic bytes and achieved GB/s against the 8 TB/s HBM peak."""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import HBM_PEAK_GBS, algorithmic_bytes  # noqa: E402
from ironcortex_amd import hip  # noqa: E402
from ironcortex_amd.gate import Router  # noqa: E402
from ironcortex_amd.region import RegionBank, RWKVRegionCell  # noqa: E402
from ironcortex_amd.wiring import hex_axial_coords, hex_neighbors  # noqa: E402


def mask_sets(S, R, k, rotate):
    """rotate mask sets [S,R]: stream s of set i takes entries i*k..(i+1)*k-1 (cyclically) of its own seeded
    permutation of the regions, so a region's weights come back only after every other region's have
    been streamed -- with R = 512 (regions 512 MiB + W_edge 728 MiB) nothing the next iteration needs
    is left in the 256 MiB Infinity Cache.  rotate = 1: the same active set every iteration."""
    g = torch.Generator().manual_seed(0)
    perms = [torch.randperm(R, generator=g) for _ in range(S)]
    sets = []
    for i in range(max(1, rotate)):
        m = torch.zeros(S, R, dtype=torch.uint8)
        for s in range(S):
            idx = [(i * k + j) % R for j in range(k)]
            m[s, perms[s][idx]] = 1
        sets.append(m)
    return sets


_FLUSH = {}


def flush_before_backward(on: bool) -> None:
    """The backward of this harness goes through the general autograd path, which writes ~40 MB of gradient
    tensors and zero-filled row buffers right before every library launch sequence.  Those dirty lines are
    written back while the first weight-streaming kernels read (measured at R=512, k=32, S=8: region_rkv_bwd
    63.7 us with them, 35.5 us without; region_out_bwd 29 -> 20 us), which prices the harness, not the
    kernel (the LoopEngine zero-fills one block per pass).  With the hook on, a 1 GiB read is queued before
    each backward entry point: the dirty lines are written back during that read and the kernels start
    from clean -- and certainly weight-free -- caches."""
    from ironcortex_amd.hip import ops
    if "orig" not in _FLUSH:
        _FLUSH["orig"] = ops._call
    if not on:
        ops._call = _FLUSH["orig"]
        return

    def call(name, a, *rest, **kw):
        if name in ("ic_region_phase_bwd", "ic_router_bwd", "ic_region_wgrad", "ic_router_wgrad"):
            if "buf" not in _FLUSH:
                _FLUSH["buf"] = torch.zeros(1 << 28, device="cuda")
            _FLUSH["buf"].sum()
        return _FLUSH["orig"](name, a, *rest, **kw)

    ops._call = call


MARK_BASE = 1000  # marker launches (k_kwta_rows with 4 * (MARK_BASE + id) rows): id = point index before the timed
                  # iterations of a point, 4000 + index after them; scripts/sweep_grid.py splits a kernel trace by them


def mark(idx: int) -> None:
    from ironcortex_amd.utils import KWTA
    with torch.no_grad():
        KWTA(torch.zeros(4 * (MARK_BASE + idx), 8, device="cuda"), 2)


def sweep(Rs, ks, Ss, d=256, iters=20, backward=False, rotate=1, events=True, markers=False):
    """Yields one dict per (R, S, k) point.  events=False: no HIP-event bracketing (the run is being
    traced by rocprofv3 instead); the row then only carries the algorithmic bytes."""
    dev = torch.device("cuda", torch.cuda.current_device())
    for R in Rs:
        torch.manual_seed(0)
        nb, coords = hex_neighbors(R), hex_axial_coords(R).to(dev)
        cells = [RWKVRegionCell(d, _shared_bank=True) for _ in range(R)]
        bank = RegionBank(cells, 0, R - 1)
        for c in cells:
            c.to(dev)
        bank.to_device()
        router = Router(nb, d, R, enable_precision_routed_messages=PRECISION).to(dev)
        nin = torch.ones(d, device=dev)
        for S in Ss:
            for k in ks:
                masks = [m.to(dev) for m in mask_sets(S, R, k, rotate)]
                Hs = [torch.randn(S, R, d, device=dev) * m.unsqueeze(-1) for m in masks]
                sensor = torch.randn(S, d, device=dev)
                bank.reset_state(dev, S)
                bank.num, bank.den = torch.rand(S, R, d, device=dev), torch.rand(S, R, d, device=dev) + 0.5
                bank.pred = 0.1 * torch.randn(S, R, d, device=dev)
                bank.dt = torch.randint(0, 8, (S, R), dtype=torch.int32, device=dev)
                state0 = (bank.num, bank.den, bank.pred, bank.dt)

                def run(grad, i=0):
                    mask, H = masks[i % len(masks)], Hs[i % len(masks)]
                    bank.num, bank.den, bank.pred, bank.dt = state0
                    if grad:
                        Hg = H.clone().requires_grad_(True)
                        M = router.route(Hg, mask, coords)
                        out = bank.phase(M, sensor, nin, mask, 0.5)
                        (out.sum() + bank.num.sum()).backward()
                        for p in list(router.parameters()) + [p for c in cells for p in c.parameters()]:
                            p.grad = None
                    else:
                        with torch.no_grad():
                            M = router.route(H, mask, coords)
                            bank.phase(M, sensor, nin, mask, 0.5)

                for i in range(3):
                    run(backward, i)
                torch.cuda.synchronize()
                if events:
                    hip.prof_select("")
                    hip.prof_enable(True)
                    hip.prof_read()
                if markers:
                    mark(_POINT[0])
                for i in range(iters):
                    run(backward, 3 + i)
                if markers:
                    mark(4000 + _POINT[0])
                    _POINT[0] += 1
                torch.cuda.synchronize()
                prof = hip.prof_read() if events else {}
                hip.prof_enable(False)
                n_sets = float(len(masks))  # per-launch averages over the rotation
                uniq = sum(int(m.any(0).sum()) for m in masks) / n_sets
                e_uniq = sum(int(m.bool()[:, router.src_idx].any(0).sum()) for m in masks) / n_sets
                e_pairs = sum(int(m.bool()[:, router.src_idx].sum()) for m in masks) / n_sets
                q_uniq = sum(int(torch.zeros(R, device=dev).index_add_(0, router.dst_idx, m.bool()[:, router.src_idx].any(0).float()).gt(0).sum())
                             for m in masks) / n_sets
                pairs = sum(int(m.sum()) for m in masks) / n_sets
                st = dict(u_f=uniq, n_f=pairs, u_b=uniq, n_b=pairs, e=e_uniq, ep=e_pairs, q=q_uniq, R=R, S=S, afd=False, prec=PRECISION,
                          fsplit=(os.environ.get("IC_FWD_SPLIT") or ("1" if S >= 32 else "0"))[0] != "0")
                row = {"R": R, "S": S, "k": k, "d": d, "rotate": len(masks), "backward": backward,
                       "unique_regions": uniq, "unique_edges": e_uniq, "pairs": pairs, "edge_pairs": e_pairs,
                       "weights_MiB": {"regions": R * 4 * d * d * 4 / 2**20, "W_edge": router.E * d * d * 4 / 2**20},
                       "kernels": {}}
                for name, (n, ms) in sorted(prof.items(), key=lambda kv: -kv[1][1]):
                    us = ms / n * 1e3
                    ab = algorithmic_bytes(name, d, st)
                    row["kernels"][name] = {"avg_us": round(us, 2), "alg_MB": round(ab / 1e6, 3),
                                            "GBs": round(ab / (us * 1e-6) / 1e9, 1) if ab else None,
                                            "frac": round(ab / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4) if ab else None}
                if not events:  # priced for the summariser that reads rocprofv3's tables (scripts/sweep_profile.py)
                    row["alg_bytes"] = {name: algorithmic_bytes(name, d, st) for name in SWEEP_KERNELS}
                yield row
        del bank, router, cells
        torch.cuda.empty_cache()


_POINT = [0]
PRECISION = False  # --precision: precision-routed messages (key / query GEMVs + softmax combine)
SWEEP_KERNELS = ("region_pro", "region_rkv", "region_state", "region_out", "region_fin", "router_fwd", "router_edge_fwd", "router_combine", "region_out_bwd", "region_rkv_bwd",
                 "region_mid_bwd", "region_pro_bwd", "router_mat_bwd", "router_gather_bwd", "wgrad_region", "wgrad_edge")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--R", type=int, nargs="+", default=[64, 128, 256, 512])
    ap.add_argument("--k", type=int, nargs="+", default=[4, 8, 16, 32])
    ap.add_argument("--S", type=int, nargs="+", default=[1, 8])
    ap.add_argument("--d", type=int, default=256)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--backward", action="store_true")
    ap.add_argument("--rotate", type=int, default=1, help="cycle through this many disjoint active sets")
    ap.add_argument("--no-events", action="store_true", help="no HIP-event bracketing (run under rocprofv3)")
    ap.add_argument("--precision", action="store_true")
    ap.add_argument("--flush-bwd", action="store_true", help="1 GiB read before every backward entry point (see flush_before_backward)")
    ap.add_argument("--markers", action="store_true", help="marker launches around every point's timed iterations (scripts/sweep_grid.py)")
    args = ap.parse_args()
    global PRECISION
    PRECISION = args.precision
    flush_before_backward(args.flush_bwd)
    for row in sweep(args.R, args.k, args.S, args.d, args.iters, args.backward, args.rotate, not args.no_events, args.markers):
        print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
