#!/usr/bin/env python3
"""by_direction + precision routing, forward only: the per-edge precision kernels (messages, keys and queries in one
launch) against the two shared-matrix GEMMs over the (stream, edge) pairs + the queries launch.  R=512, d=256, k_active
of the regions on per stream.  Run once as is and once with IC_SHARED_GEMM_MIN=1000000000 (never the GEMMs).
One JSON line per point: microseconds per ic_router_fwd call by kernel (HIP events around every launch)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ironcortex_amd import hip  # noqa: E402
from ironcortex_amd.gate import Router  # noqa: E402
from ironcortex_amd.wiring import hex_axial_coords, hex_neighbors  # noqa: E402

dev = torch.device("cuda:0")
d, R = 256, 512
torch.manual_seed(0)
router = Router(hex_neighbors(R), d, R, edge_transform_mode="by_direction", enable_precision_routed_messages=True).to(dev)
coords = hex_axial_coords(R).to(dev)
for k_act, Ss in ((32, (8, 16, 64)), (128, (8, 64))):
    for S in Ss:
        mask = torch.zeros(S, R, dtype=torch.uint8, device=dev)
        for s in range(S):
            mask[s, torch.randperm(R, device=dev)[:k_act]] = 1
        H = torch.randn(S, R, d, device=dev) * mask.unsqueeze(-1)
        with torch.no_grad():
            for _ in range(5):
                router.route(H, mask, coords)
            torch.cuda.synchronize()
            hip.prof_select("")
            hip.prof_enable(True)
            hip.prof_read()
            n = 30
            for _ in range(n):
                router.route(H, mask, coords)
            torch.cuda.synchronize()
            prof = hip.prof_read()
            hip.prof_enable(False)
        us = {k: round(ms / n * 1e3, 2) for k, (c, ms) in prof.items()}
        print(json.dumps({"IC_SHARED_GEMM_MIN": os.environ.get("IC_SHARED_GEMM_MIN"), "R": R, "k_active": k_act, "S": S,
                          "pairs": S * router.E, "us_per_call": us, "total_us": round(sum(us.values()), 2)}), flush=True)
