#!/usr/bin/env python3
"""by_direction forward at small column counts: the shared-matrix GEMM against the per-edge kernels (IC_SHARED_GEMM_MIN
decides in the library; run once with IC_SHARED_GEMM_MIN=0 and once with a huge value).  Forward only, k_active
regions on per stream, d = 256.  One JSON line per point: microseconds per ic_router_fwd call by kernel
(HIP events around every launch)."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ironcortex_amd import hip  # noqa: E402
from ironcortex_amd.gate import Router  # noqa: E402
from ironcortex_amd.wiring import hex_axial_coords, hex_neighbors  # noqa: E402

dev = torch.device("cuda:0")
d = 256
POINTS = ((36, 8, (1, 4, 16, 64)), (64, 8, (1, 4, 16, 64)), (512, 32, (1, 2, 4, 8)))
if os.environ.get("IC_BYDIR_LARGE") == "1":
    POINTS = ((64, 8, (128, 256, 512)), (512, 32, (16, 32, 64)), (512, 128, (8, 16, 64)))
for R, k_act, Ss in POINTS:
    torch.manual_seed(0)
    router = Router(hex_neighbors(R), d, R, edge_transform_mode="by_direction").to(dev)
    coords = hex_axial_coords(R).to(dev)
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
            n = 50
            for _ in range(n):
                router.route(H, mask, coords)
            torch.cuda.synchronize()
            prof = hip.prof_read()
            hip.prof_enable(False)
        print(json.dumps({"IC_SHARED_GEMM_MIN": os.environ.get("IC_SHARED_GEMM_MIN"), "R": R, "k_active": k_act, "S": S, "columns": S * R,
                          "us_per_call": {k: round(ms / n * 1e3, 2) for k, (c, ms) in prof.items()}}), flush=True)
