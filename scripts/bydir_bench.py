#!/usr/bin/env python3
"""edge_transform_mode="by_direction" router (the MFMA GEMM path, csrc/router_mfma.hip): forward and
backward at R regions, S streams, d=256, a fraction `p` of the regions active per stream.  Prints one
JSON line per point: per-kernel average duration (HIP events), the GEMM's flops and achieved TFLOP/s
against the dense fp32 matrix peak.  Run under `rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES
GRBM_GUI_ACTIVE -- python3 scripts/bydir_bench.py --no-events` for the MFMA-busy counters."""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ironcortex_amd import hip  # noqa: E402
from ironcortex_amd.gate import Router  # noqa: E402
from ironcortex_amd.wiring import hex_axial_coords, hex_neighbors  # noqa: E402

FP32_MFMA_PEAK_TFLOPS = 157.3  # 256 CUs x 4 SIMDs x 64 flop/cycle x 2.4 GHz (v_mfma_f32_32x32x2_f32: 4096 flop / 64 cycles)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--R", type=int, default=512)
    ap.add_argument("--S", type=int, nargs="+", default=[8, 64])
    ap.add_argument("--d", type=int, default=256)
    ap.add_argument("--p", type=float, default=0.5)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--no-events", action="store_true")
    args = ap.parse_args()
    dev = torch.device("cuda", torch.cuda.current_device())
    R, d = args.R, args.d
    torch.manual_seed(0)
    router = Router(hex_neighbors(R), d, R, edge_transform_mode="by_direction").to(dev)
    coords = hex_axial_coords(R).to(dev)
    for S in args.S:
        mask = (torch.rand(S, R, device=dev) < args.p).to(torch.uint8)
        H = torch.randn(S, R, d, device=dev) * mask.unsqueeze(-1)

        def run():
            Hg = H.clone().requires_grad_(True)
            M = router.route(Hg, mask, coords)
            M.sum().backward()
            for p_ in router.parameters():
                p_.grad = None

        for _ in range(3):
            run()
        torch.cuda.synchronize()
        if not args.no_events:
            hip.prof_select("")
            hip.prof_enable(True)
            hip.prof_read()
        for _ in range(args.iters):
            run()
        torch.cuda.synchronize()
        prof = hip.prof_read() if not args.no_events else {}
        hip.prof_enable(False)
        n_cols = S * R
        flops_fwd = 2.0 * d * (router.nW * d) * n_cols              # [d, 6d] x [6d, S*R], zero columns included
        pairs = float(mask.bool()[:, router.src_idx].sum())
        flops_wgrad = 2.0 * d * d * pairs                            # sum over directions of [d, P_k] x [P_k, d]
        row = {"sorted_columns": bool(hip.ops.SHARED_SORT and n_cols >= hip.ops.SHARED_SORT_MIN), "R": R, "S": S, "d": d, "active_fraction": args.p, "columns": n_cols, "active_edge_pairs": pairs,
               "fwd_GFLOP": flops_fwd / 1e9, "wgrad_GFLOP": flops_wgrad / 1e9, "peak_TFLOPs": FP32_MFMA_PEAK_TFLOPS, "kernels": {}}
        for name, (n, ms) in sorted(prof.items(), key=lambda kv: -kv[1][1]):
            us = ms / args.iters * 1e3  # per call of the op (the weight gradient is up to three launches: list, GEMM, reduce)
            k = {"us_per_call": round(us, 2), "launches_per_call": n / args.iters}
            # router_mfma = the forward GEMM + the data-gradient GEMM (+ their column sorts, ops.SHARED_SORT): "TFLOPs" prices
            # ONE dense forward (zero columns included, the figure of earlier rounds), "useful_TFLOPs" the multiplications of
            # active (edge, stream) pairs only, forward and data gradient together
            fl = {"router_mfma": flops_fwd, "wgrad_mfma": flops_wgrad}.get(name)
            if fl:
                k["TFLOPs"] = round(fl / (us * 1e-6) / 1e12, 2)
                k["frac_of_fp32_mfma_peak"] = round(k["TFLOPs"] / FP32_MFMA_PEAK_TFLOPS, 4)
            if name == "router_mfma":
                k["useful_TFLOPs"] = round(2 * flops_wgrad / (us * 1e-6) / 1e12, 2)
            row["kernels"][name] = k
        print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
