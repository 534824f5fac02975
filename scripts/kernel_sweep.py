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


def sweep(Rs, ks, Ss, d=256, iters=20, backward=False):
    """Yields one dict per (R, S, k) point."""
    dev = torch.device("cuda", torch.cuda.current_device())
    for R in Rs:
        torch.manual_seed(0)
        nb, coords = hex_neighbors(R), hex_axial_coords(R).to(dev)
        cells = [RWKVRegionCell(d, _shared_bank=True) for _ in range(R)]
        bank = RegionBank(cells, 0, R - 1)
        for c in cells:
            c.to(dev)
        bank.to_device()
        router = Router(nb, d, R).to(dev)
        nin = torch.ones(d, device=dev)
        for S in Ss:
            for k in ks:
                g = torch.Generator().manual_seed(0)
                mask = torch.zeros(S, R, dtype=torch.uint8)
                for s in range(S):
                    mask[s, torch.randperm(R, generator=g)[:k]] = 1
                mask = mask.to(dev)
                H = torch.randn(S, R, d, device=dev) * mask.unsqueeze(-1)
                sensor = torch.randn(S, d, device=dev)
                bank.reset_state(dev, S)
                bank.num, bank.den = torch.rand(S, R, d, device=dev), torch.rand(S, R, d, device=dev) + 0.5
                bank.pred = 0.1 * torch.randn(S, R, d, device=dev)
                bank.dt = torch.randint(0, 8, (S, R), dtype=torch.int32, device=dev)
                state0 = (bank.num, bank.den, bank.pred, bank.dt)

                def run(grad):
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

                for _ in range(3):
                    run(backward)
                torch.cuda.synchronize()
                hip.prof_select("")
                hip.prof_enable(True)
                hip.prof_read()
                for _ in range(iters):
                    run(backward)
                prof = hip.prof_read()
                hip.prof_enable(False)
                uniq = int(mask.any(0).sum())
                src_act = mask.bool()[:, router.src_idx]
                e_uniq = int(src_act.any(0).sum())
                pairs = int(mask.sum())
                st = dict(u_f=uniq, n_f=pairs, u_b=uniq, n_b=pairs, e=e_uniq, q=0, R=R, afd=False, prec=False)
                row = {"R": R, "S": S, "k": k, "d": d, "unique_regions": uniq, "unique_edges": e_uniq,
                       "weights_MiB": {"regions": R * 4 * d * d * 4 / 2**20, "W_edge": router.E * d * d * 4 / 2**20},
                       "kernels": {}}
                for name, (n, ms) in sorted(prof.items(), key=lambda kv: -kv[1][1]):
                    us = ms / n * 1e3
                    ab = algorithmic_bytes(name, d, st)
                    row["kernels"][name] = {"avg_us": round(us, 2), "alg_MB": round(ab / 1e6, 3),
                                            "GBs": round(ab / (us * 1e-6) / 1e9, 1) if ab else None,
                                            "frac": round(ab / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4) if ab else None}
                yield row
        del bank, router, cells
        torch.cuda.empty_cache()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--R", type=int, nargs="+", default=[64, 128, 256, 512])
    ap.add_argument("--k", type=int, nargs="+", default=[4, 8, 16, 32])
    ap.add_argument("--S", type=int, nargs="+", default=[1, 8])
    ap.add_argument("--d", type=int, default=256)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--backward", action="store_true")
    args = ap.parse_args()
    for row in sweep(args.R, args.k, args.S, args.d, args.iters, args.backward):
        print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
