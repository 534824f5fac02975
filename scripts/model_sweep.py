#!/usr/bin/env python3
"""Model-level point of BASELINE configs[2]: forward_inner_step region-steps/s of the whole
reasoning loop (token mixer once per sequence, gate -> router -> region bank -> heads per inner step) at R in {64, 512},
k_active in {8, 32}, d = 256, sequences of 1024 tokens (max_T = 2048), forward only (no_grad), N sequences per call,
in strict mode (the reference's schedule, device-steered + graph replay) and in multi-stream mode (lockstep loops).
One JSON line per point.  IC_SWEEP_EDGE=by_direction|factorized selects the edge-transform mode (default per_edge),
IC_SWEEP_R="512" the map sizes, IC_SWEEP_N the sequences per call."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ironcortex_amd as ic  # noqa: E402

dev = torch.device("cuda:0")
N, T, REPS = int(os.environ.get("IC_SWEEP_N", "16")), 1024, 5
EDGE = os.environ.get("IC_SWEEP_EDGE", "per_edge")
for R in [int(x) for x in os.environ.get("IC_SWEEP_R", "64 512").split()]:
    for k_active in (8, 32):
        cfg = ic.CortexConfig(R=R, d=256, V=256, K_inner=8, B_br=2, k_active=k_active, max_T=2048, edge_transform_mode=EDGE)
        torch.manual_seed(0)
        model = ic.CortexReasoner(ic.hex_neighbors(R), ic.hex_axial_coords(R), {"sensor": 0, "motor": R - 1}, cfg).to(dev)
        with torch.no_grad():  # tie-free gate scores from the first inner step on
            model.gate.gain_ema.copy_(torch.rand(R, device=dev) * 0.2)
            model.gate.homeo.copy_(torch.rand(R, device=dev) * 0.05)
        tokens = torch.randint(0, 255, (N, T), device=dev)
        focus = torch.rand(N, T, device=dev) < 0.3
        for mode in ("strict", "multi"):
            model.stream_mode = mode
            with torch.no_grad():
                for _ in range(2):
                    model.reasoning_loop_batch(tokens, cfg.K_inner, focus)
                model.count_units(True)
                i0 = model.counters["inner_steps"]
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(REPS):
                    model.reasoning_loop_batch(tokens, cfg.K_inner, focus)
                torch.cuda.synchronize()
                el = time.perf_counter() - t0
                u = model.read_units()
                inner = model.counters["inner_steps"] - i0
                # the token mixer's share: the same batch through sensor_vectors_batch alone
                t1 = time.perf_counter()
                for _ in range(REPS):
                    model.sensor_vectors_batch(tokens, focus)
                torch.cuda.synchronize()
                el_mix = time.perf_counter() - t1
                model.count_units(False)
            print(json.dumps({"edge": EDGE, "R": R, "k_active": k_active, "d": 256, "T": T, "N": N, "mode": mode, "ms_per_call": 1e3 * el / REPS,
                              "mixer_ms_per_call": 1e3 * el_mix / REPS, "inner_steps_per_s": inner / el,
                              "region_steps_per_s": u["region_steps"] / el, "active_edges_per_s": u["active_edges"] / el,
                              "tokens_per_s": N * T * REPS / el,
                              "region_steps_per_inner_step": u["region_steps"] / max(1, inner)}), flush=True)
        del model
        torch.cuda.empty_cache()
