#!/usr/bin/env python3
"""bench.py — train_step tokens/sec of the IronCortex hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

A "step" is one forward-forward train_step (corrupt -> reasoning loops over [pos;neg] ->
losses -> backward -> Hebbian nudges -> AdamW) on one batch of synthetic byte sequences
resident on the device.  N=1 workload: BASELINE.json configs[1], the ~10M-parameter
train_tiny_shakespeare_10m.py config (R=10, d=256, all five experimental flags on), B=8
sequences of T=256 per GPU.  N>1: one process per GPU (torchrun), batch sharded (weak scaling),
one RCCL exchange per step (ironcortex_amd/parallel.py).  Strict-sequential mode: every rank
runs the reference's own schedule (samples one after another, region state carried over).

Prints ONE JSON line on rank 0 with the contract fields plus
  roofline      the library kernel with the largest summed duration (found in the last warm-up
                step with every kernel bracketed), timed live with HIP events on its launch stream
                over the timed region (ic_prof_*), against the HBM peak;
  cpu_baseline  the CPU oracle model (reference-shaped restatement, kind "port") timed on this
                box's host cores on a bounded sample of the same workload (N=1 only).
"""
from __future__ import annotations

import argparse
import json
import os
import random
import sys
import time

# Data-parallel runs: two hardware queues per process instead of HIP's four.  A train_step is one long chain of dependent
# launches on one stream plus a handful of collectives on the backend's; with the default four queues the loops of a
# data-parallel step ran 0.9-1.3 ms longer than the same graph without a process group (one-rank RCCL group,
# scripts/host_gpu_timeline.py --dp: step 15.3 ms, with two queues 14.0 ms, with one 15.6 ms; without data parallelism
# the setting changes nothing).  Must be in the environment before the HIP runtime starts.  IC_HW_QUEUES=<n> chooses
# another count, IC_HW_QUEUES=keep leaves the environment's GPU_MAX_HW_QUEUES alone (the GPU boxes export 4).
if (int(os.environ.get("WORLD_SIZE", "1")) > 1 or os.environ.get("IC_FORCE_DP") == "1") and os.environ.get("IC_HW_QUEUES") != "keep":
    os.environ["GPU_MAX_HW_QUEUES"] = os.environ.get("IC_HW_QUEUES", "2")

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PROF_STRIDE = 7        # the dominant kernel's launches are event-bracketed 1 in 7 in the measurement pass after the timed region
                       # (coprime with K_inner = 8 and with the 3 GEMV launches per iteration: a stride of 8 sampled the SAME
                       # iteration of every loop -- at R=512 the first one, whose router has no active sender yet)
EVENT_STEPS = 3        # steps of that pass
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s achievable)


def make_config(name: str):
    from ironcortex_amd import CortexConfig
    if name == "10m":  # reference train_tiny_shakespeare_10m.py:32-48
        return CortexConfig(R=10, d=256, V=256, K_inner=8, B_br=2, k_active=8, max_T=512,
                            enable_adaptive_filter_dynamics=True, enable_precision_routed_messages=True,
                            enable_radial_tangential_updates=True, enable_afa_attention=True,
                            enable_ff_energy_alignment=True, enable_energy_verifier=True,
                            enable_forward_forward=True, debug_metrics_every_n_steps=2)
    if name == "default":  # reference train_tiny_shakespeare.py:59-67
        return CortexConfig(R=36, d=256, V=256, K_inner=8, B_br=2, k_active=8, max_T=512)
    if name in ("r512", "r64"):  # BASELINE configs[2], model level: the 10M flags on a large map (not a `value` workload)
        return CortexConfig(R=int(name[1:]), d=256, V=256, K_inner=8, B_br=2, k_active=32, max_T=512,
                            enable_adaptive_filter_dynamics=True, enable_precision_routed_messages=True,
                            enable_radial_tangential_updates=True, enable_afa_attention=True,
                            enable_ff_energy_alignment=True, enable_energy_verifier=True,
                            enable_forward_forward=True, debug_metrics_every_n_steps=2)
    if name in ("bydir36", "fact36"):  # the default script config in a shared-matrix edge mode (A/B workloads, not `value`;
        # forward-forward off: the reference's train_step raises AttributeError with it in these modes)
        return CortexConfig(R=36, d=256, V=256, K_inner=8, B_br=2, k_active=8, max_T=512, enable_forward_forward=False,
                            edge_transform_mode="by_direction" if name == "bydir36" else "factorized")
    raise SystemExit(f"unknown --config {name}")


WORKLOAD = {"10m": "train_tiny_shakespeare_10m.py config (R=10,d=256,all flags on)",
            "default": "train_tiny_shakespeare.py default config (R=36,d=256)",
            "r512": "configs[2] model level: 10M-script flags on R=512, k_active=32, d=256",
            "r64": "configs[2] model level: 10M-script flags on R=64, k_active=32, d=256",
            "bydir36": "default script config (R=36,d=256), edge_transform_mode=by_direction, forward-forward off",
            "fact36": "default script config (R=36,d=256), edge_transform_mode=factorized, forward-forward off"}


def algorithmic_bytes(kernel: str, d: int, st: dict) -> float:
    """fp32 bytes one launch of a library kernel must move (u*16d^2-class
    weight bytes + n*vectors), from per-launch averages counted on the device:
      n_f / n_b  (stream, region) pairs served by a forward / backward region launch,
      u_f / u_b  distinct regions among them (a launch reads a region's weights once for all its streams),
      e          distinct active edges, ep  (stream, edge) pairs, q  destinations with an active in-edge,
      S, R       streams and regions of the launch.
    Kernels whose bytes follow from their shapes alone (the GEMV chains, the heads) are priced by the
    library itself at launch time (ic_prof_bytes_read, see kernel_bytes()); 0.0 = no model."""
    afd = st.get("afd", False)
    dd, vec = 4.0 * d * d, 4.0 * d
    S, R = st.get("S", 1), st.get("R", 0)
    n_f, u_f, n_b, u_b = st.get("n_f", 0), st.get("u_f", 0), st.get("n_b", 0), st.get("u_b", 0)
    e, ep, q, prec = st.get("e", 0), st.get("ep", st.get("e", 0)), st.get("q", 0), st.get("prec", False)
    table = {
        # forward region phase
        "region_pro": n_f * 6 * vec,                                    # M, pred, norm_w (+sensor) in; u1, x out
        "region_rkv": u_f * 3 * dd + n_f * (10 + (5 if afd else 0)) * vec,
        "region_out": u_f * dd + n_f * 3 * vec,
        # active rows: hpre in, H / gate / pred out; idle rows: M, pred, num, den in, H, pred, num, den out (state is functional)
        "region_fin": n_f * 4 * vec + (S * R - n_f) * 8 * vec + S * R * 9,
        # backward region phase (weights re-read; dW deferred to wgrad_region)
        "region_out_bwd": u_b * dd + n_b * 4 * vec,
        "region_rkv_bwd": u_b * 3 * dd + n_b * (16 + (6 if afd else 0)) * vec,
        "region_pro_bwd": n_b * 10 * vec,
        # split r/k/v backward (many streams per workgroup): per active pair 8 state / activation / incoming-gradient
        # rows + the o-projection's 8 slab rows in, 3 tape rows + 3 gradient rows out (+5 under AFD)
        "region_mid_bwd": n_b * (22 + (5 if afd else 0)) * vec,
        "region_fin_bwd": n_b * 4 * vec,
        "wgrad_region": u_b * 4 * dd * 2 + n_b * 6 * vec,               # dW read-modify-write once per touched region + tape
        # router
        "router_fwd": e * (dd + vec) + R * S * vec,
        "router_edge_fwd": e * (2 * dd + 3 * vec) + q * (dd + 2 * vec),
        "router_combine": ep * 3 * vec + S * R * vec,
        "router_comb_bwd": ep * 5 * vec + S * R * 2 * vec,
        "router_mat_bwd": e * ((2 if prec else 1) * dd + 4 * vec) + (q * (dd + 2 * vec) if prec else 0),
        "router_gather_bwd": ep * vec + S * R * vec,
        "wgrad_edge": e * dd * 2 + ep * 2 * vec,
        # gate / primitives
        "gate_scores": S * R * vec, "gate_select": S * R * (vec if R <= 64 else 8.0),
        "kwta_rows": st.get("rows", 0) * 3 * vec, "goodness_fwd": st.get("rows", 0) * vec, "goodness_bwd": st.get("rows", 0) * 2 * vec,
        "hebbian": st.get("e_upd", 0) * dd * 2,
    }
    # the deferred weight gradients as pair-list GEMMs (LoopEngine passes with >= 64 tape slots): region matrices in two
    # calls (r/k/v, o), edge matrices in one per kind (messages; + keys, queries with precision routing), each call a
    # list kernel and a GEMM kernel under ONE kernel id -> the same bytes as the row-slice kernels, averaged per launch
    n_calls = 2 + (3 if prec else 1)
    table["wgrad_mfma"] = (table["wgrad_region"] + (n_calls - 2) * table["wgrad_edge"]) / (2 * n_calls)
    if st.get("fsplit"):  # >= 32 streams per launch: k_region_rkvG (weights, x row in, 3 raw rows out) + k_region_state
        table["region_rkv"] = u_f * 3 * dd + n_f * 4 * vec
        table["region_state"] = n_f * (10 + (3 if afd else 0)) * vec
    if st.get("routed"):  # ic_loop_step_fwd: the router's combine rides in the r/k/v kernel (per launch: e_f active
        # incoming edges -> message, key, precision rows; q_f destinations -> state and query rows in, M row out)
        table["region_rkv"] += (3 * st["e_f"] + 3 * st["q_f"]) * vec
    return table.get(kernel, 0.0)


# flops per (query, key) pair of the attention kernels (fp32 MFMA, dense peak 157.3 TFLOP/s): scores + probabilities x
# values forward; both transposed products per operand backward (scripts/mixer_bench.py)
ATTN_FLOPS = {"attn_fwd": lambda dqk, dv: 2 * (2 * dqk + dv), "attn_bwd_kv": lambda dqk, dv: 2 * (2 * dqk + 2 * dv),
              "attn_bwd_q": lambda dqk, dv: 2 * (2 * dqk + dv)}


def pmc_traffic(kernel: str, cfg_name: str):
    """HBM-side bytes per launch of `kernel` from the committed rocprofv3 --pmc passes of this same
    command (profiles/r03_pmc_traffic.json; PMC collection serialises kernels, so it is a
    separate run, never the timed one).  None when no measurement is on file for the config."""
    path = os.path.join(ROOT, "profiles", "r03_pmc_traffic.json")
    if cfg_name != "10m" or not os.path.exists(path):
        return None
    with open(path) as f:
        k = json.load(f).get("kernels", {}).get(kernel)
    return k["traffic_bytes_per_launch"] if k else None


def cpu_baseline(cfg_name: str, B: int, T: int, seed: int):
    """Reference-shaped oracle train_step on the host cores (1 warm-up + 2 timed steps)."""
    from oracle import model as OM
    cfg = make_config(cfg_name)
    torch.manual_seed(seed)
    random.seed(seed)
    model = OM.build(cfg)
    opt = torch.optim.AdamW(model.parameters(), lr=1e-3)
    tokens = torch.randint(0, 255, (B, T))
    OM.oracle_train_step(model, opt, tokens, OM.LossWeights())
    n, t0 = 2, time.perf_counter()
    for _ in range(n):
        OM.oracle_train_step(model, opt, tokens, OM.LossWeights())
    dt = (time.perf_counter() - t0) / n
    threads = torch.get_num_threads()
    torch.set_num_threads(1)  # the 1-thread number beside the all-core one (one more step, already warm)
    try:
        t0 = time.perf_counter()
        OM.oracle_train_step(model, opt, tokens, OM.LossWeights())
        dt1 = time.perf_counter() - t0
    finally:
        torch.set_num_threads(threads)
    return {"value": B * T / dt, "unit": "tokens/s", "cores": threads, "kind": "port",
            "one_thread": {"value": B * T / dt1, "s_per_step": dt1},
            "sample": f"2 train_steps of the full B={B},T={T} batch after 1 warm-up ({dt:.2f} s/step), "
                      f"oracle/model.py (reference-shaped per-region Python loop), torch {torch.__version__} CPU",
            "s_per_step": dt, "host_cpus": os.cpu_count()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--config", default="10m")
    ap.add_argument("--batch", type=int, default=8, help="sequences per GPU")
    ap.add_argument("--seq", type=int, default=256)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--no-kernel-sweep", action="store_true")
    ap.add_argument("--no-multi-stream", action="store_true")
    args = ap.parse_args()

    if os.environ.get("IC_HANG_DUMP"):  # debugging aid: dump every thread's stack after N seconds and exit
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["IC_HANG_DUMP"]), exit=True)
    import ironcortex_amd as ic
    from ironcortex_amd import hip

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N > 1")
    if os.environ.get("IC_DIST_BACKEND", "nccl") != "nccl":
        local_rank = local_rank % max(1, torch.cuda.device_count())  # rehearsal: ranks may share a GPU
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dp = None
    if world > 1 or os.environ.get("IC_FORCE_DP") == "1":  # (IC_FORCE_DP=1 under torch.distributed.run --nproc-per-node 1:
        # the data-parallel path on RCCL with a one-rank group, to rehearse it on a one-GPU box)
        import torch.distributed as dist
        from ironcortex_amd.parallel import DataParallel
        # RCCL over xGMI; IC_DIST_BACKEND=gloo only to rehearse the N>1 code path on a one-GPU box
        backend = os.environ.get("IC_DIST_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
        dp = DataParallel()

    cfg = make_config(args.config)
    torch.manual_seed(1234)
    random.seed(1234 + rank)
    R = cfg.R
    model = ic.CortexReasoner(ic.hex_neighbors(R), ic.hex_axial_coords(R), {"sensor": 0, "motor": R - 1}, cfg).to(dev)
    if dp is not None:
        dp.broadcast_module(model)
    # the reference's optimizer (train_tiny_shakespeare.py:86: AdamW, lr 1e-3), in torch's fused multi-tensor
    # implementation: same update (scripts/opt_fused_check.py: parameters equal to 2e-7 after a step, identical
    # losses), one kernel per chunk instead of ~15 foreach launches (-1.3 ms per step).  IC_BENCH_FUSED_ADAMW=0: foreach
    opt = torch.optim.AdamW(model.parameters(), lr=1e-3, fused=os.environ.get("IC_BENCH_FUSED_ADAMW", "1") != "0" or None)
    n_params = sum(p.numel() for p in model.parameters())
    gen = torch.Generator().manual_seed(1234 + rank)
    B, T = args.batch, args.seq
    tokens = torch.randint(0, 255, (B, T), generator=gen).to(dev)  # uniform bytes, mask id 255 excluded
    lw = ic.LossWeights()

    def sync_all():
        torch.cuda.synchronize()
        if dp is not None:
            import torch.distributed as dist
            dist.barrier()
            torch.cuda.synchronize()

    timing = not args.no_kernel_timing
    survey, survey_bytes = {}, {}
    for w in range(args.warmup):
        last = timing and w == args.warmup - 1
        if last:  # untimed survey step: every library kernel bracketed by events, to find the dominant one
            torch.cuda.synchronize()
            hip.prof_select("")
            hip.prof_enable(True)
            hip.prof_read()
            hip.prof_bytes_read()
        ic.train_step(model, opt, tokens, lw, dev, dp=dp)
        if last:
            survey = hip.prof_read()
            survey_bytes = hip.prof_bytes_read()
            hip.prof_enable(False)
    dominant = None
    if timing and survey:
        # the dominant kernel = the largest summed duration among ALL library kernels (every one has a byte model:
        # algorithmic_bytes() or the library's own launch-time count)
        dominant = max(survey, key=lambda k: survey[k][1])
    # timed region: the product path as it ships -- no event bracketing, no unit counters (the reasoning loops of a
    # step are replayed as ONE captured HIP graph, whose kernels cannot be bracketed individually)
    sync_all()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        metrics = ic.train_step(model, opt, tokens, lw, dev, dp=dp)
    sync_all()
    elapsed = time.perf_counter() - t0
    # measurement pass, right after the timed region, same process, same workload: EVENT_STEPS more steps with the
    # dominant kernel bracketed by HIP events (1 launch in PROF_STRIDE, launched call by call) and the device-side
    # unit counters on -> roofline.achieved and the per-step unit rates
    prof, lib_bytes = {}, {}
    model.count_units(True)
    if timing:
        hip.prof_select(dominant or "", stride=PROF_STRIDE)
        hip.prof_enable(True)
        hip.prof_read()
        hip.prof_bytes_read()
    sync_all()
    inner0 = model.counters["inner_steps"]
    t1 = time.perf_counter()
    for _ in range(EVENT_STEPS):
        ic.train_step(model, opt, tokens, lw, dev, dp=dp)
    sync_all()
    elapsed_units = time.perf_counter() - t1
    if timing:
        prof = hip.prof_read()
        lib_bytes = hip.prof_bytes_read()
        hip.prof_enable(False)
        hip.prof_select("")
    units = model.read_units()
    units["inner_steps"] -= inner0
    model.count_units(False)
    unit_scale = args.steps / EVENT_STEPS  # unit counts of EVENT_STEPS steps -> per timed region
    for k_ in ("region_steps", "inner_steps"):
        units[k_ + "_timed"] = units[k_] * unit_scale
    if dp is not None:
        import torch.distributed as dist
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- second, named measurement: the same train_step in multi-stream mode: the 2B loops of a
    # step as independent lockstep streams.  NOT the reference's schedule (no sample-to-sample state carry), so
    # it never becomes `value`; each stream is parity-checked against a batch-1 oracle run (tests/test_hip_multistream.py)
    multi = None
    if not args.no_multi_stream:
        model.count_units(False)
        model.stream_mode = "multi"
        for _ in range(max(2, min(args.warmup, 3))):
            ic.train_step(model, opt, tokens, lw, dev, dp=dp)
        sync_all()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            m2 = ic.train_step(model, opt, tokens, lw, dev, dp=dp)
        sync_all()
        el2 = time.perf_counter() - t0
        model.stream_mode = "strict"
        if dp is not None:
            import torch.distributed as dist
            t = torch.tensor([el2], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el2 = float(t.item())
        multi = {"value": world * B * T * args.steps / el2, "unit": "tokens/s", "ms_per_step": 1e3 * el2 / args.steps,
                 "steps": args.steps, "loss_total": m2["total"],
                 "mode": "multi-stream: the 2B reasoning loops of a train_step as independent streams in lockstep, every "
                         "one from the region state the step starts with (strict mode carries it from sample to sample)"}

    if rank == 0:
        tok_s = world * B * T * args.steps / elapsed
        out = {
            "metric": "train_step tokens/sec (fwd+bwd)", "value": tok_s, "unit": "tokens/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": WORKLOAD[args.config], "per_gpu_batch": B, "global_batch": B * world,
                       "seq_len": T, "params": n_params, "mode": "strict-sequential (reference schedule)",
                       "parallelism": f"dp{world}",
                       **({"gate_ties": "fresh model of R > 16 regions: all-equal gate scores at inner step 0; exact ties are broken "
                                        "by the stable order (score desc, index asc), not by the reference's unstable argsort "
                                        "(declared deviation, DESIGN §3)"} if cfg.R > 16 else {})},
            # (units counted over the EVENT_STEPS steps after the timed region: same tokens, same schedule)
            "region_steps_per_s": units["region_steps_timed"] / elapsed * world,
            "inner_steps_per_s": units["inner_steps_timed"] / elapsed * world,
            "loss_total": metrics["total"],
        }
        roof = None
        if prof and dominant:
            nb = 1 if cfg.train_deterministic_inner_loop else max(1, cfg.B_br)
            lf, lb = max(1.0, units["fwd_region_launches"]), max(1.0, units["bwd_region_launches"])
            lrb = max(1.0, units["bwd_router_launches"])
            prec = bool(cfg.enable_precision_routed_messages)
            fwd = dict(S=nb, R=cfg.R, n_f=units["region_steps"] / lf, u_f=units["fwd_unique_regions"] / lf,
                       e=units["fwd_unique_edges"] / lf, ep=units["active_edges"] / lf, q=units["fwd_unique_dst"] / lf,
                       afd=cfg.enable_adaptive_filter_dynamics, prec=prec,
                       routed=bool(prec and cfg.d <= 256 and not cfg.train_deterministic_inner_loop
                                   and cfg.B_br > 1 and os.environ.get("IC_FUSE_STEP", "1") != "0"),
                       e_f=units["active_edges"] / lf, q_f=units["dst_regions"] / lf)
            bwd = dict(S=2 * B, R=cfg.R, n_b=units["bwd_pairs"] / lb, u_b=units["bwd_unique_regions"] / lb,
                       e=units["bwd_unique_edges"] / lrb, ep=units["bwd_edge_pairs"] / lrb, q=units["bwd_dst_regions"] / lrb,
                       afd=cfg.enable_adaptive_filter_dynamics, prec=prec, rows=2 * B * cfg.R,
                       e_upd=units["hebbian_edges"] / max(1.0, units["hebbian_launches"]))

            def bytes_per_launch(name, libb):
                """algorithmic bytes of one launch of `name`: the library's own count for shape-determined kernels
                (+ the mask-dependent region-head rows of the heads kernel), else the unit model."""
                if name in libb and libb[name][0]:
                    ab = libb[name][1] / libb[name][0]
                    if name == "token_rows":  # + one V x d head per active non-motor region of each stream
                        ab += max(0.0, fwd["n_f"] - nb) * cfg.V * 4.0 * cfg.d
                    return ab
                st = bwd if (name.endswith("_bwd") or name.startswith("wgrad") or name in ("hebbian", "goodness_fwd", "goodness_bwd")) else fwd
                return algorithmic_bytes(name, cfg.d, st)

            launches, ms = prof[dominant]
            ab = bytes_per_launch(dominant, lib_bytes)
            avg_s = ms / launches * 1e-3
            achieved = ab / avg_s / 1e9
            lib_ms = sum(ms_ for _, ms_ in survey.values())
            roof = {"bound": "hbm", "kernel": "k_" + dominant, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": achieved / HBM_PEAK_GBS, "traffic": pmc_traffic(dominant, args.config), "launches": launches,
                    "launch_sampling": f"HIP events on 1 launch in {PROF_STRIDE} over {EVENT_STEPS} further steps of the same workload "
                                       "right after the timed region (which replays the loops as one captured HIP graph: its "
                                       "kernels cannot be bracketed one by one; rocprofv3 --kernel-trace of this command, "
                                       "profiles/, times them inside the graph)",
                    "avg_launch_us": avg_s * 1e6, "algorithmic_bytes_per_launch": ab,
                    "dominant_by": "largest summed duration among ALL library kernels in the survey step",
                    "units_per_launch": {k: round(v, 3) for k, v in (bwd if dominant.endswith("_bwd") else fwd).items()
                                         if isinstance(v, float)},
                    "library_kernel_ms_per_step": lib_ms,
                    "note": "working set fits the 256 MB Infinity Cache; launches are latency-bound at this size "
                            "(see profiles/ for the R=64..512 kernel sweep)"}
            # every library kernel of the survey step, priced the same way
            table = {}
            for k, (n_, ms_) in sorted(survey.items(), key=lambda kv: -kv[1][1]):
                abk = bytes_per_launch(k, survey_bytes)
                us = ms_ / n_ * 1e3
                table[k] = {"ms_per_step": round(ms_, 4), "launches_per_step": n_, "avg_us": round(us, 2),
                            "alg_KB_per_launch": round(abk / 1e3, 1),
                            "GBs": round(abk / (us * 1e-6) / 1e9, 1) if abk else None}
                if k in ATTN_FLOPS:  # MFMA-bound kernels of the token mixer: flops, not bytes
                    hd = cfg.d // 4
                    dqk = hd if cfg.enable_afa_attention else hd + 64
                    Ta = T + (0 if cfg.enable_afa_attention else sum(len(range(S_ // 2, T, S_)) for S_ in (128, 32)) + 8)
                    fl = ATTN_FLOPS[k](dqk, hd) * 2 * B * 4 * Ta * Ta  # per (query, key) pair x sequences x heads x positions^2
                    table[k].update(alg_GFLOP_per_launch=round(fl / 1e9, 3), TFLOPs=round(fl / (us * 1e-6) / 1e12, 1),
                                    frac_of_fp32_mfma_peak=round(fl / (us * 1e-6) / 1e12 / 157.3, 3))
            out["kernels"] = table
            out["kernels_source"] = ("HIP events on every library kernel during the last warm-up step; algorithmic bytes "
                                     "from bench.algorithmic_bytes x device-counted units, or the library's launch-time count")
        out["roofline"] = roof
        # the same kernels where bytes, not launch latency, bound them (BASELINE configs[2], one point:
        # R=512, k=32, S=8 streams, d=256, forward): HIP events on every launch, ~2 s
        if world == 1 and not args.no_kernel_sweep and args.config == "10m":
            try:
                sys.path.insert(0, os.path.join(ROOT, "scripts"))
                from kernel_sweep import sweep
                pt = next(iter(sweep([512], [32], [8], 256, 10, False)))
                out["kernel_sweep"] = {"point": "R=512,k=32,S=8,d=256 forward (synthetic, BASELINE configs[2])",
                                       "peak_GBs": HBM_PEAK_GBS,
                                       "kernels": {k: v for k, v in pt["kernels"].items() if v["GBs"]}}
            except Exception as e:  # never fail the headline line for the extra measurement
                out["kernel_sweep"] = {"error": repr(e)}
        cb = None
        if world == 1 and not args.no_cpu_baseline:
            cb = cpu_baseline(args.config, B, T, 1234)
        out["cpu_baseline"] = cb
        out["multi_stream"] = multi
        print(json.dumps(out))
    if dp is not None:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
