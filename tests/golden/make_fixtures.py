"""Generate golden vectors by running the REAL reference (read-only at /root/reference).

Run once in the build container (the reference never travels to the GPU box):

    cd /tmp && PYTHONPATH=/root/reference:/root/repo PYTHONDONTWRITEBYTECODE=1 MPLBACKEND=Agg \
        python /root/repo/tests/golden/make_fixtures.py

Only OUTPUTS are stored (tests/golden/*.npz); inputs come from tests/golden/inputs.py.
Large matrix gradients are stored as two probe projections (dW @ p, q @ dW).
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from tests.golden import inputs as gi  # noqa: E402

from ironcortex.gate import Gate, Router  # noqa: E402  (the reference package)
from ironcortex.region import RWKVRegionCell  # noqa: E402
from ironcortex.utils import KWTA, nms_topk  # noqa: E402
from ironcortex.wiring import hex_axial_coords, hex_neighbors  # noqa: E402


def npy(t):
    return t.detach().cpu().numpy().copy()  # copy: fixtures must not alias live parameters


def save(name, blob):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **blob)
    print(f"wrote {path}: {len(blob)} arrays, {os.path.getsize(path)/1024:.1f} KiB")


# ------------------------------------------------------------------ F1/F3/F4: single region step
def build_cell(case, t):
    name, d, m_time, dt, afd, rt, noise_mode, step_pos = case
    cell = RWKVRegionCell(
        d, m_time_pairs=m_time, enable_adaptive_filter_dynamics=afd,
        enable_radial_tangential_updates=rt, afd_noise_mode=noise_mode,
    )
    with torch.no_grad():
        cell.r_lin.weight.copy_(t["Wr"]); cell.k_lin.weight.copy_(t["Wk"])
        cell.v_lin.weight.copy_(t["Wv"]); cell.o_lin.weight.copy_(t["Wo"])
        cell.norm.weight.copy_(t["norm_w"]); cell.raw_decay.copy_(t["raw_decay"])
        cell.process_noise_param.copy_(t["pn_param"]); cell.obs_noise_param.copy_(t["on_param"])
        cell.radius_process_noise_param.copy_(t["rpn_param"])
        cell.radius_obs_noise_param.copy_(t["ron_param"])
    return cell


def region_fixture():
    blob = {}
    for ci, case in enumerate(gi.REGION_CASES):
        name, d, m_time, dt, afd, rt, noise_mode, step_pos = case
        t = gi.region_inputs(1000 + 100 * ci, d, noise_mode)
        cell = build_cell(case, t)
        leaves = {k: t[k].clone().requires_grad_(True) for k in ("num", "den", "var", "pred", "radius", "radius_var", "x_in")}
        cell.state_num = leaves["num"] * 1.0
        cell.state_den = leaves["den"] * 1.0
        cell.state_var = leaves["var"] * 1.0
        cell.pred = leaves["pred"] * 1.0
        cell.radius = leaves["radius"] * 1.0
        cell.radius_var = leaves["radius_var"] * 1.0
        cell.surprise_ema = t["surprise"].clone()
        cell.dt = dt
        h = cell.step(leaves["x_in"], step_pos)
        loss = (h * t["g_h"]).sum() + (cell.state_num * t["g_num"]).sum() + (cell.state_den * t["g_den"]).sum()
        if afd:
            loss = loss + (cell.state_var * t["g_var"]).sum()
        if rt:
            loss = loss + (cell.radius * t["g_radius"]).sum() + (cell.radius_var * t["g_radius_var"]).sum()
        loss.backward()
        pre = f"{name}."
        blob[pre + "h"] = npy(h)
        blob[pre + "num"] = npy(cell.state_num); blob[pre + "den"] = npy(cell.state_den)
        blob[pre + "var"] = npy(cell.state_var); blob[pre + "pred"] = npy(cell.pred)
        blob[pre + "radius"] = npy(cell.radius); blob[pre + "radius_var"] = npy(cell.radius_var)
        blob[pre + "surprise"] = npy(cell.surprise_ema)
        blob[pre + "dt"] = np.array(cell.dt)
        if rt:
            blob[pre + "last_dir"] = npy(cell.last_dir); blob[pre + "last_norm"] = npy(cell.last_norm)
        for k, leaf in leaves.items():
            if leaf.grad is not None:
                blob[pre + "g." + k] = npy(leaf.grad)
        for pname, par in (("norm_w", cell.norm.weight), ("raw_decay", cell.raw_decay),
                           ("pn_param", cell.process_noise_param), ("on_param", cell.obs_noise_param),
                           ("rpn_param", cell.radius_process_noise_param),
                           ("ron_param", cell.radius_obs_noise_param)):
            if par.grad is not None:
                blob[pre + "g." + pname] = npy(par.grad)
        for pname, lin in (("Wr", cell.r_lin), ("Wk", cell.k_lin), ("Wv", cell.v_lin), ("Wo", cell.o_lin)):
            g = lin.weight.grad
            if d <= 32:
                blob[pre + "g." + pname] = npy(g)
            blob[pre + "gp." + pname] = npy(g @ t["probe_p"])
            blob[pre + "gq." + pname] = npy(t["probe_q"] @ g)
    save("region.npz", blob)


# ------------------------------------------------------------------ F2: 4-step chains with grads
def chain_fixture():
    blob = {}
    for ci, (d, afd, rt) in enumerate([(8, True, False), (32, False, False), (32, True, True)]):
        t = gi.region_inputs(5000 + 100 * ci, d, "scalar")
        case = (f"chain{ci}", d, 16, 0, afd, rt, "scalar", 0.0)
        cell = build_cell(case, t)
        xs = gi.det(5050 + ci, 4, d).requires_grad_(True)
        outs = []
        for i in range(4):
            if i == 2:  # a skipped micro-step in the middle exercises dt>0 and the trace
                cell.predict(gi.det(5060 + ci, d))
                cell.skip()
            outs.append(cell.step(xs[i], i / 3.0))
        loss = sum((o * gi.det(5070 + ci + 10 * i, d)).sum() for i, o in enumerate(outs))
        loss.backward()
        pre = f"chain{ci}."
        blob[pre + "outs"] = npy(torch.stack(outs))
        blob[pre + "num"] = npy(cell.state_num); blob[pre + "den"] = npy(cell.state_den)
        blob[pre + "var"] = npy(cell.state_var); blob[pre + "radius"] = npy(cell.radius)
        blob[pre + "g.xs"] = npy(xs.grad)
        for pname, lin in (("Wr", cell.r_lin), ("Wk", cell.k_lin), ("Wv", cell.v_lin), ("Wo", cell.o_lin)):
            blob[pre + "g." + pname] = npy(lin.weight.grad)
        blob[pre + "g.norm_w"] = npy(cell.norm.weight.grad)
        blob[pre + "g.raw_decay"] = npy(cell.raw_decay.grad)
        if afd:
            blob[pre + "g.pn_param"] = npy(cell.process_noise_param.grad)
            blob[pre + "g.on_param"] = npy(cell.obs_noise_param.grad)
        if rt:
            blob[pre + "g.rpn_param"] = npy(cell.radius_process_noise_param.grad)
            blob[pre + "g.ron_param"] = npy(cell.radius_obs_noise_param.grad)
    save("chain.npz", blob)


# ------------------------------------------------------------------ F5: KWTA
def kwta_fixture():
    blob = {}
    for i, (d, k) in enumerate(gi.KWTA_CASES):
        x = gi.kwta_rows(7000 + i, d)
        blob[f"hard.{d}.{k}"] = npy(KWTA(x, k))
        blob[f"soft.{d}.{k}"] = npy(KWTA(x, k, soft=True, temp=0.7))
    blob["zero_k"] = npy(KWTA(gi.kwta_rows(7100, 32), 0))
    save("kwta.npz", blob)


# ------------------------------------------------------------------ F6: gate
def gate_fixture():
    blob = {}
    for R in gi.GATE_R:
        nb = hex_neighbors(R)
        io = {"sensor": 0, "motor": R - 1}
        gate = Gate(R, nb, io)
        gate.gain_ema.copy_(gi.det(8000 + R, R, scale=0.2))
        gate.homeo.copy_(gi.det(8001 + R, R, scale=0.05))
        H = gi.det(8002 + R, R, 16)
        H[R // 2:] = 0.0  # half the regions silent, like a real inner step
        for fi, (focus_any, vb, um) in enumerate([(True, 0.13, 0.41), (False, -0.2, 0.0)]):
            gate.value_bias = vb
            focus = torch.zeros(5, dtype=torch.bool)
            focus[2] = focus_any
            sc = gate.score_regions(H, focus, um)
            blob[f"scores.{R}.{fi}"] = npy(sc)
            for k_active in (2, 4, 8, 32):
                for burst in (0, max(1, R // 8)):
                    m = gate.select_k(sc, k_active, burst)
                    blob[f"mask.{R}.{fi}.{k_active}.{burst}"] = npy(m)
        # exact-tie inputs: record the reference's (unstable) walk for documentation
        z = torch.zeros(R)
        blob[f"tie_zero.{R}"] = np.array(nms_topk(z, 6, nb))
        z2 = z.clone(); z2[0] = 0.5; z2[R - 1] = 0.5
        blob[f"tie_two.{R}"] = np.array(nms_topk(z2, 6, nb))
    save("gate.npz", blob)


# ------------------------------------------------------------------ F7: router
def router_fixture():
    blob = {}
    for ci, (name, R, d, mode, prec) in enumerate(gi.ROUTER_CASES):
        nb = hex_neighbors(R)
        coords = hex_axial_coords(R)
        router = Router(nb, d, R, enable_precision_routed_messages=prec, edge_transform_mode=mode)
        E = len(router.edge_keys)
        t = gi.router_inputs(9000 + 100 * ci, R, d, E, mode)
        with torch.no_grad():
            router.W_reg.copy_(t["W_reg"]); router.beta_cos.copy_(t["beta_cos"]); router.beta_sin.copy_(t["beta_sin"])
            for r in range(R):
                router.query_lin[r].weight.copy_(t["query_W"][r])
            if mode == "per_edge":
                for e, k in enumerate(router.edge_keys):
                    router.W_edge[k].weight.copy_(t["W_edge"][e]); router.key_lin[k].weight.copy_(t["key_W"][e])
                    router.raw_P_edge[k].copy_(t["raw_P"][e])
            elif mode == "by_direction":
                for i in range(6):
                    router.W_dir[i].weight.copy_(t["W_dir"][i]); router.key_lin_dir[i].weight.copy_(t["key_W"][i])
                    router.raw_P_dir[i].copy_(t["raw_P"][i])
            else:
                router.U.weight.copy_(t["U"]); router.V.weight.copy_(t["V"])
                for e, k in enumerate(router.edge_keys):
                    router.w_edge[k].copy_(t["w_edge"][e]); router.key_lin[k].weight.copy_(t["key_W"][e])
                    router.raw_P_edge[k].copy_(t["raw_P"][e])
        H = t["H"].clone().requires_grad_(True)
        M = router.messages(H, t["mask_prev"], coords)
        (M * t["g_M"]).sum().backward()
        pre = name + "."
        blob[pre + "M"] = npy(M)
        blob[pre + "g.H"] = npy(H.grad)
        w = torch.zeros(E)
        for e, k in enumerate(router.edge_keys):
            if k in router.last_weights:
                w[e] = router.last_weights[k]
        blob[pre + "weights"] = npy(w)
        blob[pre + "g.beta_cos"] = npy(router.beta_cos.grad); blob[pre + "g.beta_sin"] = npy(router.beta_sin.grad)
        zeros = torch.zeros(d, d)
        if mode == "per_edge":
            blob[pre + "g.W_edge"] = npy(torch.stack([router.W_edge[k].weight.grad if router.W_edge[k].weight.grad is not None else zeros for k in router.edge_keys]))
        elif mode == "by_direction":
            blob[pre + "g.W_dir"] = npy(torch.stack([l.weight.grad if l.weight.grad is not None else zeros for l in router.W_dir]))
        else:
            blob[pre + "g.U"] = npy(router.U.weight.grad); blob[pre + "g.V"] = npy(router.V.weight.grad)
            blob[pre + "g.w_edge"] = npy(torch.stack([router.w_edge[k].grad if router.w_edge[k].grad is not None else torch.zeros(d) for k in router.edge_keys]))
        if prec:
            blob[pre + "g.query_W"] = npy(torch.stack([l.weight.grad if l.weight.grad is not None else zeros for l in router.query_lin]))
            if mode == "by_direction":
                blob[pre + "g.key_W"] = npy(torch.stack([l.weight.grad if l.weight.grad is not None else zeros for l in router.key_lin_dir]))
                blob[pre + "g.raw_P"] = npy(torch.stack([p.grad if p.grad is not None else torch.zeros(d) for p in router.raw_P_dir]))
            else:
                blob[pre + "g.key_W"] = npy(torch.stack([router.key_lin[k].weight.grad if router.key_lin[k].weight.grad is not None else zeros for k in router.edge_keys]))
                blob[pre + "g.raw_P"] = npy(torch.stack([router.raw_P_edge[k].grad if router.raw_P_edge[k].grad is not None else torch.zeros(d) for k in router.edge_keys]))
    # reference known-answer: precision routing down-weights the inconsistent neighbour
    # (tests/test_router_precision.py:46-65)
    R, d = 3, 4
    router = Router([[1], [0, 2], [1]], d, R, enable_precision_routed_messages=True)
    router.fb_alpha = 0.0
    for r in range(R):
        router.query_lin[r].weight.data = torch.eye(d)
    for lin in router.key_lin.values():
        lin.weight.data = torch.eye(d)
    for lin in router.W_edge.values():
        lin.weight.data = torch.eye(d) * 0.5
    for p in router.raw_P_edge.values():
        p.data.fill_(1.0)
    H = torch.zeros(R, d); H[2] = 5.0; H[0] = 0.25
    M = router.messages(H, torch.ones(R, dtype=torch.bool), torch.zeros(R, 2))
    blob["line3.M"] = npy(M)
    blob["line3.w01"] = np.array(router.last_weights["0->1"]); blob["line3.w21"] = np.array(router.last_weights["2->1"])
    save("router.npz", blob)


# ------------------------------------------------------------------ F8: wiring
def wiring_fixture():
    blob = {}
    for R in (1, 2, 7, 10, 36, 64, 128, 256, 512):
        blob[f"coords.{R}"] = npy(hex_axial_coords(R))
        nb = hex_neighbors(R)
        flat = np.full((R, 6), -1, dtype=np.int32)
        for i, row in enumerate(nb):
            flat[i, : len(row)] = row
        blob[f"nbr.{R}"] = flat
    save("wiring.npz", blob)


# ------------------------------------------------------------------ F9/F10/F11: whole model
MODEL_CASES = {
    "default": dict(R=7, d=32, V=32, K_inner=3, B_br=2, k_active=4, max_T=64),
    "det": dict(R=7, d=32, V=32, K_inner=4, B_br=2, k_active=4, max_T=64, train_deterministic_inner_loop=True),
    "allflags": dict(R=10, d=32, V=32, K_inner=3, B_br=2, k_active=5, max_T=64,
                     enable_adaptive_filter_dynamics=True, enable_precision_routed_messages=True,
                     enable_radial_tangential_updates=True, enable_afa_attention=True,
                     enable_ff_energy_alignment=True, surprise_lambda=0.05, tau_kappa=0.01),
}
# corners of the configuration space (same fixture layout; `corner` on the command line)
CORNER_CASES = {
    "kactive2": dict(R=7, d=32, V=32, K_inner=3, B_br=2, k_active=2, max_T=64),  # k_total = 0 (utils.py:118-120)
    "softkwta": dict(R=7, d=32, V=32, K_inner=3, B_br=2, k_active=4, max_T=64, kwta_soft_mode=True, kwta_k_start=12,
                     kwta_k_schedule=10, kwta_soft_temp=0.5),
    "nopred_nokwta": dict(R=7, d=32, V=32, K_inner=3, B_br=2, k_active=4, max_T=64, use_predictive_trace=False,
                          disable_kwta_during_gating=True),
    "afdvec": dict(R=10, d=32, V=32, K_inner=3, B_br=2, k_active=5, max_T=64, enable_adaptive_filter_dynamics=True,
                   afd_noise_mode="vector", enable_radial_tangential_updates=True, kwta_k=6, router_vectorized=False),
    "noff_surprise": dict(R=7, d=32, V=32, K_inner=3, B_br=2, k_active=4, max_T=64, enable_forward_forward=False,
                          enable_adaptive_filter_dynamics=True, enable_ff_energy_alignment=True, surprise_lambda=0.1,
                          surprise_lambda_schedule=4, init_decay=0.5),
}
MODEL_B, MODEL_T = 2, 24


def corner_fixture():
    model_fixture(CORNER_CASES)


def shared_fixture():
    """by_direction / factorized router modes through the whole model."""
    import random
    from ironcortex import CortexConfig, CortexReasoner, LossWeights, train_step
    from tests.model_cases import SHARED_CASES, SHARED_RAISE
    model_fixture(SHARED_CASES)
    # forward-forward loss on: the reference's Hebbian block raises (training.py:240-243) -- record when, and what the
    # step had already changed by then (gain_ema / tau are updated before the backward, homeo and the optimizer after)
    blob = {}
    for name, kw in SHARED_RAISE.items():
        torch.manual_seed(11); random.seed(11)
        cfg = CortexConfig(**kw)
        model = CortexReasoner(hex_neighbors(cfg.R), hex_axial_coords(cfg.R), {"sensor": 0, "motor": cfg.R - 1}, cfg)
        for k, v in model.state_dict().items():
            blob[f"{name}.sd.{k}"] = npy(v)
        tokens = (gi.det(31, MODEL_B, MODEL_T, kind="rand") * (cfg.V - 1)).long()
        opt = torch.optim.AdamW(model.parameters(), lr=1e-3)
        torch.manual_seed(77); random.seed(77)
        raised_at, err = -1, ""
        for step in range(3):
            try:
                out = train_step(model, opt, tokens, LossWeights(), torch.device("cpu"))
                for k, v in out.items():
                    blob[f"{name}.step{step}.{k}"] = np.array(v, dtype=np.float64)
            except AttributeError as e:
                raised_at, err = step, type(e).__name__
                break
        blob[f"{name}.raised_at"] = np.array(raised_at)
        blob[f"{name}.gain_ema"] = npy(model.gate.gain_ema)
        blob[f"{name}.homeo"] = npy(model.gate.homeo)
        blob[f"{name}.tau"] = npy(torch.stack([f.tau for f in model.reg_ff]))
        print(name, "raised at step", raised_at, err)
    save("shared_raise.npz", blob)


def model_fixture(cases=None):
    import copy
    import random
    from ironcortex import CortexConfig, CortexReasoner, LossWeights, train_step
    for name, kw in (cases or MODEL_CASES).items():
        torch.manual_seed(11); random.seed(11)
        cfg = CortexConfig(**kw)
        model = CortexReasoner(hex_neighbors(cfg.R), hex_axial_coords(cfg.R), {"sensor": 0, "motor": cfg.R - 1}, cfg)
        blob = {"sd." + k: npy(v) for k, v in model.state_dict().items()}
        tokens = (gi.det(31, MODEL_B, MODEL_T, kind="rand") * (cfg.V - 1)).long()
        focus = gi.det(32, MODEL_B, MODEL_T, kind="rand") < 0.3
        probe = copy.deepcopy(model)
        with torch.no_grad():
            H, m, logits, traces, en, ent = probe.reasoning_loop_batch(tokens, cfg.K_inner, focus)
        blob["loop.H"] = npy(H); blob["loop.mask"] = npy(m); blob["loop.logits"] = npy(logits)
        blob["loop.trace0"] = npy(torch.stack(traces[0])); blob["loop.energy"] = npy(en)
        blob["loop.num"] = npy(torch.stack([r.state_num for r in probe.regions]))
        blob["loop.pred"] = npy(torch.stack([r.pred for r in probe.regions]))
        opt = torch.optim.AdamW(model.parameters(), lr=1e-3)
        torch.manual_seed(77); random.seed(77)
        for step in range(2):
            out = train_step(model, opt, tokens, LossWeights(), torch.device("cpu"))
            for k, v in out.items():
                blob[f"step{step}.{k}"] = np.array(v, dtype=np.float64)
            if step == 0:
                names = []
                for n, p in model.named_parameters():
                    if p.grad is not None:
                        names.append(n)
                        blob["gnorm." + n] = np.array(float(p.grad.norm()))
                        if p.numel() <= 4096 and (n.startswith("regions.0.") or n.startswith("regions.6.") or "W_edge.1->0" in n
                                                  or n.startswith("norm_in") or n.startswith("router.beta") or "query_lin.1." in n
                                                  or n.startswith("rtdd") or "raw_P_edge.1->0" in n or "key_lin.1->0" in n
                                                  or n.startswith(("router.W_dir", "router.U.", "router.V.", "router.key_lin_dir",
                                                                   "router.raw_P_dir")) or "w_edge.1->0" in n):
                            blob["grad." + n] = npy(p.grad)
                blob["grad_names"] = np.array(names)
        blob["final.gain_ema"] = npy(model.gate.gain_ema); blob["final.homeo"] = npy(model.gate.homeo)
        blob["final.tau"] = npy(torch.stack([f.tau for f in model.reg_ff]))
        for n, p in model.named_parameters():
            blob["pnorm." + n] = np.array(float(p.detach().norm()))
        save(f"model_{name}.npz", blob)


# ------------------------------------------------------------------ evaluate_perplexity (evaluation.py:14-52)
from tests.golden.make_eval import EVAL_T  # noqa: E402


def eval_fixture():
    """Reference evaluate_perplexity on the small whole-model cases (state_dicts of model_*.npz)."""
    from ironcortex import CortexConfig, CortexReasoner
    from ironcortex.evaluation import evaluate_perplexity
    blob = {}
    for name, kw in MODEL_CASES.items():
        gold = np.load(os.path.join(HERE, f"model_{name}.npz"))
        cfg = CortexConfig(**kw)
        torch.manual_seed(5)
        model = CortexReasoner(hex_neighbors(cfg.R), hex_axial_coords(cfg.R), {"sensor": 0, "motor": cfg.R - 1}, cfg)
        model.load_state_dict({k[3:]: torch.as_tensor(gold[k]) for k in gold.files if k.startswith("sd.")})
        tokens = (gi.det(31, MODEL_B, MODEL_T, kind="rand") * (cfg.V - 1)).long()[:, :EVAL_T]
        out = evaluate_perplexity(model, tokens, torch.device("cpu"))
        blob[name + ".cross_entropy"] = np.array(out["cross_entropy"]); blob[name + ".perplexity"] = np.array(out["perplexity"])
        blob[name + ".num"] = npy(torch.stack([r.state_num for r in model.regions]))  # state carried across the calls
    save("eval.npz", blob)


# ------------------------------------------------------------------ generation (BASELINE configs[4])
def gen_fixture():
    """Reference generate / diffusion_generate (generation.py:11-88, diffusion.py:20-77) with
    enable_afa_attention=True under tests/golden/inputs.py::DeterministicDraws: per sampling step the
    distribution handed to torch.multinomial and the sample drawn, plus the returned tokens.  Cases: the small
    all-flags model and the exact 10M bench model with the scripts' call (prompt "ROMEO:", 100 tokens, 4 diffusion
    steps; train_tiny_shakespeare.py:36-37,184-202)."""
    import random
    from ironcortex import CortexConfig, CortexReasoner
    from ironcortex.diffusion import DiffusionConfig, diffusion_generate
    from ironcortex.generation import generate
    from tests.model_cases import BIG_CASES, load_case
    blob = {}
    for name, T_total, iters in (("allflags", 16, 3), ("bench10m", 100, 8)):
        gold, cfg_amd, sd, _tok, _foc = load_case(name)
        kw = MODEL_CASES[name] if name in MODEL_CASES else BIG_CASES[name][0]
        cfg = CortexConfig(**kw)
        prompt = torch.tensor([t % (cfg.V - 2) + 1 for t in gi.GEN_PROMPT], dtype=torch.long)
        for fn_name in ("generate", "diffusion"):
            torch.manual_seed(5); random.seed(5)
            model = CortexReasoner(hex_neighbors(cfg.R), hex_axial_coords(cfg.R), {"sensor": 0, "motor": cfg.R - 1}, cfg)
            model.load_state_dict(sd)
            model.eval()
            torch.manual_seed(123)
            with gi.DeterministicDraws(900) as dd:
                if fn_name == "generate":
                    out = generate(model, prompt, T_total=T_total, max_outer_iters=iters)
                else:
                    out = diffusion_generate(model, prompt, T_total=T_total, diff_cfg=DiffusionConfig(steps=4))
            pre = f"{name}.{fn_name}."
            blob[pre + "tokens"] = npy(out)
            blob[pre + "probs"] = np.stack([p.numpy() for p in dd.probs])
            blob[pre + "samples"] = np.stack([s_.numpy() for s_ in dd.samples])
            blob[pre + "num"] = npy(torch.stack([r.state_num for r in model.regions]))
    save("generation.npz", blob)


# ------------------------------------------------------------------ F9/F10 at full size
def bigmodel_fixture():
    """The exact bench model (10M script config, T=256) and the default script config (R=36, d=256,
    T=64).  The state_dict is the reference's own initialisation with every large tensor replaced by
    a seeded uniform of the same mean / std (tests/golden/inputs.py::synth_tensor), so that it can be
    regenerated instead of stored; large gradients are stored as two probe projections."""
    import copy
    import random
    from ironcortex import CortexConfig, CortexReasoner, LossWeights, train_step
    from tests.model_cases import BIG_CASES, big_inputs, grad_probes
    for name, (kw, B, T) in BIG_CASES.items():
        torch.manual_seed(11); random.seed(11)
        cfg = CortexConfig(**kw)
        model = CortexReasoner(hex_neighbors(cfg.R), hex_axial_coords(cfg.R), {"sensor": 0, "motor": cfg.R - 1}, cfg)
        if name == "r36":  # tie-free gate scores from the first inner step on
            model.gate.gain_ema.copy_(gi.det(8100, cfg.R, scale=0.2))
            model.gate.homeo.copy_(gi.det(8101, cfg.R, scale=0.05))
        blob, new = {}, {}
        sd = model.state_dict()
        blob["sd_order"] = np.array(list(sd.keys()))
        for idx, (k, v) in enumerate(sd.items()):
            if v.is_floating_point() and v.numel() > 4096:
                if bool(((v == 0) | (v == 1)).all()):  # k-hot token codes (sdr.py:14-29)
                    blob["sdbits." + k] = np.packbits(npy(v).astype(np.uint8).reshape(-1))
                    blob["sdshape." + k] = np.array(v.shape)
                    new[k] = v.clone()
                else:
                    mean, std = float(v.mean()), float(v.std())
                    blob["sdgen." + k] = np.array([idx, mean, std, *v.shape], dtype=np.float64)
                    new[k] = gi.synth_tensor(idx, v.shape, mean, std)
            else:
                blob["sd." + k] = npy(v)
                new[k] = v.clone()
        model.load_state_dict(new)
        tokens, focus = big_inputs(name)
        probe = copy.deepcopy(model)
        with torch.no_grad():
            H, m, logits, traces, en, ent = probe.reasoning_loop_batch(tokens, cfg.K_inner, focus)
        blob["loop.H"] = npy(H); blob["loop.mask"] = npy(m); blob["loop.logits"] = npy(logits)
        blob["loop.trace0"] = npy(torch.stack(traces[0])); blob["loop.energy"] = npy(en)
        blob["loop.num"] = npy(torch.stack([r.state_num for r in probe.regions]))
        blob["loop.pred"] = npy(torch.stack([r.pred for r in probe.regions]))
        opt = torch.optim.AdamW(model.parameters(), lr=1e-3)
        torch.manual_seed(77); random.seed(77)
        for step in range(2):
            out = train_step(model, opt, tokens, LossWeights(), torch.device("cpu"))
            for k, v in out.items():
                blob[f"step{step}.{k}"] = np.array(v, dtype=np.float64)
            if step == 0:
                names = []
                for n, p in model.named_parameters():
                    if p.grad is None:
                        continue
                    names.append(n)
                    blob["gnorm." + n] = np.array(float(p.grad.norm()))
                    if p.numel() <= 4096:
                        blob["grad." + n] = npy(p.grad)
                    elif p.dim() == 2:
                        pc, pr = grad_probes(p.shape[1], p.shape[0])
                        blob["gp." + n] = npy(p.grad @ pc); blob["gq." + n] = npy(pr @ p.grad)
                blob["grad_names"] = np.array(names)
        blob["final.gain_ema"] = npy(model.gate.gain_ema); blob["final.homeo"] = npy(model.gate.homeo)
        blob["final.tau"] = npy(torch.stack([f.tau for f in model.reg_ff]))
        for n, p in model.named_parameters():
            blob["pnorm." + n] = np.array(float(p.detach().norm()))
        save(f"model_{name}.npz", blob)


if __name__ == "__main__":
    torch.manual_seed(0)
    which = sys.argv[1:] or ["region", "chain", "kwta", "gate", "router", "wiring", "model", "bigmodel", "eval", "gen"]
    for w in which:
        globals()[w + "_fixture"]()
