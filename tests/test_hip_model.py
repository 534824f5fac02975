"""GPU whole-model parity: the product CortexReasoner/train_step (HIP hot path) against the
REAL reference's recorded outputs, from the same state_dict, tokens and seeds.

Cases: three small configs (d=32) and two FULL-SIZE ones -- the exact bench model (10M script
config, R=10, d=256, T=256) and the default script config (R=36, d=256, T=64).  Tolerances are
the TOL_* constants below (a few 1e-6, set from the errors observed on MI355X and logged by
tests/parity_log.py).  Region selections and KWTA winners must be identical."""
import copy
import os

import numpy as np
import pytest
import torch

from tests import parity_log as PL
from tests.golden import inputs as gi
from tests.model_cases import BIG_CASES, CORNER_CASES, MODEL_CASES, SHARED_CASES, grad_probes, load_case, seed_train

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
KEYS = ("ff", "rtd", "denoise", "critic", "verify", "E_pos", "E_neg", "E_gap", "rtd_acc", "denoise_acc",
        "grad_norm", "active_frac", "lambda_s", "ce_last", "total")


def build(cfg, sd):
    import ironcortex_amd as ic
    model = ic.CortexReasoner(ic.hex_neighbors(cfg.R), ic.hex_axial_coords(cfg.R), {"sensor": 0, "motor": cfg.R - 1}, cfg)
    assert set(model.state_dict().keys()) == set(sd.keys())
    model.load_state_dict(sd)
    return model.to(DEV)


ALL_CASES = list(MODEL_CASES) + list(CORNER_CASES) + list(SHARED_CASES) + list(BIG_CASES)
# Tolerances (relative to the reference tensor's largest magnitude; see tests/parity_log.py).
# TOL_* are ~3x the worst error OBSERVED on MI355X over all cases (profiles/r02_parity_errors.jsonl);
# the north_star asks for "about 1e-5 relative" and every forward quantity meets it.
# Observed worst (round 2, all five cases, both host paths): states 7.0e-7, metrics 9.6e-7, gradient
# norms 1.3e-5 (verify.net.1.weight, a PyTorch MLP off the hot path; hot-path parameters <= 2e-6),
# gradient tensors / probes 5.3e-6, gain_ema 4.8e-7, tau 2.3e-6, parameter norms 1.2e-6.
TOL_STATE = 3e-6     # region states, traces, logits, energies after up to 64 chained inner steps
TOL_METRIC = 5e-6    # the 15 train_step metrics
TOL_GNORM = 4e-5     # per-parameter gradient norms
TOL_GRAD = 2e-5      # gradient tensors / probe projections
TOL_BUF = 1e-5       # gain_ema, tau after two steps
TOL_PNORM = 5e-6     # parameter norms after two AdamW steps


@pytest.mark.parametrize("engine", [True, False], ids=["engine", "general"])
@pytest.mark.parametrize("name", ALL_CASES)
def test_reasoning_loop_vs_reference(name, engine):
    """engine: the LoopEngine (ironcortex_amd/engine.py); general: one autograd Function per call."""
    gold, cfg, sd, tokens, focus = load_case(name)
    model = build(cfg, sd)
    model.engine.enabled = engine
    with torch.no_grad():
        H, m, logits, traces, en, ent = model.reasoning_loop_batch(tokens.to(DEV), cfg.K_inner, focus.to(DEV))
    assert np.array_equal(m.cpu().numpy(), gold["loop.mask"])
    assert np.array_equal(H.cpu().numpy() != 0, gold["loop.H"] != 0)
    tag = f"loop/{name}/{'engine' if engine else 'general'}"
    PL.check(tag, "H", H, gold["loop.H"], TOL_STATE); PL.check(tag, "logits", logits, gold["loop.logits"], TOL_STATE)
    PL.check(tag, "trace0", torch.stack(traces[0]), gold["loop.trace0"], TOL_STATE)
    PL.check(tag, "energy", en, gold["loop.energy"], TOL_STATE)
    PL.check(tag, "num", model.bank.num[0], gold["loop.num"], TOL_STATE)
    PL.check(tag, "pred", model.bank.pred[0], gold["loop.pred"], TOL_STATE)
    # the drop-in attribute views agree with the bank
    PL.check(tag, "num", model.regions[0].state_num, gold["loop.num"][0], TOL_STATE)
    PL.flush(tag)


@pytest.mark.parametrize("engine", [True, False], ids=["engine", "general"])
@pytest.mark.parametrize("name", ALL_CASES)
def test_two_train_steps_vs_reference(name, engine):
    import ironcortex_amd as ic
    gold, cfg, sd, tokens, focus = load_case(name)
    model = build(cfg, sd)
    model.engine.enabled = engine
    opt = torch.optim.AdamW(model.parameters(), lr=1e-3)
    seed_train()
    tag = f"train/{name}/{'engine' if engine else 'general'}"
    for step in range(2):
        out = ic.train_step(model, opt, tokens.to(DEV), ic.LossWeights(), torch.device(DEV))
        for k in KEYS:
            PL.check(tag, f"metric:{step}.{k}", out[k], float(gold[f"step{step}.{k}"]), TOL_METRIC, floor=1e-5)
        if step == 0:
            names = set(gold["grad_names"].tolist())
            params = dict(model.named_parameters())
            got = {n for n, p in params.items() if p.grad is not None}
            assert got == names, (sorted(got - names)[:8], sorted(names - got)[:8])
            for n in names:
                PL.check(tag, f"gnorm:{n}", float(params[n].grad.norm()), float(gold["gnorm." + n]), TOL_GNORM, floor=1e-6)
            for k in gold.files:
                if k.startswith("grad."):
                    PL.check(tag, f"grad:{k[5:]}", params[k[5:]].grad, gold[k], TOL_GRAD, floor=1e-6)
                elif k.startswith("gp."):
                    g = params[k[3:]].grad.cpu()
                    pc, pr = grad_probes(g.shape[1], g.shape[0])
                    PL.check(tag, f"gprobe:{k[3:]}", g @ pc, gold[k], TOL_GRAD, floor=1e-6)
                    PL.check(tag, f"gprobe:{k[3:]}", pr @ g, gold["gq." + k[3:]], TOL_GRAD, floor=1e-6)
    PL.check(tag, "gain_ema", model.gate.gain_ema, gold["final.gain_ema"], TOL_BUF)
    PL.check(tag, "homeo", model.gate.homeo, gold["final.homeo"], 1e-5)
    PL.check(tag, "tau", torch.stack([f.tau for f in model.reg_ff]), gold["final.tau"], TOL_BUF)
    for n, p in model.named_parameters():
        # a gradient that is analytically zero (e.g. the key bias of a softmax attention) is pure
        # rounding noise; Adam turns its SIGN into a +-lr step, so such parameters are not comparable
        if ("gnorm." + n) in gold.files and float(gold["gnorm." + n]) < 1e-6:
            continue
        # c_attn.bias = [q | k | v] biases: the KEY third has an analytically zero gradient (softmax is invariant to a
        # shift of all scores of a query), so its Adam steps are ratios of pure rounding noise (|step| anywhere in
        # [0, lr]); the norm of the whole vector is compared, with an absolute floor that covers that third
        floor = 1e-5 if n.endswith("c_attn.bias") else 1e-6
        PL.check(tag, f"pnorm:{n}", float(p.detach().norm()), float(gold["pnorm." + n]), TOL_PNORM, floor=floor)
    PL.flush(tag)


@pytest.mark.parametrize("name", ["bydir", "factorized"])
def test_shared_modes_engine_backward_on_mfma_dgrad(name):
    """The LoopEngine's batched backward addresses its streams through in_slot / sk; the shared-matrix modes' input
    gradient must run on the MFMA GEMM there too (k_router_shared_dgrad_mfma), not on the per-edge transposed GEMVs +
    gather (the gradients themselves are compared with the reference's in test_two_train_steps_vs_reference)."""
    import ironcortex_amd as ic
    from ironcortex_amd import hip
    if name not in SHARED_CASES:
        pytest.skip(f"no case {name}")
    gold, cfg, sd, tokens, focus = load_case(name)
    model = build(cfg, sd)
    opt = torch.optim.AdamW(model.parameters(), lr=1e-3)
    seed_train()
    hip.prof_select("")
    hip.prof_enable(True)
    hip.prof_read()
    try:
        ic.train_step(model, opt, tokens.to(DEV), ic.LossWeights(), torch.device(DEV))
        torch.cuda.synchronize()
        prof = hip.prof_read()
    finally:
        hip.prof_enable(False)
    assert prof.get("router_mfma", (0, 0))[0] > 0, sorted(prof)
    assert "router_mat_bwd" not in prof and "router_gather_bwd" not in prof, sorted(prof)


def test_checkpoint_roundtrip(tmp_path):
    """reference tests/test_checkpoint.py: bit-equal state after save/load, keys unchanged."""
    import ironcortex_amd as ic
    gold, cfg, sd, tokens, focus = load_case("default")
    model = build(cfg, sd)
    opt = torch.optim.AdamW(model.parameters(), lr=1e-3)
    seed_train()
    ic.train_step(model, opt, tokens.to(DEV), ic.LossWeights(), torch.device(DEV))
    path = str(tmp_path / "ck.pt")
    ic.save_checkpoint(model, opt, 3, path)
    model2 = build(cfg, sd)
    opt2 = torch.optim.AdamW(model2.parameters(), lr=1e-3)
    # the reference's call (train_tiny_shakespeare.py:95, tests/test_checkpoint.py:35)
    assert ic.load_checkpoint(path, model2, opt2, map_location=torch.device(DEV)) == 3
    assert ic.load_checkpoint(path) == 3  # model and optimizer are optional (checkpoint.py:37-42)
    for (k, a), (_, b) in zip(model.state_dict().items(), model2.state_dict().items()):
        assert torch.equal(a, b), k
    # the arena layout survived load_state_dict: kernels see the loaded weights
    assert model2.regions[1].r_lin.weight.data_ptr() == model2.bank.flat.stacked("W")[4].data_ptr()


@pytest.mark.parametrize("name", list(MODEL_CASES))
def test_evaluate_perplexity_vs_reference(name):
    """reference evaluation.py:14-52 run by tests/golden/make_fixtures.py::eval_fixture: same dict,
    same values, region state carried across the T-1 calls the same way; ValueError for T < 2."""
    import ironcortex_amd as ic
    from tests.golden.make_eval import EVAL_T
    gold, cfg, sd, tokens, focus = load_case(name)
    ev = np.load(os.path.join(os.path.dirname(__file__), "golden", "eval.npz"))
    model = build(cfg, sd)
    out = ic.evaluate_perplexity(model, tokens[:, :EVAL_T], torch.device(DEV))
    assert set(out) == {"cross_entropy", "perplexity"}
    tag = f"eval/{name}"
    PL.check(tag, "cross_entropy", out["cross_entropy"], float(ev[name + ".cross_entropy"]), TOL_METRIC)
    PL.check(tag, "perplexity", out["perplexity"], float(ev[name + ".perplexity"]), TOL_METRIC)
    PL.check(tag, "num", model.bank.num[0], ev[name + ".num"], TOL_STATE)
    PL.flush(tag)
    with pytest.raises(ValueError):
        ic.evaluate_perplexity(model, tokens[:, :1], torch.device(DEV))


@pytest.mark.parametrize("engine", [True, False], ids=["engine", "general"])
@pytest.mark.parametrize("fn_name", ["generate", "diffusion"])
@pytest.mark.parametrize("name,T_total,iters", [("allflags", 16, 3), ("bench10m", 100, 8)])
def test_generation_vs_reference(name, T_total, iters, fn_name, engine):
    """BASELINE configs[4]: generate / diffusion_generate with enable_afa_attention=True against the REAL
    reference's run (tests/golden/make_fixtures.py::gen_fixture): at every sampling step the distribution
    handed to torch.multinomial (the sample itself is replayed from the reference run, all other draws come
    from the same seeded CPU generator: tests/golden/inputs.py::DeterministicDraws), the number of steps, the
    returned tokens and the region state left behind.  reference generation.py:11-88, diffusion.py:20-77."""
    import ironcortex_amd as ic
    gold, cfg, sd, _tok, _foc = load_case(name)
    gen = np.load(os.path.join(os.path.dirname(__file__), "golden", "generation.npz"))
    pre = f"{name}.{fn_name}."
    model = build(cfg, sd)
    model.engine.enabled = engine
    model.eval()
    prompt = torch.tensor([t % (cfg.V - 2) + 1 for t in gi.GEN_PROMPT], dtype=torch.long, device=DEV)
    torch.manual_seed(123)
    with gi.DeterministicDraws(900, replay=gen[pre + "samples"]) as dd:
        if fn_name == "generate":
            out = ic.generate(model, prompt, T_total=T_total, max_outer_iters=iters)
        else:
            out = ic.diffusion_generate(model, prompt, T_total=T_total, diff_cfg=ic.DiffusionConfig(steps=4))
    ref_probs = gen[pre + "probs"]
    assert len(dd.probs) == ref_probs.shape[0], "number of sampling steps differs"
    tag = f"generation/{name}/{fn_name}/{'engine' if engine else 'general'}"
    for i, p in enumerate(dd.probs):
        PL.check(tag, f"probs:step{i}", p, ref_probs[i], TOL_STATE)
    assert np.array_equal(out.cpu().numpy(), gen[pre + "tokens"])
    PL.check(tag, "num", model.bank.num[0], gen[pre + "num"], TOL_STATE)
    PL.flush(tag)


def test_generation_smoke():
    """reference tests/test_smoke.py:19-44: ranges only."""
    import ironcortex_amd as ic
    gold, cfg, sd, tokens, focus = load_case("allflags")
    model = build(cfg, sd)
    prompt = torch.tensor([3, 4, 5], device=DEV)
    out = ic.generate(model, prompt, T_total=12, max_outer_iters=2)
    assert out.shape == (12,) and int(out.min()) >= 0 and int(out.max()) < cfg.V
    out2 = ic.diffusion_generate(model, prompt, T_total=12, diff_cfg=ic.DiffusionConfig(steps=2))
    assert out2.shape == (12,) and torch.equal(out2[:3], prompt)
    t = model.telemetry()
    assert "regions" in t and len(t["regions"]) == cfg.R


@pytest.mark.parametrize("name", list(MODEL_CASES))
def test_batched_backward_equals_loop_by_loop(name):
    """train_step differentiates all its loops as one node, step t of every loop as the streams of
    one launch (engine.backward_batch).  Same gradients as replaying the loops one after another
    (engine.backward), which is what the fallback does; and the batched path is the one taken."""
    import ironcortex_amd as ic
    gold, cfg, sd, tokens, focus = load_case(name)
    grads, metrics = [], []
    for force_loops in (False, True):
        model = build(cfg, sd)
        if force_loops:
            model.engine._batch_uniform = lambda bt: False
        opt = torch.optim.SGD(model.parameters(), lr=0.0)
        seed_train()
        out = ic.train_step(model, opt, tokens.to(DEV), ic.LossWeights(), torch.device(DEV))
        c = model.engine.counters
        if cfg.train_deterministic_inner_loop or cfg.B_br >= 1:
            assert (c["loop_bwd"] > 0) == force_loops and (c["batch_bwd"] > 0) == (not force_loops), c
        grads.append({n: (None if p.grad is None else p.grad.clone()) for n, p in model.named_parameters()})
        metrics.append(out)
    assert {n for n, g in grads[0].items() if g is not None} == {n for n, g in grads[1].items() if g is not None}
    for n, g in grads[0].items():
        if g is not None:  # same kernels per (loop, step); only the order of the parameter-gradient sums differs
            torch.testing.assert_close(g, grads[1][n], rtol=1e-4, atol=1e-6 * max(1.0, float(grads[1][n].abs().max())), msg=n)
    for k in KEYS:
        assert metrics[0][k] == pytest.approx(metrics[1][k], rel=1e-5, abs=1e-6), k


def test_train_step_recovers_after_a_failed_backward():
    """A backward pass that raises must not poison the next train_step (GradCoordinator
    kept _pending set, so later passes skipped begin_pass() and never published gradients)."""
    import ironcortex_amd as ic
    gold, cfg, sd, tokens, focus = load_case("allflags")
    model = build(cfg, sd)
    opt = torch.optim.SGD(model.parameters(), lr=0.0)
    eng = model.engine
    orig = eng.backward_batch

    def boom(bt, gH):
        eng.model._coord.begin()  # the pass has started when the failure comes
        raise RuntimeError("injected")

    eng.backward_batch = boom
    seed_train()
    with pytest.raises(RuntimeError, match="injected"):
        ic.train_step(model, opt, tokens.to(DEV), ic.LossWeights(), torch.device(DEV))
    assert model._coord._pending  # the stale flag the advisor described
    eng.backward_batch = orig
    out = ic.train_step(model, opt, tokens.to(DEV), ic.LossWeights(), torch.device(DEV))
    assert not model._coord._pending and model.bank.tape.pos == 0 and eng._pass is None
    got = {n for n, p in model.named_parameters() if p.grad is not None}
    assert {"regions.0.r_lin.weight", "regions.0.norm.weight", "norm_in.weight", "router.beta_cos"} <= got
    assert float(model.regions[0].r_lin.weight.grad.abs().sum()) > 0 and out["grad_norm"] > 0
    # and the step after that equals the same step of a model that never failed
    ref = build(cfg, sd)
    for k in ("num", "den", "var", "pred", "radius", "radius_var", "surprise", "dt"):
        setattr(ref.bank, k, getattr(model.bank, k).clone())
    ref.bank.S = model.bank.S
    ref.work.slots = model.work.slots.clone()
    ref.load_state_dict(model.state_dict())
    for a, b in zip(ref.reg_ff, model.reg_ff):
        a.tau = b.tau.clone()
    ref._train_step, ref._lambda_s = model._train_step, model._lambda_s
    for k in ("E_pos_mean", "E_neg_mean"):  # non-persistent buffers
        if hasattr(model, k):
            setattr(ref, k, getattr(model, k).clone())
    opt2 = torch.optim.SGD(ref.parameters(), lr=0.0)
    outs = []
    for mdl, o in ((model, opt), (ref, opt2)):
        seed_train()
        outs.append(ic.train_step(mdl, o, tokens.to(DEV), ic.LossWeights(), torch.device(DEV)))
    for k in KEYS:
        assert outs[0][k] == outs[1][k], k
    for (n, p), (_, q) in zip(model.named_parameters(), ref.named_parameters()):
        assert (p.grad is None) == (q.grad is None) and (p.grad is None or torch.equal(p.grad, q.grad)), n


def test_early_halt_engine_equals_general_path(monkeypatch):
    """Loops that halt early (utils.py:147-151; forced here at k >= 1 of K_inner = 3): the engine and the general
    path agree, the planner is not run for an iteration that will not happen, and train_step still
    differentiates all loops at once (a halted loop leaves its final state in the last slot of its
    block, where the next loop's first step looks for it)."""
    import ironcortex_amd as ic
    from ironcortex_amd import engine as E, model as M
    halt = lambda scores, u_mean, k, K_inner: k >= 1
    monkeypatch.setattr(E, "should_halt", halt)
    monkeypatch.setattr(M, "should_halt", halt)
    monkeypatch.setattr(E, "STEER", False)  # a patched halting rule only exists on the host (device-steered halting: test_hip_steer.py)
    gold, cfg, sd, tokens, focus = load_case("allflags")
    outs = []
    for engine in (True, False):
        model = build(cfg, sd)
        model.engine.enabled = engine
        with torch.no_grad():
            H, m, logits, traces, en, ent = model.reasoning_loop_batch(tokens.to(DEV), cfg.K_inner, focus.to(DEV))
        outs.append((H, m, logits, [len(t) for t in traces]))
        assert cfg.K_inner == 3 and model.counters["inner_steps"] == tokens.shape[0] * 2 * cfg.B_br  # k = 0, 1
    assert torch.equal(outs[0][1], outs[1][1]) and outs[0][3] == outs[1][3]
    torch.testing.assert_close(outs[0][0], outs[1][0], rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(outs[0][2], outs[1][2], rtol=1e-4, atol=1e-5)
    grads = []
    for engine in (True, False):
        model = build(cfg, sd)
        model.engine.enabled = engine
        opt = torch.optim.SGD(model.parameters(), lr=0.0)
        seed_train()
        out = ic.train_step(model, opt, tokens.to(DEV), ic.LossWeights(), torch.device(DEV))
        if engine:
            assert model.engine.counters["batch_bwd"] > 0 and model.engine.counters["loop_bwd"] == 0
        grads.append(({n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}, out))
    assert set(grads[0][0]) == set(grads[1][0])
    for n, g in grads[0][0].items():
        torch.testing.assert_close(g, grads[1][0][n], rtol=2e-3, atol=2e-5 * max(1.0, float(g.abs().max())), msg=n)
    for k in KEYS:
        assert grads[0][1][k] == pytest.approx(grads[1][1][k], rel=1e-3, abs=1e-4), k


@pytest.mark.parametrize("name", ["default", "allflags"])
def test_deferred_winner_equals_immediate(name, monkeypatch):
    """A recorded loop of a batch reads its winner after the next loop's first launch is queued
    (LoopEngine.resolve_pending) instead of waiting for its own last scores: same outputs, masks,
    traces, losses and gradients, bit for bit, as with the wait (IC_DEFER_SYNC=0), and the
    deferral is the path taken."""
    import ironcortex_amd as ic
    from ironcortex_amd import engine as E
    gold, cfg, sd, tokens, focus = load_case(name)
    res = []
    monkeypatch.setattr(E, "STEER", False)  # (the host-steered loops: device-steered ones never wait at all)
    for defer in (True, False):
        monkeypatch.setattr(E, "DEFER_SYNC", defer)
        model = build(cfg, sd)
        n_resolved = [0]
        orig = model.engine.resolve_pending

        def counted(orig=orig, n=n_resolved, eng=model.engine):
            n[0] += eng._pending is not None
            orig()

        model.engine.resolve_pending = counted
        opt = torch.optim.SGD(model.parameters(), lr=0.0)
        seed_train()
        out = ic.train_step(model, opt, tokens.to(DEV), ic.LossWeights(), torch.device(DEV))
        assert (n_resolved[0] > 0) == defer, n_resolved
        assert model.engine._pending is None
        grads = {n: (None if p.grad is None else p.grad.clone()) for n, p in model.named_parameters()}
        model.zero_grad()
        seed_train()
        H, m, logits, traces, en, ent = model.reasoning_loop_batch(tokens.to(DEV), cfg.K_inner, focus.to(DEV))
        res.append((out, grads, H.detach().clone(), m.clone(), logits.detach().clone(), [[t.clone() for t in tr] for tr in traces]))
    (o0, g0, H0, m0, l0, t0), (o1, g1, H1, m1, l1, t1) = res
    for k in KEYS:
        assert o0[k] == o1[k], k
    for n in g0:
        assert (g0[n] is None) == (g1[n] is None) and (g0[n] is None or torch.equal(g0[n], g1[n])), n
    assert torch.equal(H0, H1) and torch.equal(m0, m1) and m0.dtype == torch.bool and torch.equal(l0, l1)
    assert [len(t) for t in t0] == [len(t) for t in t1]
    for a, b in zip(t0, t1):
        for x, y in zip(a, b):
            assert torch.equal(x, y)


def test_train_trajectory_equals_oracle_over_six_steps():
    """Beyond the two-step fixtures: six consecutive train_steps of the 10M config (all flags on; B=2, T=64) from one
    seed against the oracle model on the CPU (itself pinned to the real reference's two-step fixtures by
    test_oracle_model.py) -- same weights, tokens and random streams.  Every step's metrics agree to 1e-4 (observed:
    <= 2e-5 over the first eleven steps, scripts/trajectory.py / profiles/r02c_trajectory_40steps.txt; after that a
    selection flips somewhere and the two runs part, as any two fp32 runs of these dynamics do)."""
    import random

    import ironcortex_amd as ic
    from oracle import model as OM
    gold, cfg, sd, tokens, focus = load_case("bench10m")
    tokens = tokens[:2, :64]
    om = OM.build(cfg)
    om.load_state_dict(sd)
    model = build(cfg, sd)
    runs = []
    for which in ("oracle", "hip"):
        torch.manual_seed(1)
        random.seed(1)
        if which == "oracle":
            opt = torch.optim.AdamW(om.parameters(), lr=1e-3)
            runs.append([OM.oracle_train_step(om, opt, tokens, OM.LossWeights()) for _ in range(6)])
        else:
            opt = torch.optim.AdamW(model.parameters(), lr=1e-3)
            runs.append([ic.train_step(model, opt, tokens.to(DEV), ic.LossWeights(), torch.device(DEV)) for _ in range(6)])
    tag = "train/trajectory6"
    for step, (o, h) in enumerate(zip(*runs)):
        for k in KEYS:
            PL.check(tag, f"metric:{step}.{k}", h[k], float(o[k]), 1e-4, floor=1e-4)
    PL.flush(tag)


@pytest.mark.parametrize("kw,T", [
    (dict(R=7, d=32, V=32, K_inner=3, B_br=2, k_active=2, max_T=64), 24),    # k_total = 0: nms_topk still picks one (utils.py:118-120; reference tests/test_smoke.py:22)
    (dict(R=4, d=32, V=32, K_inner=3, B_br=2, k_active=4, max_T=64), 24),    # a 4-region map: every region selected or suppressed
    (dict(R=7, d=32, V=32, K_inner=2, B_br=1, k_active=4, max_T=64), 2),     # two-token sequences, one branch
    (dict(R=10, d=64, V=32, K_inner=3, B_br=2, k_active=5, max_T=64, enable_afa_attention=True,
          enable_precision_routed_messages=True), 17),                      # odd length, 16-wide attention heads, precision routing
    (dict(R=7, d=32, V=32, K_inner=3, B_br=2, k_active=4, max_T=64, kwta_soft_mode=True, kwta_k_start=12,
          kwta_k_schedule=10, kwta_soft_temp=0.5), 24),                      # soft KWTA warm-up with a winner-count schedule (region.py:238-248)
    (dict(R=7, d=32, V=32, K_inner=3, B_br=2, k_active=4, max_T=64, use_predictive_trace=False), 24),
    (dict(R=7, d=32, V=32, K_inner=3, B_br=2, k_active=4, max_T=64, disable_kwta_during_gating=True), 24),
], ids=["k_active2", "R4", "T2_onebranch", "d64_afa_precision", "kwta_soft_schedule", "no_pred_trace", "no_kwta"])
def test_corner_configs_two_train_steps_equal_oracle(kw, T):
    """Corners the fixtures do not hold, checked the way test_train_trajectory_* is: two train_steps of the product
    against the oracle model from the same weights, tokens and random streams (metrics to 1e-4)."""
    import random

    import ironcortex_amd as ic
    from ironcortex_amd.config import CortexConfig
    from oracle import model as OM
    cfg = CortexConfig(**kw)
    torch.manual_seed(11)
    random.seed(11)
    om = OM.build(cfg)
    with torch.no_grad():  # tie-free gate buffers (exact ties follow the CPU sort's order)
        om.gate.gain_ema.copy_(gi.det(61, cfg.R, scale=0.1))
        om.gate.homeo.copy_(gi.det(62, cfg.R, scale=0.1).abs())
    sd = {k: v.clone() for k, v in om.state_dict().items()}
    model = build(cfg, sd)
    tokens = (gi.det(63, 2, T, kind="rand") * (cfg.V - 1)).long()
    runs = []
    for which in ("oracle", "hip"):
        torch.manual_seed(2)
        random.seed(2)
        if which == "oracle":
            opt = torch.optim.AdamW(om.parameters(), lr=1e-3)
            runs.append([OM.oracle_train_step(om, opt, tokens, OM.LossWeights()) for _ in range(2)])
        else:
            opt = torch.optim.AdamW(model.parameters(), lr=1e-3)
            runs.append([ic.train_step(model, opt, tokens.to(DEV), ic.LossWeights(), torch.device(DEV)) for _ in range(2)])
    tag = "train/corner/" + "_".join(f"{k}{v}" for k, v in kw.items() if k in ("R", "k_active", "B_br")) + f"_T{T}"
    for step, (o, h) in enumerate(zip(*runs)):
        for k in KEYS:
            # step 1 runs on parameters that AdamW moved by lr * g / sqrt(v): for a parameter whose step-0 gradient is
            # small, the fp32 summation order of that gradient decides the step (observed after the round-3 change of
            # ic_linear_bwd's input-gradient order: 1.1e-4 absolute on E_neg of the no_kwta corner, 1e-5 before)
            PL.check(tag, f"metric:{step}.{k}", h[k], float(o[k]), 1e-4, floor=1e-4 if step == 0 else 2.5e-4)
    PL.flush(tag)


@pytest.mark.parametrize("engine", [True, False], ids=["engine", "general"])
@pytest.mark.parametrize("name", ["bydir_ff", "factorized_ff"])
def test_shared_modes_raise_where_the_reference_does(name, engine):
    """Forward-forward loss on with a shared-matrix router: the reference's Hebbian block indexes router.W_edge
    (training.py:240-243), which these modes do not define -- AttributeError at the first step with an update due
    (step 1 in the reference's recorded run, tests/golden/shared_raise.npz), after that step's gain / tau updates,
    before homeostasis and the optimizer."""
    import random
    import ironcortex_amd as ic
    from ironcortex_amd.config import CortexConfig
    from tests.model_cases import GOLD, MODEL_B, MODEL_T, SHARED_RAISE
    gold = np.load(os.path.join(GOLD, "shared_raise.npz"))
    cfg = CortexConfig(**SHARED_RAISE[name])
    sd = {k[len(name) + 4:]: torch.as_tensor(gold[k]) for k in gold.files if k.startswith(name + ".sd.")}
    model = build(cfg, sd)
    model.engine.enabled = engine
    tokens = (gi.det(31, MODEL_B, MODEL_T, kind="rand") * (cfg.V - 1)).long().to(DEV)
    opt = torch.optim.AdamW(model.parameters(), lr=1e-3)
    seed_train()
    raised_at = -1
    tag = f"train/{name}/{'engine' if engine else 'general'}"
    for step in range(3):
        try:
            out = ic.train_step(model, opt, tokens, ic.LossWeights(), torch.device(DEV))
        except AttributeError:
            raised_at = step
            break
        for k in KEYS:
            PL.check(tag, f"metric:{step}.{k}", out[k], float(gold[f"{name}.step{step}.{k}"]), TOL_METRIC, floor=1e-6)
    assert raised_at == int(gold[name + ".raised_at"]) == 1
    PL.check(tag, "gain_ema", model.gate.gain_ema, gold[name + ".gain_ema"], TOL_BUF)
    PL.check(tag, "homeo", model.gate.homeo, gold[name + ".homeo"], TOL_BUF)
    PL.check(tag, "tau", torch.stack([f.tau for f in model.reg_ff]), gold[name + ".tau"], TOL_BUF)
    PL.flush(tag)
    # the failed step must not poison the next one (the coordinator / tapes are reset by train_step)
    with pytest.raises(AttributeError):
        ic.train_step(model, opt, tokens, ic.LossWeights(), torch.device(DEV))


def test_three_branches_raise_as_in_the_reference():
    """The reference keeps the top two branches and then indexes B_br of them (model.py:388-445): B_br = 3 raises
    IndexError in its second outer iteration.  A limitation to mirror, not to fix (SURVEY §8a T2 note)."""
    import ironcortex_amd as ic
    from ironcortex_amd.config import CortexConfig
    cfg = CortexConfig(R=7, d=32, V=32, K_inner=3, B_br=3, k_active=4, max_T=64)
    torch.manual_seed(12)
    model = ic.CortexReasoner(ic.hex_neighbors(cfg.R), ic.hex_axial_coords(cfg.R), {"sensor": 0, "motor": cfg.R - 1}, cfg).to(DEV)
    tokens = (gi.det(64, 1, 16, kind="rand") * (cfg.V - 1)).long().to(DEV)
    for engine in (True, False):
        model.engine.enabled = engine
        with pytest.raises(IndexError), torch.no_grad():
            model.reasoning_loop_batch(tokens, cfg.K_inner, torch.zeros_like(tokens, dtype=torch.bool))


def test_branching_loop_without_verifier_raises_as_in_the_reference():
    """enable_energy_verifier=False leaves model.verify None and the reference's branching loop calls it
    (model.py:425): TypeError.  Both host paths raise the same type."""
    import ironcortex_amd as ic
    from ironcortex_amd.config import CortexConfig
    cfg = CortexConfig(R=7, d=32, V=32, K_inner=2, B_br=2, k_active=4, max_T=64, enable_energy_verifier=False)
    torch.manual_seed(13)
    model = ic.CortexReasoner(ic.hex_neighbors(cfg.R), ic.hex_axial_coords(cfg.R), {"sensor": 0, "motor": cfg.R - 1}, cfg).to(DEV)
    tokens = (gi.det(65, 1, 16, kind="rand") * (cfg.V - 1)).long().to(DEV)
    for engine in (True, False):
        model.engine.enabled = engine
        with pytest.raises(TypeError), torch.no_grad():
            model.reasoning_loop_batch(tokens, cfg.K_inner, torch.zeros_like(tokens, dtype=torch.bool))
