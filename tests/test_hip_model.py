"""GPU whole-model parity: the product CortexReasoner/train_step (HIP hot path) against the
REAL reference's recorded outputs, from the same state_dict, tokens and seeds.

Tolerances: states/logits rtol 1e-4 atol 1e-5; the 15 train_step metrics rel 1e-3 (they sum
thousands of fp32 terms through 24-64 chained inner steps); gradient norms rel 2e-3.
Region selections and KWTA winners must be identical."""
import copy

import numpy as np
import pytest
import torch

from tests.model_cases import MODEL_CASES, load_case, seed_train

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


@pytest.mark.parametrize("engine", [True, False], ids=["engine", "general"])
@pytest.mark.parametrize("name", list(MODEL_CASES))
def test_reasoning_loop_vs_reference(name, engine):
    """engine: the LoopEngine (ironcortex_amd/engine.py); general: one autograd Function per call."""
    gold, cfg, sd, tokens, focus = load_case(name)
    model = build(cfg, sd)
    model.engine.enabled = engine
    with torch.no_grad():
        H, m, logits, traces, en, ent = model.reasoning_loop_batch(tokens.to(DEV), cfg.K_inner, focus.to(DEV))
    assert np.array_equal(m.cpu().numpy(), gold["loop.mask"])
    assert np.array_equal(H.cpu().numpy() != 0, gold["loop.H"] != 0)
    close = lambda a, b: torch.testing.assert_close(a.cpu(), torch.as_tensor(b), rtol=1e-4, atol=1e-5)
    close(H, gold["loop.H"]); close(logits, gold["loop.logits"]); close(torch.stack(traces[0]), gold["loop.trace0"])
    close(en, gold["loop.energy"])
    close(model.bank.num[0], gold["loop.num"]); close(model.bank.pred[0], gold["loop.pred"])
    # the drop-in attribute views agree with the bank
    close(model.regions[0].state_num, gold["loop.num"][0])


@pytest.mark.parametrize("engine", [True, False], ids=["engine", "general"])
@pytest.mark.parametrize("name", list(MODEL_CASES))
def test_two_train_steps_vs_reference(name, engine):
    import ironcortex_amd as ic
    gold, cfg, sd, tokens, focus = load_case(name)
    model = build(cfg, sd)
    model.engine.enabled = engine
    opt = torch.optim.AdamW(model.parameters(), lr=1e-3)
    seed_train()
    for step in range(2):
        out = ic.train_step(model, opt, tokens.to(DEV), ic.LossWeights(), torch.device(DEV))
        for k in KEYS:
            ref = float(gold[f"step{step}.{k}"])
            assert out[k] == pytest.approx(ref, rel=1e-3, abs=1e-4), (step, k, out[k], ref)
        if step == 0:
            names = set(gold["grad_names"].tolist())
            got = {n for n, p in model.named_parameters() if p.grad is not None}
            assert got == names, (sorted(got - names)[:8], sorted(names - got)[:8])
            for n, p in model.named_parameters():
                if p.grad is not None:
                    assert float(p.grad.norm()) == pytest.approx(float(gold["gnorm." + n]), rel=2e-3, abs=1e-5), n
            for k in gold.files:
                if k.startswith("grad."):
                    g = dict(model.named_parameters())[k[5:]].grad
                    torch.testing.assert_close(g.cpu(), torch.as_tensor(gold[k]), rtol=2e-3, atol=2e-5)
    torch.testing.assert_close(model.gate.gain_ema.cpu(), torch.as_tensor(gold["final.gain_ema"]), rtol=1e-3, atol=1e-5)
    torch.testing.assert_close(model.gate.homeo.cpu(), torch.as_tensor(gold["final.homeo"]), rtol=1e-5, atol=1e-7)
    torch.testing.assert_close(torch.stack([f.tau for f in model.reg_ff]).cpu(), torch.as_tensor(gold["final.tau"]), rtol=1e-3, atol=1e-5)
    for n, p in model.named_parameters():
        # a gradient that is analytically zero (e.g. the key bias of a softmax attention) is pure
        # rounding noise; Adam turns its SIGN into a +-lr step, so such parameters are not comparable
        if ("gnorm." + n) in gold.files and float(gold["gnorm." + n]) < 1e-6:
            continue
        assert float(p.detach().norm()) == pytest.approx(float(gold["pnorm." + n]), rel=1e-4, abs=1e-6), n


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
    assert ic.load_checkpoint(model2, opt2, path, device=torch.device(DEV)) == 3
    for (k, a), (_, b) in zip(model.state_dict().items(), model2.state_dict().items()):
        assert torch.equal(a, b), k
    # the arena layout survived load_state_dict: kernels see the loaded weights
    assert model2.regions[1].r_lin.weight.data_ptr() == model2.bank.flat.stacked("W")[4].data_ptr()


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
