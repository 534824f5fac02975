"""Device-steered reasoning loops (LoopEngine.run_steered + ic_loop_steer) against the host-steered loops
(LoopEngine.run / _drive): the same kernels on the same inputs with the outer-iteration decisions of reference
model.py:371-452 taken on the device instead of the host -- every result must be IDENTICAL, bit for bit
(selections, states, metrics, gradients), including loops that halt early (utils.py:147-151)."""
import copy

import numpy as np
import pytest
import torch

from tests.model_cases import BIG_CASES, CORNER_CASES, MODEL_CASES, load_case, seed_train

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def build(cfg, sd):
    import ironcortex_amd as ic
    model = ic.CortexReasoner(ic.hex_neighbors(cfg.R), ic.hex_axial_coords(cfg.R), {"sensor": 0, "motor": cfg.R - 1}, cfg)
    model.load_state_dict(sd)
    return model.to(DEV)


def run_steps(name, steer, n_steps=2, sharpen=0.0):
    import ironcortex_amd as ic
    from ironcortex_amd import engine as E
    gold, cfg, sd, tokens, focus = load_case(name)
    model = build(cfg, sd)
    if sharpen:  # confident token heads: u_mean < 0.05, so should_halt fires at k = max(1, K_inner // 2)
        with torch.no_grad():
            for p in model.lm_head.parameters():
                p.mul_(sharpen)
            for h in model.region_heads:
                h.weight.mul_(sharpen)
    opt = torch.optim.AdamW(model.parameters(), lr=1e-3)
    seed_train()
    old = E.STEER
    E.STEER = steer
    try:
        out = []
        for _ in range(n_steps):
            met = ic.train_step(model, opt, tokens.to(DEV), ic.LossWeights(), torch.device(DEV))
            grads = {k: (p.grad.clone() if p.grad is not None else None) for k, p in model.named_parameters()}
            state = {k: getattr(model.bank, k).clone() for k in ("num", "den", "pred", "dt", "var", "radius", "surprise")}
            out.append((met, grads, state, model.work.slots.clone(), float(model.gate.value_bias)))
        counters = dict(model.engine.counters)
        counters["early_flags"] = model._coord.early_used
        with torch.no_grad():
            loop = model.reasoning_loop_batch(tokens.to(DEV), cfg.K_inner, focus.to(DEV))
        return out, counters, loop, model.counters["inner_steps"]
    finally:
        E.STEER = old


def assert_same(a, b):
    for (ma, ga, sa, wa, va), (mb, gb, sb_, wb, vb) in zip(a, b):
        assert ma == mb, (ma, mb)
        assert set(ga) == set(gb)
        for k in ga:
            assert (ga[k] is None) == (gb[k] is None), k
            if ga[k] is not None:
                assert torch.equal(ga[k], gb[k]), k
        for k in sa:
            assert torch.equal(sa[k], sb_[k]), k
        assert torch.equal(wa, wb)
        assert va == vb


@pytest.mark.parametrize("name", ["default", "allflags", "kactive2", "softkwta", "afdvec", "bydir", "bydir_prec", "factorized", "bench10m"])
def test_steered_train_steps_equal_host_steered(name):
    a, ca, la, na = run_steps(name, True)
    b, cb, lb, nb = run_steps(name, False)
    assert ca.get("steered", 0) >= 2 and cb.get("steered", 0) == 0
    # the steered backward hands the touched flags over before it runs (GradCoordinator.offer_early; conftest makes every
    # such pass compare them with the flags the weight-gradient launches raise); the factorized tapes are not covered
    assert ca["early_flags"] == (0 if name == "factorized" else 2) and cb["early_flags"] == 0
    assert_same(a, b)
    assert na == nb
    for x, y in zip(la[:3], lb[:3]):  # the no-grad batch call: H, masks, logits
        assert torch.equal(x, y)
    for ta, tb in zip(la[3], lb[3]):
        assert len(ta) == len(tb)
        for x, y in zip(ta, tb):
            assert torch.equal(x, y)


def test_steered_loops_replay_one_captured_graph():
    """Three steps of the bench model: the launch sequence of the loops is captured once (HIP graph) and replayed;
    same results as launching call by call."""
    from ironcortex_amd import engine as E
    a, ca, la, _ = run_steps("bench10m", True, n_steps=3)
    assert ca.get("steered", 0) == 3 and ca.get("graph_captures", 0) == 1, ca
    old = E.GRAPH
    E.GRAPH = False
    try:
        b, cb, lb, _ = run_steps("bench10m", True, n_steps=3)
    finally:
        E.GRAPH = old
    assert cb.get("steered", 0) == 3 and cb.get("graph_captures", 0) == 0, cb
    assert_same(a, b)
    for x, y in zip(la[:3], lb[:3]):
        assert torch.equal(x, y)


def test_steered_loops_that_halt_early():
    """Sharpened token heads drive the uncertainty below 0.05: every loop halts after iteration K_inner // 2 and the
    launches queued behind it must pass the state through untouched."""
    a, ca, la, na = run_steps("bench10m", True, sharpen=60.0)
    b, cb, lb, nb = run_steps("bench10m", False, sharpen=60.0)
    gold, cfg, *_ = load_case("bench10m")
    assert na == nb and na < 2 * 2 * 2 * cfg.K_inner * cfg.B_br + 2 * cfg.K_inner * cfg.B_br, (na, "no loop halted early")
    assert_same(a, b)
    for x, y in zip(la[:3], lb[:3]):
        assert torch.equal(x, y)
