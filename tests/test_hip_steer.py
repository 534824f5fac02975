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


def run_steps(name, steer, n_steps=2, sharpen=0.0, split=False):
    import ironcortex_amd as ic
    from ironcortex_amd import engine as E
    from ironcortex_amd import training as T
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
    old, old_split = E.STEER, T.SPLIT
    E.STEER = steer
    # split=False: the token mixer sees the whole 2B batch in one call on both paths (train_step's split prologue gives
    # the steered path two calls of B rows: the same values, but the mixer's weight gradients summed in two parts)
    T.SPLIT = split
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
        E.STEER, T.SPLIT = old, old_split


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


@pytest.mark.parametrize("name", ["default", "bench10m"])
def test_split_prologue_equals_one_batch(name):
    """train_step's split prologue (the clean sequences' loops queued first, the corrupted sequences drawn and mixed while
    they run; training.SPLIT) against the one-batch order: the same launches on the same values.  Only the token
    mixer's weight gradients are summed in two parts instead of one GEMM over 2B rows, hence a tolerance on the
    gradients (1e-5 relative to the largest entry) instead of equality, and of 2e-6 on the first step's metrics."""
    a, ca, la, na = run_steps(name, True, split=True)
    b, cb, lb, nb = run_steps(name, True, split=False)
    assert ca["steered"] == cb["steered"] and na == nb and ca["early_flags"] == cb["early_flags"] == 2
    (ma, ga, sa, wa, va), (mb, gb, sb_, wb, vb) = a[0], b[0]
    for key in ma:  # (the mixer's per-row statistics move by an ulp with the number of rows in the call)
        assert abs(ma[key] - mb[key]) <= (1e-5 if key == "grad_norm" else 2e-6) * max(1.0, abs(mb[key])), (key, ma[key], mb[key])
    for k in ga:
        assert (ga[k] is None) == (gb[k] is None), k
        if ga[k] is not None:
            tol = 1e-5 * float(gb[k].abs().max()) + 1e-9
            assert float((ga[k] - gb[k]).abs().max()) <= tol, (k, float((ga[k] - gb[k]).abs().max()), tol)
    for k in sa:
        assert torch.allclose(sa[k].float(), sb_[k].float(), rtol=1e-5, atol=1e-7), k
    assert torch.allclose(wa, wb, rtol=1e-5, atol=1e-7) and abs(va - vb) <= 1e-6 * max(1.0, abs(vb))
    for key in a[1][0]:  # second step: behind one optimizer update of slightly different gradients
        assert abs(a[1][0][key] - b[1][0][key]) <= 2e-3 * max(1.0, abs(b[1][0][key])), (key, a[1][0][key], b[1][0][key])


def test_graph_capture_failure_falls_back_to_plain_launches(monkeypatch):
    """A runtime that refuses the capture (here: ic_graph_begin made to fail) must not cost the step: the loops are
    launched call by call from then on, same results."""
    import warnings
    from ironcortex_amd.hip import lib
    a, ca, la, _ = run_steps("default", True)
    L = lib()
    monkeypatch.setattr(L, "ic_graph_begin", lambda h: -3)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        b, cb, lb, _ = run_steps("default", True)
    assert any("graph capture failed" in str(x.message) for x in w)
    assert ca.get("graph_captures", 0) >= 1 and cb.get("graph_captures", 0) == 0 and cb["steered"] == ca["steered"]
    assert_same(a, b)


@pytest.mark.parametrize("name", ["bench10m", "r36"])
def test_wgrad_gemm_equals_row_slice_kernels(name, monkeypatch):
    """The deferred weight gradients of a step's loops through the pair-list MFMA GEMM (tapes of >= 64 (slot, stream)
    pairs: region matrices as 4R pseudo edges, edge matrices, keys, queries) against the row-slice kernels on the same
    tapes: same sums in another order -- 2e-6 of each gradient's largest entry, identical None-ness."""
    from ironcortex_amd.hip import ops
    n0 = ops.WGRAD_GEMM_CALLS[0]
    a, *_ = run_steps(name, True, n_steps=1)
    assert ops.WGRAD_GEMM_CALLS[0] >= n0 + 2, "the GEMM path was not taken"
    monkeypatch.setattr(ops, "WGRAD_GEMM", False)
    n1 = ops.WGRAD_GEMM_CALLS[0]
    b, *_ = run_steps(name, True, n_steps=1)
    assert ops.WGRAD_GEMM_CALLS[0] == n1
    ga, gb = a[0][1], b[0][1]
    worst = 0.0
    for k in ga:
        assert (ga[k] is None) == (gb[k] is None), k
        if ga[k] is not None:
            scale = float(gb[k].abs().max())
            err = float((ga[k] - gb[k]).abs().max())
            assert err <= 2e-6 * scale + 1e-12, (k, err, scale)
            worst = max(worst, err / max(scale, 1e-30))
    assert {k: v for k, v in a[0][0].items() if k != "grad_norm"} == {k: v for k, v in b[0][0].items() if k != "grad_norm"}
