"""Multi-stream mode (model.stream_mode = "multi", LoopEngine.run_lockstep): the samples of a
batch are INDEPENDENT reasoning loops advanced in lockstep, every one from the region state and workspace
the bank holds when the call starts.

Forward: every stream is checked against an oracle (oracle/model.py, CPU) batch-1 reasoning_loop run
from the same initial state -- a non-trivial one, left behind by a warm-up loop on both sides.
Backward: the gradients of one lockstep batch equal the sum over the loops of the strict engine's
batch-1 run (itself pinned to the real reference's gradients by test_hip_model.py) from that state.
Strict mode stays the mode of every fixture comparison and of bench.py's `value`."""
import copy

import numpy as np
import pytest
import torch

from oracle import model as OM
from tests import parity_log as PL
from tests.golden import inputs as gi
from tests.model_cases import load_case
from tests.test_hip_model import DEV, TOL_GRAD, TOL_STATE, build

pytestmark = pytest.mark.gpu
STATE = ("num", "den", "dt", "var", "surprise", "pred", "radius", "radius_var")


def _samples(cfg, N, T):
    tokens = (gi.det(41, N, T, kind="rand") * (cfg.V - 1)).long()
    focus = gi.det(42, N, T, kind="rand") < 0.3
    focus[1] = False  # one sample with no focus at all (the gate's focus_any term is per stream)
    return tokens, focus


@pytest.mark.parametrize("name,N", [("default", 5), ("det", 4), ("allflags", 5), ("bench10m", 3)])
def test_each_stream_equals_oracle_batch1_from_same_state(name, N):
    gold, cfg, sd, tok2, foc2 = load_case(name)
    T = min(tok2.shape[1], 64)
    tokens, focus = _samples(cfg, N, T)
    K = cfg.K_inner
    # ---- oracle: warm-up loop, then one batch-1 run per sample, each from a copy of the warmed model
    om = OM.build(cfg)
    om.load_state_dict(sd)
    want = []
    with torch.no_grad():
        om.reasoning_loop_batch(tok2[:1, :T], K, foc2[:1, :T])
        for j in range(N):
            o = copy.deepcopy(om)
            o.detach_state()
            H0, m0 = o.zeros_state()
            H, m, lg, tr = o.reasoning_loop(tokens[j], K, focus[j], m0, H0)
            want.append((H, m, lg, tr, o))
    # ---- product: same warm-up (strict: one sample), then all N samples as lockstep streams
    model = build(cfg, sd)
    with torch.no_grad():
        model.reasoning_loop_batch(tok2[:1, :T].to(DEV), K, foc2[:1, :T].to(DEV))
        model.stream_mode = "multi"
        n0 = model.engine.counters.get("lockstep", 0)
        H, m, logits, traces, en, ent = model.reasoning_loop_batch(tokens.to(DEV), K, focus.to(DEV))
    assert model.engine.counters.get("lockstep", 0) == n0 + 1  # the lockstep path is the one that ran
    tag = f"multistream/{name}"
    for j, (Hw, mw, lw, tw, o) in enumerate(want):
        assert np.array_equal(m[j].cpu().numpy(), mw.numpy()), j          # identical region selections
        assert np.array_equal(H[j].cpu().numpy() != 0, Hw.numpy() != 0), j  # identical KWTA winners
        PL.check(tag, f"H[{j}]", H[j], Hw, TOL_STATE)
        PL.check(tag, f"logits[{j}]", logits[j], lw, TOL_STATE)
        assert len(traces[j]) == len(tw), (j, len(traces[j]), len(tw))  # same number of outer iterations (halting)
        PL.check(tag, f"traces[{j}]", torch.stack(traces[j]), torch.stack(tw), TOL_STATE)
    # the bank continues from the LAST loop's final state, the workspace from its winner's
    o = want[-1][4]
    PL.check(tag, "num", model.bank.num[0], torch.stack([c.st["num"] for c in o.regions]), TOL_STATE)
    PL.check(tag, "slots", model.work.slots, o.work.slots, TOL_STATE)
    PL.flush(tag)


@pytest.mark.parametrize("name,N", [("default", 4), ("det", 3), ("allflags", 4), ("bench10m", 3)])
def test_lockstep_gradients_equal_sum_of_batch1_runs(name, N):
    gold, cfg, sd, tok2, foc2 = load_case(name)
    T = min(tok2.shape[1], 64)
    tokens, focus = _samples(cfg, N, T)
    tokens, focus = tokens.to(DEV), focus.to(DEV)
    K = cfg.K_inner
    probe = gi.det(43, N, cfg.R, cfg.d).to(DEV)
    grads, outs = [], []
    for mode in ("multi", "strict"):
        model = build(cfg, sd)
        with torch.no_grad():
            model.reasoning_loop_batch(tok2[:1, :T].to(DEV), K, foc2[:1, :T].to(DEV))
        model.engine._rescue_state()
        start = {k: getattr(model.bank, k).clone() for k in STATE}
        slots = model.work.slots.clone()
        model.engine.new_step()
        model._logits_detached = True  # what train_step promises: no gradient through the inner-step logits
        try:
            if mode == "multi":
                model.stream_mode = "multi"
                H = model.reasoning_loop_batch(tokens, K, focus)[0]
                (H * probe).sum().backward()
                assert model.engine.counters["batch_bwd"] == 1 and model.engine.counters["loop_bwd"] == 0
            else:
                Hs = []
                for j in range(N):
                    for k, v in start.items():
                        setattr(model.bank, k, v.clone())
                    model.work.slots = slots.clone()
                    Hj = model.reasoning_loop_batch(tokens[j:j + 1], K, focus[j:j + 1])[0]
                    (Hj[0] * probe[j]).sum().backward()
                    Hs.append(Hj[0].detach())
                H = torch.stack(Hs)
        finally:
            model._logits_detached = False
        outs.append(H.detach())
        grads.append({n: (None if p.grad is None else p.grad.clone()) for n, p in model.named_parameters()})
    tag = f"multistream_grad/{name}"
    PL.check(tag, "H", outs[0], outs[1], TOL_STATE)
    have = [{n for n, g in gr.items() if g is not None} for gr in grads]
    assert have[0] == have[1], (sorted(have[0] - have[1])[:6], sorted(have[1] - have[0])[:6])
    assert any("regions." in n for n in have[0]) and any("router." in n for n in have[0])
    for n in sorted(have[0]):
        PL.check(tag, f"grad:{n}", grads[0][n], grads[1][n], TOL_GRAD, floor=1e-6)
    PL.flush(tag)


def test_train_step_multi_mode_runs_and_strict_is_untouched():
    """train_step in multi-stream mode: finite metrics, every loop differentiated by the batched backward;
    with B = 1 sample per stream group and a fresh (all-zero) state, the first positive loop is the same
    computation in both modes, so ff-positive goodness of sample 0 agrees -- checked through the whole
    metric set of a 1-sample batch whose two loops (positive, negative) both start from the zero state
    in multi mode but chain in strict mode: the metrics differ only through the negative loop's start."""
    import ironcortex_amd as ic
    from tests.model_cases import seed_train
    gold, cfg, sd, tokens, focus = load_case("allflags")
    out = {}
    for mode in ("strict", "multi"):
        model = build(cfg, sd)
        model.stream_mode = mode
        opt = torch.optim.AdamW(model.parameters(), lr=1e-3)
        seed_train()
        out[mode] = ic.train_step(model, opt, tokens.to(DEV), ic.LossWeights(), torch.device(DEV))
        assert all(np.isfinite(float(v)) for k, v in out[mode].items() if isinstance(v, (int, float))), out[mode]
        if mode == "multi":
            assert model.engine.counters.get("lockstep", 0) >= 1 and model.engine.counters["loop_bwd"] == 0
    # strict mode still reproduces the real reference's first step exactly as before
    for k in ("ff", "total", "grad_norm"):
        assert out["strict"][k] == pytest.approx(float(gold[f"step0.{k}"]), rel=2e-5, abs=1e-6), k


@pytest.mark.parametrize("variant", ["by_direction", "by_direction_edge_kernels", "groups"])
def test_multistream_by_direction_and_launch_groups(variant, monkeypatch):
    """Two corners of the lockstep path, each stream again against an oracle batch-1 run from the same warmed state:
    the shared-matrix router (edge_transform_mode="by_direction": the MFMA forward reads each stream's parent block
    through in_slot), and a batch larger than one launch group (every group restarts from the state the call began
    with; the bank ends with the last loop's)."""
    import ironcortex_amd as ic
    from ironcortex_amd.config import CortexConfig
    from tests.model_cases import MODEL_CASES
    kw = dict(MODEL_CASES["default"])
    if variant.startswith("by_direction"):
        kw["edge_transform_mode"] = "by_direction"
        # the GEMM at this size only when asked for (IC_RF_SHARED_GEMM); the library's own choice below 8192 columns is
        # the per-edge kernels reading the shared matrices through widx (second variant)
        from ironcortex_amd.hip import ops
        monkeypatch.setattr(ops, "SHARED_GEMM_ALWAYS", variant == "by_direction")
    cfg = CortexConfig(**kw)
    torch.manual_seed(5)
    ref = ic.CortexReasoner(ic.hex_neighbors(cfg.R), ic.hex_axial_coords(cfg.R), {"sensor": 0, "motor": cfg.R - 1}, cfg)
    with torch.no_grad():  # tie-free gate buffers (the reference's order on exact ties is the CPU sort's)
        ref.gate.gain_ema.copy_(gi.det(51, cfg.R, scale=0.1))
        ref.gate.homeo.copy_(gi.det(52, cfg.R, scale=0.1).abs())
    sd = {k: v.clone() for k, v in ref.state_dict().items()}
    N, T, K = 5, 24, cfg.K_inner
    tokens, focus = _samples(cfg, N, T)
    warm_t, warm_f = tokens[:1].flip(1), focus[:1]
    om = OM.build(cfg)
    om.load_state_dict(sd)
    want = []
    with torch.no_grad():
        om.reasoning_loop_batch(warm_t, K, warm_f)
        for j in range(N):
            o = copy.deepcopy(om)
            o.detach_state()
            H0, m0 = o.zeros_state()
            want.append(o.reasoning_loop(tokens[j], K, focus[j], m0, H0) + (o,))
    model = build(cfg, sd)
    if variant == "groups":
        monkeypatch.setattr(type(model.engine), "MAX_LOCKSTEP_STREAMS", 4)  # 2 loops x 2 branches per launch group
    with torch.no_grad():
        model.reasoning_loop_batch(warm_t.to(DEV), K, warm_f.to(DEV))
        model.stream_mode = "multi"
        n0 = model.engine.counters["lockstep"]
        H, m, logits, traces, en, ent = model.reasoning_loop_batch(tokens.to(DEV), K, focus.to(DEV))
    assert model.engine.counters["lockstep"] - n0 == (3 if variant == "groups" else 1)
    tag = f"multistream/{variant}"
    for j, (Hw, mw, lw, tw, o) in enumerate(want):
        assert np.array_equal(m[j].cpu().numpy(), mw.numpy()), j
        assert np.array_equal(H[j].cpu().numpy() != 0, Hw.numpy() != 0), j
        PL.check(tag, f"H[{j}]", H[j], Hw, TOL_STATE)
        PL.check(tag, f"logits[{j}]", logits[j], lw, TOL_STATE)
        PL.check(tag, f"traces[{j}]", torch.stack(traces[j]), torch.stack(tw), TOL_STATE)
    PL.check(tag, "num", model.bank.num[0], torch.stack([c.st["num"] for c in want[-1][4].regions]), TOL_STATE)
    PL.check(tag, "slots", model.work.slots, want[-1][4].work.slots, TOL_STATE)
    PL.flush(tag)
