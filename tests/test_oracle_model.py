"""The reference-shaped oracle model reproduces the REAL reference's whole-step outputs
(reasoning loop + two train_steps) from the same state_dict, tokens and seeds.  CPU only."""
import copy

import numpy as np
import pytest
import torch

from oracle import model as OM
from tests.golden import inputs as gi
from tests.model_cases import BIG_CASES, CORNER_CASES, MODEL_CASES, SHARED_CASES, grad_probes, load_case, seed_train

KEYS = ("ff", "rtd", "denoise", "critic", "verify", "E_pos", "E_neg", "E_gap", "rtd_acc", "denoise_acc",
        "grad_norm", "active_frac", "lambda_s", "ce_last", "total")


@pytest.mark.parametrize("name", list(MODEL_CASES) + list(CORNER_CASES) + list(SHARED_CASES) + list(BIG_CASES))
def test_state_dict_keys_and_loop(name):
    gold, cfg, sd, tokens, focus = load_case(name)
    model = OM.build(cfg)
    assert set(model.state_dict().keys()) == set(sd.keys())  # F11: same names
    for k, v in model.state_dict().items():
        assert tuple(v.shape) == tuple(sd[k].shape), k
    model.load_state_dict(sd)
    with torch.no_grad():
        H, m, logits, traces, en, ent = copy.deepcopy(model).reasoning_loop_batch(tokens, cfg.K_inner, focus)
    assert np.array_equal(m.numpy(), gold["loop.mask"])  # identical region selections
    assert np.array_equal(H.numpy() != 0, gold["loop.H"] != 0)  # identical KWTA winners
    torch.testing.assert_close(H, torch.as_tensor(gold["loop.H"]), rtol=2e-5, atol=2e-6)
    torch.testing.assert_close(logits, torch.as_tensor(gold["loop.logits"]), rtol=2e-5, atol=2e-6)
    torch.testing.assert_close(torch.stack(traces[0]), torch.as_tensor(gold["loop.trace0"]), rtol=2e-5, atol=2e-6)
    torch.testing.assert_close(en, torch.as_tensor(gold["loop.energy"]), rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("name", list(MODEL_CASES) + list(CORNER_CASES) + list(SHARED_CASES) + list(BIG_CASES))
def test_two_train_steps(name):
    gold, cfg, sd, tokens, focus = load_case(name)
    model = OM.build(cfg)
    model.load_state_dict(sd)
    opt = torch.optim.AdamW(model.parameters(), lr=1e-3)
    seed_train()
    for step in range(2):
        out, grads = OM.oracle_train_step(model, opt, tokens, OM.LossWeights(), return_grads=True)
        for k in KEYS:
            ref = float(gold[f"step{step}.{k}"])
            assert out[k] == pytest.approx(ref, rel=2e-4, abs=2e-5), (step, k, out[k], ref)
        if step == 0:
            names = set(gold["grad_names"].tolist())
            assert set(grads.keys()) == names  # same parameters receive a gradient (None elsewhere)
            for n in names:
                assert float(grads[n].norm()) == pytest.approx(float(gold["gnorm." + n]), rel=5e-4, abs=1e-6), n
            for k in gold.files:
                if k.startswith("grad."):
                    torch.testing.assert_close(grads[k[5:]], torch.as_tensor(gold[k]), rtol=5e-4, atol=1e-6)
                elif k.startswith("gp."):  # full-size cases: large matrix gradients as two probe projections
                    g = grads[k[3:]]
                    pc, pr = grad_probes(g.shape[1], g.shape[0])
                    torch.testing.assert_close(g @ pc, torch.as_tensor(gold[k]), rtol=5e-4, atol=1e-5)
                    torch.testing.assert_close(pr @ g, torch.as_tensor(gold["gq." + k[3:]]), rtol=5e-4, atol=1e-5)
    torch.testing.assert_close(model.gate.gain_ema, torch.as_tensor(gold["final.gain_ema"]), rtol=1e-4, atol=1e-6)
    torch.testing.assert_close(model.gate.homeo, torch.as_tensor(gold["final.homeo"]), rtol=1e-5, atol=1e-7)
    torch.testing.assert_close(torch.stack([f.tau for f in model.reg_ff]), torch.as_tensor(gold["final.tau"]), rtol=1e-4, atol=1e-6)
    for n, p in model.named_parameters():
        if ("gnorm." + n) in gold.files and float(gold["gnorm." + n]) < 1e-6:
            continue  # first-step gradient analytically zero: Adam makes the next one a +-lr step of rounding noise
        # (Adam divides by |g|: parameters that start at 0 with tiny gradients, e.g. raw_decay at d=256, turn a
        # 1e-6 relative gradient difference into 1e-5 of the parameter)
        # (1e-4: softkwta's plan.mlp_p.net.0.bias, a zero-initialised bias after two Adam steps, sits at 6.5e-5)
        assert float(p.detach().norm()) == pytest.approx(float(gold["pnorm." + n]), rel=1e-4, abs=1e-7), n


@pytest.mark.parametrize("name", list(MODEL_CASES))
def test_evaluate_perplexity(name):
    """oracle restatement of evaluation.py:14-52 == the reference's recorded values."""
    import os
    from tests.golden.make_eval import EVAL_T
    gold, cfg, sd, tokens, focus = load_case(name)
    ev = np.load(os.path.join(os.path.dirname(__file__), "golden", "eval.npz"))
    model = OM.build(cfg)
    model.load_state_dict(sd)
    out = OM.oracle_evaluate_perplexity(model, tokens[:, :EVAL_T])
    assert out["cross_entropy"] == pytest.approx(float(ev[name + ".cross_entropy"]), rel=1e-6)
    assert out["perplexity"] == pytest.approx(float(ev[name + ".perplexity"]), rel=1e-6)
    num = torch.stack([c.st["num"] for c in model.regions])
    torch.testing.assert_close(num, torch.as_tensor(ev[name + ".num"]), rtol=1e-5, atol=1e-7)


@pytest.mark.parametrize("name", ["bydir_ff", "factorized_ff"])
def test_shared_modes_raise_where_the_reference_does(name):
    """Forward-forward loss on with a shared-matrix router: the reference's Hebbian block indexes router.W_edge
    (training.py:240-243), which these modes do not define -- AttributeError at the first step with an update due
    (step 1 here), after that step's gain / tau updates.  tests/golden/shared_raise.npz is the reference's run."""
    import os
    import random
    from oracle import model as OM
    from ironcortex_amd.config import CortexConfig
    from tests.model_cases import GOLD, MODEL_B, MODEL_T, SHARED_RAISE
    gold = np.load(os.path.join(GOLD, "shared_raise.npz"))
    cfg = CortexConfig(**SHARED_RAISE[name])
    om = OM.build(cfg)
    om.load_state_dict({k[len(name) + 4:]: torch.as_tensor(gold[k]) for k in gold.files if k.startswith(name + ".sd.")})
    tokens = (gi.det(31, MODEL_B, MODEL_T, kind="rand") * (cfg.V - 1)).long()
    opt = torch.optim.AdamW(om.parameters(), lr=1e-3)
    torch.manual_seed(77); random.seed(77)
    raised_at = -1
    for step in range(3):
        try:
            out = OM.oracle_train_step(om, opt, tokens, OM.LossWeights())
        except AttributeError:
            raised_at = step
            break
        for k, v in out.items():
            assert v == pytest.approx(float(gold[f"{name}.step{step}.{k}"]), rel=2e-5, abs=2e-6), (step, k)
    assert raised_at == int(gold[name + ".raised_at"]) == 1
    torch.testing.assert_close(om.gate.gain_ema, torch.as_tensor(gold[name + ".gain_ema"]), rtol=1e-5, atol=1e-7)
    torch.testing.assert_close(om.gate.homeo, torch.as_tensor(gold[name + ".homeo"]), rtol=1e-5, atol=1e-7)
    torch.testing.assert_close(torch.stack([f.tau for f in om.reg_ff]), torch.as_tensor(gold[name + ".tau"]), rtol=1e-5, atol=1e-7)
