"""Pins the CPU oracle (oracle/hotpath.py) against golden vectors produced by the real
reference (tests/golden/make_fixtures.py).  CPU only."""
import os

import numpy as np
import pytest
import torch

from oracle import hotpath as O
from tests.golden import inputs as gi

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def load(name):
    return np.load(os.path.join(GOLD, name))


def close(a, b, rtol=1e-5, atol=1e-6):
    a = torch.as_tensor(np.asarray(a), dtype=torch.float32) if not isinstance(a, torch.Tensor) else a.detach()
    b = torch.as_tensor(np.asarray(b), dtype=torch.float32)
    assert a.shape == b.shape, (a.shape, b.shape)
    torch.testing.assert_close(a, b, rtol=rtol, atol=atol)


def run_region_case(case, seed):
    name, d, m_time, dt, afd, rt, noise_mode, step_pos = case
    t = gi.region_inputs(seed, d, noise_mode)
    leaves = {k: t[k].clone().requires_grad_(True) for k in
              ("num", "den", "var", "pred", "radius", "radius_var", "x_in", "Wr", "Wk", "Wv", "Wo",
               "norm_w", "raw_decay", "pn_param", "on_param", "rpn_param", "ron_param")}
    p = {k: leaves[k] for k in ("Wr", "Wk", "Wv", "Wo", "norm_w", "raw_decay", "pn_param", "on_param",
                                "rpn_param", "ron_param")}
    p["W_time"] = O.time_freqs(d, m_time)
    st = {k: leaves[k] for k in ("num", "den", "var", "pred", "radius", "radius_var")}
    st["surprise"] = t["surprise"]
    h, new = O.region_step(p, st, leaves["x_in"], step_pos, dt=dt, afd=afd, rt=rt,
                           kwta_k=max(1, d // 8))
    loss = (h * t["g_h"]).sum() + (new["num"] * t["g_num"]).sum() + (new["den"] * t["g_den"]).sum()
    if afd:
        loss = loss + (new["var"] * t["g_var"]).sum()
    if rt:
        loss = loss + (new["radius"] * t["g_radius"]).sum() + (new["radius_var"] * t["g_radius_var"]).sum()
    loss.backward()
    return t, leaves, h, new


@pytest.mark.parametrize("ci", range(len(gi.REGION_CASES)))
def test_region_step_matches_reference(ci):
    gold = load("region.npz")
    case = gi.REGION_CASES[ci]
    name, d, m_time, dt, afd, rt, noise_mode, step_pos = case
    t, leaves, h, new = run_region_case(case, 1000 + 100 * ci)
    pre = name + "."
    close(h, gold[pre + "h"])
    # identical KWTA index selection
    assert np.array_equal(h.detach().numpy() != 0, gold[pre + "h"] != 0)
    for k in ("num", "den", "var", "pred", "radius", "radius_var"):
        close(new[k], gold[pre + k])
    if afd:
        close(new["surprise"], gold[pre + "surprise"])
    if rt:
        close(new["last_dir"], gold[pre + "last_dir"])
    for k in ("num", "den", "pred", "x_in", "norm_w", "raw_decay"):
        close(leaves[k].grad, gold[pre + "g." + k], rtol=2e-5, atol=2e-6)
    if afd:
        for k in ("var", "pn_param", "on_param"):
            close(leaves[k].grad, gold[pre + "g." + k], rtol=2e-5, atol=2e-6)
    if rt:
        for k in ("radius", "radius_var", "rpn_param", "ron_param"):
            close(leaves[k].grad, gold[pre + "g." + k], rtol=2e-5, atol=2e-6)
    for k in ("Wr", "Wk", "Wv", "Wo"):
        g = leaves[k].grad
        if d <= 32:
            close(g, gold[pre + "g." + k], rtol=2e-5, atol=2e-6)
        close(g @ t["probe_p"], gold[pre + "gp." + k], rtol=1e-4, atol=1e-5)
        close(t["probe_q"] @ g, gold[pre + "gq." + k], rtol=1e-4, atol=1e-5)


def test_region_manual_step_known_answer():
    """Reference tests/test_rwkv_region.py:8-32 restated: with zero state the default step is
    h = KWTA(x + Wo (sigmoid(Wr x) * v)) where num/den = v exactly."""
    d = 8
    t = gi.region_inputs(42, d)
    p = {k: t[k] for k in ("Wr", "Wk", "Wv", "Wo", "norm_w", "raw_decay")}
    p["W_time"] = torch.zeros(0)
    st = {"num": torch.zeros(d), "den": torch.zeros(d), "var": torch.zeros(d), "pred": torch.zeros(d),
          "radius": torch.zeros(1), "radius_var": torch.zeros(1), "surprise": torch.tensor(0.0)}
    h, new = O.region_step(p, st, t["x_in"], 0.0, dt=0, kwta_k=1)
    x = O.rmsnorm(t["x_in"], t["norm_w"])
    w = torch.exp(t["Wk"] @ x)
    v = t["Wv"] @ x
    y = torch.sigmoid(t["Wr"] @ x) * (w * v / (w + 1e-8))
    expect = O.kwta(x + t["Wo"] @ y, 1)
    close(h, expect.numpy(), atol=1e-6)
    assert int((h != 0).sum()) == 1
    assert torch.all(new["var"] == 0)


def test_fast_forward_closed_form():
    """tests/test_rwkv_region.py:101-115: dt skipped steps decay num/den by lam^dt."""
    d = 4
    raw = gi.det(3, d)
    p = {"Wr": torch.zeros(d, d), "Wk": torch.zeros(d, d), "Wv": torch.zeros(d, d), "Wo": torch.zeros(d, d),
         "norm_w": torch.ones(d), "raw_decay": raw, "W_time": torch.zeros(0),
         "pn_param": torch.tensor(-4.0), "on_param": torch.tensor(0.0)}
    st = {"num": torch.ones(d), "den": torch.ones(d), "var": torch.full((d,), 0.5), "pred": torch.zeros(d),
          "radius": torch.zeros(1), "radius_var": torch.zeros(1), "surprise": torch.tensor(0.0)}
    _, new = O.region_step(p, st, torch.ones(d), 0.0, dt=2, kwta_k=1)
    lam = torch.exp(-torch.nn.functional.softplus(raw) * 3)
    # W = 0  =>  k = 0, w = 1, v = 0: num' = lam, den' = lam + 1
    close(new["num"], lam.numpy())
    close(new["den"], (lam + 1).numpy())


@pytest.mark.parametrize("ci", range(3))
def test_chain_matches_reference(ci):
    gold = load("chain.npz")
    d, afd, rt = [(8, True, False), (32, False, False), (32, True, True)][ci]
    t = gi.region_inputs(5000 + 100 * ci, d, "scalar")
    names = ("Wr", "Wk", "Wv", "Wo", "norm_w", "raw_decay", "pn_param", "on_param", "rpn_param", "ron_param")
    p = {k: t[k].clone().requires_grad_(True) for k in names}
    p["W_time"] = O.time_freqs(d, 16)
    xs = gi.det(5050 + ci, 4, d).requires_grad_(True)
    st = {"num": torch.zeros(d), "den": torch.zeros(d), "var": torch.zeros(d), "pred": torch.zeros(d),
          "radius": torch.zeros(1), "radius_var": torch.zeros(1), "surprise": torch.tensor(0.0)}
    outs, dt = [], 0
    for i in range(4):
        if i == 2:
            st = dict(st)
            st["pred"] = O.predict_trace(st["pred"], gi.det(5060 + ci, d))
            dt += 1
        h, st = O.region_step(p, st, xs[i], i / 3.0, dt=dt, afd=afd, rt=rt, kwta_k=max(1, d // 8))
        dt = 0
        outs.append(h)
    loss = sum((o * gi.det(5070 + ci + 10 * i, d)).sum() for i, o in enumerate(outs))
    loss.backward()
    pre = f"chain{ci}."
    close(torch.stack(outs), gold[pre + "outs"])
    close(st["num"], gold[pre + "num"]); close(st["den"], gold[pre + "den"])
    close(st["var"], gold[pre + "var"]); close(st["radius"], gold[pre + "radius"])
    close(xs.grad, gold[pre + "g.xs"], rtol=5e-5, atol=5e-6)
    for k in ("Wr", "Wk", "Wv", "Wo", "norm_w", "raw_decay"):
        close(p[k].grad, gold[pre + "g." + k], rtol=5e-5, atol=5e-6)
    if afd:
        for k in ("pn_param", "on_param"):
            close(p[k].grad, gold[pre + "g." + k], rtol=5e-5, atol=5e-6)
    if rt:
        for k in ("rpn_param", "ron_param"):
            close(p[k].grad, gold[pre + "g." + k], rtol=5e-5, atol=5e-6)


@pytest.mark.parametrize("i", range(len(gi.KWTA_CASES)))
def test_kwta_matches_reference(i):
    gold = load("kwta.npz")
    d, k = gi.KWTA_CASES[i]
    x = gi.kwta_rows(7000 + i, d)
    assert np.array_equal(O.kwta(x, k).numpy(), gold[f"hard.{d}.{k}"])  # bit-exact
    close(O.kwta(x, k, soft=True, temp=0.7), gold[f"soft.{d}.{k}"], atol=1e-7)
    assert np.array_equal(O.kwta(gi.kwta_rows(7100, 32), 0).numpy(), gold["zero_k"])


def test_kwta_reference_known_answers():
    """tests/test_kwta.py:6-12."""
    x = torch.tensor([[3.0, 2.0, 1.0], [1.0, -2.0, 0.5]])
    y = O.kwta(x, 1)
    assert y.nonzero().shape[0] == 2
    assert torch.equal(y, torch.tensor([[3.0, 0, 0], [0, -2.0, 0]]))
    assert O.kwta(x, 1, soft=True, temp=1.0).abs().sum() > y.abs().sum()


@pytest.mark.parametrize("R", gi.GATE_R)
def test_gate_matches_reference(R):
    gold = load("gate.npz")
    nb = O.hex_neighbors(R)
    gain, homeo = gi.det(8000 + R, R, scale=0.2), gi.det(8001 + R, R, scale=0.05)
    H = gi.det(8002 + R, R, 16)
    H[R // 2:] = 0.0
    for fi, (focus_any, vb, um) in enumerate([(True, 0.13, 0.41), (False, -0.2, 0.0)]):
        sc = O.gate_scores(H, gain, homeo, focus_any, vb, um, 0, R - 1)
        assert np.array_equal(sc.numpy(), gold[f"scores.{R}.{fi}"])  # same ops, same order: bit-exact
        for k_active in (2, 4, 8, 32):
            for burst in (0, max(1, R // 8)):
                ref = gold[f"mask.{R}.{fi}.{k_active}.{burst}"]
                # the reference's own (unstable) order
                m_ref = O.gate_select(sc, k_active, burst, nb, 0, R - 1, stable=False)
                assert np.array_equal(m_ref.numpy(), ref)
                # the build's declared total order: must agree whenever the walk saw no exact ties
                m = O.gate_select(sc, k_active, burst, nb, 0, R - 1, stable=True)
                walked = torch.sort(sc, descending=True).values
                if torch.unique(walked).numel() == walked.numel():
                    assert np.array_equal(m.numpy(), ref)


def test_gate_tie_policy_documented():
    """Exact ties: torch's default argsort is unstable for R > 16, the build uses (score desc,
    index asc).  For R <= 16 (incl. the R=10 bench config) the two coincide."""
    gold = load("gate.npz")
    for R in (4, 7, 10):
        nb = O.hex_neighbors(R)
        assert O.nms_topk(torch.zeros(R), 6, nb, stable=True) == gold[f"tie_zero.{R}"].tolist()
        z = torch.zeros(R); z[0] = 0.5; z[R - 1] = 0.5
        assert O.nms_topk(z, 6, nb, stable=True) == gold[f"tie_two.{R}"].tolist()
    # R > 16: the reference's walk over exactly tied scores follows torch.argsort's unstable order (declared
    # deviation).  What must still hold between its recorded picks and the build's: the same number of
    # picks, the same scores picked (the strictly larger scores always come first), and both sets free of neighbours.
    for R in (36, 64, 512):
        nb = O.hex_neighbors(R)
        for name, z in (("tie_zero", torch.zeros(R)), ("tie_two", None)):
            if z is None:
                z = torch.zeros(R); z[0] = 0.5; z[R - 1] = 0.5
            ref, got = gold[f"{name}.{R}"].tolist(), O.nms_topk(z, 6, nb, stable=True)
            assert len(got) == len(ref), (R, name, got, ref)
            assert sorted(z[got].tolist()) == sorted(z[ref].tolist()), (R, name)
            for picks in (ref, got):
                assert all(q not in nb[p_] for p_ in picks for q in picks), (R, name, picks)
            if name == "tie_two":  # the two strictly larger scores lead both walks, in index order on their tie
                assert set(got[:2]) == set(ref[:2]) == {0, R - 1}
    # k_total == 0 still selects one region (utils.py:118-120)
    assert len(O.nms_topk(torch.tensor([0.1, 0.9, 0.3]), 0, [[1], [0, 2], [1]])) == 1


@pytest.mark.parametrize("R", (1, 2, 7, 10, 36, 64, 128, 256, 512))
def test_wiring_matches_reference(R):
    gold = load("wiring.npz")
    assert np.array_equal(O.hex_axial_coords(R).numpy(), gold[f"coords.{R}"])
    nb = O.hex_neighbors(R)
    flat = np.full((R, 6), -1, dtype=np.int32)
    for i, row in enumerate(nb):
        flat[i, : len(row)] = row
    assert np.array_equal(flat, gold[f"nbr.{R}"])


def test_wiring_known_answers():
    """tests/test_wiring.py:6-20."""
    c = O.hex_axial_coords(7).tolist()
    assert c == [[0, 0], [1, 0], [1, -1], [0, -1], [-1, 0], [-1, 1], [0, 1]]
    nb = O.hex_neighbors(7)
    assert set(nb[0]) == {1, 2, 3, 4, 5, 6}
    assert set(nb[1]) == {0, 2, 6}


def router_params(t, mode):
    names = {"per_edge": ("W_edge",), "by_direction": ("W_dir",), "factorized": ("U", "V", "w_edge")}[mode]
    P = {k: t[k].clone().requires_grad_(True) for k in names + ("key_W", "raw_P", "query_W", "beta_cos", "beta_sin")}
    P["W_reg"] = t["W_reg"]
    return P


@pytest.mark.parametrize("ci", range(len(gi.ROUTER_CASES)))
def test_router_matches_reference(ci):
    gold = load("router.npz")
    name, R, d, mode, prec = gi.ROUTER_CASES[ci]
    nb, coords = O.hex_neighbors(R), O.hex_axial_coords(R)
    E = sum(len(x) for x in nb)
    t = gi.router_inputs(9000 + 100 * ci, R, d, E, mode)
    P = router_params(t, mode)
    H = t["H"].clone().requires_grad_(True)
    M, w = O.router_messages(P, H, t["mask_prev"], coords, nb, mode=mode, precision=prec)
    (M * t["g_M"]).sum().backward()
    pre = name + "."
    close(M, gold[pre + "M"], rtol=2e-5, atol=2e-6)
    close(H.grad, gold[pre + "g.H"], rtol=5e-5, atol=5e-6)
    if prec:
        close(w, gold[pre + "weights"], rtol=2e-5, atol=1e-6)
    close(P["beta_cos"].grad, gold[pre + "g.beta_cos"], rtol=1e-4, atol=1e-5)
    close(P["beta_sin"].grad, gold[pre + "g.beta_sin"], rtol=1e-4, atol=1e-5)
    wname = {"per_edge": ("W_edge",), "by_direction": ("W_dir",), "factorized": ("U", "V", "w_edge")}[mode]
    for k in wname:
        close(P[k].grad, gold[pre + "g." + k], rtol=5e-5, atol=5e-6)
    if prec:
        for k in ("query_W", "key_W", "raw_P"):
            g = P[k].grad if P[k].grad is not None else torch.zeros_like(P[k])
            close(g, gold[pre + "g." + k], rtol=5e-5, atol=5e-6)


def test_router_precision_downweights_inconsistent():
    """tests/test_router_precision.py:46-65 against the recorded reference weights."""
    gold = load("router.npz")
    R, d = 3, 4
    nb = [[1], [0, 2], [1]]
    E = 4
    P = {"W_edge": torch.eye(d).repeat(E, 1, 1) * 0.5, "key_W": torch.eye(d).repeat(E, 1, 1),
         "raw_P": torch.ones(E, d), "query_W": torch.eye(d).repeat(R, 1, 1),
         "W_reg": torch.zeros(8, 2), "beta_cos": torch.zeros(8), "beta_sin": torch.zeros(8)}
    H = torch.zeros(R, d); H[2] = 5.0; H[0] = 0.25
    M, w = O.router_messages(P, H, torch.ones(R, dtype=torch.bool), torch.zeros(R, 2), nb,
                             precision=True, fb_alpha=0.0)
    close(M, gold["line3.M"], atol=1e-6)
    # edges in dst-major order: 1->0, 0->1, 2->1, 1->2
    close(w[1], gold["line3.w01"], atol=1e-7); close(w[2], gold["line3.w21"], atol=1e-7)
    assert w[1] > w[2]


def test_region_phase_equals_single_steps():
    """The batched oracle entry (used to check the HIP phase) is just the loop of single steps."""
    S, R, d = 2, 5, 16
    W = gi.det(1, R, 4, d, d, kind="xavier")
    P = {"W": W, "norm_w": gi.det(2, R, d, scale=0.1, shift=1.0), "raw_decay": gi.det(3, R, d, scale=0.5),
         "W_time": O.time_freqs(d)}
    state = {"num": gi.det(4, S, R, d), "den": gi.det(5, S, R, d, kind="rand", shift=0.5),
             "var": torch.zeros(S, R, d), "pred": gi.det(6, S, R, d, scale=0.1),
             "radius": torch.zeros(S, R), "radius_var": torch.zeros(S, R), "surprise": torch.zeros(S, R),
             "dt": torch.tensor([[0, 1, 2, 0, 3], [1, 0, 0, 2, 0]], dtype=torch.int32)}
    M, sensor = gi.det(7, S, R, d), gi.det(8, S, d)
    mask = torch.tensor([[1, 0, 1, 1, 0], [0, 1, 1, 0, 1]], dtype=torch.bool)
    nin = gi.det(9, d, scale=0.1, shift=1.0)
    H, new = O.region_phase(P, state, M, sensor, mask, 0.5, norm_in_w=nin, sensor_idx=0, motor_idx=4, kwta_k=2)
    assert torch.all(H[~mask] == 0)
    assert torch.equal(new["dt"], torch.tensor([[0, 2, 0, 0, 4], [2, 0, 0, 3, 0]], dtype=torch.int32))
    close(new["pred"][0, 1], (0.95 * state["pred"][0, 1] + (1 - 0.95) * M[0, 1]).numpy())
    assert torch.all(new["pred"][mask] == 0)
    assert torch.equal(new["num"][~mask], state["num"][~mask])
    assert ((H[mask] != 0).sum(-1) == 2).all()
