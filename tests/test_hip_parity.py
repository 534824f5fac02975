"""GPU parity: the HIP path (through the C ABI) against the golden vectors of the reference and
against the CPU oracle on the same seeded inputs.

Tolerances (fp32, GEMV reductions in a different order than the CPU BLAS):
  activations/state  rtol 2e-5, atol 2e-6;   gradients  rtol 1e-4, atol 1e-5 * max(1, max|ref|).
KWTA / gate selections must be identical.
"""
import os

import numpy as np
import pytest
import torch

from tests.golden import inputs as gi

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
DEV = "cuda:0"


def load(name):
    return np.load(os.path.join(GOLD, name))


def close(a, b, rtol=2e-5, atol=2e-6, what=""):
    a = a.detach().float().cpu() if isinstance(a, torch.Tensor) else torch.as_tensor(np.asarray(a), dtype=torch.float32)
    b = b.detach().float().cpu() if isinstance(b, torch.Tensor) else torch.as_tensor(np.asarray(b), dtype=torch.float32)
    a, b = a.reshape(b.shape) if a.numel() == b.numel() else a, b
    torch.testing.assert_close(a, b, rtol=rtol, atol=atol, msg=lambda m: f"{what}: {m}")


def gclose(a, b, what=""):
    """Gradients: rtol 1e-4, atol 1e-5 x max(1, max|ref|) (sums over rows cancel, so the absolute
    floor scales with the tensor's magnitude)."""
    bb = b.detach().float().cpu() if isinstance(b, torch.Tensor) else torch.as_tensor(np.asarray(b), dtype=torch.float32)
    scale = max(1.0, float(bb.abs().max())) if bb.numel() else 1.0
    close(a, b, rtol=1e-4, atol=1e-5 * scale, what=what)


def test_device_is_gfx950():
    import ctypes
    from ironcortex_amd import hip
    buf = ctypes.create_string_buffer(64)
    rc = hip.lib().ic_device_check(buf, 64)
    assert rc == 0, (buf.value, hip.lib().ic_last_error())


# --------------------------------------------------------------------------- region cell
def make_cell(case, t):
    from ironcortex_amd.region import RWKVRegionCell
    name, d, m_time, dt, afd, rt, noise_mode, step_pos = case
    cell = RWKVRegionCell(d, m_time_pairs=m_time, enable_adaptive_filter_dynamics=afd,
                          enable_radial_tangential_updates=rt, afd_noise_mode=noise_mode).to(DEV)
    with torch.no_grad():
        cell.r_lin.weight.copy_(t["Wr"]); cell.k_lin.weight.copy_(t["Wk"])
        cell.v_lin.weight.copy_(t["Wv"]); cell.o_lin.weight.copy_(t["Wo"])
        cell.norm.weight.copy_(t["norm_w"]); cell.raw_decay.copy_(t["raw_decay"])
        cell.process_noise_param.copy_(t["pn_param"]); cell.obs_noise_param.copy_(t["on_param"])
        cell.radius_process_noise_param.copy_(t["rpn_param"]); cell.radius_obs_noise_param.copy_(t["ron_param"])
    return cell


@pytest.mark.parametrize("ci", range(len(gi.REGION_CASES)))
def test_region_step_vs_reference(ci):
    gold = load("region.npz")
    case = gi.REGION_CASES[ci]
    name, d, m_time, dt, afd, rt, noise_mode, step_pos = case
    t = gi.region_inputs(1000 + 100 * ci, d, noise_mode)
    cell = make_cell(case, t)
    leaves = {k: t[k].clone().to(DEV).requires_grad_(True) for k in ("num", "den", "var", "pred", "radius", "radius_var", "x_in")}
    cell.state_num = leaves["num"] * 1.0
    cell.state_den = leaves["den"] * 1.0
    cell.state_var = leaves["var"] * 1.0
    cell.pred = leaves["pred"] * 1.0
    cell.radius = leaves["radius"] * 1.0
    cell.radius_var = leaves["radius_var"] * 1.0
    cell.surprise_ema = t["surprise"]
    cell.dt = dt
    g = {k: t[k].to(DEV) for k in ("g_h", "g_num", "g_den", "g_var", "g_radius", "g_radius_var")}
    h = cell.step(leaves["x_in"], step_pos)
    loss = (h * g["g_h"]).sum() + (cell.state_num * g["g_num"]).sum() + (cell.state_den * g["g_den"]).sum()
    if afd:
        loss = loss + (cell.state_var * g["g_var"]).sum()
    if rt:
        loss = loss + (cell.radius * g["g_radius"]).sum() + (cell.radius_var * g["g_radius_var"]).sum()
    loss.backward()
    pre = name + "."
    assert np.array_equal(h.detach().cpu().numpy() != 0, gold[pre + "h"] != 0), "KWTA winners differ"
    close(h, gold[pre + "h"], what="h")
    close(cell.state_num, gold[pre + "num"], what="num"); close(cell.state_den, gold[pre + "den"], what="den")
    close(cell.pred, gold[pre + "pred"], what="pred")
    assert cell.dt == int(gold[pre + "dt"])
    if afd:
        close(cell.state_var, gold[pre + "var"], what="var")
        close(cell.surprise_ema, gold[pre + "surprise"], rtol=1e-4, what="surprise")
    if rt:
        close(cell.radius, gold[pre + "radius"], what="radius"); close(cell.radius_var, gold[pre + "radius_var"], what="rvar")
        close(cell.last_dir, gold[pre + "last_dir"], what="last_dir"); close(cell.last_norm, gold[pre + "last_norm"], what="last_norm")
    for k in ("num", "den", "pred", "x_in"):
        gclose(leaves[k].grad, gold[pre + "g." + k], what="g." + k)
    gclose(cell.norm.weight.grad, gold[pre + "g.norm_w"], what="g.norm_w")
    gclose(cell.raw_decay.grad, gold[pre + "g.raw_decay"], what="g.raw_decay")
    if afd:
        gclose(leaves["var"].grad, gold[pre + "g.var"], what="g.var")
        gclose(cell.process_noise_param.grad, gold[pre + "g.pn_param"], what="g.pn")
        gclose(cell.obs_noise_param.grad, gold[pre + "g.on_param"], what="g.on")
    else:
        assert cell.process_noise_param.grad is None
    if rt:
        gclose(leaves["radius"].grad, gold[pre + "g.radius"], what="g.radius")
        gclose(leaves["radius_var"].grad, gold[pre + "g.radius_var"], what="g.radius_var")
        gclose(cell.radius_process_noise_param.grad, gold[pre + "g.rpn_param"], what="g.rpn")
        gclose(cell.radius_obs_noise_param.grad, gold[pre + "g.ron_param"], what="g.ron")
    pp, pq = t["probe_p"].to(DEV), t["probe_q"].to(DEV)
    for k, lin in (("Wr", cell.r_lin), ("Wk", cell.k_lin), ("Wv", cell.v_lin), ("Wo", cell.o_lin)):
        gw = lin.weight.grad
        assert gw is not None, k
        if d <= 32:
            gclose(gw, gold[pre + "g." + k], what="g." + k)
        close(gw @ pp, gold[pre + "gp." + k], rtol=2e-4, atol=2e-5, what="gp." + k)
        close(pq @ gw, gold[pre + "gq." + k], rtol=2e-4, atol=2e-5, what="gq." + k)


@pytest.mark.parametrize("ci", range(3))
def test_region_chain_vs_reference(ci):
    gold = load("chain.npz")
    d, afd, rt = [(8, True, False), (32, False, False), (32, True, True)][ci]
    t = gi.region_inputs(5000 + 100 * ci, d, "scalar")
    cell = make_cell((f"chain{ci}", d, 16, 0, afd, rt, "scalar", 0.0), t)
    xs = gi.det(5050 + ci, 4, d).to(DEV).requires_grad_(True)
    outs = []
    for i in range(4):
        if i == 2:
            cell.predict(gi.det(5060 + ci, d).to(DEV))
            cell.skip()
        outs.append(cell.step(xs[i], i / 3.0))
    loss = sum((o * gi.det(5070 + ci + 10 * i, d).to(DEV)).sum() for i, o in enumerate(outs))
    loss.backward()
    pre = f"chain{ci}."
    close(torch.stack(outs), gold[pre + "outs"], what="outs")
    close(cell.state_num, gold[pre + "num"], what="num"); close(cell.state_den, gold[pre + "den"], what="den")
    gclose(xs.grad, gold[pre + "g.xs"], what="g.xs")
    for k, lin in (("Wr", cell.r_lin), ("Wk", cell.k_lin), ("Wv", cell.v_lin), ("Wo", cell.o_lin)):
        gclose(lin.weight.grad, gold[pre + "g." + k], what="g." + k)
    gclose(cell.norm.weight.grad, gold[pre + "g.norm_w"], what="g.norm_w")
    gclose(cell.raw_decay.grad, gold[pre + "g.raw_decay"], what="g.raw_decay")
    if afd:
        gclose(cell.process_noise_param.grad, gold[pre + "g.pn_param"], what="g.pn")
        gclose(cell.obs_noise_param.grad, gold[pre + "g.on_param"], what="g.on")
    if rt:
        gclose(cell.radius_process_noise_param.grad, gold[pre + "g.rpn_param"], what="g.rpn")
        gclose(cell.radius_obs_noise_param.grad, gold[pre + "g.ron_param"], what="g.ron")


# --------------------------------------------------------------------------- batched region phase vs oracle
def build_bank(R, d, afd, rt, noise_mode, seed):
    from ironcortex_amd.region import RWKVRegionCell, RegionBank
    cells = [RWKVRegionCell(d, enable_adaptive_filter_dynamics=afd, enable_radial_tangential_updates=rt,
                            afd_noise_mode=noise_mode, _shared_bank=True) for _ in range(R)]
    bank = RegionBank(cells, sensor_idx=0, motor_idx=R - 1)
    P = {"W": gi.det(seed, R, 4, d, d, kind="xavier"), "norm_w": gi.det(seed + 1, R, d, scale=0.1, shift=1.0),
         "raw_decay": gi.det(seed + 2, R, d, scale=0.5)}
    ns = (R,) if noise_mode == "scalar" else (R, d)
    P["pn_param"] = gi.det(seed + 3, *ns, scale=0.5, shift=-3.0)
    P["on_param"] = gi.det(seed + 4, *ns, scale=0.5)
    P["rpn_param"] = gi.det(seed + 5, R, scale=0.5, shift=-3.0)
    P["ron_param"] = gi.det(seed + 6, R, scale=0.5)
    for c in cells:
        c.to(DEV)
    bank.to_device()
    with torch.no_grad():
        for r, c in enumerate(cells):
            for j, lin in enumerate((c.r_lin, c.k_lin, c.v_lin, c.o_lin)):
                lin.weight.copy_(P["W"][r, j])
            c.norm.weight.copy_(P["norm_w"][r]); c.raw_decay.copy_(P["raw_decay"][r])
            c.process_noise_param.copy_(P["pn_param"][r]); c.obs_noise_param.copy_(P["on_param"][r])
            c.radius_process_noise_param.copy_(P["rpn_param"][r]); c.radius_obs_noise_param.copy_(P["ron_param"][r])
    return bank, cells, P


@pytest.mark.parametrize("S,R,d,afd,rt,noise", [
    (1, 7, 32, False, False, "scalar"), (3, 7, 32, True, True, "scalar"), (2, 10, 256, True, True, "vector"),
    (1, 36, 256, False, False, "scalar"), (2, 5, 128, True, False, "scalar"), (2, 4, 512, False, True, "scalar"),
])
def test_region_phase_vs_oracle(S, R, d, afd, rt, noise):
    from oracle import hotpath as O
    seed = 300 + R + d
    bank, cells, P = build_bank(R, d, afd, rt, noise, seed)
    st = {"num": gi.det(seed + 10, S, R, d, scale=0.5), "den": gi.det(seed + 11, S, R, d, kind="rand", shift=0.5),
          "var": gi.det(seed + 12, S, R, d, kind="rand", scale=0.45, shift=0.05), "pred": gi.det(seed + 13, S, R, d, scale=0.1),
          "radius": gi.det(seed + 14, S, R, kind="rand", shift=0.5), "radius_var": gi.det(seed + 15, S, R, kind="rand", scale=0.5, shift=0.1),
          "surprise": gi.det(seed + 16, S, R, kind="rand"),
          "dt": (gi.det(seed + 17, S, R, kind="rand") * 4).to(torch.int32)}
    M, sensor = gi.det(seed + 18, S, R, d), gi.det(seed + 19, S, d)
    mask = gi.det(seed + 20, S, R, kind="rand") < 0.55
    mask[:, 0] = True
    nin = gi.det(seed + 21, d, scale=0.1, shift=1.0)
    cot = {k: gi.det(seed + 30 + i, *st[k].shape, scale=0.3) for i, k in enumerate(("num", "den", "var", "pred", "radius", "radius_var"))}
    gH = gi.det(seed + 40, S, R, d)
    kw = max(1, d // 8)
    # ---- oracle (CPU autograd)
    Pc = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    Pc["W_time"] = O.time_freqs(d)
    stc = {k: (v.clone().requires_grad_(True) if v.dtype.is_floating_point and k != "surprise" else v.clone()) for k, v in st.items()}
    Mc, sc, ninc = M.clone().requires_grad_(True), sensor.clone().requires_grad_(True), nin.clone().requires_grad_(True)
    Hc, newc = O.region_phase(Pc, stc, Mc, sc, mask, 0.4285714, norm_in_w=ninc, sensor_idx=0, motor_idx=R - 1,
                              afd=afd, rt=rt, kwta_k=kw)
    loss = (Hc * gH).sum() + sum((newc[k] * cot[k]).sum() for k in cot)
    loss.backward()
    # ---- HIP
    bank.reset_state(torch.device(DEV), S)
    leaves = {k: st[k].clone().to(DEV).requires_grad_(True) for k in ("num", "den", "var", "pred", "radius", "radius_var")}
    for k, v in leaves.items():
        setattr(bank, k, v * 1.0)
    bank.surprise = st["surprise"].to(DEV)
    bank.dt = st["dt"].to(DEV)
    Mg, sg, ning = M.to(DEV).requires_grad_(True), sensor.to(DEV).requires_grad_(True), nin.to(DEV).requires_grad_(True)
    Hg = bank.phase(Mg, sg, ning, mask.to(torch.uint8).to(DEV), 0.4285714)
    lossg = (Hg * gH.to(DEV)).sum() + (bank.num * cot["num"].to(DEV)).sum() + (bank.den * cot["den"].to(DEV)).sum() \
        + (bank.pred * cot["pred"].to(DEV)).sum()
    if afd:
        lossg = lossg + (bank.var * cot["var"].to(DEV)).sum()
    if rt:
        lossg = lossg + (bank.radius * cot["radius"].to(DEV)).sum() + (bank.radius_var * cot["radius_var"].to(DEV)).sum()
    lossg.backward()
    assert torch.equal((Hg != 0).cpu(), Hc != 0), "KWTA winners differ"
    close(Hg, Hc, what="H")
    for k in ("num", "den", "pred"):
        close(getattr(bank, k), newc[k], what=k)
    assert torch.equal(bank.dt.cpu(), newc["dt"])
    if afd:
        close(bank.var, newc["var"], what="var"); close(bank.surprise, newc["surprise"], rtol=1e-4, what="surprise")
    if rt:
        close(bank.radius, newc["radius"], what="radius"); close(bank.radius_var, newc["radius_var"], what="radius_var")
    gclose(Mg.grad, Mc.grad, what="g.M"); gclose(sg.grad, sc.grad, what="g.sensor"); gclose(ning.grad, ninc.grad, what="g.norm_in")
    names = ["num", "den", "pred"] + (["var"] if afd else []) + (["radius", "radius_var"] if rt else [])
    for k in names:
        gclose(leaves[k].grad, stc[k].grad, what="g." + k)
    active = mask.any(0)
    for r, c in enumerate(cells):
        if not bool(active[r]):
            assert c.r_lin.weight.grad is None and c.norm.weight.grad is None  # untouched -> None (AdamW skip)
            continue
        for j, lin in enumerate((c.r_lin, c.k_lin, c.v_lin, c.o_lin)):
            gclose(lin.weight.grad, Pc["W"].grad[r, j], what=f"g.W[{r},{j}]")
        gclose(c.norm.weight.grad, Pc["norm_w"].grad[r], what="g.norm_w")
        gclose(c.raw_decay.grad, Pc["raw_decay"].grad[r], what="g.raw_decay")
        if afd:
            gclose(c.process_noise_param.grad, Pc["pn_param"].grad[r], what="g.pn")
            gclose(c.obs_noise_param.grad, Pc["on_param"].grad[r], what="g.on")
        if rt:
            gclose(c.radius_process_noise_param.grad, Pc["rpn_param"].grad[r], what="g.rpn")
            gclose(c.radius_obs_noise_param.grad, Pc["ron_param"].grad[r], what="g.ron")


def test_region_soft_kwta_and_schedule():
    """tests/test_kwta.py:15-38 on the GPU + soft-mode gradient vs oracle."""
    from ironcortex_amd.region import RWKVRegionCell
    from oracle import hotpath as O
    d = 32
    t = gi.region_inputs(77, d)
    cell = RWKVRegionCell(d, m_time_pairs=0, kwta_k=2, kwta_k_start=16, kwta_k_schedule=10,
                          kwta_soft_mode=True, kwta_soft_temp=5.0).to(DEV)
    with torch.no_grad():
        for lin, k in ((cell.r_lin, "Wr"), (cell.k_lin, "Wk"), (cell.v_lin, "Wv"), (cell.o_lin, "Wo")):
            lin.weight.copy_(t[k])
    x = t["x_in"].to(DEV).requires_grad_(True)
    cell.global_step = 4
    h = cell.step(x, 0.0)
    (h * t["g_h"].to(DEV)).sum().backward()
    k, soft, temp = O.kwta_schedule(2, 16, 10, 4, True, 5.0)
    assert soft
    p = {k2: t[k2].clone() for k2 in ("Wr", "Wk", "Wv", "Wo")}
    p.update(norm_w=torch.ones(d), raw_decay=torch.zeros(d), W_time=torch.zeros(0))
    st = {"num": torch.zeros(d), "den": torch.zeros(d), "var": torch.zeros(d), "pred": torch.zeros(d),
          "radius": torch.zeros(1), "radius_var": torch.zeros(1), "surprise": torch.tensor(0.0)}
    xc = t["x_in"].clone().requires_grad_(True)
    hc, _ = O.region_step(p, st, xc, 0.0, dt=0, kwta_k=k, kwta_soft=True, kwta_temp=temp)
    (hc * t["g_h"]).sum().backward()
    close(h, hc, what="soft h")
    gclose(x.grad, xc.grad, what="soft g.x")
    cell.global_step = 10
    h1 = cell.step(t["x_in"].to(DEV), 0.0)
    assert int((h1 != 0).sum()) == 2
    cell2 = RWKVRegionCell(d, m_time_pairs=0, disable_kwta=True).to(DEV)
    assert int((cell2.step(t["x_in"].to(DEV), 0.0) != 0).sum()) == d


# --------------------------------------------------------------------------- KWTA / RMSNorm / goodness
@pytest.mark.parametrize("i", range(len(gi.KWTA_CASES)))
def test_kwta_vs_reference(i):
    from ironcortex_amd.utils import KWTA
    gold = load("kwta.npz")
    d, k = gi.KWTA_CASES[i]
    x = gi.kwta_rows(7000 + i, d).to(DEV)
    assert np.array_equal(KWTA(x, k).cpu().numpy(), gold[f"hard.{d}.{k}"])  # bit-exact incl. ties
    close(KWTA(x, k, soft=True, temp=0.7), gold[f"soft.{d}.{k}"], atol=1e-6, what="soft")
    assert np.array_equal(KWTA(gi.kwta_rows(7100, 32).to(DEV), 0).cpu().numpy(), gold["zero_k"])
    from oracle import hotpath as O
    xg = x.clone().requires_grad_(True)
    KWTA(xg, k).sum().backward()
    xc = gi.kwta_rows(7000 + i, d).requires_grad_(True)
    O.kwta(xc, k).sum().backward()
    assert torch.equal(xg.grad.cpu(), xc.grad)


def test_kwta_large_rows_property():
    """Full-size property check: exactly k winners on tie-free rows, all >= every loser."""
    from ironcortex_amd.utils import KWTA
    x = torch.randn(4096, 256, device=DEV)
    y = KWTA(x, 32)
    nz = (y != 0)
    assert torch.all(nz.sum(-1) == 32)
    lo = torch.where(nz, x.abs(), torch.full_like(x, float("inf"))).min(-1).values
    hi = torch.where(~nz, x.abs(), torch.zeros_like(x)).max(-1).values
    assert torch.all(lo >= hi)
    assert torch.equal(y[nz], x[nz])


def test_rmsnorm_and_goodness_vs_oracle():
    from ironcortex_amd.utils import RMSNorm, goodness
    from ironcortex_amd.hip import ops
    from oracle import hotpath as O
    for d in (8, 128, 256, 512):
        x = gi.det(d, 5, 3, d)
        w = gi.det(d + 1, d, scale=0.1, shift=1.0)
        norm = RMSNorm(d).to(DEV)
        with torch.no_grad():
            norm.weight.copy_(w)
        xg = x.to(DEV).requires_grad_(True)
        y = norm(xg)
        (y * gi.det(d + 2, 5, 3, d).to(DEV)).sum().backward()
        xc, wc = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
        yc = O.rmsnorm(xc, wc)
        (yc * gi.det(d + 2, 5, 3, d)).sum().backward()
        close(y, yc, what="rmsnorm"); gclose(xg.grad, xc.grad, what="g.x"); gclose(norm.weight.grad, wc.grad, what="g.w")
        hg = x.to(DEV).requires_grad_(True)
        g = ops.GoodnessFn.apply(hg)
        (g * gi.det(d + 3, 5, 3).to(DEV)).sum().backward()
        hc = x.clone().requires_grad_(True)
        gc = O.ff_goodness(hc)
        (gc * gi.det(d + 3, 5, 3)).sum().backward()
        close(g, gc, what="goodness"); gclose(hg.grad, hc.grad, what="g.H")
        close(goodness(x.to(DEV)), gc.mean(), what="goodness scalar")


# --------------------------------------------------------------------------- gate
@pytest.mark.parametrize("R", gi.GATE_R)
def test_gate_vs_reference(R):
    from ironcortex_amd.gate import Gate
    from ironcortex_amd.wiring import hex_neighbors
    gold = load("gate.npz")
    nb = hex_neighbors(R)
    gate = Gate(R, nb, {"sensor": 0, "motor": R - 1}).to(DEV)
    gate.gain_ema.copy_(gi.det(8000 + R, R, scale=0.2)); gate.homeo.copy_(gi.det(8001 + R, R, scale=0.05))
    H = gi.det(8002 + R, R, 16)
    H[R // 2:] = 0.0
    H = H.to(DEV)
    for fi, (focus_any, vb, um) in enumerate([(True, 0.13, 0.41), (False, -0.2, 0.0)]):
        gate.value_bias = vb
        focus = torch.zeros(5, dtype=torch.bool, device=DEV)
        focus[2] = focus_any
        sc = gate.score_regions(H, focus, um)
        ref_sc = torch.as_tensor(gold[f"scores.{R}.{fi}"])
        close(sc, ref_sc, rtol=1e-6, atol=1e-7, what="scores")
        distinct = torch.unique(ref_sc).numel() == R
        for k_active in (2, 4, 8, 32):
            for burst in (0, max(1, R // 8)):
                ref = gold[f"mask.{R}.{fi}.{k_active}.{burst}"]
                # same scores in -> identical selection whenever the reference saw no exact ties
                m = gate.select_k(ref_sc.to(DEV), k_active, burst)
                if distinct or R <= 16:
                    assert np.array_equal(m.cpu().numpy(), ref), (R, fi, k_active, burst)
                # fused path on the GPU's own scores
                _, m2 = gate.score_select(H.unsqueeze(0), focus.any().to(torch.uint8).reshape(1), um, k_active, burst)
                if distinct and torch.equal(sc.cpu(), ref_sc):
                    assert np.array_equal(m2[0].bool().cpu().numpy(), ref)


def test_gate_select_vs_oracle_random():
    """Stable total order (score desc, index asc) + greedy suppression == oracle, incl. ties, k=0."""
    from ironcortex_amd.gate import Gate
    from ironcortex_amd.wiring import hex_neighbors
    from oracle import hotpath as O
    # (above 64 regions ic_gate_select ranks with a sorting network padded to a power of two and walks 64 candidates at a
    # time: sizes on both sides of the chunk and padding boundaries)
    for R in (4, 10, 36, 64, 65, 100, 128, 300, 512, 1000):
        nb = hex_neighbors(R)
        gate = Gate(R, nb, {"sensor": 0, "motor": R - 1}).to(DEV)
        for trial in range(4):
            sc = gi.det(900 + R + trial, R)
            if trial % 2:
                sc = torch.round(sc * 2) / 2  # many ties
            for k_active, burst in ((0, 0), (2, 0), (3, 1), (8, 4), (32, 64), (32, R // 8), (200, 0)):
                ref = O.gate_select(sc, k_active, burst, nb, 0, R - 1, stable=True)
                got = gate.select_k(sc.to(DEV), k_active, burst)
                assert torch.equal(got.cpu(), ref), (R, trial, k_active, burst)


def test_gate_select_large_map_corner_cases():
    """Above 64 regions, on a symmetric adjacency, ic_gate_select finds the walk's selections in parallel rounds
    (ic_gate.h).  Corners: a monotone ridge of scores (as many rounds as the map is long), constant scores (index
    order alone decides), an ASYMMETRIC adjacency (the kernel notices and takes the sorted walk), a map without edges
    (every region is a candidate), and several streams with their own k in one launch."""
    from ironcortex_amd.gate import Gate
    from ironcortex_amd.wiring import hex_neighbors
    from oracle import hotpath as O
    for R in (100, 512, 1024):
        maps = {"hex": hex_neighbors(R), "directed": [[(i + 1) % R, (i + 7) % R] for i in range(R)], "no-edges": [[] for _ in range(R)],
                "path": [[j for j in (i - 1, i + 1) if 0 <= j < R] for i in range(R)]}
        for name, nb in maps.items():
            gate = Gate(R, nb, {"sensor": 0, "motor": R - 1}).to(DEV)
            ramp = torch.arange(R, dtype=torch.float32)
            for what, sc in (("down", -ramp), ("up", ramp), ("flat", torch.zeros(R)), ("rand", gi.det(950 + R, R)),
                             ("coarse", torch.round(gi.det(951 + R, R) * 3))):
                for k_active, burst in ((0, 0), (1, 0), (3, 0), (32, 0), (32, R // 8), (R, 0)):
                    ref = O.gate_select(sc, k_active, burst, nb, 0, R - 1, stable=True)
                    got = gate.select_k(sc.to(DEV), k_active, burst)
                    assert torch.equal(got.cpu(), ref), (R, name, what, k_active, burst)


# --------------------------------------------------------------------------- router
def make_router(name, R, d, mode, prec, t):
    from ironcortex_amd.gate import Router
    from ironcortex_amd.wiring import hex_neighbors
    router = Router(hex_neighbors(R), d, R, enable_precision_routed_messages=prec, edge_transform_mode=mode).to(DEV)
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
    return router


@pytest.mark.parametrize("ci", range(len(gi.ROUTER_CASES)))
def test_router_vs_reference(ci):
    from ironcortex_amd.wiring import hex_axial_coords, hex_neighbors
    gold = load("router.npz")
    name, R, d, mode, prec = gi.ROUTER_CASES[ci]
    E = sum(len(x) for x in hex_neighbors(R))
    t = gi.router_inputs(9000 + 100 * ci, R, d, E, mode)
    router = make_router(name, R, d, mode, prec, t)
    coords = hex_axial_coords(R).to(DEV)
    H = t["H"].to(DEV).requires_grad_(True)
    M = router.messages(H, t["mask_prev"].to(DEV), coords)
    (M * t["g_M"].to(DEV)).sum().backward()
    pre = name + "."
    close(M, gold[pre + "M"], what="M")
    gclose(H.grad, gold[pre + "g.H"], what="g.H")
    gclose(router.beta_cos.grad, gold[pre + "g.beta_cos"], what="g.beta_cos")
    gclose(router.beta_sin.grad, gold[pre + "g.beta_sin"], what="g.beta_sin")
    zeros = torch.zeros(d, d)
    getg = lambda p, z: p.grad.cpu() if p.grad is not None else z
    if mode == "per_edge":
        gw = torch.stack([getg(router.W_edge[k].weight, zeros) for k in router.edge_keys])
        gclose(gw, gold[pre + "g.W_edge"], what="g.W_edge")
        # inactive senders leave their edge weights without a gradient
        src_active = t["mask_prev"][router.src_idx.cpu()]
        for e, k in enumerate(router.edge_keys):
            assert (router.W_edge[k].weight.grad is not None) == bool(src_active[e]), k
    elif mode == "by_direction":
        gclose(torch.stack([getg(l.weight, zeros) for l in router.W_dir]), gold[pre + "g.W_dir"], what="g.W_dir")
    else:  # factorized: shared U, V on rocBLAS GEMMs + per-edge diagonal
        gclose(router.U.weight.grad, gold[pre + "g.U"], what="g.U")
        gclose(router.V.weight.grad, gold[pre + "g.V"], what="g.V")
        gclose(torch.stack([getg(router.w_edge[k], torch.zeros(d)) for k in router.edge_keys]), gold[pre + "g.w_edge"], what="g.w_edge")
        src_active = t["mask_prev"][router.src_idx.cpu()]
        for e, k in enumerate(router.edge_keys):
            assert (router.w_edge[k].grad is not None) == bool(src_active[e]), k
    if prec:
        w = torch.zeros(E)
        for e, k in enumerate(router.edge_keys):
            w[e] = router.last_weights.get(k, 0.0)
        close(w, gold[pre + "weights"], rtol=5e-5, atol=1e-6, what="weights")
        gclose(torch.stack([getg(l.weight, zeros) for l in router.query_lin]), gold[pre + "g.query_W"], what="g.query")
        if mode in ("per_edge", "factorized"):
            gclose(torch.stack([getg(router.key_lin[k].weight, zeros) for k in router.edge_keys]), gold[pre + "g.key_W"], what="g.key")
            gclose(torch.stack([getg(router.raw_P_edge[k], torch.zeros(d)) for k in router.edge_keys]), gold[pre + "g.raw_P"], what="g.rawP")
        else:
            gclose(torch.stack([getg(l.weight, zeros) for l in router.key_lin_dir]), gold[pre + "g.key_W"], what="g.key")
            gclose(torch.stack([getg(p, torch.zeros(d)) for p in router.raw_P_dir]), gold[pre + "g.raw_P"], what="g.rawP")


def test_router_precision_known_answer():
    """tests/test_router_precision.py:46-65 run unmodified in spirit (incl. `.weight.data = eye`)."""
    from ironcortex_amd.gate import Router
    gold = load("router.npz")
    R, d = 3, 4
    router = Router([[1], [0, 2], [1]], d, R, enable_precision_routed_messages=True).to(DEV)
    router.fb_alpha = 0.0
    for r in range(R):
        router.query_lin[r].weight.data = torch.eye(d, device=DEV)
    for lin in router.key_lin.values():
        lin.weight.data = torch.eye(d, device=DEV)
    for lin in router.W_edge.values():
        lin.weight.data = torch.eye(d, device=DEV) * 0.5
    for p in router.raw_P_edge.values():
        p.data.fill_(1.0)
    H = torch.zeros(R, d, device=DEV); H[2] = 5.0; H[0] = 0.25
    M = router.messages(H, torch.ones(R, dtype=torch.bool, device=DEV), torch.zeros(R, 2, device=DEV))
    close(M, gold["line3.M"], atol=1e-6, what="M")
    assert router.last_weights["0->1"] > router.last_weights["2->1"]
    close(torch.tensor(router.last_weights["0->1"]), gold["line3.w01"], atol=1e-6)


def test_router_multistream_and_empty():
    """S > 1 streams with different activity == per-stream oracle; no active sender -> zeros."""
    from ironcortex_amd.wiring import hex_axial_coords, hex_neighbors
    from oracle import hotpath as O
    R, d = 10, 64
    nb = hex_neighbors(R)
    E = sum(len(x) for x in nb)
    # S = 3: stream-batched kernel; S = 12 (default routing): the LDS-accumulator kernel for many streams
    for prec, S in ((False, 3), (True, 3), (False, 12)):
        t = gi.router_inputs(4242, R, d, E, "per_edge")
        router = make_router("ms", R, d, "per_edge", prec, t)
        coords = hex_axial_coords(R)
        H = gi.det(4300, S, R, d)
        mask = gi.det(4301, S, R, kind="rand") < 0.5
        mask[2] = False
        Hg = H.to(DEV).requires_grad_(True)
        M = router.route(Hg, mask.to(torch.uint8).to(DEV), coords.to(DEV))
        gM = gi.det(4302, S, R, d)
        (M * gM.to(DEV)).sum().backward()
        assert torch.all(M[2] == 0)
        P = {k: t[k].clone().requires_grad_(True) for k in ("W_edge", "key_W", "raw_P", "query_W", "beta_cos", "beta_sin")}
        P["W_reg"] = t["W_reg"]
        Hc = H.clone().requires_grad_(True)
        Mc = torch.stack([O.router_messages(P, Hc[s], mask[s], coords, nb, precision=prec)[0] for s in range(S)])
        (Mc * gM).sum().backward()
        close(M, Mc, what="M"); gclose(Hg.grad, Hc.grad, what="g.H")
        gw = torch.stack([router.W_edge[k].weight.grad.cpu() if router.W_edge[k].weight.grad is not None else torch.zeros(d, d) for k in router.edge_keys])
        gclose(gw, P["W_edge"].grad, what="g.W_edge")


def test_hebbian_vs_oracle():
    from ironcortex_amd.hip import ops
    from ironcortex_amd.wiring import hex_neighbors
    from oracle import hotpath as O
    R, d, B = 7, 32, 3
    nb = hex_neighbors(R)
    src, dst = O.edge_list(nb)
    E = len(src)
    W = gi.det(1, E, d, d, kind="xavier")
    H = gi.det(2, B, R, d)
    mask = gi.det(3, B, R, kind="rand") < 0.7
    gains = gi.det(4, R).tolist()
    ref = O.hebbian_updates(W, nb, gains, H, mask)
    coef = torch.zeros(B, E)
    for e in range(E):
        g = gains[dst[e]]
        if g > 0:
            for b in range(B):
                if mask[b, dst[e]] and mask[b, src[e]]:
                    coef[b, e] = 1e-3 * g
    Wg = W.clone().to(DEV)
    i32 = lambda x: torch.tensor(x, dtype=torch.int32, device=DEV)
    ops.hebbian_rank1(Wg, i32(dst), i32(src), i32(list(range(E))), coef.to(DEV), H.to(DEV))
    close(Wg, ref, rtol=1e-6, atol=1e-7, what="hebbian")
