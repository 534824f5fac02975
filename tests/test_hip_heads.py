"""GPU parity of the loop-heads entry points (ic_heads_fwd, ic_gemv_chain, ic_verify_score,
ic_steer_select) through the C ABI against the CPU oracle model's heads (oracle/model.py, which
is pinned to the reference by tests/test_oracle_model.py) on the same weights and inputs.

Tolerances: rtol 2e-5 / atol 2e-6 on vectors (GEMV sums in a different order than the CPU BLAS);
the workspace KWTA winners must be identical.
"""
import ctypes

import pytest
import torch
import torch.nn.functional as F

from tests.model_cases import load_case

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def close(a, b, rtol=2e-5, atol=2e-6, what=""):
    torch.testing.assert_close(a.detach().float().cpu().reshape(b.shape), b.detach().float(), rtol=rtol, atol=atol,
                               msg=lambda m: f"{what}: {m}")


def build_pair(name):
    """name of a fixture case, or "<case>@R": that case's configuration on a map of R regions with seeded fresh weights
    (no reference fixture: product and oracle share the weights) -- maps above 16 regions take the heads kernels'
    ballot-driven walks over the active rows instead of the all-rows loops."""
    import dataclasses
    import ironcortex_amd as ic
    from oracle import model as OM
    if "@" in name:
        base, R = name.split("@")
        _g, cfg, _sd, _t, _f = load_case(base)
        cfg = dataclasses.replace(cfg, R=int(R))
        torch.manual_seed(1234)
        prod = ic.CortexReasoner(ic.hex_neighbors(cfg.R), ic.hex_axial_coords(cfg.R), {"sensor": 0, "motor": cfg.R - 1}, cfg)
        sd = {k: v.detach().clone() for k, v in prod.state_dict().items()}
        orc = OM.build(cfg)
        orc.load_state_dict(sd)
        return cfg, prod.to(DEV), orc
    gold, cfg, sd, tokens, focus = load_case(name)
    prod = ic.CortexReasoner(ic.hex_neighbors(cfg.R), ic.hex_axial_coords(cfg.R), {"sensor": 0, "motor": cfg.R - 1}, cfg)
    prod.load_state_dict(sd)
    prod = prod.to(DEV)
    orc = OM.build(cfg)
    orc.load_state_dict(sd)
    return cfg, prod, orc


def step_inputs(cfg, seed, n_active):
    g = torch.Generator().manual_seed(seed)
    R, d = cfg.R, cfg.d
    mask = torch.zeros(R, dtype=torch.bool)
    mask[torch.randperm(R, generator=g)[:n_active]] = True
    mask[0] = mask[R - 1] = True  # sensor / motor are forced on (gate.py:78-80)
    H = torch.randn(R, d, generator=g) * (torch.rand(R, d, generator=g) < 0.2) * mask.unsqueeze(-1)
    H_par = torch.randn(R, d, generator=g) * (torch.rand(R, d, generator=g) < 0.2)
    slots = torch.randn(8, d, generator=g) * (torch.rand(8, d, generator=g) < 0.25)
    surprise = torch.rand(R, generator=g)
    return H, H_par, mask, slots, surprise


@pytest.mark.parametrize("name,n_active", [("default", 4), ("allflags", 10), ("allflags", 2), ("allflags@40", 13),
                                           ("allflags@150", 40), ("allflags@150", 2)])
def test_heads_fwd_vs_oracle(name, n_active):
    from ironcortex_amd import hip
    from ironcortex_amd.utils import context_logprob
    from oracle import hotpath as O
    cfg, prod, orc = build_pair(name)
    R, d, V, motor = cfg.R, cfg.d, cfg.V, cfg.R - 1
    H, H_par, mask, slots, surprise = step_inputs(cfg, 5 + n_active, n_active)
    # ---- oracle: reference model.py:296-318 as restated in OracleReasoner.forward_inner_step
    with torch.no_grad():
        orc.work.slots = slots.clone()
        active = mask.nonzero(as_tuple=False).squeeze(-1)
        delta = torch.zeros_like(slots)
        n_write = min(active.numel(), 8)
        delta[:n_write] = H[active[:n_write]]
        orc.work.write(delta)
        parts = [orc.lm_head(H[motor])[0]] + [orc.region_heads[r](H[r]) for r in active.tolist() if r != motor]
        logits = torch.stack(parts, 0).mean(0)
        p = F.softmax(logits, -1)
        stats = torch.stack([1.0 - p.max(), context_logprob(logits),
                             O.ff_goodness(H).mean() - O.ff_goodness(H_par).mean(), surprise.mean()])
    # ---- HIP
    hs = prod.engine._heads_setup(torch.device(DEV))
    ha, buf = hs["ha"], hs["buf"]
    t = {k: v.to(DEV).contiguous() for k, v in dict(H=H, H_par=H_par, slots=slots, surprise=surprise).items()}
    mk = mask.to(torch.uint8).to(DEV)
    out_slots, out_logits = torch.zeros(8, d, device=DEV), torch.zeros(V, device=DEV)
    st = torch.zeros(4, device=DEV)
    for k, v in dict(H=t["H"], H_par=t["H_par"], mask=mk, slots_in=t["slots"], slots_out=out_slots,
                     surprise=t["surprise"], logits=out_logits, stats=st).items():
        hip.set_field(ha, k, v)
    ha.aux_surprise = buf["aux"].data_ptr() + 4
    hip.check(hip.lib().ic_heads_fwd(ctypes.byref(ha), hip.stream_handle()), "ic_heads_fwd")
    torch.cuda.synchronize()
    assert torch.equal(out_slots.cpu() != 0, orc.work.slots != 0), "workspace KWTA winners differ"
    close(out_slots, orc.work.slots, what="slots")
    close(out_logits, logits, what="logits")
    close(buf["probs"][0], p, what="probs")
    close(st, stats, rtol=1e-4, atol=1e-6, what="stats")
    close(buf["aux"][0, 1], surprise.mean(), what="aux surprise")
    # the product's differentiable PyTorch heads (general path) agree with the kernel
    with torch.no_grad():
        prod.work.slots = t["slots"].clone()
        lg2, _rtd, _ws, _m = prod._heads(t["H"], mk)
    close(lg2, logits, what="general-path logits")
    close(prod.work.slots, orc.work.slots, what="general-path slots")


@pytest.mark.parametrize("name", ["allflags", "allflags@40"])
def test_steer_chain_and_verify_score_vs_oracle(name):
    from ironcortex_amd import hip
    cfg, prod, orc = build_pair(name)
    R, d, V, motor = cfg.R, cfg.d, cfg.V, cfg.R - 1
    dev = torch.device(DEV)
    hs = prod.engine._heads_setup(dev)
    buf, P, L, st = hs["buf"], hs["P"], hip.lib(), hip.stream_handle()
    cands = [step_inputs(cfg, 40 + c, 3 + c) for c in range(2)]
    scores = torch.tensor([0.25, 0.75])
    # ---- oracle: model.py:371-386 on the better candidate
    H, _, mask, _, _ = cands[1]
    with torch.no_grad():
        act = mask.nonzero(as_tuple=False).squeeze(-1)
        ws_prev = H[act].mean(0)
        _p, g = orc.plan(ws_prev, H[motor])
        v_hat = orc.critic(ws_prev, g)
    sa = hs["sa"]
    keep = []
    for c, (Hc, _, mc, _, _) in enumerate(cands):
        hd, md = Hc.to(DEV).contiguous(), mc.to(torch.uint8).to(DEV)
        keep += [hd, md]
        sa.H_cand[c], sa.mask_cand[c] = hd.data_ptr(), md.data_ptr()
    sa.n_cand = 2
    sc = scores.to(DEV)
    sa.scores = sc.data_ptr()
    hip.check(L.ic_steer_select(ctypes.byref(sa), st) or L.ic_gemv_chain(hs["steer"], 6, st), "steer")
    torch.cuda.synchronize()
    close(buf["ws_prev"][0], ws_prev, what="ws_prev")
    close(buf["g"][0], g, what="subgoal g")
    close(buf["ls"][40], v_hat, what="v_hat")
    # no scores -> candidate 0; no active region -> zeros (model.py:374-375)
    sa.scores = None
    zero_mask = torch.zeros(R, dtype=torch.uint8, device=DEV)
    sa.mask_cand[0] = zero_mask.data_ptr()
    hip.check(L.ic_steer_select(ctypes.byref(sa), st), "steer")
    torch.cuda.synchronize()
    assert float(buf["ws_prev"][0].abs().max()) == 0.0
    close(buf["motor"], cands[0][0][motor], what="motor row")
    # ---- verifier energy + branch score (energy.py:12-30, model.py:425-438)
    g2 = torch.Generator().manual_seed(9)
    motor_state = torch.randn(d, generator=g2)
    probs = F.softmax(torch.randn(V, generator=g2), -1)
    aux = torch.rand(3, generator=g2)
    stats = torch.randn(4, generator=g2)
    with torch.no_grad():
        E = orc.verify(motor_state, probs, aux=aux)
        score = 0.6 * stats[2] + 0.2 * (-E) + 0.2 * stats[1]
    ms = motor_state.to(DEV)
    buf["probs"][0].copy_(probs)
    buf["aux"][0, :3].copy_(aux)
    stats_d, score_d = stats.to(DEV), torch.zeros(1, device=DEV)
    ver = hs["ver"]
    ver[0].x0 = ms.data_ptr()
    ver[2].y = buf["ls"].data_ptr() + 48 * 4
    hip.check(L.ic_verify_score(ver, 3, stats_d.data_ptr(), score_d.data_ptr(), st), "ic_verify_score")
    torch.cuda.synchronize()
    close(buf["ls"][48], E, rtol=1e-4, atol=1e-5, what="verifier energy")
    close(score_d[0], score, rtol=1e-4, atol=1e-5, what="branch score")
    # the standalone score entry point
    hip.check(L.ic_branch_score(stats_d.data_ptr(), buf["ls"].data_ptr() + 48 * 4, score_d.data_ptr(), st), "ic_branch_score")
    torch.cuda.synchronize()
    close(score_d[0], score, rtol=1e-4, atol=1e-5, what="ic_branch_score")


def test_gemv_chain_rejects_bad_descriptors():
    from ironcortex_amd import hip
    G = hip.CLASSES["ic_gemv_desc"]
    d = (G * 1)()
    rc = hip.lib().ic_gemv_chain(d, 1, hip.stream_handle())
    assert rc == hip.ENUMS["IC_EINVAL"] and b"ic_gemv_chain" in hip.lib().ic_last_error()


@pytest.mark.parametrize("name", ["allflags", "allflags@40", "allflags@150"])
def test_strided_streams_equal_separate_calls(name):
    """The batched-branch launches: gate + router with the two streams' inputs in REVERSED slot
    order (negative stream stride) must equal two single-stream calls; same for the batched heads
    and verifier chain."""
    import ironcortex_amd as ic
    from ironcortex_amd import hip
    from ironcortex_amd.hip import ops
    cfg, prod, _ = build_pair(name)
    R, d, V = cfg.R, cfg.d, cfg.V
    dev = torch.device(DEV)
    L, st = hip.lib(), hip.stream_handle()
    g = torch.Generator().manual_seed(3)
    H2 = (torch.randn(2, R, d, generator=g) * (torch.rand(2, R, d, generator=g) < 0.3)).to(DEV)
    mp2 = (torch.rand(2, R, generator=g) < 0.6).to(torch.uint8).to(DEV)
    mp2[:, 0] = 1
    gate, router = prod.gate, prod.router
    off, nbr = gate._tables(dev)
    tables = router._edge_tables(prod.reg_coords)
    ebias = router._edge_bias(prod.reg_coords).detach().contiguous()
    E = router.E

    def run(S, Hptr, mptr, hs, ms, fused=False):
        out = dict(scores=torch.zeros(S, R, device=DEV), mask=torch.zeros(S, R, dtype=torch.uint8, device=DEV),
                   M=torch.zeros(S, R, d, device=DEV), msg=torch.zeros(S, E, d, device=DEV),
                   keyv=torch.zeros(S, E, d, device=DEV), q=torch.zeros(S, R, d, device=DEV), wgt=torch.zeros(S, E, device=DEV))
        fa = torch.ones(S, dtype=torch.uint8, device=DEV)
        ga = hip.new("ic_gate_args", S=S, R=R, d=d, sensor_idx=0, motor_idx=R - 1, k_total=5, force_io=1, w1=1.0, w2=0.3,
                     w3=0.5, w4=0.2, t5=0.1, t6=0.05, gain_ema=gate.gain_ema, homeo=gate.homeo, focus_any=fa,
                     nbr_off=off, nbr=nbr, scores=out["scores"], mask=out["mask"])
        ga.H, ga.H_sstride = Hptr, hs
        spec = ops.RouterSpec(R, d, E, router.nW, True)
        ra = ops._router_args(spec, tables, router._params(), S)
        ra.ebias, ra.H, ra.mask_prev, ra.H_sstride, ra.mask_sstride = ebias.data_ptr(), Hptr, mptr, hs, ms
        for k in ("M", "msg", "keyv", "q", "wgt"):
            hip.set_field(ra, k, out[k])
        if fused:  # gate riding in the router's launch
            hip.check(L.ic_front_fwd(ctypes.byref(ga), ctypes.byref(ra), st), "front")
        else:
            hip.check(L.ic_gate_fwd(ctypes.byref(ga), st), "gate")
            hip.check(L.ic_router_fwd(ctypes.byref(ra), st), "router")
        torch.cuda.synchronize()
        return out

    # streams in reversed slot order: stream 0 reads H2[1], stream 1 reads H2[0]
    both = run(2, H2[1].data_ptr(), mp2[1].data_ptr(), -R * d, -R)
    front = run(2, H2[1].data_ptr(), mp2[1].data_ptr(), -R * d, -R, fused=True)
    for k in both:
        assert torch.equal(front[k], both[k]), ("ic_front_fwd != ic_gate_fwd + ic_router_fwd", k)
    for s_, src in ((0, 1), (1, 0)):
        one = run(1, H2[src].data_ptr(), mp2[src].data_ptr(), 0, 0)
        for k in one:
            assert torch.equal(both[k][s_], one[k][0]), (k, s_)

    # batched heads + verifier == one step at a time
    hs_ = prod.engine._heads_setup(dev)
    ha, ver, buf = hs_["ha"], hs_["ver"], hs_["buf"]
    Hc = (torch.randn(2, R, d, generator=g) * (torch.rand(2, R, d, generator=g) < 0.2)).to(DEV)
    mk = (torch.rand(2, R, generator=g) < 0.5).to(torch.uint8).to(DEV)
    mk[:, 0] = mk[:, R - 1] = 1
    slots_in = torch.randn(2, 8, d, generator=g).to(DEV)
    sur = torch.rand(2, R, generator=g).to(DEV)

    steer = hs_["steer"]

    def run_heads(S, first, hp, hps, sp, sps, plan=False):
        o = dict(slots=torch.zeros(S, 8, d, device=DEV), logits=torch.zeros(S, V, device=DEV))
        ls = torch.zeros(64, device=DEV)
        ha.S, ha.Hpar_sstride, ha.slots_in_sstride = S, hps, sps
        ha.H, ha.mask, ha.surprise = Hc[first].data_ptr(), mk[first].data_ptr(), sur[first].data_ptr()
        ha.H_par, ha.slots_in = hp, sp
        ha.slots_out, ha.logits, ha.stats = o["slots"].data_ptr(), o["logits"].data_ptr(), ls.data_ptr() + 32
        ha.aux_surprise = buf["aux"].data_ptr() + 4
        ha.ws_mean = buf["ws_prev"].data_ptr()  # each stream's planner input comes out of the first launch
        buf["ws_prev"].zero_(); buf["p_h"].zero_()
        buf["p"].zero_()
        steer[0].nbatch = steer[1].nbatch = S
        if plan:  # the first two planner layers ride in the two launches of the heads
            hip.check(L.ic_heads_gemv_fwd(ctypes.byref(ha), ctypes.byref(steer[0]), ctypes.byref(steer[1]), st), "heads+gemv")
        else:
            hip.check(L.ic_heads_fwd(ctypes.byref(ha), st), "heads")
            hip.check(L.ic_gemv_chain(steer, 2, st), "planner layers 0, 1")
        steer[0].nbatch = steer[1].nbatch = 0
        ha.ws_mean = None
        o["ws_prev"], o["p_h"], o["p"] = buf["ws_prev"].clone(), buf["p_h"].clone(), buf["p"].clone()
        ver[0].x0 = Hc[first].data_ptr() + (R - 1) * d * 4
        ver[2].y = ls.data_ptr() + 48 * 4
        for g_ in ver:
            g_.nbatch = S
        hip.check(L.ic_verify_score(ver, 3, ls.data_ptr() + 32, ls.data_ptr(), st), "verify")
        torch.cuda.synchronize()
        o["ls"] = ls.clone()
        return o

    both = run_heads(2, 0, H2[1].data_ptr(), -R * d, slots_in[1].data_ptr(), -8 * d)
    rode = run_heads(2, 0, H2[1].data_ptr(), -R * d, slots_in[1].data_ptr(), -8 * d, plan=True)
    for k in both:
        assert torch.equal(both[k], rode[k]), ("ic_heads_gemv_fwd != ic_heads_fwd + ic_gemv_chain", k)
    on = mk.bool()
    for b_ in range(2):  # ws_mean == mean of the active rows (model.py:344-349)
        close(both["ws_prev"][b_], Hc[b_][on[b_]].mean(0).cpu(), what="ws_mean")
    assert float(both["p_h"].abs().max()) > 0 and float(both["p"].abs().max()) > 0
    for s_, src in ((0, 1), (1, 0)):
        # single call for step s_ (its scratch is block 0, its statistics land at ls[8:12], score ls[0])
        ha_H, ha_m, ha_s = Hc[s_], mk[s_], sur[s_]
        o = dict(slots=torch.zeros(1, 8, d, device=DEV), logits=torch.zeros(1, V, device=DEV))
        ls = torch.zeros(64, device=DEV)
        ha.S, ha.Hpar_sstride, ha.slots_in_sstride = 1, 0, 0
        ha.H, ha.mask, ha.surprise = ha_H.data_ptr(), ha_m.data_ptr(), ha_s.data_ptr()
        ha.H_par, ha.slots_in = H2[src].data_ptr(), slots_in[src].data_ptr()
        ha.slots_out, ha.logits, ha.stats = o["slots"].data_ptr(), o["logits"].data_ptr(), ls.data_ptr() + 32
        hip.check(L.ic_heads_fwd(ctypes.byref(ha), st), "heads")
        ver[0].x0 = ha_H.data_ptr() + (R - 1) * d * 4
        ver[2].y = ls.data_ptr() + 48 * 4
        for g_ in ver:
            g_.nbatch = 1
        hip.check(L.ic_verify_score(ver, 3, ls.data_ptr() + 32, ls.data_ptr(), st), "verify")
        torch.cuda.synchronize()
        assert torch.equal(both["slots"][s_], o["slots"][0]) and torch.equal(both["logits"][s_], o["logits"][0])
        assert torch.equal(both["ls"][8 + 4 * s_: 12 + 4 * s_], ls[8:12]), "statistics"
        # (a single small chain runs in the one-workgroup kernel, the batched one layer by layer:
        # different summation order, so tolerance instead of bit equality)
        close(both["ls"][s_], ls[0].cpu(), rtol=1e-5, atol=1e-6, what="score")
        close(both["ls"][48 + s_], ls[48].cpu(), rtol=1e-5, atol=1e-6, what="energy")


def test_planner_for_both_candidates_in_the_verifier_launches():
    """The engine's merged launches (ic_gemv_multi: verifier layer i next to planner layer i, both
    batched over the two branches) == the two chains run separately, bit for bit."""
    from ironcortex_amd import hip
    cfg, prod, _ = build_pair("allflags")
    R, d, V = cfg.R, cfg.d, cfg.V
    dev = torch.device(DEV)
    L, st = hip.lib(), hip.stream_handle()
    hs_ = prod.engine._heads_setup(dev)
    steer, ver, buf, pairs, tail = hs_["steer"], hs_["ver"], hs_["buf"], hs_["pairs"], hs_["tail"]
    g = torch.Generator().manual_seed(21)
    Hc = (torch.randn(2, R, d, generator=g) * (torch.rand(2, R, d, generator=g) < 0.3)).to(DEV)
    ws = torch.randn(2, d, generator=g).to(DEV)
    pr = torch.softmax(torch.randn(2, V, generator=g), -1).to(DEV)
    stats = torch.tensor([0.0, -1.5, 0.25, 0.0, 0.0, -2.0, 0.5, 0.0], device=DEV)
    outs = []
    for merged in (False, True):
        for k in ("p_h", "p", "g_h", "g", "c_h", "h1", "h2", "ls"):
            buf[k].zero_()
        buf["ws_prev"].copy_(ws)
        buf["probs"].copy_(pr)
        ls = buf["ls"]
        ls[8:16].copy_(stats)
        ver[0].x0 = Hc.data_ptr() + (R - 1) * d * 4
        ver[2].y = ls.data_ptr() + 48 * 4
        for g_ in list(ver) + list(steer):
            g_.nbatch = 2
        if merged:
            hip.check(L.ic_gemv_chain(steer, 2, st), "planner layers 0, 1 (ride in the heads' launches in the engine)")
            for i in range(3):
                pairs[i][0], pairs[i][1] = ver[i], steer[i + 2]
                hip.check(L.ic_gemv_multi(pairs[i], 2, ls.data_ptr() + 32, ls.data_ptr() if i == 2 else None, st), "multi")
            tail[0] = steer[5]
            hip.check(L.ic_gemv_chain(tail, 1, st), "tail")
        else:
            hip.check(L.ic_verify_score(ver, 3, ls.data_ptr() + 32, ls.data_ptr(), st), "verify")
            hip.check(L.ic_gemv_chain(steer, 6, st), "chain")
        torch.cuda.synchronize()
        outs.append({k: buf[k].clone() for k in ("p_h", "p", "g_h", "g", "c_h", "h1", "h2", "ls")})
    for g_ in list(ver) + list(steer):
        g_.nbatch = 0
    for k in outs[0]:
        assert torch.equal(outs[0][k], outs[1][k]), k
    assert float(outs[1]["ls"][40]) != float(outs[1]["ls"][41]) and float(outs[1]["ls"][0]) != 0.0
