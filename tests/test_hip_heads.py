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
    import ironcortex_amd as ic
    from oracle import model as OM
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


@pytest.mark.parametrize("name,n_active", [("default", 4), ("allflags", 10), ("allflags", 2)])
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


def test_steer_chain_and_verify_score_vs_oracle():
    from ironcortex_amd import hip
    cfg, prod, orc = build_pair("allflags")
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
    close(buf["ws_prev"], ws_prev, what="ws_prev")
    close(buf["g"], g, what="subgoal g")
    close(buf["ls"][40], v_hat, what="v_hat")
    # no scores -> candidate 0; no active region -> zeros (model.py:374-375)
    sa.scores = None
    zero_mask = torch.zeros(R, dtype=torch.uint8, device=DEV)
    sa.mask_cand[0] = zero_mask.data_ptr()
    hip.check(L.ic_steer_select(ctypes.byref(sa), st), "steer")
    torch.cuda.synchronize()
    assert float(buf["ws_prev"].abs().max()) == 0.0
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
