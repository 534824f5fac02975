"""IC_F_CHAIN (the branches of an outer iteration as one launch sequence) == the same steps run one
call after another, bit for bit: the arithmetic of a step is unchanged, only where its input
state comes from (registers / the previous stream's rows instead of a second call).
Reference semantics: model.py:388-392 runs the branches one after another on the same cells."""
import ctypes

import pytest
import torch

from tests.golden import inputs as gi

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _phase(P, spec, flags, S, state, M, sensor, mask, step_pos, kw):
    from ironcortex_amd.hip import ops, lib, check, stream_handle, ENUMS
    R, d = spec.R, spec.d
    a = ops._region_args(spec, P, S, step_pos, kw, flags, 1.0)
    z = lambda *s, dt=torch.float32: torch.zeros(*s, device=DEV, dtype=dt)
    out = {k: z(S, R, d) for k in ("H", "num_o", "den_o", "var_o", "pred_o", "x", "u1", "r_s", "k_s", "v_s", "y",
                                   "hpre", "gate", "sterm")}
    out.update({k: z(S, R) for k in ("radius_o", "radius_var_o", "surprise_o", "ynorm")})
    out["inv_s"] = z(S, R, 2)
    out["dt_o"] = z(S, R, dt=torch.int32)
    chain = bool(flags & ENUMS["IC_F_CHAIN"])
    tl = 1 if chain else S
    out["last_dir"], out["last_norm"] = z(tl, R, d), z(tl, R)
    tensors = dict(M=M, sensor=sensor, mask=mask, **state, **out)
    for k, t in tensors.items():
        setattr(a, k, t.data_ptr())
    rc = lib().ic_region_phase_fwd(ctypes.byref(a), stream_handle())
    check(rc, "ic_region_phase_fwd")
    torch.cuda.synchronize()
    return out


@pytest.mark.parametrize("R,d,S,noise_vec,rows", [(10, 256, 2, False, 0), (10, 256, 3, True, 0), (5, 512, 3, False, 0),
                                                 (7, 32, 4, False, 0), (6, 256, 2, False, 16)])
def test_chain_equals_sequential_calls(R, d, S, noise_vec, rows):
    from ironcortex_amd.hip import ops, ENUMS
    from oracle import hotpath as O
    seed = 900 + R + d + S
    E = ENUMS
    flags = E["IC_F_AFD"] | E["IC_F_RT"] | E["IC_F_PRED"] | E["IC_F_NORM_IN"] | (E["IC_F_NOISE_VEC"] if noise_vec else 0)
    ns = (R, d) if noise_vec else (R,)
    P = {"W": gi.det(seed, R, 4, d, d, kind="xavier"), "norm_w": gi.det(seed + 1, R, d, scale=0.1, shift=1.0),
         "raw_decay": gi.det(seed + 2, R, d, scale=0.5), "pn_param": gi.det(seed + 3, *ns, scale=0.5, shift=-3.0),
         "on_param": gi.det(seed + 4, *ns, scale=0.5), "rpn_param": gi.det(seed + 5, R, scale=0.5, shift=-3.0),
         "ron_param": gi.det(seed + 6, R, scale=0.5), "norm_in_w": gi.det(seed + 7, d, scale=0.1, shift=1.0),
         "W_time": O.time_freqs(d)}
    P = {k: v.to(DEV).contiguous() for k, v in P.items()}
    spec = ops.RegionPhaseSpec(R, d, P["W_time"].numel(), flags, sensor_idx=0, motor_idx=R - 1)
    state0 = {"num": gi.det(seed + 10, 1, R, d, scale=0.5), "den": gi.det(seed + 11, 1, R, d, kind="rand", shift=0.5),
              "var": gi.det(seed + 12, 1, R, d, kind="rand", scale=0.45, shift=0.05),
              "pred": gi.det(seed + 13, 1, R, d, scale=0.1), "radius": gi.det(seed + 14, 1, R, kind="rand", shift=0.5),
              "radius_var": gi.det(seed + 15, 1, R, kind="rand", scale=0.5, shift=0.1),
              "surprise": gi.det(seed + 16, 1, R, kind="rand"),
              "dt": (gi.det(seed + 17, 1, R, kind="rand") * 4).to(torch.int32)}
    state0 = {k: v.to(DEV).contiguous() for k, v in state0.items()}
    M = gi.det(seed + 18, S, R, d).to(DEV)
    sensor = gi.det(seed + 19, 1, d).to(DEV)
    mask = (gi.det(seed + 20, S, R, kind="rand") < 0.5)
    mask[:, 0] = True          # active in every stream
    mask[:, 1] = False         # never active
    mask[0, 2], mask[1:, 2] = False, True   # (0,1,..)
    mask[0, 3], mask[1:, 3] = True, False   # (1,0,..)
    mask = mask.to(torch.uint8).to(DEV).contiguous()
    kw = max(1, d // 8)
    old_rows = ops.ROWS_FWD
    ops.ROWS_FWD = rows
    try:
        # ---- one call per step, state handed over on the host
        seq, st = [], dict(state0)
        for s in range(S):
            o = _phase(P, spec, flags, 1, st, M[s:s + 1].contiguous(), sensor, mask[s:s + 1].contiguous(), 0.4285714, kw)
            seq.append(o)
            st = {"num": o["num_o"], "den": o["den_o"], "var": o["var_o"], "pred": o["pred_o"], "radius": o["radius_o"],
                  "radius_var": o["radius_var_o"], "surprise": o["surprise_o"], "dt": o["dt_o"]}
        # ---- the same steps as streams of one chained call
        ch = _phase(P, spec, flags | E["IC_F_CHAIN"], S, state0, M, sensor, mask, 0.4285714, kw)
    finally:
        ops.ROWS_FWD = old_rows
    act = mask.bool()
    for k in ("H", "num_o", "den_o", "var_o", "pred_o", "radius_o", "radius_var_o", "surprise_o", "dt_o"):
        ref = torch.cat([o[k] for o in seq], 0)
        assert torch.equal(ch[k], ref), f"{k}: chained call differs from sequential calls"
    for k in ("x", "u1", "r_s", "k_s", "v_s", "y", "hpre", "gate", "sterm", "ynorm", "inv_s"):  # saved: active rows
        ref = torch.cat([o[k] for o in seq], 0)
        assert torch.equal(ch[k][act], ref[act]), f"saved {k}: chained call differs on active rows"
    # telemetry of the one stream: the last step that had the region active
    for r in range(R):
        on = [s for s in range(S) if bool(act[s, r])]
        if on:
            assert torch.equal(ch["last_dir"][0, r], seq[on[-1]]["last_dir"][0, r])
            assert torch.equal(ch["last_norm"][0, r], seq[on[-1]]["last_norm"][0, r])


@pytest.mark.parametrize("S,k_total,d", [(2, 5, 32), (3, 3, 32), (6, 7, 32), (2, 4, 256), (3, 5, 256), (16, 3, 256)])
def test_loop_step_equals_front_then_region(S, k_total, d):
    """ic_loop_step_fwd (router combine inside the region's r/k/v kernel) == ic_front_fwd followed
    by ic_region_phase_fwd, bit for bit, including M and the edge weights of sat-out rows."""
    import ironcortex_amd as ic
    from ironcortex_amd import hip
    from ironcortex_amd.hip import ops, ENUMS
    from tests.model_cases import load_case, MODEL_CASES
    if d == 32:
        _, cfg, sd, _, _ = load_case("allflags")
    else:
        cfg, sd = ic.CortexConfig(**dict(MODEL_CASES["allflags"], d=d, R=7)), None
    torch.manual_seed(5)
    prod = ic.CortexReasoner(ic.hex_neighbors(cfg.R), ic.hex_axial_coords(cfg.R), {"sensor": 0, "motor": cfg.R - 1}, cfg)
    if sd is not None:
        prod.load_state_dict(sd)
    prod = prod.to(DEV)
    R = cfg.R
    assert cfg.d == d
    dev = torch.device(DEV)
    L, st = hip.lib(), hip.stream_handle()
    g = torch.Generator().manual_seed(40 + S)
    Hin = (torch.randn(S, R, d, generator=g) * (torch.rand(S, R, d, generator=g) < 0.3)).to(DEV)
    mprev = (torch.rand(S, R, generator=g) < 0.6).to(torch.uint8)
    mprev[:, 0] = 1
    mprev[S - 1] = 0  # a stream whose senders were all off
    mprev = mprev.to(DEV)
    gate, router, bank = prod.gate, prod.router, prod.bank
    assert router.enable_precision_routed_messages
    off, nbr = gate._tables(dev)
    tables = router._edge_tables(prod.reg_coords)
    ebias = router._edge_bias(prod.reg_coords).detach().contiguous()
    E = router.E
    bank.reset_state(dev, 1)
    kw_k, kw_soft, kw_temp = bank.kwta_now()
    flags = bank._flags(kw_soft, bank.cells[0].disable_kwta, True) | ENUMS["IC_F_CHAIN"]
    P = dict(bank.params())
    P["norm_in_w"] = prod.norm_in.weight
    sensor = torch.randn(1, d, generator=g).to(DEV)
    state = {"num": torch.randn(1, R, d, generator=g) * 0.5, "den": torch.rand(1, R, d, generator=g) + 0.5,
             "var": torch.rand(1, R, d, generator=g) * 0.45 + 0.05, "pred": torch.randn(1, R, d, generator=g) * 0.1,
             "radius": torch.rand(1, R, generator=g) + 0.5, "radius_var": torch.rand(1, R, generator=g) * 0.5 + 0.1,
             "surprise": torch.rand(1, R, generator=g), "dt": (torch.rand(1, R, generator=g) * 4).to(torch.int32)}
    state = {k: v.to(DEV).contiguous() for k, v in state.items()}

    def run(fused: bool):
        z = lambda *s, dt=torch.float32: torch.zeros(*s, device=DEV, dtype=dt)
        out = dict(scores=z(S, R), mask=z(S, R, dt=torch.uint8), M=z(S, R, d), msg=z(S, E, d), keyv=z(S, E, d),
                   q=z(S, R, d), wgt=z(S, E))
        fo = torch.ones(S, dtype=torch.uint8, device=DEV)
        ga = hip.new("ic_gate_args", S=S, R=R, d=d, sensor_idx=0, motor_idx=R - 1, k_total=k_total, force_io=1, w1=1.0,
                     w2=0.3, w3=0.5, w4=0.2, t5=0.1, t6=0.05, gain_ema=gate.gain_ema, homeo=gate.homeo, focus_any=fo,
                     nbr_off=off, nbr=nbr, scores=out["scores"], mask=out["mask"], H=Hin)
        ra = ops._router_args(ops.RouterSpec(R, d, E, router.nW, True), tables, router._params(), S)
        ra.ebias, ra.H, ra.mask_prev = ebias.data_ptr(), Hin.data_ptr(), mprev.data_ptr()
        for k in ("M", "msg", "keyv", "q", "wgt"):
            hip.set_field(ra, k, out[k])
        a = ops._region_args(bank.spec, P, S, 0.4285714, kw_k, flags, kw_temp)
        reg = {k: z(S, R, d) for k in ("H", "num_o", "den_o", "var_o", "pred_o", "x", "u1", "r_s", "k_s", "v_s", "y",
                                       "hpre", "gate", "sterm")}
        reg.update({k: z(S, R) for k in ("radius_o", "radius_var_o", "surprise_o", "ynorm")})
        reg["inv_s"], reg["dt_o"] = z(S, R, 2), z(S, R, dt=torch.int32)
        reg["last_dir"], reg["last_norm"] = z(1, R, d), z(1, R)
        for k, t in dict(sensor=sensor, **state, **reg).items():
            setattr(a, k, t.data_ptr())
        a.M, a.mask = out["M"].data_ptr(), out["mask"].data_ptr()
        if fused:
            hip.check(L.ic_loop_step_fwd(ctypes.byref(ga), ctypes.byref(ra), ctypes.byref(a), st), "ic_loop_step_fwd")
        else:
            hip.check(L.ic_front_fwd(ctypes.byref(ga), ctypes.byref(ra), st), "ic_front_fwd")
            hip.check(L.ic_region_phase_fwd(ctypes.byref(a), st), "ic_region_phase_fwd")
        torch.cuda.synchronize()
        out.update(reg)
        return out

    two, one = run(False), run(True)
    act = two["mask"].bool()
    assert 0 < int(act.sum()) < S * R, "want active and sat-out rows"
    for k in two:
        assert torch.equal(one[k], two[k]), f"{k}: ic_loop_step_fwd differs from ic_front_fwd + ic_region_phase_fwd"
