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
