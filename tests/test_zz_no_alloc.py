"""ic_heads_fwd_static (the one entry point that allocates nothing: its inter-launch buffers live
in a static device array) == ic_heads_fwd, bit for bit, through the C ABI -- on two calls in a row,
the second with a smaller input (the array keeps the first call's contents), and it rejects sizes
beyond its stated maximum.  Collected last (test_zz_*): the round's end runs pytest -x."""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _case(seed, S, R, d, V, Kf, n_slots):
    g = torch.Generator().manual_seed(seed)
    rn = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(DEV)
    mask = (torch.rand(S, R, generator=g) < 0.5)
    mask[:, 0] = mask[:, R - 1] = True
    sparse = lambda *s: (torch.randn(*s, generator=g) * (torch.rand(*s, generator=g) < 0.2)).to(DEV)
    t = dict(H=sparse(S, R, d) * mask.to(DEV).unsqueeze(-1), H_par=sparse(S, R, d), mask=mask.to(torch.uint8).to(DEV),
             slots_in=sparse(S, n_slots, d), lm_gate_W=rn(Kf, d, sc=0.1), lm_gate_b=rn(Kf, sc=0.1),
             lm_W=rn(Kf, V, d, sc=0.06), lm_b=rn(Kf, V, sc=0.1), rh_W=rn(R, V, d, sc=0.06), rh_b=rn(R, V, sc=0.1),
             surprise=torch.rand(S, R, generator=g).to(DEV))
    return t


def _run(fn_name, t, S, R, d, V, Kf, n_slots, with_scratch):
    from ironcortex_amd import hip
    z = lambda *s: torch.full(s, float("nan"), device=DEV)
    out = dict(slots_out=z(S, n_slots, d), logits=z(S, V), probs=z(S, V), stats=z(S, 4))
    a = hip.new("ic_heads_args", R=R, d=d, V=V, Kf=Kf, n_slots=n_slots, motor_idx=R - 1, kwta_k=max(1, d // 4), S=S)
    for k, v in {**t, **out}.items():
        hip.set_field(a, k, v)
    keep = []
    if with_scratch:
        keep = [z(S, Kf + Kf * V), z(S, R, V), z(S, 2 * R)]
        a.z, a.per_region, a.gsq = (x.data_ptr() for x in keep)
    rc = getattr(hip.lib(), fn_name)(ctypes.byref(a), hip.stream_handle())
    torch.cuda.synchronize()
    return rc, out


def test_heads_fwd_static_equals_heads_fwd_twice_then_smaller():
    from ironcortex_amd import hip
    name = ctypes.create_string_buffer(64)
    hip.check(hip.lib().ic_device_check(name, 64), "ic_device_check")  # the set-up call resolves the array's address
    # first call: two streams of the bench model's shape; second call: a smaller model, one stream
    for seed, dims in ((1, dict(S=2, R=10, d=256, V=256, Kf=8, n_slots=8)), (2, dict(S=1, R=7, d=64, V=48, Kf=4, n_slots=8))):
        t = _case(seed, **dims)
        rc0, ref = _run("ic_heads_fwd", t, with_scratch=True, **dims)
        rc1, got = _run("ic_heads_fwd_static", t, with_scratch=False, **dims)
        hip.check(rc0, "ic_heads_fwd"); hip.check(rc1, "ic_heads_fwd_static")
        for k in ref:
            assert not torch.isnan(got[k]).any(), k
            assert torch.equal(ref[k], got[k]), (dims, k)


def test_heads_fwd_static_rejects_oversize():
    from ironcortex_amd import hip
    dims = dict(S=8, R=512, d=64, V=256, Kf=8, n_slots=8)  # 8 * 512 * 256 floats of per_region alone = the whole array
    need = 8 * (8 + 8 * 256) + 8 * 512 * 256 + 8 * 2 * 512
    assert need > hip.ENUMS["IC_HEADS_STATIC_FLOATS"]
    t = _case(3, **dims)
    rc, _ = _run("ic_heads_fwd_static", t, with_scratch=False, **dims)
    assert rc == hip.ENUMS["IC_EINVAL"]
    msg = hip.lib().ic_last_error().decode()
    assert "scratch floats" in msg and "ic_heads_fwd" in msg, msg
