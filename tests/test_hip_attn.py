"""Token-mixer attention kernels (ic_attn_fwd / ic_attn_bwd, csrc/attn_mfma.hip) against the
plain PyTorch fp32 formulation of the same arithmetic (reference iron_rope.py:201-235 and
attention/adaptive_filter_attention.py:60-88, which materialise the [B,H,T,T] tensors).  The whole-model fixtures
(test_hip_model.py: logits, traces and gradients of the real reference, anchors and ragged padding included) run
through the same kernels; here the op is isolated at the sizes the bench and the sweep use.
Tolerance: 2e-5 of the tensor's largest magnitude (fp32 sums over up to 1064 keys in a different order)."""
import math

import pytest
import torch

from ironcortex_amd.constants import EPS_LOG
from tests.golden import inputs as gi

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TOL = 2e-5


def close(a, b, what, tol=TOL):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    err = float((a - b).abs().max()) / max(float(b.abs().max()), 1e-6)
    assert err <= tol, f"{what}: {err:.3e}"


def torch_attention(q, k, v, valid_len=None, alpha=None, lag_dt=1.0):
    T = q.shape[2]
    s = q @ k.transpose(-2, -1)
    idx = torch.arange(T, device=q.device, dtype=torch.float32)
    if alpha is not None:
        s = s * torch.exp(-alpha * (idx.view(T, 1) - idx.view(1, T)).abs() * lag_dt)
    vk = None
    if valid_len is not None:
        vk = torch.arange(T, device=q.device).unsqueeze(0) < valid_len.unsqueeze(1)
        s = s.masked_fill(~vk[:, None, None, :], float("-inf"))
    p = torch.softmax(s, dim=-1)
    lg = (p + EPS_LOG).log()
    if vk is not None:
        lg = lg * vk[:, None, None, :].to(lg.dtype)
    return p @ v, torch.stack([lg.sum(-1), (p * lg).sum(-1)], dim=-1).detach()


@pytest.mark.parametrize("B,H,T,dqk,dv,ragged,afa", [
    (3, 4, 300, 128, 64, True, False),    # IronRoPE at d=256: 64 head + 64 bias columns, anchors ragged
    (2, 4, 256, 64, 64, False, True),     # AFA at d=256
    (2, 2, 1064, 128, 64, True, False),   # sweep length (seq 1024 + anchors)
    (2, 4, 70, 72, 8, True, False),       # d=32 fixtures: 8-wide heads + 64 bias columns, zero-padded to (96, 32)
    (1, 4, 33, 32, 32, False, True),      # d=128 default width, AFA, one key past a tile
    (2, 4, 129, 8, 8, False, True),       # 8-wide heads alone, one query past a workgroup's block
    (1, 2, 200, 192, 128, False, False),  # d=512 widths
    (16, 4, 520, 128, 64, True, False),   # the bench's batch (16 sequences x 4 heads), ragged
    (16, 4, 520, 64, 64, False, True),
])
def test_attention_forward_backward_vs_torch(B, H, T, dqk, dv, ragged, afa):
    from ironcortex_amd.hip import ops
    seed = 9100 + T + dqk
    q = gi.det(seed, B, H, T, dqk, scale=0.35).to(DEV).requires_grad_(True)
    k = gi.det(seed + 1, B, H, T, dqk, scale=0.35).to(DEV).requires_grad_(True)
    v = gi.det(seed + 2, B, H, T, dv).to(DEV).requires_grad_(True)
    probe = gi.det(seed + 3, B, H, T, dv).to(DEV)
    valid = None
    if ragged:
        valid = torch.tensor([T - 7 * b for b in range(B)], device=DEV)
        probe = probe * (torch.arange(T, device=DEV).view(1, 1, T, 1) < valid.view(B, 1, 1, 1))  # padded rows: dO = 0 and != 0 both legal
    alpha = torch.tensor(0.03, device=DEV, requires_grad=True) if afa else None
    out, stats = ops.attention(q, k, v, valid_len=valid, alpha=alpha, lag_dt=0.7, eps_log=EPS_LOG)
    (out * probe).sum().backward()
    got = [t.grad.clone() for t in (q, k, v)] + ([alpha.grad.clone()] if afa else [])
    for t in (q, k, v) + ((alpha,) if afa else ()):
        t.grad = None
    want_out, want_stats = torch_attention(q, k, v, valid, alpha, 0.7)
    (want_out * probe).sum().backward()
    want = [t.grad for t in (q, k, v)] + ([alpha.grad] if afa else [])
    vq = torch.ones(B, 1, T, 1, device=DEV) if valid is None else (torch.arange(T, device=DEV).view(1, 1, T, 1) < valid.view(B, 1, 1, 1)).float()
    close(out * vq, want_out * vq, "out")
    close(out, want_out, "out (padded query rows too)")
    close(stats * vq, want_stats * vq, "stats", tol=1e-4)  # sums of ~T logs of magnitude ~7
    for g, w, n in zip(got, want, ("dq", "dk", "dv", "dalpha")):
        close(g, w, n, tol=1e-4 if n == "dalpha" else TOL)
    # deterministic: a second backward gives the same bits (no float atomics)
    out2, _ = ops.attention(q.detach().requires_grad_(True), k.detach(), v.detach(), valid_len=valid,
                            alpha=None if alpha is None else alpha.detach(), lag_dt=0.7, eps_log=EPS_LOG)
    assert torch.equal(out2, out)


@pytest.mark.parametrize("d,T,ragged", [(256, 200, True), (32, 50, True), (128, 64, False)])
def test_ironrope_module_hip_equals_torch_formulation(d, T, ragged):
    """IronRoPESelfAttention.forward: the kernel path against the same module's [B,H,T,T] formulation (outputs,
    the attention-energy statistic, and every parameter / input gradient)."""
    from ironcortex_amd.mixer import IronRoPESelfAttention
    torch.manual_seed(3)
    att = IronRoPESelfAttention(n_embd=d, n_head=4, block_size=512, rope_m=64, rope_coord_dim=2, fb_m=32).to(DEV)
    B = 3
    x = gi.det(9300 + d, B, T, d).to(DEV)
    pos = torch.arange(T, device=DEV, dtype=torch.float32)
    coords = torch.stack([pos, pos / T], dim=-1).unsqueeze(0).expand(B, T, 2).contiguous()
    um = (gi.det(9301, B, T, kind="rand") < 0.4).to(DEV)
    valid = torch.tensor([T, T - 5, T - 11], device=DEV) if ragged else None
    probe = gi.det(9302, B, T, d).to(DEV)
    res = []
    for path in ("hip", "torch"):
        xx = x.clone().requires_grad_(True)
        if path == "torch":  # an all-True context mask is the identity and selects the [T,T] formulation
            y = att(xx, coords=coords, update_mask=um, context_mask=torch.ones(B, T, dtype=torch.bool, device=DEV), valid_len=valid)
        else:
            y = att(xx, coords=coords, update_mask=um, valid_len=valid)
        (y * probe).sum().backward()
        res.append((y.detach(), att.last_attn_energy_b.clone(), xx.grad.clone(),
                    {n: p.grad.clone() for n, p in att.named_parameters()}))
        att.zero_grad()
    close(res[0][0], res[1][0], "y")
    close(res[0][1], res[1][1], "attention energy", tol=1e-5)
    close(res[0][2], res[1][2], "dx", tol=1e-4)
    for n in res[0][3]:
        close(res[0][3][n], res[1][3][n], "grad " + n, tol=1e-4)


@pytest.mark.parametrize("d,T", [(256, 256), (32, 40)])
def test_afa_module_hip_equals_torch_formulation(d, T):
    from ironcortex_amd.mixer import AdaptiveFilterAttention
    torch.manual_seed(4)
    att = AdaptiveFilterAttention(d_model=d, n_head=4, alpha=0.05).to(DEV)
    x = gi.det(9400 + d, 2, T, d).to(DEV)
    probe = gi.det(9401, 2, T, d).to(DEV)
    res = []
    for path in ("hip", "torch"):
        xx = x.clone().requires_grad_(True)
        y = att(xx) if path == "hip" else att(xx, mask=torch.ones(2, T, T, device=DEV))
        (y * probe).sum().backward()
        res.append((y.detach(), att.last_attn_energy_b.clone(), att.last_attn_entropy_b.clone(), att.last_kernel_norm.clone(),
                    xx.grad.clone(), {n: p.grad.clone() for n, p in att.named_parameters()}))
        att.zero_grad()
    close(res[0][0], res[1][0], "y")
    close(res[0][1], res[1][1], "energy", tol=1e-5)
    close(res[0][2], res[1][2], "entropy", tol=1e-5)
    close(res[0][3], res[1][3], "kernel norm", tol=1e-5)
    close(res[0][4], res[1][4], "dx", tol=1e-4)
    for n in res[0][5]:
        close(res[0][5][n], res[1][5][n], "grad " + n, tol=1e-4)


def test_heads_wider_than_the_kernels_keep_the_torch_formulation():
    """d = 1024 with 4 heads (256-wide heads, beyond the kernels' 128 value columns): both attention modules still
    run (the [T,T] formulation), forward and backward."""
    from ironcortex_amd.hip import ops
    from ironcortex_amd.mixer import AdaptiveFilterAttention, IronRoPESelfAttention
    assert not ops.attention_fits(256, 256) and ops.attention_fits(72, 8) and ops.attention_fits(128, 128)
    x = gi.det(9500, 2, 40, 1024).to(DEV).requires_grad_(True)
    for att in (IronRoPESelfAttention(n_embd=1024, n_head=4, block_size=64).to(DEV), AdaptiveFilterAttention(1024, 4).to(DEV)):
        y = att(x)
        y.sum().backward()
        assert torch.isfinite(y).all() and torch.isfinite(x.grad).all()
        assert torch.isfinite(att.last_attn_energy_b).all()


# ---------------------------------------------------------------- fused linear layers of the loss heads (ic_linear_bwd)
@pytest.mark.parametrize("B,n,rows,act", [(8, 256, 256, "gelu"), (16, 515, 256, "silu"), (8, 512, 1, "none"),
                                           (3, 256, 2, "none"), (1, 40, 33, "gelu"), (16, 512, 512, "gelu"),
                                           (2, 64, 64, "silu"), (2, 32, 2, "none"), (2, 64, 1, "none"), (5, 96, 35, "gelu"),
                                           (16, 1024, 64, "silu"), (6, 30, 70, "gelu")])
def test_linear_act_forward_backward_vs_torch(B, n, rows, act):
    """y = act(x W^T + b) on the library (forward ic_gemv_chain, backward ic_linear_bwd) against F.linear + the
    activation under torch autograd: outputs and all three gradients, 1e-5 of the largest magnitude; twice the same bits."""
    import torch.nn.functional as F
    from ironcortex_amd.hip import ops
    seed = 9600 + n + rows
    x = gi.det(seed, B, n).to(DEV).requires_grad_(True)
    W = gi.det(seed + 1, rows, n, kind="xavier").to(DEV).requires_grad_(True)
    b = gi.det(seed + 2, rows, scale=0.1).to(DEV).requires_grad_(True)
    probe = gi.det(seed + 3, B, rows).to(DEV)
    fn = {"gelu": F.gelu, "silu": F.silu, "none": lambda t: t}[act]
    y = ops.linear_act(x, W, b, act)
    (y * probe).sum().backward()
    got = [t.grad.clone() for t in (x, W, b)]
    for t in (x, W, b):
        t.grad = None
    yw = fn(F.linear(x, W, b))
    (yw * probe).sum().backward()
    close(y, yw, "y", tol=1e-5)
    for g, t, nm in zip(got, (x, W, b), ("gx", "gW", "gb")):
        close(g, t.grad, nm, tol=1e-5)
    for t in (x, W, b):
        t.grad = None
    y2 = ops.linear_act(x, W, b, act)
    (y2 * probe).sum().backward()
    assert torch.equal(y2, y) and all(torch.equal(g, t.grad) for g, t in zip(got, (x, W, b)))
