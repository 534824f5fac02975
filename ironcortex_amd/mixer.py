"""Token mixer that pools the focused positions into the sensor vector.  Parameter
names follow reference iron_rope.py:100-374 and attention/adaptive_filter_attention.py so checkpoints load
unchanged.  The attention core (scores, softmax, statistics, value product and their backward) runs on the
library's fp32-MFMA kernels (ic_attn_fwd / ic_attn_bwd, csrc/attn_mfma.hip): no [B,H,T,T] tensor exists.

How the reference's arithmetic reaches that kernel without changing results:
  * the Fourier relative bias b(p_q - p_k) is a rank-2m feature product (cos/sin addition theorem), so it
    rides as 2m extra q / k columns instead of a [B,T,T,m] tensor; 1/sqrt(hd) is folded into q;
  * no [1,1,max_T,max_T] mask buffer for the bidirectional case; padded anchor rows are a per-sample key count.
Options the mixer never uses (causal, an explicit context mask, attention dropout > 0, an AFA mask) and heads
wider than the kernels' 128 columns (d = 1024 with 4 heads) keep the plain PyTorch formulation of the same
arithmetic (`*_torch`), which is also the tests' fp32 reference."""
from __future__ import annotations

import math
from typing import List, Optional, Tuple

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from .constants import EPS_DIV, EPS_LOG, MAX_EXP
from .hip import ops
from .utils import FastAttrs, extract_spans, init_weights, masked_mean


def make_freq_bank(m: int, d_coord: int, kind: str = "log", base: float = 10000.0, sigma: float = 1.0,
                   device=None, dtype=None) -> torch.Tensor:
    """[m, d_coord] frequency bank (iron_rope.py:16-37)."""
    if kind == "log":
        col = torch.logspace(math.log(1.0 / base), 0.0, steps=m, base=math.e, device=device, dtype=dtype)
        return col.unsqueeze(1).repeat(1, d_coord)
    if kind == "gaussian":
        return torch.randn(m, d_coord, device=device, dtype=dtype) * sigma
    raise ValueError(f"Unknown freq kind: {kind}")


def sincos(W: torch.Tensor, p: torch.Tensor) -> torch.Tensor:
    theta = p @ W.T
    return torch.cat([torch.sin(theta), torch.cos(theta)], dim=-1)


def rope_rotate_pairs(x: torch.Tensor, cos_th: torch.Tensor, sin_th: torch.Tensor, m_pairs: int) -> torch.Tensor:
    """Rotate channel pairs (2j, 2j+1), j < m_pairs (iron_rope.py:49-66)."""
    if m_pairs <= 0:
        return x
    head, tail = x[..., : 2 * m_pairs], x[..., 2 * m_pairs:]
    even, odd = head[..., 0::2], head[..., 1::2]
    rot = torch.stack((even * cos_th - odd * sin_th, even * sin_th + odd * cos_th), dim=-1)
    return torch.cat([rot.flatten(-2), tail], dim=-1)


def fourier_features(p: torch.Tensor, W, beta_cos, beta_sin, scale: float = 1.0):
    """(left, right) with left[q] . right[k] = scale * sum_m cos(S)beta_cos + sin(S)beta_sin, S = (p_q - p_k) W^T
    (iron_rope.py:69-97): left [B,T,2m] (headwise betas: [B,H,T,2m]), right [B,T,2m]."""
    if W.shape[1] != p.shape[-1]:
        W = W.expand(W.shape[0], p.shape[-1])
    a = p @ W.T
    c, s = torch.cos(a), torch.sin(a)
    right = torch.cat([c, s], dim=-1)
    if beta_cos.dim() == 2:
        bc, bs = beta_cos[None, :, None, :], beta_sin[None, :, None, :]
        c, s = c.unsqueeze(1), s.unsqueeze(1)
    else:
        bc, bs = beta_cos, beta_sin
    left = torch.cat([c * bc + s * bs, s * bc - c * bs], dim=-1) * scale
    return left, right


def relative_fourier_bias(p_q, p_k, W, beta_cos, beta_sin, scale: float = 1.0) -> torch.Tensor:
    """b[q,k] = scale * sum_m cos(S)beta_cos + sin(S)beta_sin, S = (p_q - p_k) W^T
    (iron_rope.py:69-97), evaluated without the [B,Tq,Tk,m] intermediate."""
    if W.shape[1] != p_q.shape[-1]:  # the reference's einsum broadcasts a size-1 coordinate axis
        W = W.expand(W.shape[0], p_q.shape[-1])
    aq, ak = p_q @ W.T, p_k @ W.T  # [B,T,m]
    cq, sq, ck, sk = torch.cos(aq), torch.sin(aq), torch.cos(ak), torch.sin(ak)
    if beta_cos.dim() == 2:  # headwise [H,m]
        bc, bs = beta_cos[None, :, None, :], beta_sin[None, :, None, :]
        cq, sq = cq.unsqueeze(1), sq.unsqueeze(1)
        left = torch.cat([cq * bc + sq * bs, sq * bc - cq * bs], dim=-1)
        right = torch.cat([ck, sk], dim=-1).unsqueeze(1)
        return scale * (left @ right.transpose(-2, -1))
    left = torch.cat([cq * beta_cos + sq * beta_sin, sq * beta_cos - cq * beta_sin], dim=-1)
    right = torch.cat([ck, sk], dim=-1)
    return (scale * (left @ right.transpose(-2, -1))).unsqueeze(1)


class IronRoPESelfAttention(nn.Module):
    """Bidirectional self-attention with rotary q/k and a Fourier relative bias
    (iron_rope.py:100-251).  Only positions in update_mask are changed."""

    def __init__(self, n_embd: int, n_head: int, block_size: int, use_iron_rope: bool = True,
                 rope_m: int = 64, rope_coord_dim: int = 1, rope_kind: str = "log", rope_base: float = 10000.0,
                 rope_sigma: float = 1.0, use_fourier_bias: bool = True, fb_m: int = 32, fb_coord_dim: int = 1,
                 fb_kind: str = "gaussian", fb_base: float = 10000.0, fb_sigma: float = 1.0,
                 fb_headwise: bool = False, attn_pdrop: float = 0.0, resid_pdrop: float = 0.0,
                 causal: bool = False):
        super().__init__()
        assert n_embd % n_head == 0
        self.n_head, self.head_dim = n_head, n_embd // n_head
        self.block_size, self.causal = block_size, causal
        self.c_attn = nn.Linear(n_embd, 3 * n_embd)
        self.c_proj = nn.Linear(n_embd, n_embd)
        self.adrop, self.rdrop = nn.Dropout(attn_pdrop), nn.Dropout(resid_pdrop)
        self.register_buffer("last_attn_energy", torch.tensor(0.0), persistent=False)
        self.use_rope = use_iron_rope
        self.rope_m = min(self.head_dim // 2, rope_m) if use_iron_rope else 0
        if self.rope_m > 0:
            self.register_buffer("W_rope", make_freq_bank(self.rope_m, rope_coord_dim, rope_kind, rope_base, rope_sigma))
        self.use_fb = use_fourier_bias
        self.fb_m = fb_m if use_fourier_bias else 0
        if self.fb_m > 0:
            self.register_buffer("W_fb", make_freq_bank(self.fb_m, fb_coord_dim, fb_kind, fb_base, fb_sigma))
            shape = (n_head, self.fb_m) if fb_headwise else (self.fb_m,)
            self.beta_cos = nn.Parameter(torch.zeros(shape))
            self.beta_sin = nn.Parameter(torch.zeros(shape))
            nn.init.normal_(self.beta_cos, std=0.01)
            nn.init.normal_(self.beta_sin, std=0.01)
            self.fb_scale = 1.0 / math.sqrt(self.fb_m)

    def forward(self, x, coords=None, update_mask=None, context_mask=None, dropout: bool = True,
                valid_len: Optional[torch.Tensor] = None):
        """valid_len [B] (batched sensor vectors only): positions >= valid_len[b] are padding;
        they are masked out as keys and left out of the per-sample attention energy, so every
        sample sees exactly what a batch-of-one call on its own rows would see."""
        B, T, C = x.shape
        assert T <= self.block_size
        if coords is None:
            coords = torch.arange(T, device=x.device, dtype=torch.float32).view(1, T, 1).expand(B, T, 1)
        elif coords.dim() == 2:
            coords = coords.unsqueeze(0).expand(B, -1, -1)
        q, k, v = self.c_attn(x).split(C, dim=2)
        shp = (B, T, self.n_head, self.head_dim)
        q, k, v = (t.view(shp).transpose(1, 2) for t in (q, k, v))
        if self.rope_m > 0:
            theta = coords @ self.W_rope.T
            c, s = torch.cos(theta).unsqueeze(1), torch.sin(theta).unsqueeze(1)
            q, k = rope_rotate_pairs(q, c, s, self.rope_m), rope_rotate_pairs(k, c, s, self.rope_m)
        plain = not self.causal and context_mask is None and not (dropout and self.training and self.adrop.p > 0)
        plain = plain and ops.attention_fits(self.head_dim + 2 * self.fb_m, self.head_dim)  # heads wider than 128: [T,T] path
        if plain:  # the library's kernels
            y = self._attend_hip(q, k, v, coords, valid_len)
        else:
            y = self._attend_torch(q, k, v, coords, context_mask, valid_len, dropout)
        y = self.c_proj(y.transpose(1, 2).contiguous().view(B, T, C))
        if dropout:
            y = self.rdrop(y)
        if update_mask is not None:
            return x + y * update_mask.unsqueeze(-1).to(y.dtype)
        return x + y

    def _attend_hip(self, q, k, v, coords, valid_len):
        """iron_rope.py:201-235 through ic_attn_fwd: bias features appended to the q / k rows."""
        B, H, T, hd = q.shape
        q = q * (1.0 / math.sqrt(self.head_dim))
        if self.fb_m > 0:
            left, right = fourier_features(coords, self.W_fb, self.beta_cos, self.beta_sin, self.fb_scale)
            left = left if left.dim() == 4 else left.unsqueeze(1)
            q = torch.cat([q, left.expand(B, H, T, -1)], dim=-1)
            k = torch.cat([k, right.unsqueeze(1).expand(B, H, T, -1)], dim=-1)
        out, stats = ops.attention(q, k, v, valid_len=valid_len, eps_log=EPS_LOG)
        s0 = stats[..., 0]
        if valid_len is None:
            self.last_attn_energy_b = -s0.sum(dim=(1, 2)) / float(H * T * T)
            self.last_attn_energy = self.last_attn_energy_b.mean()
        else:  # valid queries only; the kernel already left the padding keys out
            vq = (torch.arange(T, device=q.device).unsqueeze(0) < valid_len.unsqueeze(1)).to(s0.dtype)
            cnt = valid_len.to(s0.dtype).pow(2) * H
            self.last_attn_energy_b = -(s0 * vq.unsqueeze(1)).sum(dim=(1, 2)) / cnt
            self.last_attn_energy = self.last_attn_energy_b.mean()
        return out

    def _attend_torch(self, q, k, v, coords, context_mask, valid_len, dropout):
        """The same arithmetic with the [B,H,T,T] tensors in memory (options the kernels do not take; the
        tests' fp32 reference)."""
        T = q.shape[2]
        att = (q @ k.transpose(-2, -1)) * (1.0 / math.sqrt(self.head_dim))
        if self.fb_m > 0:
            att = att + relative_fourier_bias(coords, coords, self.W_fb, self.beta_cos, self.beta_sin, self.fb_scale)
        if self.causal:
            keep = torch.ones(T, T, dtype=torch.bool, device=x.device).tril()
            att = att.masked_fill(~keep, float("-inf"))
        x = q  # (device / dtype of the masks below)
        valid = None
        if valid_len is not None:
            valid = torch.arange(T, device=x.device).unsqueeze(0) < valid_len.unsqueeze(1)  # [B,T]
            context_mask = valid if context_mask is None else (context_mask.to(torch.bool) & valid)
        if context_mask is not None:
            att = att.masked_fill(~context_mask[:, None, None, :].to(torch.bool), float("-inf"))
        att = F.softmax(att, dim=-1)
        with torch.no_grad():
            logp = (att + EPS_LOG).log()
            if valid is None:
                self.last_attn_energy_b = -logp.mean(dim=(1, 2, 3))
            else:
                both = (valid.unsqueeze(2) & valid.unsqueeze(1)).unsqueeze(1).to(logp.dtype)  # [B,1,T,T]
                cnt = valid_len.to(logp.dtype).pow(2) * self.n_head
                self.last_attn_energy_b = -(logp * both).sum(dim=(1, 2, 3)) / cnt
            self.last_attn_energy = self.last_attn_energy_b.mean() if valid is not None else -logp.mean()
        if dropout:
            att = self.adrop(att)
        return att @ v


class AdaptiveFilterAttention(FastAttrs, nn.Module):
    """Dot-product attention times exp(-alpha |lag| dt) (adaptive_filter_attention.py:8-127)."""
    _plain_attrs = frozenset({"last_attn_entropy", "_kernel_norm"})

    def __init__(self, d_model: int, n_head: int, dt: float = 1.0, alpha: float = 0.0, *,
                 debug_exact: bool = False, exact_threshold: int = 64):
        super().__init__()
        assert d_model % n_head == 0
        self.d_model, self.n_head, self.head_dim = d_model, n_head, d_model // n_head
        self.scale = self.head_dim ** -0.5
        self.dt = dt
        self.alpha = nn.Parameter(torch.tensor(alpha))
        self.debug_exact, self.exact_threshold = debug_exact, exact_threshold
        self.q_proj, self.k_proj = nn.Linear(d_model, d_model), nn.Linear(d_model, d_model)
        self.v_proj, self.out_proj = nn.Linear(d_model, d_model), nn.Linear(d_model, d_model)
        self.register_buffer("last_attn_energy", torch.tensor(0.0), persistent=False)
        self._kernel_norm = torch.tensor(0.0)  # a tensor, or (alpha at the time, T) of the last kernel call: formed on demand
        self.last_attn_entropy = torch.tensor(0.0)

    @property
    def last_kernel_norm(self) -> torch.Tensor:
        """||exp(-alpha |i-j| dt)||_F of the last call (adaptive_filter_attention.py:73; telemetry only).  The kernel path
        keeps (a copy of alpha, T) and forms the value when somebody asks: ten small launches per token-mixer call
        otherwise, on the host-bound stretch of a train_step before its loops start."""
        kn = self._kernel_norm
        if isinstance(kn, tuple):
            a, T = kn
            with torch.no_grad():
                lag = torch.arange(1, T, device=a.device, dtype=torch.float32)
                w = torch.exp(-2.0 * a * lag * self.dt) * (T - lag)  # ||decay||_F^2 = T + 2 sum_lag (T-lag) e^(-2 a lag dt)
                kn = self._kernel_norm = (T + 2.0 * w.sum()).sqrt()
        return kn

    @last_kernel_norm.setter
    def last_kernel_norm(self, value) -> None:
        self._kernel_norm = value

    def _attend(self, q, k, v, mask):
        if mask is None and ops.attention_fits(self.head_dim, self.head_dim):  # (heads wider than 128: [T,T] path)
            return self._attend_hip(q, k, v)
        return self._attend_torch(q, k, v, mask)

    def _attend_hip(self, q, k, v):
        """adaptive_filter_attention.py:60-88 through ic_attn_fwd (scores * exp(-alpha |i-j| dt) in the kernel)."""
        B, H, T, _ = q.shape
        out, stats = ops.attention(q * self.scale, k, v, alpha=self.alpha, lag_dt=float(self.dt), eps_log=EPS_LOG)
        with torch.no_grad():
            self._kernel_norm = (self.alpha.detach().clone(), T)  # (see last_kernel_norm)
            s0, s1 = stats[..., 0], stats[..., 1]
            self.last_attn_energy_b = -s0.sum(dim=(1, 2)) / float(H * T * T)
            self.last_attn_entropy_b = -s1.sum(dim=(1, 2)) / float(H * T)
            self.last_attn_energy = self.last_attn_energy_b.mean()
            self.last_attn_entropy = self.last_attn_entropy_b.mean()
        return out

    def _attend_torch(self, q, k, v, mask):
        T = q.size(-2)
        idx = torch.arange(T, device=q.device, dtype=torch.float32)
        lag = (idx.view(T, 1) - idx.view(1, T)).abs()
        decay = torch.exp(-self.alpha * lag * self.dt)  # gradient reaches alpha exactly as via the gather
        self.last_kernel_norm = decay.norm().detach()
        scores = (q @ k.transpose(-2, -1)) * self.scale * decay
        if mask is not None:
            if mask.dim() == 3:
                mask = mask.unsqueeze(1)
            scores = scores.masked_fill(mask == 0, -MAX_EXP)
        attn = F.softmax(scores, dim=-1)
        if mask is not None:
            attn = attn * mask
            attn = attn / attn.sum(dim=-1, keepdim=True).clamp_min(EPS_DIV)
        logp = (attn + EPS_LOG).log()
        self.last_attn_energy = (-logp.mean()).detach()
        self.last_attn_entropy = (-(logp * attn).sum(dim=-1).mean()).detach()
        # per-sample values for batched sensor vectors (equal to the scalars when B == 1)
        self.last_attn_energy_b = (-logp.mean(dim=(1, 2, 3))).detach()
        self.last_attn_entropy_b = (-(logp * attn).sum(dim=-1).mean(dim=(1, 2))).detach()
        return attn @ v

    def forward(self, x, mask=None):
        B, T, _ = x.shape
        shp = (B, T, self.n_head, self.head_dim)
        q, k, v = (p(x).view(shp).transpose(1, 2) for p in (self.q_proj, self.k_proj, self.v_proj))
        out = self._attend(q, k, v, mask)  # the reference's kernel path is the exact path (:89-98)
        return self.out_proj(out.transpose(1, 2).contiguous().view(B, T, self.d_model))

    def telemetry(self) -> dict:
        return {"alpha": float(self.alpha.detach()), "kernel_norm": float(self.last_kernel_norm),
                "attn_entropy_mean": float(self.last_attn_entropy)}


class LocalTokenMixer(FastAttrs, nn.Module):
    """Anchor tokens + one attention block; returns the mean of the focused positions
    (iron_rope.py:254-374)."""
    _fast_buffers = frozenset({"last_energy"})

    def __init__(self, d: int, n_head: int = 4, block_size: int = 8192, m_tok: int = 64, use_afa: bool = False):
        super().__init__()
        if use_afa:
            self.attn = AdaptiveFilterAttention(d_model=d, n_head=n_head)
            self.use_anchors = False
            self.register_parameter("anch_embed", None)
        else:
            self.attn = IronRoPESelfAttention(n_embd=d, n_head=n_head, block_size=block_size, use_iron_rope=True,
                                              rope_m=m_tok, rope_coord_dim=2, use_fourier_bias=True, fb_m=32,
                                              fb_coord_dim=1, causal=False)
            self.use_anchors = True
            self.anch_embed = nn.Parameter(torch.randn(3, d) * 0.02)  # centre / span start / span end
        self.register_buffer("last_energy", torch.tensor(0.0), persistent=False)
        self.apply(init_weights)

    @torch.no_grad()
    def build_anchors(self, T: int, focus_mask, strides=(128, 32)) -> List[Tuple[int, int]]:
        if not self.use_anchors:
            return []
        anchors = [(i, 0) for S in strides for i in range(S // 2, T, S)]
        if focus_mask is not None:
            for start, end in extract_spans(focus_mask):  # one host read of the focus bits
                anchors += [(start, 1), (max(start, end - 1), 2)]
        return anchors

    def forward(self, tok_emb, pos_coords, focus_mask, ws_slots: Optional[torch.Tensor] = None,
                use_dropout: bool = True) -> torch.Tensor:
        B, T, d = tok_emb.shape
        dev = tok_emb.device
        if not self.use_anchors:
            pooled = masked_mean(self.attn(tok_emb), focus_mask)
            self.last_energy = self.attn.last_attn_energy.detach()
            self.last_energy_b = self.attn.last_attn_energy_b
            return pooled
        # anchors of all samples laid out on the host (one read of the focus bits, one upload): the stride
        # anchors are the same for every row, then (span start, kind 1), (span last, kind 2) per focus span
        base = [i for S in (128, 32) for i in range(S // 2, T, S)]
        nb = len(base)
        if focus_mask is not None:
            fh = focus_mask.detach().cpu().numpy().astype(np.int8)
            edge = np.diff(np.pad(fh, ((0, 0), (1, 1))), axis=1)  # +1 at a span's start, -1 one past its end
            rs, starts = np.nonzero(edge == 1)
            _re, ends = np.nonzero(edge == -1)  # row-major: the j-th end belongs to the j-th start
            n_span = np.bincount(rs, minlength=B)
        else:
            rs = starts = ends = np.zeros(0, dtype=np.int64)
            n_span = np.zeros(B, dtype=np.int64)
        lens = nb + 2 * n_span
        n_max = int(lens.max()) if B else 0
        ragged = bool((lens != n_max).any())
        Ta = T + n_max
        X = torch.zeros(B, Ta, d, device=dev)
        C = torch.zeros(B, Ta, 2, device=dev)
        UM = torch.zeros(B, Ta, dtype=torch.bool, device=dev)
        X[:, :T], C[:, :T], UM[:, :T] = tok_emb, pos_coords, focus_mask
        if n_max > 0:
            packed = np.zeros((3, B, n_max), dtype=np.int64)  # position, kind, valid
            packed[0, :, :nb] = np.asarray(base, dtype=np.int64)
            packed[2, :, :nb] = 1
            if len(rs):
                j = np.arange(len(rs)) - np.concatenate([[0], np.cumsum(n_span)])[rs]  # span number within its row
                col = nb + 2 * j
                packed[0, rs, col], packed[1, rs, col], packed[2, rs, col] = starts, 1, 1
                packed[0, rs, col + 1], packed[1, rs, col + 1], packed[2, rs, col + 1] = np.maximum(starts, ends - 1), 2, 1
            pk = torch.from_numpy(packed).to(dev)
            ok = pk[2].to(torch.float32)
            posf = pk[0].to(torch.float32)
            # (a [*,3] one-hot product, not anch_embed[kind]: the gather's backward is an index_put over three
            # rows shared by every anchor of the batch -- 0.6 ms at T=256, 2.1 ms at T=1024 of serialised adds)
            kinds = F.one_hot(pk[1], 3).to(torch.float32) * ok.unsqueeze(-1)
            X[:, T:] = kinds @ self.anch_embed
            C[:, T:] = torch.stack([posf, posf / (T + EPS_DIV)], dim=-1) * ok.unsqueeze(-1)
        valid_len = torch.from_numpy(T + lens).to(dev) if ragged else None
        Y = self.attn(X, coords=C, update_mask=UM, context_mask=None, dropout=use_dropout, valid_len=valid_len)
        pooled = masked_mean(Y[:, :T, :], focus_mask)
        self.last_energy = self.attn.last_attn_energy.detach()
        self.last_energy_b = self.attn.last_attn_energy_b
        return pooled
