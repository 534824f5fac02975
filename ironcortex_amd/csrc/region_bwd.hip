// Region phase, backward.  Three launches mirror the forward in reverse (four under soft KWTA):
//   [fin_bwd (wave/row)   soft KWTA only: KWTA backward -> ghpre, inactive rows' pass-through.
//                         With hard KWTA ghpre = gH * gate is recomputed by its two readers and
//                         the pass-through moves into pro_bwd]
//   out_bwd  (slice x r)  W_o^T ghpre -> slab, tape g_o / operand
//   rkv_bwd  (slice x r)  slab reduce -> gy (+ radial-tangential scalar chain), state/WKV/AFD
//                         backward per channel, W_{r,k,v}^T g -> slab2, tape
//   pro_bwd  (wave/row)   slab2 reduce + residual -> two RMSNorm backwards -> gM, gpred;
//                         inactive rows: trace/state pass-through grads
// Weight gradients are deferred: see k_wgrad_region (one launch per backward pass).
// Autograd of reference region.py:168-249, model.py:198-204,275-290.
#include "ic_common.h"

namespace {

// Streams may be spread out (ic_region_bwd_args.sk): stream s lives at stream index s*sk of every
// [S,R,..] array of the call.
__device__ __forceinline__ int stream_step(const ic_region_bwd_args& b) { return b.sk > 1 ? b.sk : 1; }
// The slice kernels split the streams over gridDim.z (workgroup z serves streams z, z+nz, ...):
// with many streams and few regions the launch is bounded by the serial stream loop of a
// workgroup, not by the weight rows it re-reads from L2.
__device__ __forceinline__ bool region_has_work(const uint8_t* mask, int S, int R, int r, int sk) {
  const int lane = threadIdx.x & 63;  // lanes test 64 of this workgroup's streams per round trip
  for (int sc = blockIdx.z; sc < S; sc += 64 * gridDim.z) {
    const int s = sc + lane * gridDim.z;
    if (__ballot(s < S && mask[(size_t)s * sk * R + r] != 0)) return true;
  }
  return false;
}
// wave-per-row kernels: local row index -> (stream, region, row in the spread-out arrays)
__device__ __forceinline__ bool spread_row(const ic_region_bwd_args& b, int lrow, int& s, int& r, int& row) {
  const ic_region_args& a = b.f;
  if (lrow >= a.S * a.R) return false;
  s = lrow / a.R;
  r = lrow % a.R;
  row = s * stream_step(b) * a.R + r;
  return true;
}

template <int DCH>
__device__ __forceinline__ void load_opt(RowFrag<DCH>& f, const float* base, size_t off, int d,
                                         int lane) {
  if (base)
    load_row(f, base + off, d, lane);
  else
    zero_frag(f);
}

// Inactive (stream, region) rows: the state passes through, the trace is an EMA of M
// (region.py:140-146): gM = (1-alpha) gpred_o, gpred = alpha gpred_o.
template <int DCH>
__device__ __forceinline__ void inactive_passthrough(const ic_region_bwd_args& b, int row, int lane,
                                                     float one_minus_alpha) {
  const ic_region_args& a = b.f;
  const int d = a.d;
  const size_t off = (size_t)row * d;
  RowFrag<DCH> g, t, gn, gd, gv;
  load_opt(gn, b.gnum_o, off, d, lane);
  load_opt(gd, b.gden_o, off, d, lane);
  if (a.flags & IC_F_AFD) load_opt(gv, b.gvar_o, off, d, lane);
  if (a.flags & IC_F_PRED) {
    load_opt(g, b.gpred_o, off, d, lane);
    const float al = (float)a.pred_alpha;
#pragma unroll
    for (int j = 0; j < DCH; ++j) {
      t.v[j] = make_float4(al * g.v[j].x, al * g.v[j].y, al * g.v[j].z, al * g.v[j].w);
      g.v[j] = make_float4(one_minus_alpha * g.v[j].x, one_minus_alpha * g.v[j].y,
                           one_minus_alpha * g.v[j].z, one_minus_alpha * g.v[j].w);
    }
    store_row(t, b.gpred + off, d, lane);
    store_row(g, b.gM + off, d, lane);
  } else {
    zero_frag(g);
    store_row(g, b.gM + off, d, lane);
  }
  store_row(gn, b.gnum + off, d, lane);
  store_row(gd, b.gden + off, d, lane);
  if (a.flags & IC_F_AFD) store_row(gv, b.gvar + off, d, lane);
  if ((a.flags & IC_F_RT) && lane == 0) {
    b.gradius[row] = b.gradius_o ? b.gradius_o[row] : 0.f;
    b.gradius_var[row] = b.gradius_var_o ? b.gradius_var_o[row] : 0.f;
  }
}

// ---------------------------------------------------------------- finish backward
template <int DCH>
__global__ __launch_bounds__(256) void k_region_fin_bwd(ic_region_bwd_args b, float one_minus_alpha) {
  const ic_region_args& a = b.f;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int s_, r_, row;
  if (!spread_row(b, blockIdx.x * 4 + wave, s_, r_, row)) return;
  const int d = a.d;
  const size_t off = (size_t)row * d;
  if (a.mask[row]) {
    RowFrag<DCH> g, gt, h;
    load_opt(g, b.gH, off, d, lane);
    load_row(gt, a.gate + off, d, lane);
    const bool soft = (a.flags & IC_F_KWTA_SOFT) && !(a.flags & IC_F_NOKWTA) && a.kwta_k > 0;
    if (!soft) {
#pragma unroll
      for (int j = 0; j < DCH; ++j) {
        g.v[j].x *= gt.v[j].x; g.v[j].y *= gt.v[j].y; g.v[j].z *= gt.v[j].z; g.v[j].w *= gt.v[j].w;
      }
    } else {
      // out_j = h_j * G_j, G_j = sigmoid((|h_j| - thr)/T) below the threshold, 1 at/above it;
      // thr = |h_k*| carries gradient to the k-th winner (utils.py:53-58).
      load_row(h, a.hpre + off, d, lane);
      const float T = a.kwta_temp;
      float gthr = 0.f, thr = 0.f;
      float hv[4 * DCH], gv[4 * DCH], Gv[4 * DCH];
#pragma unroll
      for (int j = 0; j < DCH; ++j) {
        hv[4 * j] = h.v[j].x; hv[4 * j + 1] = h.v[j].y; hv[4 * j + 2] = h.v[j].z; hv[4 * j + 3] = h.v[j].w;
        gv[4 * j] = g.v[j].x; gv[4 * j + 1] = g.v[j].y; gv[4 * j + 2] = g.v[j].z; gv[4 * j + 3] = g.v[j].w;
        Gv[4 * j] = gt.v[j].x; Gv[4 * j + 1] = gt.v[j].y; Gv[4 * j + 2] = gt.v[j].z; Gv[4 * j + 3] = gt.v[j].w;
      }
      // threshold = smallest |h| among the winners (gate == 1)
      float tmin = INFINITY;
#pragma unroll
      for (int e = 0; e < 4 * DCH; ++e) {
        const int c = 4 * lane + 256 * (e >> 2) + (e & 3);
        if (c < d && Gv[e] >= 1.0f) tmin = fminf(tmin, fabsf(hv[e]));
      }
      thr = -wave_max(-tmin);
#pragma unroll
      for (int e = 0; e < 4 * DCH; ++e) {
        const int c = 4 * lane + 256 * (e >> 2) + (e & 3);
        float out = 0.f;
        if (c < d) {
          if (Gv[e] >= 1.0f) {
            out = gv[e];
          } else {
            const float sp = Gv[e] * (1.0f - Gv[e]) / T;
            const float sg = hv[e] > 0.f ? 1.f : (hv[e] < 0.f ? -1.f : 0.f);
            out = gv[e] * (Gv[e] + hv[e] * sp * sg);
            gthr -= gv[e] * hv[e] * sp;
          }
        }
        gv[e] = out;
      }
      gthr = wave_sum(gthr);
      // first element whose |h| equals the threshold receives gthr * sign(h)
      int first = 1 << 30;
#pragma unroll
      for (int e = 0; e < 4 * DCH; ++e) {
        const int c = 4 * lane + 256 * (e >> 2) + (e & 3);
        if (c < d && Gv[e] >= 1.0f && fabsf(hv[e]) == thr) first = min(first, c);
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) first = min(first, __shfl_xor(first, o, 64));
#pragma unroll
      for (int e = 0; e < 4 * DCH; ++e) {
        const int c = 4 * lane + 256 * (e >> 2) + (e & 3);
        if (c == first) gv[e] += gthr * (hv[e] > 0.f ? 1.f : (hv[e] < 0.f ? -1.f : 0.f));
      }
#pragma unroll
      for (int j = 0; j < DCH; ++j)
        g.v[j] = make_float4(gv[4 * j], gv[4 * j + 1], gv[4 * j + 2], gv[4 * j + 3]);
    }
    store_row(g, b.ghpre + off, d, lane);
  } else {
    inactive_passthrough<DCH>(b, row, lane, one_minus_alpha);
  }
}

// ---------------------------------------------------------------- o-projection backward
template <int DCH, int RPW>
__global__ __launch_bounds__(256) void k_region_out_bwd(ic_region_bwd_args b, int fold) {
  const ic_region_args& a = b.f;
  const int r = blockIdx.y;
  const int sk = stream_step(b);
  if (!region_has_work(a.mask, a.S, a.R, r, sk)) return;
  __shared__ float red[4][DCH * 256 + 1];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int d = a.d;
  const int row0 = blockIdx.x * (4 * RPW) + wave * RPW;
  const size_t dd = (size_t)d * d;
  const float* Wo = a.W + ((size_t)r * 4 + 3) * dd;
  RowFrag<DCH> wo[RPW];
#pragma unroll
  for (int i = 0; i < RPW; ++i) {
    if (row0 + i < d)
      load_row(wo[i], Wo + (size_t)(row0 + i) * d, d, lane);
    else
      zero_frag(wo[i]);
  }
  const bool rt = a.flags & IC_F_RT;
  const int SR = a.S * sk * a.R;
  // This workgroup's streams (z, z + nz, ...) are tested 64 at a time and the active ones walked
  // from the ballot; the next stream's operands (the y row, its norm and radius, and the wave's
  // RPW incoming gradients, one per lane) are requested before the current one is reduced.
  struct Ops {
    RowFrag<DCH> op;
    float nrm, scale, g;
  };
  auto request = [&](int s, Ops& o) {
    const int row = s * sk * a.R + r;
    const size_t off = (size_t)row * d;
    load_row(o.op, a.y + off, d, lane);
    o.nrm = o.scale = 1.0f;
    if (rt) {
      o.nrm = a.ynorm[row];
      o.scale = a.radius_o[row];
    }
    const int ch = row0 + (lane < RPW ? lane : 0);
    o.g = 0.f;
    if (lane < RPW && ch < d)
      o.g = fold ? (b.gH ? b.gH[off + ch] * a.gate[off + ch] : 0.f) : b.ghpre[off + ch];
  };
  const int nz = gridDim.z;
  for (int sc = blockIdx.z; sc < a.S; sc += 64 * nz) {
    const int sl = sc + lane * nz;
    unsigned long long act = __ballot(sl < a.S && a.mask[(size_t)sl * sk * a.R + r] != 0);
    if (!act) continue;
    Ops cur, nxt;
    int s_cur = sc + (__ffsll((long long)act) - 1) * nz;
    act &= act - 1;
    request(s_cur, cur);
    while (s_cur >= 0) {
      const int s_nxt = act ? sc + (__ffsll((long long)act) - 1) * nz : -1;
      act &= act - 1;
      if (s_nxt >= 0) request(s_nxt, nxt);
      const int row = s_cur * sk * a.R + r;
      const size_t off = (size_t)row * d;
      RowFrag<DCH> op = cur.op, acc;
      const float scale = cur.scale;
      if (rt) {
        const float nrm = cur.nrm;
#pragma unroll
        for (int j = 0; j < DCH; ++j) {
          op.v[j].x /= nrm; op.v[j].y /= nrm; op.v[j].z /= nrm; op.v[j].w /= nrm;
        }
      }
      zero_frag(acc);
      float part_rt = 0.f;
#pragma unroll
      for (int i = 0; i < RPW; ++i) {
        const int ch = row0 + i;
        const float g = __shfl(cur.g, i, 64);
        if (ch < d) {  // wave-uniform
          float gi = g;
          if (rt) {
            const float hd = wave_sum(dot_frag(wo[i], op));
            part_rt += g * hd;
            gi = scale * g;
          }
          if (lane == 0) b.tape_g[((size_t)row * 4 + 3) * d + ch] = gi;
          axpy_frag(acc, gi, wo[i]);
        }
      }
      // cross-wave reduction of the partial W_o^T g
#pragma unroll
      for (int j = 0; j < DCH; ++j) {
        float* p = &red[wave][256 * j + 4 * lane];
        p[0] = acc.v[j].x; p[1] = acc.v[j].y; p[2] = acc.v[j].z; p[3] = acc.v[j].w;
      }
      if (lane == 0) red[wave][DCH * 256] = part_rt;
      __syncthreads();
      if (wave == 0) {
        RowFrag<DCH> tot;
#pragma unroll
        for (int j = 0; j < DCH; ++j) {
          const int c = 256 * j + 4 * lane;
          tot.v[j].x = (red[0][c] + red[1][c]) + (red[2][c] + red[3][c]);
          tot.v[j].y = (red[0][c + 1] + red[1][c + 1]) + (red[2][c + 1] + red[3][c + 1]);
          tot.v[j].z = (red[0][c + 2] + red[1][c + 2]) + (red[2][c + 2] + red[3][c + 2]);
          tot.v[j].w = (red[0][c + 3] + red[1][c + 3]) + (red[2][c + 3] + red[3][c + 3]);
        }
        store_row(tot, b.slab + ((size_t)blockIdx.x * SR + row) * d, d, lane);
        if (rt && lane == 0) {
          const int q = DCH * 256;
          b.slab_rt[(size_t)blockIdx.x * SR + row] = (red[0][q] + red[1][q]) + (red[2][q] + red[3][q]);
        }
        if (blockIdx.x == 0) store_row(op, b.tape_o + off, d, lane);
      }
      __syncthreads();
      cur = nxt;
      s_cur = s_nxt;
    }
  }
}

// ---------------------------------------------------------------- middle: gy + RT scalars
// Radial-tangential (region.py:219-235): the full row g = d/d dir (already scaled by radius')
// becomes gy = g / n + coef * y and the scalar Kalman chain of the radius is differentiated.
// Every wave that needs gy runs this redundantly (8 KB of slab rows from L2); `writer` stores
// the scalar outputs once per row.
template <int DCH>
__device__ __forceinline__ void rt_gy_row(const ic_region_bwd_args& b, RowFrag<DCH>& g, int row, int r,
                                          int nslab, int lane, bool writer) {
  const ic_region_args& a = b.f;
  const int d = a.d, SR = a.S * stream_step(b) * a.R;
  const size_t off = (size_t)row * d;
  RowFrag<DCH> y;
  load_row(y, a.y + off, d, lane);
  // one lane per slab fetches its partial d/d radius' (one round trip), summed in slab order
  float part = 0.f;
  for (int s0 = 0; s0 < nslab; s0 += 64)
    if (s0 + lane < nslab) part = b.slab_rt[(size_t)(s0 + lane) * SR + row];
  const float n = a.ynorm[row];
  const float rho = a.radius[row];
  const float pn_p = a.rpn_param[r], on_p = a.ron_param[r];
  const float rv = a.radius_var[row];
  const float Grho_o = b.gradius_o ? b.gradius_o[row] : 0.f;
  const float Grv = b.gradius_var_o ? b.gradius_var_o[row] : 0.f;
  zero_frag(g);
  sum_slabs(g, b.slab + off, (size_t)SR * d, nslab, d, lane);
  float g_rho_h = 0.f;
  for (int sl = 0; sl < nslab; ++sl) g_rho_h += __shfl(part, sl, 64);
  const float ynrm = sqrtf(wave_sum(dot_frag(y, y)));
  const float gdy = wave_sum(dot_frag(g, y));
  const float pv = rv + ic_noise(pn_p);
  const float on = ic_noise(on_p);
  const float D = pv + on + IC_EPS_DIV;
  const float gain = pv / D;
  const float mu = (float)0.2;
  const float Grho = Grho_o + g_rho_h;
  const float g_gain = Grho * mu * (n - rho) - Grv * pv;
  const float g_pv = Grv * (1.0f - gain) + g_gain * (on + IC_EPS_DIV) / (D * D);
  const float g_on = -g_gain * pv / (D * D);
  float g_n = Grho * mu * gain - gdy / (n * n);
  if (!(ynrm >= IC_EPS_DIV)) g_n = 0.f;  // clamp_min blocks the gradient
  const float coef = ynrm > 0.f ? g_n / ynrm : 0.f;
#pragma unroll
  for (int j = 0; j < DCH; ++j) {
    g.v[j].x = g.v[j].x / n + coef * y.v[j].x; g.v[j].y = g.v[j].y / n + coef * y.v[j].y;
    g.v[j].z = g.v[j].z / n + coef * y.v[j].z; g.v[j].w = g.v[j].w / n + coef * y.v[j].w;
  }
  if (writer && lane == 0) {
    b.gradius[row] = Grho * (1.0f - mu * gain);
    b.gradius_var[row] = g_pv;
    b.g_rnoise_rows[row * 2 + 0] = g_pv * ic_noise_grad(pn_p);
    b.g_rnoise_rows[row * 2 + 1] = g_on * ic_noise_grad(on_p);
  }
}

// ---------------------------------------------------------------- r,k,v + state backward
template <int DCH, int RPW>
__global__ __launch_bounds__(256) void k_region_rkv_bwd(ic_region_bwd_args b, int nslab) {
  const ic_region_args& a = b.f;
  const int r = blockIdx.y;
  const int sk = stream_step(b);
  if (!region_has_work(a.mask, a.S, a.R, r, sk)) return;
  __shared__ float red[4][DCH * 256];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int d = a.d;
  const int row0 = blockIdx.x * (4 * RPW) + wave * RPW;
  const size_t dd = (size_t)d * d;
  const float* Wr = a.W + (size_t)r * 4 * dd;
  const float* Wk = Wr + dd;
  const float* Wv = Wk + dd;
  RowFrag<DCH> wr[RPW], wk[RPW], wv[RPW];
#pragma unroll
  for (int i = 0; i < RPW; ++i) {
    if (row0 + i < d) {
      load_row(wr[i], Wr + (size_t)(row0 + i) * d, d, lane);
      load_row(wk[i], Wk + (size_t)(row0 + i) * d, d, lane);
      load_row(wv[i], Wv + (size_t)(row0 + i) * d, d, lane);
    } else {
      zero_frag(wr[i]); zero_frag(wk[i]); zero_frag(wv[i]);
    }
  }
  const bool afd = a.flags & IC_F_AFD, vec = a.flags & IC_F_NOISE_VEC, rt = a.flags & IC_F_RT;
  const int SR = a.S * sk * a.R;
  // lane l < RPW owns channel row0 + l; what depends on (region, channel) only is read once
  const int chr = row0 + (lane < RPW ? lane : 0);
  const bool valid = lane < RPW && chr < d;  // rotation pairs (even, odd lane) are valid together
  const int ch = valid ? chr : 0;            // clamp loads
  const float rd = a.raw_decay[(size_t)r * d + ch];
  float pn_p = 0.f, on_p = 0.f;
  if (afd) {
    pn_p = vec ? a.pn_param[(size_t)r * d + ch] : a.pn_param[r];
    on_p = vec ? a.on_param[(size_t)r * d + ch] : a.on_param[r];
  }
  const bool rot = ch < 2 * a.m_time;
  const float th = rot ? a.step_pos * a.W_time[ch >> 1] : 0.f;
  // Streams of this workgroup (z, z + nz, ...) are tested 64 at a time; the active ones are walked
  // from the ballot with the next stream's per-channel operands (one batch of loads) in flight
  // while the current stream is reduced.  Every wave sees the same ballot: barriers stay uniform.
  struct Ops {
    float num_in, den_in, rr, kk, vv, gno, gdo, var_in, gvo, t[8];
    int dt;
  };
  auto request = [&](int s, Ops& o) {
    const int row = s * sk * a.R + r;
    const size_t idx = (size_t)row * d + ch;
    o.dt = a.dt[row];
    o.num_in = a.num[idx]; o.den_in = a.den[idx];
    o.rr = a.r_s[idx]; o.kk = a.k_s[idx]; o.vv = a.v_s[idx];
    o.gno = b.gnum_o ? b.gnum_o[idx] : 0.f;
    o.gdo = b.gden_o ? b.gden_o[idx] : 0.f;
    o.var_in = o.gvo = 0.f;
    if (afd) {
      o.var_in = a.var[idx];
      o.gvo = b.gvar_o ? b.gvar_o[idx] : 0.f;
    }
    if (!rt) {  // gy of this channel: the o-projection's slabs (first eight; the rest at use)
#pragma unroll
      for (int u = 0; u < 8; ++u) o.t[u] = u < nslab ? b.slab[((size_t)u * SR + row) * d + ch] : 0.f;
    }
  };
  const int nz = gridDim.z;
  for (int sc = blockIdx.z; sc < a.S; sc += 64 * nz) {
   const int sl = sc + lane * nz;
   unsigned long long act = __ballot(sl < a.S && a.mask[(size_t)sl * sk * a.R + r] != 0);
   if (!act) continue;
   Ops cur, nxt;
   int s_cur = sc + (__ffsll((long long)act) - 1) * nz;
   act &= act - 1;
   request(s_cur, cur);
   while (s_cur >= 0) {
    const int s_nxt = act ? sc + (__ffsll((long long)act) - 1) * nz : -1;
    act &= act - 1;
    if (s_nxt >= 0) request(s_nxt, nxt);
    const int row = s_cur * sk * a.R + r;
    const size_t off = (size_t)row * d;
    const float steps = (float)(cur.dt + 1);
    RowFrag<DCH> acc;
    {
      const size_t idx = off + ch;
      const float num_in = cur.num_in, den_in = cur.den_in, rr = cur.rr, kk = cur.kk, vv = cur.vv;
      const float gno = cur.gno, gdo = cur.gdo, var_in = cur.var_in, gvo = cur.gvo;
      // ---- gy of this channel: the o-projection's slabs, reduced here (no launch of its own)
      float gy;
      if (rt) {
        rt_gy_row<DCH>(b, acc, row, r, nslab, lane, blockIdx.x == 0 && wave == 0);
#pragma unroll
        for (int j = 0; j < DCH; ++j) {
          float* p = &red[wave][256 * j + 4 * lane];
          p[0] = acc.v[j].x; p[1] = acc.v[j].y; p[2] = acc.v[j].z; p[3] = acc.v[j].w;
        }
        __syncthreads();
        gy = red[wave][ch];
        __syncthreads();
      } else {
        gy = 0.f;
#pragma unroll
        for (int u = 0; u < 8; ++u)
          if (u < nslab) gy += cur.t[u];
        for (int s0 = 8; s0 < nslab; s0 += 8) {
          float t[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) t[u] = s0 + u < nslab ? b.slab[((size_t)(s0 + u) * SR + row) * d + ch] : 0.f;
#pragma unroll
          for (int u = 0; u < 8; ++u)
            if (s0 + u < nslab) gy += t[u];
        }
      }
      zero_frag(acc);
      // ---- recompute the forward for this channel
      const float rate = -ic_softplus(rd);
      const float ex = expf(rate * steps);
      const float lam = fmaxf(ex, IC_EPS_DIV);
      const float numd = num_in * lam, dend = den_in * lam;
      const float w = expf(fminf(kk, IC_MAX_EXP));
      float g_r, g_v, g_w, g_numd, g_dend, g_lam, g_k_extra = 0.f, g_var_in = 0.f;
      float g_pn_p = 0.f, g_on_p = 0.f;
      if (afd) {
        const float pn = ic_noise(pn_p), on = ic_noise(on_p);
        const float vard = var_in * (lam * lam) + pn * steps;
        const float dpe = dend + IC_EPS_DIV;
        const float prd = numd / dpe;
        const float resid = w * vv - prd;
        const float e = expf(fminf(-kk, IC_MAX_EXP));
        const float obs = e * on + IC_EPS_DIV;
        const float T = vard + obs;
        const float graw = vard / T;
        const float gain = fminf(fmaxf(graw, IC_EPS_DIV), 1.0f);
        const float st = prd + gain * resid;
        const float den_n = dend + w;
        g_r = gy * st;
        const float g_st = gy * rr + gno * den_n;
        const float g_den_n = gno * st + gdo;
        g_dend = g_den_n;
        g_w = g_den_n;
        const float g_gain = g_st * resid - gvo * vard;
        const float g_resid = g_st * gain;
        float g_vard = gvo * (1.0f - gain);
        const float g_graw = (graw >= IC_EPS_DIV && graw <= 1.0f) ? g_gain : 0.f;
        g_vard += g_graw * obs / (T * T);
        const float g_obs = -g_graw * vard / (T * T);
        const float g_prd = g_st - g_resid;
        g_w += g_resid * vv;
        g_v = g_resid * w;
        g_numd = g_prd / dpe;
        g_dend += -g_prd * prd / dpe;
        const float g_e = g_obs * on;
        g_on_p = g_obs * e * ic_noise_grad(on_p);
        g_k_extra = (-kk <= IC_MAX_EXP) ? -g_e * e : 0.f;
        g_var_in = g_vard * (lam * lam);
        g_lam = g_numd * num_in + g_dend * den_in + g_vard * var_in * 2.0f * lam;
        g_pn_p = g_vard * steps * ic_noise_grad(pn_p);
      } else {
        const float num_n = numd + w * vv, den_n = dend + w;
        const float dpe = den_n + IC_EPS_DIV;
        const float q = num_n / dpe;
        g_r = gy * q;
        const float g_q = gy * rr;
        g_numd = gno + g_q / dpe;
        g_dend = gdo - g_q * q / dpe;
        g_v = g_numd * w;
        g_w = g_numd * vv + g_dend;
        g_lam = g_numd * num_in + g_dend * den_in;
      }
      const float g_k = ((kk <= IC_MAX_EXP) ? g_w * w : 0.f) + g_k_extra;
      const float g_rate = (ex >= IC_EPS_DIV) ? g_lam * lam * steps : 0.f;
      const float g_rd = -g_rate * ic_softplus_grad(rd);
      const float g_rpre = g_r * rr * (1.0f - rr);
      // rotation backward (iron_rope.py:63-64)
      float g_vraw = g_v;
      const float g_vo = __shfl_xor(g_v, 1, 64);
      if (rot) {
        const float c = cosf(th), sn = sinf(th);
        g_vraw = (lane & 1) ? (-g_vo * sn + g_v * c) : (g_v * c + g_vo * sn);
      }
      if (valid) {
        b.gnum[idx] = g_numd * lam;
        b.gden[idx] = g_dend * lam;
        if (afd) {
          b.gvar[idx] = g_var_in;
          b.g_pn_rows[idx] = g_pn_p;
          b.g_on_rows[idx] = g_on_p;
        }
        b.g_raw_decay_rows[idx] = g_rd;
        float* tg = b.tape_g + (size_t)row * 4 * d;
        tg[ch] = g_rpre;
        tg[d + ch] = g_k;
        tg[2 * d + ch] = g_vraw;
      }
      // broadcast each channel's gradients and accumulate W^T g (rows past d hold zero weights)
      const float z_r = valid ? g_rpre : 0.f, z_k = valid ? g_k : 0.f, z_v = valid ? g_vraw : 0.f;
#pragma unroll
      for (int q2 = 0; q2 < RPW; ++q2) {
        axpy_frag(acc, __shfl(z_r, q2, 64), wr[q2]);
        axpy_frag(acc, __shfl(z_k, q2, 64), wk[q2]);
        axpy_frag(acc, __shfl(z_v, q2, 64), wv[q2]);
      }
    }
#pragma unroll
    for (int j = 0; j < DCH; ++j) {
      float* p = &red[wave][256 * j + 4 * lane];
      p[0] = acc.v[j].x; p[1] = acc.v[j].y; p[2] = acc.v[j].z; p[3] = acc.v[j].w;
    }
    __syncthreads();
    if (wave == 0) {
      RowFrag<DCH> tot;
#pragma unroll
      for (int j = 0; j < DCH; ++j) {
        const int c = 256 * j + 4 * lane;
        tot.v[j].x = (red[0][c] + red[1][c]) + (red[2][c] + red[3][c]);
        tot.v[j].y = (red[0][c + 1] + red[1][c + 1]) + (red[2][c + 1] + red[3][c + 1]);
        tot.v[j].z = (red[0][c + 2] + red[1][c + 2]) + (red[2][c + 2] + red[3][c + 2]);
        tot.v[j].w = (red[0][c + 3] + red[1][c + 3]) + (red[2][c + 3] + red[3][c + 3]);
      }
      store_row(tot, b.slab2 + ((size_t)blockIdx.x * SR + row) * d, d, lane);
    }
    __syncthreads();
    cur = nxt;
    s_cur = s_nxt;
   }
  }
}

// ---------------------------------------------------------------- split backward (default)
// The r/k/v backward as TWO launches: the per-channel state / WKV / AFD backward is elementwise over
// the active rows (k_region_chan_bwd: one wave per (stream, region) row, float4 lanes, every operand
// row requested in one batch), and the transposed products W_{r,k,v}^T g are a pure weight-streaming
// kernel (k_region_rkvT_bwd) that carries nothing but the weights, the accumulators of a batch of
// streams and their 3*RPW gradient scalars: 236 -> ~130 VGPRs, no dependent load after the weights,
// one LDS hand-over (two barriers) per batch of NSB streams instead of per stream, every wave
// reducing one stream of the batch.  Same arithmetic, same order of additions as k_region_rkv_bwd.
// value of lane `l` (a compile-time constant after unrolling) in every lane: v_readlane_b32 into a
// scalar register -- no LDS crossbar trip, no vector register for the broadcast
__device__ __forceinline__ float lane_bcast(float v, int l) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l));
}

struct ChanOut {
  float gnum, gden, gvar, g_pn, g_on, g_rd, g_rpre, g_k, g_v;
};

__device__ __forceinline__ ChanOut chan_bwd_one(bool afd, float steps, float rd, float pn_p, float on_p,
                                                float num_in, float den_in, float rr, float kk, float vv,
                                                float gno, float gdo, float var_in, float gvo, float gy) {
  ChanOut o;
  const float rate = -ic_softplus(rd);
  const float ex = expf(rate * steps);
  const float lam = fmaxf(ex, IC_EPS_DIV);
  const float numd = num_in * lam, dend = den_in * lam;
  const float w = expf(fminf(kk, IC_MAX_EXP));
  float g_r, g_v, g_w, g_numd, g_dend, g_lam, g_k_extra = 0.f, g_var_in = 0.f;
  float g_pn_p = 0.f, g_on_p = 0.f;
  if (afd) {
    const float pn = ic_noise(pn_p), on = ic_noise(on_p);
    const float vard = var_in * (lam * lam) + pn * steps;
    const float dpe = dend + IC_EPS_DIV;
    const float prd = numd / dpe;
    const float resid = w * vv - prd;
    const float e = expf(fminf(-kk, IC_MAX_EXP));
    const float obs = e * on + IC_EPS_DIV;
    const float T = vard + obs;
    const float graw = vard / T;
    const float gain = fminf(fmaxf(graw, IC_EPS_DIV), 1.0f);
    const float st = prd + gain * resid;
    const float den_n = dend + w;
    g_r = gy * st;
    const float g_st = gy * rr + gno * den_n;
    const float g_den_n = gno * st + gdo;
    g_dend = g_den_n;
    g_w = g_den_n;
    const float g_gain = g_st * resid - gvo * vard;
    const float g_resid = g_st * gain;
    float g_vard = gvo * (1.0f - gain);
    const float g_graw = (graw >= IC_EPS_DIV && graw <= 1.0f) ? g_gain : 0.f;
    g_vard += g_graw * obs / (T * T);
    const float g_obs = -g_graw * vard / (T * T);
    const float g_prd = g_st - g_resid;
    g_w += g_resid * vv;
    g_v = g_resid * w;
    g_numd = g_prd / dpe;
    g_dend += -g_prd * prd / dpe;
    const float g_e = g_obs * on;
    g_on_p = g_obs * e * ic_noise_grad(on_p);
    g_k_extra = (-kk <= IC_MAX_EXP) ? -g_e * e : 0.f;
    g_var_in = g_vard * (lam * lam);
    g_lam = g_numd * num_in + g_dend * den_in + g_vard * var_in * 2.0f * lam;
    g_pn_p = g_vard * steps * ic_noise_grad(pn_p);
  } else {
    const float num_n = numd + w * vv, den_n = dend + w;
    const float dpe = den_n + IC_EPS_DIV;
    const float q = num_n / dpe;
    g_r = gy * q;
    const float g_q = gy * rr;
    g_numd = gno + g_q / dpe;
    g_dend = gdo - g_q * q / dpe;
    g_v = g_numd * w;
    g_w = g_numd * vv + g_dend;
    g_lam = g_numd * num_in + g_dend * den_in;
  }
  o.g_k = ((kk <= IC_MAX_EXP) ? g_w * w : 0.f) + g_k_extra;
  const float g_rate = (ex >= IC_EPS_DIV) ? g_lam * lam * steps : 0.f;
  o.g_rd = -g_rate * ic_softplus_grad(rd);
  o.g_rpre = g_r * rr * (1.0f - rr);
  o.g_v = g_v;
  o.gnum = g_numd * lam;
  o.gden = g_dend * lam;
  o.gvar = g_var_in;
  o.g_pn = g_pn_p;
  o.g_on = g_on_p;
  return o;
}

template <int DCH>
__global__ __launch_bounds__(256) void k_region_chan_bwd(ic_region_bwd_args b, int nslab) {
  const ic_region_args& a = b.f;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int s, r, row;
  if (!spread_row(b, blockIdx.x * 4 + wave, s, r, row)) return;
  if (!a.mask[row]) return;
  const int d = a.d;
  const int SR = a.S * stream_step(b) * a.R;
  const size_t off = (size_t)row * d;
  const bool afd = a.flags & IC_F_AFD, vec = a.flags & IC_F_NOISE_VEC, rt = a.flags & IC_F_RT;
  RowFrag<DCH> num, den, rs, ks, vs, gno, gdo, var, gvo, rdv, pnv, onv, gy;
  load_row(num, a.num + off, d, lane);
  load_row(den, a.den + off, d, lane);
  load_row(rs, a.r_s + off, d, lane);
  load_row(ks, a.k_s + off, d, lane);
  load_row(vs, a.v_s + off, d, lane);
  load_opt(gno, b.gnum_o, off, d, lane);
  load_opt(gdo, b.gden_o, off, d, lane);
  load_row(rdv, a.raw_decay + (size_t)r * d, d, lane);
  zero_frag(var); zero_frag(gvo); zero_frag(pnv); zero_frag(onv);
  float pn_s = 0.f, on_s = 0.f;
  if (afd) {
    load_row(var, a.var + off, d, lane);
    load_opt(gvo, b.gvar_o, off, d, lane);
    if (vec) {
      load_row(pnv, a.pn_param + (size_t)r * d, d, lane);
      load_row(onv, a.on_param + (size_t)r * d, d, lane);
    } else {
      pn_s = a.pn_param[r];
      on_s = a.on_param[r];
    }
  }
  const float steps = (float)(a.dt[row] + 1);
  // gy: the o-projection's slabs, reduced here (radial-tangential: plus its scalar chain)
  if (rt) {
    rt_gy_row<DCH>(b, gy, row, r, nslab, lane, true);
  } else {
    zero_frag(gy);
    sum_slabs(gy, b.slab + off, (size_t)SR * d, nslab, d, lane);
  }
  RowFrag<DCH> o_num, o_den, o_var, o_pn, o_on, o_rd, t_r, t_k, t_v;
#pragma unroll
  for (int j = 0; j < DCH; ++j) {
    const int ch0 = 256 * j + 4 * lane;
#define IC_CH(c) chan_bwd_one(afd, steps, rdv.v[j].c, vec ? pnv.v[j].c : pn_s, vec ? onv.v[j].c : on_s, num.v[j].c, \
                              den.v[j].c, rs.v[j].c, ks.v[j].c, vs.v[j].c, gno.v[j].c, gdo.v[j].c, var.v[j].c,      \
                              gvo.v[j].c, gy.v[j].c)
    const ChanOut c0 = IC_CH(x), c1 = IC_CH(y), c2 = IC_CH(z), c3 = IC_CH(w);
#undef IC_CH
    o_num.v[j] = make_float4(c0.gnum, c1.gnum, c2.gnum, c3.gnum);
    o_den.v[j] = make_float4(c0.gden, c1.gden, c2.gden, c3.gden);
    o_var.v[j] = make_float4(c0.gvar, c1.gvar, c2.gvar, c3.gvar);
    o_pn.v[j] = make_float4(c0.g_pn, c1.g_pn, c2.g_pn, c3.g_pn);
    o_on.v[j] = make_float4(c0.g_on, c1.g_on, c2.g_on, c3.g_on);
    o_rd.v[j] = make_float4(c0.g_rd, c1.g_rd, c2.g_rd, c3.g_rd);
    t_r.v[j] = make_float4(c0.g_rpre, c1.g_rpre, c2.g_rpre, c3.g_rpre);
    t_k.v[j] = make_float4(c0.g_k, c1.g_k, c2.g_k, c3.g_k);
    // rotation backward (iron_rope.py:63-64): channels (2p, 2p+1) rotate together by step_pos * W_time[p]
    float v0 = c0.g_v, v1 = c1.g_v, v2 = c2.g_v, v3 = c3.g_v;
    if (ch0 < 2 * a.m_time) {
      const float th = a.step_pos * a.W_time[ch0 >> 1];
      const float c = cosf(th), sn = sinf(th);
      const float e = c0.g_v * c + c1.g_v * sn, o = -c0.g_v * sn + c1.g_v * c;
      v0 = e; v1 = o;
    }
    if (ch0 + 2 < 2 * a.m_time) {
      const float th = a.step_pos * a.W_time[(ch0 >> 1) + 1];
      const float c = cosf(th), sn = sinf(th);
      const float e = c2.g_v * c + c3.g_v * sn, o = -c2.g_v * sn + c3.g_v * c;
      v2 = e; v3 = o;
    }
    t_v.v[j] = make_float4(v0, v1, v2, v3);
  }
  store_row(o_num, b.gnum + off, d, lane);
  store_row(o_den, b.gden + off, d, lane);
  if (afd) {
    store_row(o_var, b.gvar + off, d, lane);
    store_row(o_pn, b.g_pn_rows + off, d, lane);
    store_row(o_on, b.g_on_rows + off, d, lane);
  }
  store_row(o_rd, b.g_raw_decay_rows + off, d, lane);
  float* tg = b.tape_g + (size_t)row * 4 * d;
  store_row(t_r, tg, d, lane);
  store_row(t_k, tg + d, d, lane);
  store_row(t_v, tg + 2 * d, d, lane);
}

// The batch hand-over shared by the two transposed-product kernels: every wave leaves its partial
// rows of the batch's streams in LDS, wave u sums stream u's four partials in wave order.
template <int DCH, int NSB>
struct BatchRed {
  float v[4][NSB][DCH * 256];
};

template <int DCH, int RPW, int NSB>
__global__ __launch_bounds__(256) void k_region_rkvT_bwd(ic_region_bwd_args b) {
  const ic_region_args& a = b.f;
  const int r = blockIdx.y;
  const int sk = stream_step(b);
  __shared__ BatchRed<DCH, NSB> red;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int d = a.d;
  const int row0 = blockIdx.x * (4 * RPW) + wave * RPW;
  const size_t dd = (size_t)d * d;
  const int SR = a.S * sk * a.R;
  const int nz = gridDim.z;
  RowFrag<DCH> wr[RPW], wk[RPW], wv[RPW];
  bool loaded = false;
  // lane l < 3*RPW fetches gradient scalar (matrix l / RPW, row row0 + l % RPW) of a stream
  const int zm = lane / RPW, zq = lane % RPW;
  const bool zvalid = lane < 3 * RPW && row0 + zq < d;
  for (int sc = blockIdx.z; sc < a.S; sc += 64 * nz) {
    const int sl = sc + lane * nz;
    unsigned long long act = __ballot(sl < a.S && a.mask[(size_t)sl * sk * a.R + r] != 0);
    if (!act) continue;
    if (!loaded) {  // (wave-uniform) the weights are only requested once the region is known to have work
      const float* Wr = a.W + (size_t)r * 4 * dd;
#pragma unroll
      for (int i = 0; i < RPW; ++i) {
        if (row0 + i < d) {
          load_row(wr[i], Wr + (size_t)(row0 + i) * d, d, lane);
          load_row(wk[i], Wr + dd + (size_t)(row0 + i) * d, d, lane);
          load_row(wv[i], Wr + 2 * dd + (size_t)(row0 + i) * d, d, lane);
        } else {
          zero_frag(wr[i]); zero_frag(wk[i]); zero_frag(wv[i]);
        }
      }
      loaded = true;
    }
    while (act) {
      int srow[NSB];
      float z[NSB];
#pragma unroll
      for (int u = 0; u < NSB; ++u) {
        srow[u] = -1;
        z[u] = 0.f;
        if (act) {
          const int s = sc + (__ffsll((long long)act) - 1) * nz;
          act &= act - 1;
          srow[u] = s * sk * a.R + r;
          if (zvalid) z[u] = b.tape_g[((size_t)srow[u] * 4 + zm) * d + row0 + zq];
        }
      }
#pragma unroll
      for (int u = 0; u < NSB; ++u) {
        if (srow[u] < 0) continue;  // wave-uniform
        RowFrag<DCH> acc;
        zero_frag(acc);
#pragma unroll
        for (int q = 0; q < RPW; ++q) {
          axpy_frag(acc, lane_bcast(z[u], q), wr[q]);
          axpy_frag(acc, lane_bcast(z[u], RPW + q), wk[q]);
          axpy_frag(acc, lane_bcast(z[u], 2 * RPW + q), wv[q]);
        }
#pragma unroll
        for (int j = 0; j < DCH; ++j) *reinterpret_cast<float4*>(&red.v[wave][u][256 * j + 4 * lane]) = acc.v[j];
      }
      __syncthreads();
#pragma unroll
      for (int u = 0; u < NSB; ++u) {
        if (srow[u] < 0 || (u & 3) != wave) continue;
        RowFrag<DCH> tot;
#pragma unroll
        for (int j = 0; j < DCH; ++j) {
          const int c = 256 * j + 4 * lane;
          const float4 p0 = *reinterpret_cast<const float4*>(&red.v[0][u][c]), p1 = *reinterpret_cast<const float4*>(&red.v[1][u][c]);
          const float4 p2 = *reinterpret_cast<const float4*>(&red.v[2][u][c]), p3 = *reinterpret_cast<const float4*>(&red.v[3][u][c]);
          tot.v[j] = make_float4((p0.x + p1.x) + (p2.x + p3.x), (p0.y + p1.y) + (p2.y + p3.y),
                                 (p0.z + p1.z) + (p2.z + p3.z), (p0.w + p1.w) + (p2.w + p3.w));
        }
        store_row(tot, b.slab2 + ((size_t)blockIdx.x * SR + srow[u]) * d, d, lane);
      }
      __syncthreads();
    }
  }
}

// ---------------------------------------------------------------- prologue backward
template <int DCH>
__global__ __launch_bounds__(256) void k_region_pro_bwd(ic_region_bwd_args b, int nslab, int fold,
                                                        float one_minus_alpha) {
  const ic_region_args& a = b.f;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int s, r, row;
  if (!spread_row(b, blockIdx.x * 4 + wave, s, r, row)) return;
  const int sk = stream_step(b);
  const int SR = a.S * sk * a.R;
  if (!a.mask[row]) {
    if (fold) inactive_passthrough<DCH>(b, row, lane, one_minus_alpha);
    return;
  }
  const int d = a.d;
  const size_t off = (size_t)row * d;
  const float inv_d = 1.0f / (float)d;
  RowFrag<DCH> g, t, u, w;
  if (fold) {  // hard KWTA: ghpre = gH * gate
    load_opt(g, b.gH, off, d, lane);
    load_row(t, a.gate + off, d, lane);
#pragma unroll
    for (int j = 0; j < DCH; ++j) {
      g.v[j].x *= t.v[j].x; g.v[j].y *= t.v[j].y; g.v[j].z *= t.v[j].z; g.v[j].w *= t.v[j].w;
    }
  } else {
    load_row(g, b.ghpre + off, d, lane);  // residual path h = x + ...
  }
  sum_slabs(g, b.slab2 + (size_t)row * d, (size_t)SR * d, nslab, d, lane);
  load_row(t, a.x + off, d, lane);
  store_row(t, b.tape_x + off, d, lane);
  // ---- region norm: x = nw * u1 * inv1
  load_row(u, a.u1 + off, d, lane);
  load_row(w, a.norm_w + (size_t)r * d, d, lane);
  const float inv1 = a.inv_s[row * 2 + 1];
  RowFrag<DCH> gw;
#pragma unroll
  for (int j = 0; j < DCH; ++j) {
    gw.v[j] = make_float4(g.v[j].x * u.v[j].x * inv1, g.v[j].y * u.v[j].y * inv1,
                          g.v[j].z * u.v[j].z * inv1, g.v[j].w * u.v[j].w * inv1);
    g.v[j] = make_float4(g.v[j].x * w.v[j].x, g.v[j].y * w.v[j].y, g.v[j].z * w.v[j].z,
                         g.v[j].w * w.v[j].w);
  }
  store_row(gw, b.g_norm_w_rows + off, d, lane);
  float dotp = wave_sum(dot_frag(g, u));
  float k3 = dotp * inv1 * inv1 * inv1 * inv_d;
#pragma unroll
  for (int j = 0; j < DCH; ++j) {
    g.v[j] = make_float4(g.v[j].x * inv1 - u.v[j].x * k3, g.v[j].y * inv1 - u.v[j].y * k3,
                         g.v[j].z * inv1 - u.v[j].z * k3, g.v[j].w * inv1 - u.v[j].w * k3);
  }
  if (a.flags & IC_F_PRED) store_row(g, b.gpred + off, d, lane);
  if (a.flags & IC_F_NORM_IN) {
    const bool io = a.sensor && (r == a.sensor_idx || r == a.motor_idx);
    load_row(u, a.M + off, d, lane);
    if (io) {
      load_row(t, a.sensor + (size_t)s * d, d, lane);
#pragma unroll
      for (int j = 0; j < DCH; ++j) {
        u.v[j].x += t.v[j].x; u.v[j].y += t.v[j].y; u.v[j].z += t.v[j].z; u.v[j].w += t.v[j].w;
      }
    }
    load_row(w, a.norm_in_w, d, lane);
    const float inv0 = a.inv_s[row * 2 + 0];
#pragma unroll
    for (int j = 0; j < DCH; ++j) {
      gw.v[j] = make_float4(g.v[j].x * u.v[j].x * inv0, g.v[j].y * u.v[j].y * inv0,
                            g.v[j].z * u.v[j].z * inv0, g.v[j].w * u.v[j].w * inv0);
      g.v[j] = make_float4(g.v[j].x * w.v[j].x, g.v[j].y * w.v[j].y, g.v[j].z * w.v[j].z,
                           g.v[j].w * w.v[j].w);
    }
    store_row(gw, b.g_norm_in_rows + off, d, lane);
    dotp = wave_sum(dot_frag(g, u));
    k3 = dotp * inv0 * inv0 * inv0 * inv_d;
#pragma unroll
    for (int j = 0; j < DCH; ++j) {
      g.v[j] = make_float4(g.v[j].x * inv0 - u.v[j].x * k3, g.v[j].y * inv0 - u.v[j].y * k3,
                           g.v[j].z * inv0 - u.v[j].z * k3, g.v[j].w * inv0 - u.v[j].w * k3);
    }
    if (io && b.gsens) {
      if (r == a.sensor_idx) store_row(g, b.gsens + ((size_t)s * sk * 2 + 0) * d, d, lane);
      else store_row(g, b.gsens + ((size_t)s * sk * 2 + 1) * d, d, lane);
    }
  }
  store_row(g, b.gM + off, d, lane);
}

// ---------------------------------------------------------------- deferred weight gradient
// Block (row slice of 8, matrix j, region r): dW[r,j,rows,:] += sum_{t,s} g[t,s,r,j,rows] (x) op[t,s,r,:]
template <int DCH>
__global__ __launch_bounds__(256) void k_wgrad_region(ic_wgrad_region_args a) {
  const int r = blockIdx.z, jm = blockIdx.y;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int d = a.d;
  const int row0 = blockIdx.x * 8 + wave * 2;
  if (row0 >= d) return;
  RowFrag<DCH> acc0, acc1, op;
  zero_frag(acc0);
  zero_frag(acc1);
  // (slot, stream) pairs tested 64 at a time, contributing ones walked in order with the next
  // pair's operands in flight (see k_wgrad_edge): same order of additions as a plain scan
  bool any = false;
  const int TS = a.T * a.S;
  const float* opbase = jm < 3 ? a.tape_x : a.tape_o;
  for (int base = 0; base < TS; base += 64) {
    const int ts = base + lane;
    unsigned long long act = __ballot(ts < TS && a.visited[ts / a.S] != 0 && a.mask[(size_t)ts * a.R + r] != 0);
    if (!act) continue;
    any = true;
    RowFrag<DCH> nxt;
    float g0 = 0.f, g1 = 0.f, h0 = 0.f, h1 = 0.f;
    int cur = base + __ffsll((long long)act) - 1;
    act &= act - 1;
    {
      const size_t row = (size_t)cur * a.R + r;
      const float* g = a.tape_g + (row * 4 + jm) * d;
      g0 = g[row0]; g1 = g[row0 + 1];
      load_row(op, opbase + row * d, d, lane);
    }
    while (cur >= 0) {
      const int nx = act ? base + __ffsll((long long)act) - 1 : -1;
      act &= act - 1;
      if (nx >= 0) {
        const size_t row = (size_t)nx * a.R + r;
        const float* g = a.tape_g + (row * 4 + jm) * d;
        h0 = g[row0]; h1 = g[row0 + 1];
        load_row(nxt, opbase + row * d, d, lane);
      }
      axpy_frag(acc0, g0, op);
      axpy_frag(acc1, g1, op);
      op = nxt; g0 = h0; g1 = h1;
      cur = nx;
    }
  }
  if (!any) return;
  float* out = a.gW + (((size_t)r * 4 + jm) * d + row0) * d;
  RowFrag<DCH> cur;
  load_row(cur, out, d, lane);
#pragma unroll
  for (int j = 0; j < DCH; ++j) {
    cur.v[j].x += acc0.v[j].x; cur.v[j].y += acc0.v[j].y; cur.v[j].z += acc0.v[j].z; cur.v[j].w += acc0.v[j].w;
  }
  store_row(cur, out, d, lane);
  load_row(cur, out + d, d, lane);
#pragma unroll
  for (int j = 0; j < DCH; ++j) {
    cur.v[j].x += acc1.v[j].x; cur.v[j].y += acc1.v[j].y; cur.v[j].z += acc1.v[j].z; cur.v[j].w += acc1.v[j].w;
  }
  store_row(cur, out + d, d, lane);
  if (blockIdx.x == 0 && jm == 0 && threadIdx.x == 0) a.touched[r] = 1;
}

}  // namespace

extern "C" int ic_bwd_nslab(int32_t d) { return d <= 256 ? ic_ceil_div(d, 32) : ic_ceil_div(d, 8); }

extern "C" int ic_region_phase_bwd(const ic_region_bwd_args* b, ic_stream_t stream) {
  IC_REQUIRE(b != nullptr, "ic_region_phase_bwd: NULL args");
  const ic_region_args* a = &b->f;
  IC_REQUIRE(a->S > 0 && a->R > 0, "ic_region_phase_bwd: S=%d R=%d", a->S, a->R);
  IC_REQUIRE(a->d >= 4 && a->d % 4 == 0 && a->d <= 1024, "ic_region_phase_bwd: d=%d unsupported", a->d);
  IC_REQUIRE(a->W && a->norm_w && a->raw_decay && a->M && a->mask && a->num && a->den && a->dt &&
                 a->x && a->u1 && a->inv_s && a->r_s && a->k_s && a->v_s && a->y && a->hpre && a->gate,
             "ic_region_phase_bwd: NULL forward pointer");
  IC_REQUIRE(b->gM && b->gnum && b->gden && b->g_norm_w_rows && b->g_raw_decay_rows && b->tape_g &&
                 b->tape_x && b->tape_o && b->ghpre && b->slab && b->slab2 && b->slab != b->slab2,
             "ic_region_phase_bwd: NULL gradient/workspace pointer");
  IC_REQUIRE(!(a->flags & IC_F_PRED) || b->gpred, "ic_region_phase_bwd: gpred is NULL");
  IC_REQUIRE(!(a->flags & IC_F_NORM_IN) || (a->norm_in_w && b->g_norm_in_rows),
             "ic_region_phase_bwd: norm_in buffers are NULL");
  IC_REQUIRE(!(a->flags & IC_F_AFD) || (a->var && a->pn_param && a->on_param && b->gvar &&
                                        b->g_pn_rows && b->g_on_rows),
             "ic_region_phase_bwd: AFD buffers are NULL");
  IC_REQUIRE(!(a->flags & IC_F_RT) || (a->radius && a->radius_var && a->radius_o && a->ynorm &&
                                       a->rpn_param && a->ron_param && b->gradius &&
                                       b->gradius_var && b->g_rnoise_rows && b->slab_rt),
             "ic_region_phase_bwd: radial-tangential buffers are NULL");
  IC_REQUIRE(a->m_time == 0 || a->W_time, "ic_region_phase_bwd: W_time is NULL");
  hipStream_t st = (hipStream_t)stream;
  const int rows = a->S * a->R;
  const int nslab = ic_bwd_nslab(a->d);
  // stream split of the slice kernels: aim at >= 512 workgroups, at most one stream group per stream
  int nz = 1;
  while (nz < a->S && nz < 16 && nslab * a->R * nz < 512) nz *= 2;
  const dim3 blk(256), grow(ic_ceil_div(rows, 4)), gmat(nslab, a->R, nz);
  const float oma = (float)(1.0 - a->pred_alpha);
  const bool soft = (a->flags & IC_F_KWTA_SOFT) && !(a->flags & IC_F_NOKWTA) && a->kwta_k > 0;
  const int fold = soft ? 0 : 1;
#define IC_REGION_BWD(DCH_, RPW_, NSB_)                                                                                      \
  do {                                                                                                                       \
    constexpr int DCH = DCH_;                                                                                                \
    if (!fold)                                                                                                               \
      IC_TIMED(IC_KID_region_fin_bwd, st, hipLaunchKernelGGL(k_region_fin_bwd<DCH>, grow, blk, 0, st, *b, oma));             \
    IC_TIMED(IC_KID_region_out_bwd, st, hipLaunchKernelGGL((k_region_out_bwd<DCH, RPW_>), gmat, blk, 0, st, *b, fold));      \
    if (split) {                                                                                                             \
      IC_TIMED(IC_KID_region_mid_bwd, st, hipLaunchKernelGGL(k_region_chan_bwd<DCH>, grow, blk, 0, st, *b, nslab));          \
      IC_TIMED(IC_KID_region_rkv_bwd, st, hipLaunchKernelGGL((k_region_rkvT_bwd<DCH, RPW_, NSB_>), gmat, blk, 0, st, *b));   \
    } else {                                                                                                                 \
      IC_TIMED(IC_KID_region_rkv_bwd, st, hipLaunchKernelGGL((k_region_rkv_bwd<DCH, RPW_>), gmat, blk, 0, st, *b, nslab));   \
    }                                                                                                                        \
    IC_TIMED(IC_KID_region_pro_bwd, st, hipLaunchKernelGGL(k_region_pro_bwd<DCH>, grow, blk, 0, st, *b, nslab, fold, oma));  \
  } while (0)
  // The r/k/v backward as one fused launch (fewer than 8 streams: the replay of a single loop) or split into the
  // per-channel kernel + the weight-streaming kernel.  Measured (profiles/r02b_*): R=512, k=32, S=64 fused 162 us,
  // split 20 + 85 us; S=8 fused 51 us, split 10 + 34 us; the batched backward of the 10M bench (S=16, R=10,
  // streams split over gridDim.z) fused 27 us, split 16 + 8 us.  IC_BWD_SPLIT=0/1 forces one of them (A/B timing;
  // same results bit for bit).
  static const int split_env = [] { const char* e = getenv("IC_BWD_SPLIT"); return (e && e[0]) ? (e[0] == '0' ? 0 : 1) : -1; }();
  const bool split = split_env >= 0 ? split_env == 1 : a->S >= 8;
  if (a->d <= 256)
    IC_REGION_BWD(1, 8, 4);
  else if (a->d <= 512)
    IC_REGION_BWD(2, 2, 4);
  else
    IC_REGION_BWD(4, 2, 2);
#undef IC_REGION_BWD
  IC_LAUNCH_CHECK("ic_region_phase_bwd");
  return IC_OK;
}

extern "C" int ic_region_wgrad(const ic_wgrad_region_args* a, ic_stream_t stream) {
  IC_REQUIRE(a != nullptr, "ic_region_wgrad: NULL args");
  IC_REQUIRE(a->T >= 0 && a->S > 0 && a->R > 0, "ic_region_wgrad: T=%d S=%d R=%d", a->T, a->S, a->R);
  IC_REQUIRE(a->d >= 4 && a->d % 4 == 0 && a->d <= 1024, "ic_region_wgrad: d=%d unsupported", a->d);
  IC_REQUIRE(a->tape_g && a->tape_x && a->tape_o && a->mask && a->visited && a->gW && a->touched,
             "ic_region_wgrad: NULL pointer");
  if (a->T == 0) return IC_OK;
  const dim3 grid(ic_ceil_div(a->d, 8), 4, a->R);
  IC_DISPATCH_DCH(a->d, { IC_TIMED(IC_KID_wgrad_region, (hipStream_t)stream, hipLaunchKernelGGL(k_wgrad_region<DCH>, grid, dim3(256), 0, (hipStream_t)stream, *a)); });
  IC_LAUNCH_CHECK("ic_region_wgrad");
  return IC_OK;
}
