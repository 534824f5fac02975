// Router pieces shared by router.hip and region_fwd.hip (ic_loop_step_fwd, where the
// per-destination softmax combine rides in the region kernel).  Reference gate.py:300-326.
#pragma once
#include "ic_common.h"

namespace {

// Row r of stream s of the input states / previous mask; streams may be strided (the branches of
// one outer iteration read their inputs from arbitrary step slots).
__device__ __forceinline__ const float* h_row(const ic_router_args& a, int s, int r) {
  if (a.in_slot) return a.H + ((int64_t)a.in_slot[s] * a.R + r) * a.d;
  const int64_t hs = a.H_sstride ? a.H_sstride : (int64_t)a.R * a.d;
  return a.H + (int64_t)s * hs + (size_t)r * a.d;
}
__device__ __forceinline__ bool prev_on(const ic_router_args& a, int s, int r) {
  if (a.in_slot) return a.mask_prev[(int64_t)a.in_slot[s] * a.R + r] != 0;
  const int64_t ms = a.mask_sstride ? a.mask_sstride : (int64_t)a.R;
  return a.mask_prev[(int64_t)s * ms + r] != 0;
}

// Precision-routed combine of one (stream, destination) by one wave, d <= 256 * DCH with all of
// the <= 8 incoming edges' rows in registers: acc = sum_e softmax_e(q.k_e/sqrt(d) - mah_e/2) msg_e
// over the edges whose sender was active (zeros when none).  The receiver's own rows and the edge
// tables are requested before the senders' activity is known; message, key and precision rows of
// every active edge are requested together.  store: also write M[s,r] and the edge weights.
template <int DCH>
__device__ __forceinline__ void combine_row(const ic_router_args& a, int s, int r, int lane, bool store,
                                            RowFrag<DCH>& acc) {
  const int d = a.d;
  const int row = s * a.R + r;
  const int e0 = a.dst_off[r], deg = a.dst_off[r + 1] - e0;  // <= 8, checked on the host
  RowFrag<DCH> hd, q;
  load_row(hd, h_row(a, s, r), d, lane);
  load_row(q, a.q + (size_t)row * d, d, lane);
  const float sqrt_d = sqrtf((float)d);
  int srcv[8], wix[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int e = e0 + (i < deg ? i : 0);
    srcv[i] = deg > 0 ? a.src[e] : 0;
    wix[i] = deg > 0 ? a.widx[e] : 0;
  }
  bool act[8];
  bool any = false;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    act[i] = i < deg && prev_on(a, s, srcv[i]);
    any |= act[i];
  }
  zero_frag(acc);
  if (any) {
    RowFrag<DCH> m[8], k[8], P[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (act[i]) {
        const size_t eo = ((size_t)s * a.E + e0 + i) * d;
        load_row(m[i], a.msg + eo, d, lane);
        load_row(k[i], a.keyv + eo, d, lane);
        load_row(P[i], a.raw_P + (size_t)wix[i] * d, d, lane);
      }
    }
    float logw[8];
    float mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      logw[i] = -INFINITY;
      if (act[i]) {
        const float score = wave_sum(dot_frag(q, k[i])) / sqrt_d;  // gate.py:308
        float part = 0.f;
#pragma unroll
        for (int j = 0; j < DCH; ++j) {
          const int c = 4 * lane + 256 * j;
          if (c < d) {
            float t;
            t = m[i].v[j].x - hd.v[j].x; part += (t * t) * ic_softplus(P[i].v[j].x);
            t = m[i].v[j].y - hd.v[j].y; part += (t * t) * ic_softplus(P[i].v[j].y);
            t = m[i].v[j].z - hd.v[j].z; part += (t * t) * ic_softplus(P[i].v[j].z);
            t = m[i].v[j].w - hd.v[j].w; part += (t * t) * ic_softplus(P[i].v[j].w);
          }
        }
        const float mah = fminf(wave_sum(part), IC_MAX_EXP / 0.5f);  // gate.py:310
        logw[i] = score - 0.5f * mah;
        mx = fmaxf(mx, logw[i]);
      }
    }
    float esum = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i)
      if (act[i]) esum += expf(logw[i] - mx);
    const float logZ = mx + logf(fmaxf(esum, IC_EPS_DIV));  // gate.py:316
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (i < deg) {
        float w = 0.f;
        if (act[i]) {
          w = expf(logw[i] - logZ);
#pragma unroll
          for (int j = 0; j < DCH; ++j) {
            acc.v[j].x += w * m[i].v[j].x; acc.v[j].y += w * m[i].v[j].y;
            acc.v[j].z += w * m[i].v[j].z; acc.v[j].w += w * m[i].v[j].w;
          }
        }
        if (store && lane == 0) a.wgt[(size_t)s * a.E + e0 + i] = w;
      }
    }
  } else if (store && lane == 0) {
    for (int i = 0; i < deg; ++i) a.wgt[(size_t)s * a.E + e0 + i] = 0.f;
  }
  if (store) store_row(acc, a.M + (size_t)row * d, d, lane);
}

}  // namespace
