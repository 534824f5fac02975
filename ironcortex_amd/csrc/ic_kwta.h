// KWTA device helpers shared by the region finish kernel, ic_kwta_fwd and the workspace write.
#pragma once
#include "ic_common.h"

// k-th largest |x| over a wave-distributed row, as raw float bits (utils.py:53).
// Padding lanes must hold 0.  Bit-by-bit search: largest t with count(|x| >= t) >= k.
template <int DCH>
__device__ __forceinline__ float kth_largest_abs(const RowFrag<DCH>& h, int k) {
  uint32_t u[4 * DCH];
#pragma unroll
  for (int j = 0; j < DCH; ++j) {
    u[4 * j + 0] = __float_as_uint(fabsf(h.v[j].x));
    u[4 * j + 1] = __float_as_uint(fabsf(h.v[j].y));
    u[4 * j + 2] = __float_as_uint(fabsf(h.v[j].z));
    u[4 * j + 3] = __float_as_uint(fabsf(h.v[j].w));
  }
  uint32_t t = 0;
  for (int b = 30; b >= 0; --b) {
    const uint32_t trial = t | (1u << b);
    int cnt = 0;
#pragma unroll
    for (int e = 0; e < 4 * DCH; ++e) cnt += __popcll(__ballot(u[e] >= trial));
    if (cnt >= k) t = trial;
  }
  return __uint_as_float(t);
}

__device__ __forceinline__ float kwta_gate(float v, float thr, bool soft, float temp) {
  const float av = fabsf(v);
  if (av >= thr) return 1.0f;
  return soft ? ic_sigmoid((av - thr) / temp) : 0.0f;  // utils.py:55-57
}

template <int DCH>
__device__ __forceinline__ void kwta_row(const RowFrag<DCH>& h, RowFrag<DCH>& out, RowFrag<DCH>& g,
                                         int d, int k, bool soft, float temp) {
  if (k <= 0) {  // utils.py:48-49
    zero_frag(out);
    zero_frag(g);
    return;
  }
  k = min(max(k, 1), d);
  const float thr = kth_largest_abs(h, k);
#pragma unroll
  for (int j = 0; j < DCH; ++j) {
    g.v[j].x = kwta_gate(h.v[j].x, thr, soft, temp); g.v[j].y = kwta_gate(h.v[j].y, thr, soft, temp);
    g.v[j].z = kwta_gate(h.v[j].z, thr, soft, temp); g.v[j].w = kwta_gate(h.v[j].w, thr, soft, temp);
    out.v[j].x = h.v[j].x * g.v[j].x; out.v[j].y = h.v[j].y * g.v[j].y;
    out.v[j].z = h.v[j].z * g.v[j].z; out.v[j].w = h.v[j].w * g.v[j].w;
  }
}

