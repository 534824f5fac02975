// Gate selection bodies shared by gate.hip (ic_gate_fwd) and router.hip (ic_front_fwd, where the
// gate rides in the router's launch).  Reference gate.py:54-87, utils.py:107-124.
#pragma once
#include "ic_common.h"

namespace {

// per-stream inputs of a multi-stream launch (ic_gate_args.in_slot / t5_s / t6_s / k_total_s), else the shared ones
__device__ __forceinline__ const float* gate_h_row(const ic_gate_args& a, int s, int r) {
  if (a.in_slot) return a.H + ((int64_t)a.in_slot[s] * a.R + r) * a.d;
  return a.H + (size_t)s * (a.H_sstride ? a.H_sstride : (int64_t)a.R * a.d) + (size_t)r * a.d;
}
__device__ __forceinline__ float gate_t5(const ic_gate_args& a, int s) { return a.t5_s ? a.t5_s[s] : a.t5; }
__device__ __forceinline__ float gate_t6(const ic_gate_args& a, int s) { return a.t6_s ? a.t6_s[s] : a.t6; }

// Stable descending rank by counting, then the greedy suppression walk on one lane (R <= 1024,
// the walk stops after k selections).  sc[] holds the stream's scores; all threads call.
__device__ __forceinline__ void gate_select_body(const ic_gate_args& a, int s, const float* sc, int* order,
                                                 uint8_t* supp) {
  const int R = a.R;
  for (int i = threadIdx.x; i < R; i += blockDim.x) {
    supp[i] = 0;
    a.mask[s * R + i] = 0;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < R; i += blockDim.x) {
    const float v = sc[i];
    int rank = 0;
    for (int j = 0; j < R; ++j) {
      const float u = sc[j];
      rank += (u > v) || (u == v && j < i);
    }
    order[rank] = i;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    uint8_t* mask = a.mask + s * R;
    const int kt = a.k_total_s ? a.k_total_s[s] : a.k_total;
    if (kt == IC_GATE_OFF) return;  // a loop that has halted (ic_loop_steer): nothing is selected, the mask stays empty
    int k = kt > 0 ? kt : 0;
    if (a.force_io) {  // gate.py:80-84
      mask[a.sensor_idx] = 1;
      mask[a.motor_idx] = 1;
      k = k - 2 > 0 ? k - 2 : 0;
    }
    int count = 0;
    for (int p = 0; p < R; ++p) {  // utils.py:115-123
      const int i = order[p];
      if (supp[i]) continue;
      mask[i] = 1;
      ++count;
      if (count >= k) break;
      supp[i] = 1;
      for (int q = a.nbr_off[i]; q < a.nbr_off[i + 1]; ++q) supp[a.nbr[q]] = 1;
    }
  }
}

// Scores and selection by ONE workgroup per stream for small region counts: the workgroup reads
// its R rows itself (R*d*4 bytes: 10 KB at R=10, one CU is enough) and goes straight on to the
// walk.  Saves a launch per inner step where launches, not bytes, are the cost.
template <int DCH>
__device__ __forceinline__ void gate_fused_body(const ic_gate_args& a, int s, float* sc, int* order,
                                                uint8_t* supp) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int r = wave; r < a.R; r += 4) {
    const int row = s * a.R + r;
    RowFrag<DCH> h;
    load_row(h, gate_h_row(a, s, r), a.d, lane);
    float part = 0.f;
#pragma unroll
    for (int j = 0; j < DCH; ++j)
      part += (fabsf(h.v[j].x) + fabsf(h.v[j].y)) + (fabsf(h.v[j].z) + fabsf(h.v[j].w));
    const float content = wave_sum(part) / (float)a.d;
    if (lane == 0) {
      const float focus =
          (a.focus_any && a.focus_any[s] && (r == a.sensor_idx || r == a.motor_idx)) ? 1.0f : 0.0f;
      float v = a.w1 * content;
      v = v + a.w2 * a.gain_ema[r];
      v = v + a.w3 * focus;
      v = v - a.w4 * a.homeo[r];
      v = v + gate_t5(a, s);
      v = v + gate_t6(a, s);
      a.scores[row] = v;
      sc[r] = v;
    }
  }
  __syncthreads();
  gate_select_body(a, s, sc, order, supp);
}

}  // namespace
