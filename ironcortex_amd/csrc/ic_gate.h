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
// l_off / l_nbr (optional, LDS: R + 1 and nbr_off[R] entries): the adjacency staged by the whole workgroup for the
// walk on large maps (below).  nullptr: the walk reads it from memory, one lane (small maps: a handful of selections).
// sc must be 16-byte aligned with room up to the next multiple of 4 (the ranking reads it four scores at a time).
__device__ __forceinline__ void gate_select_body(const ic_gate_args& a, int s, float* sc, int* order,
                                                 uint8_t* supp, uint16_t* l_off = nullptr, uint16_t* l_nbr = nullptr) {
  const int R = a.R, R4 = (R + 3) & ~3;
  for (int i = threadIdx.x; i < R; i += blockDim.x) {
    supp[i] = 0;
    a.mask[s * R + i] = 0;
  }
  if (threadIdx.x < R4 - R) sc[R + threadIdx.x] = -INFINITY;  // (never above, never equal to a score)
  int big_deg = 0;
  if (l_off) {
    for (int i = threadIdx.x; i <= R; i += blockDim.x) {
      l_off[i] = (uint16_t)a.nbr_off[i];
      if (i < R) big_deg |= a.nbr_off[i + 1] - a.nbr_off[i] > 8;
    }
    const int n = a.nbr_off[R];
    for (int q = threadIdx.x; q < n; q += blockDim.x) l_nbr[q] = (uint16_t)a.nbr[q];
  }
  big_deg = __syncthreads_or(big_deg);
  if (l_off && R > 64) {
    // Large maps with a SYMMETRIC adjacency (checked here, in LDS; hex maps are): no ranking and no walk.  The walk of
    // utils.py:115-123 selects the lexicographically first maximal independent set in the total order (score
    // descending, index ascending), cut after max(k, 1) selections.  Membership of a region depends only on its
    // better-placed neighbours, so the set comes out of parallel rounds: an undecided region with a selected neighbour
    // is suppressed; one with no undecided better-placed neighbour is selected (two adjacent regions never both are:
    // the order is total).  Random scores settle in O(log R) rounds of one LDS pass each; a monotone ridge of scores
    // needs as many rounds as it is long, still one pass each.  The selections in the order of the walk are the
    // members by rank AMONG THE MEMBERS (a third of the map at most on a hex map): counted only if more than the
    // limit exist.  R = 512, 32 selections: 45 -> ~12 us per launch.
    int asym = 0;
    for (int i = threadIdx.x; i < R; i += blockDim.x)
      for (int q = l_off[i]; q < l_off[i + 1]; ++q) {
        const int j = l_nbr[q];
        bool back = false;
        for (int q2 = l_off[j]; q2 < l_off[j + 1]; ++q2) back |= l_nbr[q2] == i;
        asym |= !back;
      }
    if (!__syncthreads_or(asym)) {
      const int kt = a.k_total_s ? a.k_total_s[s] : a.k_total;
      if (kt == IC_GATE_OFF) return;  // (workgroup-uniform; the mask stays empty)
      uint8_t* mask = a.mask + s * R;
      int k = kt > 0 ? kt : 0;
      if (a.force_io) {  // gate.py:80-84
        if (threadIdx.x == 0) {
          mask[a.sensor_idx] = 1;
          mask[a.motor_idx] = 1;
        }
        k = k - 2 > 0 ? k - 2 : 0;
      }
      const int limit = k > 1 ? k : 1;  // (utils.py:118-120: the count is tested after the append)
      auto better = [&](int j, int i) { return sc[j] > sc[i] || (sc[j] == sc[i] && j < i); };
      __shared__ int n_mem_s;
      __shared__ __align__(16) float msc[1024 + 4];
      __shared__ __align__(16) int mid[1024 + 4];
      int undecided;
      if ((int)blockDim.x >= R && !big_deg) {
        // One region per thread, its <= 8 neighbours and which of them are better placed held in registers: a round is
        // eight independent LDS reads of neighbour states and two barriers.
        const int i = threadIdx.x;
        const bool live = i < R;
        const float sci = live ? sc[i] : 0.f;
        unsigned int nbp[4] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu}, bm = 0;
        int deg = 0;
        if (live) {
          const int o0 = l_off[i];
          deg = l_off[i + 1] - o0;
#pragma unroll
          for (int q = 0; q < 8; ++q)
            if (q < deg) {
              const unsigned int j = l_nbr[o0 + q];
              nbp[q >> 1] = (q & 1) ? ((nbp[q >> 1] & 0xFFFFu) | (j << 16)) : ((nbp[q >> 1] & 0xFFFF0000u) | j);
              bm |= (unsigned int)better((int)j, i) << q;
            }
        }
        int my = live ? 0 : 2;  // 0 undecided, 1 selected, 2 suppressed (also in supp[i], for the neighbours)
        if (threadIdx.x == 0) n_mem_s = 0;
        do {
          int ns = 3, left = 0;
          if (my == 0) {
            bool win = true, dead = false;
#pragma unroll
            for (int q = 0; q < 8; ++q)
              if (q < deg) {
                const uint8_t st = supp[(nbp[q >> 1] >> (16 * (q & 1))) & 0xFFFFu];
                dead |= st == 1;
                win &= !(st == 0 && ((bm >> q) & 1u));
              }
            ns = dead ? 2 : (win ? 1 : 0);
            left = ns == 0;
          }
          __syncthreads();  // every state of this round has been read
          if (ns == 1 || ns == 2) {
            supp[i] = (uint8_t)ns;
            my = ns;
          }
          undecided = __syncthreads_or(left);
        } while (undecided);
        if (my == 1) {  // the members in any order; each one's rank among them only if the limit cuts
          const int t = atomicAdd(&n_mem_s, 1);
          mid[t] = i;
          msc[t] = sci;
        }
        __syncthreads();
        const int M = n_mem_s;
        if (M <= limit) {
          if (my == 1) mask[i] = 1;
          return;
        }
        if (threadIdx.x < 4) {  // pad to a multiple of four: pairs that sort behind every member
          msc[M + threadIdx.x] = -INFINITY;
          mid[M + threadIdx.x] = 0x7FFFFFFF;
        }
        __syncthreads();
        if ((int)threadIdx.x < M) {  // member t by thread t (the first waves only), four members per LDS read
          const int me = mid[threadIdx.x];
          const float v = msc[threadIdx.x];
          const float4* s4 = reinterpret_cast<const float4*>(msc);
          const int4* i4 = reinterpret_cast<const int4*>(mid);
          int rank = 0;
#pragma unroll 2
          for (int u = 0; u < (M + 3) / 4; ++u) {
            const float4 f = s4[u];
            const int4 d4 = i4[u];
            rank += (f.x > v) | ((f.x == v) & (d4.x < me));
            rank += (f.y > v) | ((f.y == v) & (d4.y < me));
            rank += (f.z > v) | ((f.z == v) & (d4.z < me));
            rank += (f.w > v) | ((f.w == v) & (d4.w < me));
          }
          if (rank < limit) mask[me] = 1;
        }
        return;
      }
      do {  // supp[i]: 0 undecided, 1 selected, 2 suppressed
        uint8_t ns[4] = {3, 3, 3, 3};  // (R <= 1024, 256 threads; 3: nothing to write)
        int left = 0;
        for (int m = 0, i = threadIdx.x; i < R; ++m, i += blockDim.x) {
          if (supp[i] != 0) continue;
          bool win = true, dead = false;
          for (int q = l_off[i]; q < l_off[i + 1]; ++q) {
            const int j = l_nbr[q];
            const uint8_t st = supp[j];
            dead |= st == 1;
            win &= !(st == 0 && better(j, i));
          }
          ns[m & 3] = dead ? 2 : (win ? 1 : 0);
          left |= !dead && !win;
        }
        __syncthreads();  // every state of this round has been read
        for (int m = 0, i = threadIdx.x; i < R; ++m, i += blockDim.x)
          if (ns[m & 3] == 1 || ns[m & 3] == 2) supp[i] = ns[m & 3];
        undecided = __syncthreads_or(left);
      } while (undecided);
      // the members, in any order, then each one's rank among them
      int* n_mem = &n_mem_s;
      if (threadIdx.x == 0) *n_mem = 0;
      __syncthreads();
      for (int i = threadIdx.x; i < R; i += blockDim.x)
        if (supp[i] == 1) order[atomicAdd(n_mem, 1)] = i;
      __syncthreads();
      const int M = *n_mem;
      for (int t = threadIdx.x; t < M; t += blockDim.x) {
        const int i = order[t];
        int rank = 0;
        if (M > limit)
          for (int u = 0; u < M; ++u) rank += better(order[u], i);
        if (rank < limit) mask[i] = 1;
      }
      return;
    }
    // Otherwise: the same total order (score descending, index ascending) from a bitonic sorting network over
    // (score, index) pairs in LDS, padded to a power of two with (-inf, index >= R) pairs, which sort behind every
    // region.  R^2 comparisons on the one CU this workgroup owns were 29 us of a launch at R = 512 and 106 us at
    // R = 1024 (4 cycles per wave instruction); the network is R/2 log^2 R compare-exchanges.
    int n = 64;
    while (n < R) n <<= 1;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
      if (i >= R) sc[i] = -INFINITY;
      order[i] = i;
    }
    __syncthreads();
    for (int k2 = 2; k2 <= n; k2 <<= 1) {
      for (int j = k2 >> 1; j > 0; j >>= 1) {
        for (int t = threadIdx.x; t < n / 2; t += blockDim.x) {
          const int i = (t / j) * 2 * j + (t % j), l = i + j;
          const float ka = sc[i], kb = sc[l];
          const int ia = order[i], ib = order[l];
          const bool a_first = (ka > kb) || (ka == kb && ia < ib);
          const bool want_a_first = (i & k2) == 0;
          if (a_first != want_a_first) {
            sc[i] = kb; sc[l] = ka;
            order[i] = ib; order[l] = ia;
          }
        }
        __syncthreads();
      }
    }
  } else
  // stable descending rank by counting (four scores per LDS read, four reads in flight)
  for (int i = threadIdx.x; i < R; i += blockDim.x) {
    const float v = sc[i];
    const float4* s4 = reinterpret_cast<const float4*>(sc);
    int rank = 0;
#pragma unroll 4
    for (int j4 = 0; j4 < R4 / 4; ++j4) {
      const float4 u = s4[j4];
      const int j = 4 * j4;
      rank += (u.x > v) || (u.x == v && j < i);
      rank += (u.y > v) || (u.y == v && j + 1 < i);
      rank += (u.z > v) || (u.z == v && j + 2 < i);
      rank += (u.w > v) || (u.w == v && j + 3 < i);
    }
    order[rank] = i;
  }
  __syncthreads();
  if (l_off && !big_deg) {
    // The walk of utils.py:115-123 by ONE WAVE, 64 candidates of the order at a time: every lane holds its candidate
    // and that candidate's neighbours (<= 8, packed in registers); a ballot of the candidates no earlier selection has
    // suppressed; the first of them is selected, its neighbour list is broadcast from its lane, the lanes whose candidate
    // is that region or one of the neighbours drop out of the ballot, nine lanes mark them for the chunks to come.
    // Same selections in the same order as the one-lane walk below, no memory round trip per selection
    // (R = 512, 97 selections: 50 us of a launch -> ~5).
    if (threadIdx.x >= 64) return;
    const int lane = threadIdx.x;
    uint8_t* mask = a.mask + s * R;
    const int kt = a.k_total_s ? a.k_total_s[s] : a.k_total;
    if (kt == IC_GATE_OFF) return;
    int k = kt > 0 ? kt : 0;
    if (a.force_io) {
      if (lane == 0) {
        mask[a.sensor_idx] = 1;
        mask[a.motor_idx] = 1;
      }
      k = k - 2 > 0 ? k - 2 : 0;
    }
    int count = 0;
    for (int p0 = 0; p0 < R; p0 += 64) {
      const int il = p0 + lane < R ? order[p0 + lane] : -1;
      unsigned int nb[4] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
      if (il >= 0) {
        const int o0 = l_off[il], deg = l_off[il + 1] - o0;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const unsigned int v = q < deg ? (unsigned int)l_nbr[o0 + q] : 0xFFFFu;
          nb[q >> 1] = (q & 1) ? ((nb[q >> 1] & 0xFFFFu) | (v << 16)) : ((nb[q >> 1] & 0xFFFF0000u) | v);
        }
      }
      unsigned long long alive = __ballot(il >= 0 && supp[il] == 0);
      while (alive) {
        const int l0 = __ffsll((long long)alive) - 1;
        const int i = __builtin_amdgcn_readlane(il, l0);
        if (lane == 0) mask[i] = 1;
        ++count;
        if (count >= k) return;  // utils.py:118-120: tested after the append
        unsigned int sel[4];
#pragma unroll
        for (int w = 0; w < 4; ++w) sel[w] = (unsigned int)__builtin_amdgcn_readlane((int)nb[w], l0);
        if (lane < 8) {  // nine lanes mark the region and its neighbours ...
          const unsigned int w = lane < 2 ? sel[0] : (lane < 4 ? sel[1] : (lane < 6 ? sel[2] : sel[3]));
          const int v = (int)((w >> (16 * (lane & 1))) & 0xFFFFu);
          if (v != 0xFFFF) supp[v] = 1;
        } else if (lane == 8) {
          supp[i] = 1;
        }
        // ... and every lane looks its own candidate up again (LDS operations of one wave complete in order): one
        // LDS round trip and ~25 instructions per selection; comparing against the eight broadcast neighbours in
        // registers was ~80 instructions of a single wave, 0.3 us
        alive &= __ballot(il >= 0 && supp[il] == 0);
      }
    }
    return;
  }
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
