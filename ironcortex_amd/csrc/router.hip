// Hex-neighbour message routing (reference gate.py:251-335), forward + backward.
//
// Edge weights are read IN PLACE from the [nW,d,d] arena through widx[e]; the reference
// torch.stack()s a fresh copy of every active edge matrix on each call (gate.py:272).
// Default path: one launch, workgroup = (row slice, destination region), incoming edges are
// walked in the reference's index_add_ order.  Precision-routed path: edge GEMVs (messages,
// keys, queries) then a wave-per-destination softmax combine.
#include "ic_common.h"
#include "ic_gate.h"
#include "ic_route.h"

namespace {

constexpr int kRowsPerBlock = 8;
constexpr int kSharedGemmMinCols = 8192;  // see shared_as_gemm

// ---------------------------------------------------------------- default forward
// Workgroup = (8-row slice, destination r); incoming edges are walked in index order (the
// reference's index_add_ order).  Low register use on purpose: most workgroups of a launch find no
// active edge and must come and go quickly.  (Tried and measured slower on MI355X at R=512: all
// <= 8 edges' rows in registers at once, 64-113 us against 28 us; weights re-read per batch of 8
// streams, 160 us against 76 us at S=8.)
// An edge's two weight rows are loaded ONCE per 64 streams and only the streams
// the edge is active in are visited (one ballot per edge); the per-stream sums live in LDS
// (each wave owns its two rows of every stream: no conflicts, no barrier until the write-out),
// accumulated edge by edge in index order.  R=512, k=32, d=256: 26 us at S=1 (1.9 TB/s), 76 us at
// S=8 (4.1 TB/s), 278 us at S=64 (2.7 TB/s).
template <int DCH, int PP = 1>
__device__ __forceinline__ void router_fwd_ms_body(const ic_router_args& a, const int r) {
  constexpr int RB = kRowsPerBlock * PP, NR = 2 * PP;  // rows per workgroup / per wave (PP > 1: large maps, see k_router_edge_fwd)
  __shared__ float acc[64][RB];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int d = a.d;
  const int row0 = blockIdx.x * RB + wave * NR;
  const bool rows_ok = row0 < d;  // d % 4 == 0: a wave's row pairs are valid together
  const int e0 = a.dst_off[r], e1 = a.dst_off[r + 1];
  const size_t dd = (size_t)d * d;
  for (int sc = 0; sc < a.S; sc += 64) {
    const int n = a.S - sc < 64 ? a.S - sc : 64;
    if (lane < n) {
#pragma unroll
      for (int i = 0; i < NR; ++i) acc[lane][wave * NR + i] = 0.f;
    }
    for (int e = e0; e < e1 && rows_ok; ++e) {
      const int src = a.src[e];
      unsigned long long act = __ballot(lane < n && prev_on(a, sc + lane, src));
      if (!act) continue;
      const float* W = a.W + (size_t)a.widx[e] * dd + (size_t)row0 * d;
      RowFrag<DCH> w[NR], h, hn;
#pragma unroll
      for (int i = 0; i < NR; ++i) {
        if (row0 + i < d) load_row(w[i], W + (size_t)i * d, d, lane);
        else zero_frag(w[i]);
      }
      const float bias = a.ebias[e];
      int si = __ffsll((long long)act) - 1;
      act &= act - 1;
      load_row(h, h_row(a, sc + si, src), d, lane);
      while (si >= 0) {
        const int sn = act ? __ffsll((long long)act) - 1 : -1;
        act &= act - 1;
        if (sn >= 0) load_row(hn, h_row(a, sc + sn, src), d, lane);  // next stream's sender row in flight
#pragma unroll
        for (int p = 0; p < PP; ++p) {
          const float m0 = wave_sum(dot_frag(w[2 * p], h)), m1 = wave_sum(dot_frag(w[2 * p + 1], h));
          if (lane == 0) {
            acc[si][wave * NR + 2 * p] += m0 * bias;  // gate.py:290,324 (w == 1)
            acc[si][wave * NR + 2 * p + 1] += m1 * bias;
          }
        }
        h = hn;
        si = sn;
      }
    }
    __syncthreads();
    // write-out: thread t -> (stream t / RB, row t % RB), 256 / RB streams per pass
    for (int base = 0; base < n; base += 256 / RB) {
      const int s_l = base + (int)threadIdx.x / RB, rr = (int)threadIdx.x % RB;
      const int row = blockIdx.x * RB + rr;
      if (s_l < n && row < d) a.M[((size_t)(sc + s_l) * a.R + r) * d + row] = acc[s_l][rr];
    }
    __syncthreads();
  }
}

template <int DCH, int PP>
__global__ __launch_bounds__(256) void k_router_fwd_ms(ic_router_args a) {
  router_fwd_ms_body<DCH, PP>(a, blockIdx.y);
}

// ---------------------------------------------------------------- precision: edge GEMVs
// y < E: message + key of edge e;  y >= E: query of region y-E.
// PP: row pairs per wave, i.e. 8 * PP rows of every matrix per workgroup.  1 where launches are the cost (the loop's
// k_front); 2 or 4 on large maps, where the (slice, edge) grid is tens of thousands of workgroups of which a fifth find
// an active sender and each asked for only 4 KB (110 k workgroups at R=512: the dispatch was the launch).  Every row's
// dot products are the same wave reductions whatever PP: bit-identical results.
template <int DCH, int PP = 1>
__device__ __forceinline__ void router_edge_body(const ic_router_args& a, const int y) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int d = a.d;
  constexpr int NR = 2 * PP;  // rows per wave
  const int row0 = blockIdx.x * (kRowsPerBlock * PP) + wave * NR;
  if (row0 >= d) return;
  const size_t dd = (size_t)d * d;
  // Streams z, z + nz, ... of this grid slice are tested 64 at a time (one mask byte per lane, one ballot) and
  // only the active ones visited, their sender rows requested two at a time.  (Asking for every stream's mask
  // byte in turn cost a dependent memory round trip per stream: 2.2 ms at R=512, k=32, S=64.)
  const int nz = gridDim.z;
  if (y < a.E) {
    const int e = y, src = a.src[e];
    RowFrag<DCH> w[NR], k[NR];
    float bias = 0.f;
    bool loaded = false;
    for (int sc = blockIdx.z; sc < a.S; sc += 64 * nz) {
      const int sl = sc + lane * nz;
      unsigned long long act = __ballot(sl < a.S && prev_on(a, sl, src));
      if (!act) continue;
      if (!loaded) {  // (wave-uniform) idle workgroups never ask for their rows
        const float* W = a.W + (size_t)a.widx[e] * dd + (size_t)row0 * d;
        const float* K = a.key_W + (size_t)a.widx[e] * dd + (size_t)row0 * d;
#pragma unroll
        for (int i = 0; i < NR; ++i) {
          if (row0 + i < d) {
            load_row(w[i], W + (size_t)i * d, d, lane);
            load_row(k[i], K + (size_t)i * d, d, lane);
          } else {
            zero_frag(w[i]);
            zero_frag(k[i]);
          }
        }
        bias = a.ebias[e];
        loaded = true;
      }
      while (act) {
        int sv[2];
        RowFrag<DCH> h[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          sv[u] = -1;
          if (act) {
            sv[u] = sc + (__ffsll((long long)act) - 1) * nz;
            act &= act - 1;
            load_row(h[u], h_row(a, sv[u], src), d, lane);
          }
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          if (sv[u] < 0) continue;  // wave-uniform
#pragma unroll
          for (int p = 0; p < PP; ++p) {
            const float m0 = wave_sum(dot_frag(w[2 * p], h[u])), m1 = wave_sum(dot_frag(w[2 * p + 1], h[u]));
            const float q0 = wave_sum(dot_frag(k[2 * p], h[u])), q1 = wave_sum(dot_frag(k[2 * p + 1], h[u]));
            if (lane < 2 && row0 + 2 * p + lane < d) {
              const size_t idx = ((size_t)sv[u] * a.E + e) * d + row0 + 2 * p + lane;
              a.msg[idx] = (lane ? m1 : m0) * bias;
              a.keyv[idx] = lane ? q1 : q0;
            }
          }
        }
      }
    }
  } else {
    const int r = y - a.E;
    const int e0 = a.dst_off[r], e1 = a.dst_off[r + 1];
    const float* Q = a.query_W + (size_t)r * dd + (size_t)row0 * d;
    RowFrag<DCH> q[NR];
    bool loaded = false;
    for (int sc = blockIdx.z; sc < a.S; sc += 64 * nz) {
      const int sl = sc + lane * nz;
      bool on = false;  // this lane's stream: any incoming edge with an active sender
      if (sl < a.S)
        for (int e = e0; e < e1; ++e) on |= prev_on(a, sl, a.src[e]);
      unsigned long long act = __ballot(on);
      if (!act) continue;
      if (!loaded) {
#pragma unroll
        for (int i = 0; i < NR; ++i) {
          if (row0 + i < d) load_row(q[i], Q + (size_t)i * d, d, lane);
          else zero_frag(q[i]);
        }
        loaded = true;
      }
      while (act) {
        int sv[2];
        RowFrag<DCH> h[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          sv[u] = -1;
          if (act) {
            sv[u] = sc + (__ffsll((long long)act) - 1) * nz;
            act &= act - 1;
            load_row(h[u], h_row(a, sv[u], r), d, lane);
          }
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          if (sv[u] < 0) continue;
#pragma unroll
          for (int p = 0; p < PP; ++p) {
            const float v0 = wave_sum(dot_frag(q[2 * p], h[u])), v1 = wave_sum(dot_frag(q[2 * p + 1], h[u]));
            if (lane < 2 && row0 + 2 * p + lane < d) a.q[((size_t)sv[u] * a.R + r) * d + row0 + 2 * p + lane] = lane ? v1 : v0;
          }
        }
      }
    }
  }
}

// y0: first (edge | region) row of the grid -- a.E for a launch over the queries alone
template <int DCH, int PP>
__global__ __launch_bounds__(256) void k_router_edge_fwd(ic_router_args a, int y0) {
  router_edge_body<DCH, PP>(a, (int)blockIdx.y + y0);
}

// ---------------------------------------------------------------- gate + router in one launch
// The gate (scores + selection of the NEW mask, gate.py:54-87) and the router (messages from the
// PREVIOUS states and mask) read the same inputs and do not depend on each other: the first
// g.S workgroup rows of the grid run the gate (one workgroup per stream), the rest the router.
template <int DCH, bool PRECISION>
__global__ __launch_bounds__(256) void k_front(ic_router_args a, ic_gate_args g) {
  if ((int)blockIdx.y < g.S) {
    if (blockIdx.x != 0 || blockIdx.z != 0) return;
    __shared__ __align__(16) float sc[1024];
    __shared__ int order[1024];
    __shared__ uint8_t supp[1024];
    gate_fused_body<DCH>(g, blockIdx.y, sc, order, supp);
    return;
  }
  if (PRECISION)
    router_edge_body<DCH>(a, (int)blockIdx.y - g.S);
  else
    router_fwd_ms_body<DCH>(a, (int)blockIdx.y - g.S);
}

// ---------------------------------------------------------------- precision: combine (wave per (s,dst))
// d > 256: edges walked one at a time (the all-at-once variant below would spill).
template <int DCH>
__global__ __launch_bounds__(256) void k_router_combine_serial(ic_router_args a) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int row = blockIdx.x * 4 + wave;
  if (row >= a.S * a.R) return;
  const int d = a.d, s = row / a.R, r = row % a.R;
  const int e0 = a.dst_off[r], e1 = a.dst_off[r + 1];
  const float sqrt_d = sqrtf((float)d);
  RowFrag<DCH> hd, q, m, k, P, acc;
  zero_frag(acc);
  float logw[8];  // hex degree <= 6; checked on the host
  float mx = -INFINITY;
  bool any = false;
  for (int e = e0; e < e1; ++e) {
    const int i = e - e0;
    logw[i] = -INFINITY;
    if (!prev_on(a, s, a.src[e])) continue;
    if (!any) {
      load_row(hd, h_row(a, s, r), d, lane);
      load_row(q, a.q + (size_t)row * d, d, lane);
      any = true;
    }
    const size_t eo = ((size_t)s * a.E + e) * d;
    load_row(m, a.msg + eo, d, lane);
    load_row(k, a.keyv + eo, d, lane);
    load_row(P, a.raw_P + (size_t)a.widx[e] * d, d, lane);
    const float score = wave_sum(dot_frag(q, k)) / sqrt_d;  // gate.py:308
    float part = 0.f;
#pragma unroll
    for (int j = 0; j < DCH; ++j) {
      const int c = 4 * lane + 256 * j;
      if (c < d) {
        float t;
        t = m.v[j].x - hd.v[j].x; part += (t * t) * ic_softplus(P.v[j].x);
        t = m.v[j].y - hd.v[j].y; part += (t * t) * ic_softplus(P.v[j].y);
        t = m.v[j].z - hd.v[j].z; part += (t * t) * ic_softplus(P.v[j].z);
        t = m.v[j].w - hd.v[j].w; part += (t * t) * ic_softplus(P.v[j].w);
      }
    }
    const float mah = fminf(wave_sum(part), IC_MAX_EXP / 0.5f);  // gate.py:310
    logw[i] = score - 0.5f * mah;
    mx = fmaxf(mx, logw[i]);
  }
  if (any) {
    float esum = 0.f;
    for (int e = e0; e < e1; ++e)
      if (logw[e - e0] != -INFINITY) esum += expf(logw[e - e0] - mx);
    const float logZ = mx + logf(fmaxf(esum, IC_EPS_DIV));  // gate.py:316
    for (int e = e0; e < e1; ++e) {
      const int i = e - e0;
      float w = 0.f;
      if (logw[i] != -INFINITY) {
        w = expf(logw[i] - logZ);
        load_row(m, a.msg + ((size_t)s * a.E + e) * d, d, lane);
#pragma unroll
        for (int j = 0; j < DCH; ++j) {
          acc.v[j].x += w * m.v[j].x; acc.v[j].y += w * m.v[j].y;
          acc.v[j].z += w * m.v[j].z; acc.v[j].w += w * m.v[j].w;
        }
      }
      if (lane == 0) a.wgt[(size_t)s * a.E + e] = w;
    }
  } else if (lane == 0) {
    for (int e = e0; e < e1; ++e) a.wgt[(size_t)s * a.E + e] = 0.f;
  }
  store_row(acc, a.M + (size_t)row * d, d, lane);
}

// d <= 256: wave per (stream, destination); the body lives in ic_route.h (shared with the routed
// region kernel of ic_loop_step_fwd).
template <int DCH>
__global__ __launch_bounds__(256) void k_router_combine(ic_router_args a) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int row = blockIdx.x * 4 + wave;
  if (row >= a.S * a.R) return;
  RowFrag<DCH> acc;
  combine_row<DCH>(a, row / a.R, row % a.R, lane, true, acc);
}

// ---------------------------------------------------------------- backward: stream addressing
// sk: stream s lives at stream index s*sk of the per-stream arrays; in_slot: where the stream's
// input block is (negative: the stream takes no part in this call).
__device__ __forceinline__ int stream_step(const ic_router_bwd_args& b) { return b.sk > 1 ? b.sk : 1; }
__device__ __forceinline__ bool stream_live(const ic_router_bwd_args& b, int s) {
  return !b.in_slot || b.in_slot[s] >= 0;
}
__device__ __forceinline__ const float* h_row(const ic_router_bwd_args& b, int s, int r) {
  if (b.in_slot) return b.f.H + ((size_t)b.in_slot[s] * b.f.R + r) * b.f.d;
  return h_row(b.f, s, r);
}
__device__ __forceinline__ bool prev_on(const ic_router_bwd_args& b, int s, int r) {
  if (b.in_slot) return b.in_slot[s] >= 0 && b.f.mask_prev[(size_t)b.in_slot[s] * b.f.R + r] != 0;
  return prev_on(b.f, s, r);
}

// ---------------------------------------------------------------- backward: per-destination stage
// Produces, for every active edge, the gradient w.r.t. the biased message (tape_gmsg), and in
// precision mode the key/query gradients and the direct d/dH_dst term.
// d > 256: the edges are walked one at a time (the all-at-once variant below would spill).
template <int DCH>
__global__ __launch_bounds__(256) void k_router_comb_bwd_serial(ic_router_bwd_args b, int fold_bias) {
  const ic_router_args& a = b.f;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int lrow = blockIdx.x * 4 + wave;
  if (lrow >= a.S * a.R) return;
  const int d = a.d, s = lrow / a.R, r = lrow % a.R;
  const int vs = s * stream_step(b), row = vs * a.R + r;
  const int e0 = a.dst_off[r], e1 = a.dst_off[r + 1];
  const bool prec = a.flags & IC_RF_PRECISION;
  if (!stream_live(b, s)) {
    if (lane == 0) {
      for (int e = e0; e < e1; ++e) b.tape_eact[(size_t)vs * a.E + e] = 0;
      if (prec) b.tape_qact[row] = 0;
    }
    return;
  }
  RowFrag<DCH> G, t;
  load_row(G, b.gM + (size_t)row * d, d, lane);
  load_row(t, h_row(b, s, r), d, lane);
  store_row(t, b.tape_H + (size_t)row * d, d, lane);
  bool any = false;
  for (int e = e0; e < e1; ++e) {
    const bool act = prev_on(b, s, a.src[e]);
    any |= act;
    if (lane == 0) b.tape_eact[(size_t)vs * a.E + e] = act ? 1 : 0;
    if (act && !prec) {
      if (fold_bias) {  // (MFMA input gradient: nothing folds the edge bias into the tape entry later)
        RowFrag<DCH> Gb = G;
        scale_frag(Gb, a.ebias[e]);
        store_row(Gb, b.tape_gmsg + ((size_t)vs * a.E + e) * d, d, lane);
      } else {
        store_row(G, b.tape_gmsg + ((size_t)vs * a.E + e) * d, d, lane);
      }
    }
  }
  if (!prec) return;
  if (lane == 0) b.tape_qact[row] = any ? 1 : 0;
  RowFrag<DCH> gHd, gq;
  zero_frag(gHd);
  zero_frag(gq);
  if (any) {
    RowFrag<DCH> hd = t, q, m, k, P;
    load_row(q, a.q + (size_t)row * d, d, lane);
    const float sqrt_d = sqrtf((float)d);
    // pass 1: g_w and the softmax inner product
    float gw[8], inner = 0.f;
    for (int e = e0; e < e1; ++e) {
      const int i = e - e0;
      gw[i] = 0.f;
      const float w = a.wgt[(size_t)vs * a.E + e];
      if (!prev_on(b, s, a.src[e])) continue;
      load_row(m, a.msg + ((size_t)vs * a.E + e) * d, d, lane);
      gw[i] = wave_sum(dot_frag(G, m));
      inner += w * gw[i];
    }
    for (int e = e0; e < e1; ++e) {
      const int i = e - e0;
      if (!prev_on(b, s, a.src[e])) continue;
      const size_t eo = ((size_t)vs * a.E + e) * d;
      const float w = a.wgt[(size_t)vs * a.E + e];
      const float g_lw = w * (gw[i] - inner);
      load_row(m, a.msg + eo, d, lane);
      load_row(k, a.keyv + eo, d, lane);
      load_row(P, a.raw_P + (size_t)a.widx[e] * d, d, lane);
      // recompute the un-clamped mahalanobis term
      float part = 0.f;
      RowFrag<DCH> res, sp;
#pragma unroll
      for (int j = 0; j < DCH; ++j) {
        res.v[j] = make_float4(m.v[j].x - hd.v[j].x, m.v[j].y - hd.v[j].y, m.v[j].z - hd.v[j].z,
                               m.v[j].w - hd.v[j].w);
        sp.v[j] = make_float4(ic_softplus(P.v[j].x), ic_softplus(P.v[j].y), ic_softplus(P.v[j].z),
                              ic_softplus(P.v[j].w));
        const int c = 4 * lane + 256 * j;
        if (c < d)
          part += (res.v[j].x * res.v[j].x) * sp.v[j].x + (res.v[j].y * res.v[j].y) * sp.v[j].y +
                  (res.v[j].z * res.v[j].z) * sp.v[j].z + (res.v[j].w * res.v[j].w) * sp.v[j].w;
      }
      const float mahraw = wave_sum(part);
      const float g_mah = (mahraw <= IC_MAX_EXP / 0.5f) ? -0.5f * g_lw : 0.f;
      const float g_sc = g_lw / sqrt_d;
      RowFrag<DCH> gm, gk, gP;
#pragma unroll
      for (int j = 0; j < DCH; ++j) {
        float4 gr = make_float4(g_mah * 2.f * res.v[j].x * sp.v[j].x, g_mah * 2.f * res.v[j].y * sp.v[j].y,
                                g_mah * 2.f * res.v[j].z * sp.v[j].z, g_mah * 2.f * res.v[j].w * sp.v[j].w);
        gm.v[j] = make_float4(w * G.v[j].x + gr.x, w * G.v[j].y + gr.y, w * G.v[j].z + gr.z,
                              w * G.v[j].w + gr.w);
        gHd.v[j].x -= gr.x; gHd.v[j].y -= gr.y; gHd.v[j].z -= gr.z; gHd.v[j].w -= gr.w;
        gk.v[j] = make_float4(g_sc * q.v[j].x, g_sc * q.v[j].y, g_sc * q.v[j].z, g_sc * q.v[j].w);
        gq.v[j].x += g_sc * k.v[j].x; gq.v[j].y += g_sc * k.v[j].y;
        gq.v[j].z += g_sc * k.v[j].z; gq.v[j].w += g_sc * k.v[j].w;
        gP.v[j] = make_float4(g_mah * res.v[j].x * res.v[j].x * ic_softplus_grad(P.v[j].x),
                              g_mah * res.v[j].y * res.v[j].y * ic_softplus_grad(P.v[j].y),
                              g_mah * res.v[j].z * res.v[j].z * ic_softplus_grad(P.v[j].z),
                              g_mah * res.v[j].w * res.v[j].w * ic_softplus_grad(P.v[j].w));
      }
      store_row(gm, b.tape_gmsg + eo, d, lane);
      store_row(gk, b.tape_gkey + eo, d, lane);
      store_row(gP, b.g_rawP_rows + eo, d, lane);
    }
  }
  store_row(gHd, b.gHd + (size_t)row * d, d, lane);
  store_row(gq, b.tape_gq + (size_t)row * d, d, lane);
}

// d <= 256: every operand of a destination's active edges is requested up front.
template <int DCH>
__global__ __launch_bounds__(256) void k_router_comb_bwd(ic_router_bwd_args b, int fold_bias) {
  const ic_router_args& a = b.f;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int lrow = blockIdx.x * 4 + wave;
  if (lrow >= a.S * a.R) return;
  const int d = a.d, s = lrow / a.R, r = lrow % a.R;
  const int vs = s * stream_step(b), row = vs * a.R + r;
  const int e0 = a.dst_off[r], deg = a.dst_off[r + 1] - e0;  // <= 8, checked on the host
  const bool prec = a.flags & IC_RF_PRECISION;
  if (!stream_live(b, s)) {
    if (lane == 0) {
      for (int i = 0; i < deg; ++i) b.tape_eact[(size_t)vs * a.E + e0 + i] = 0;
      if (prec) b.tape_qact[row] = 0;
    }
    return;
  }
  RowFrag<DCH> G, t;
  load_row(G, b.gM + (size_t)row * d, d, lane);
  load_row(t, h_row(b, s, r), d, lane);
  // activity of the incoming edges: wave-uniform loads, all requested together
  bool act[8];
  bool any = false;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    act[i] = i < deg && prev_on(b, s, a.src[e0 + (i < deg ? i : 0)]);
    any |= act[i];
  }
  store_row(t, b.tape_H + (size_t)row * d, d, lane);
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    if (i < deg) {
      const int e = e0 + i;
      if (lane == 0) b.tape_eact[(size_t)vs * a.E + e] = act[i] ? 1 : 0;
      if (act[i] && !prec) {
        if (fold_bias) {
          RowFrag<DCH> Gb = G;
          scale_frag(Gb, a.ebias[e]);
          store_row(Gb, b.tape_gmsg + ((size_t)vs * a.E + e) * d, d, lane);
        } else {
          store_row(G, b.tape_gmsg + ((size_t)vs * a.E + e) * d, d, lane);
        }
      }
    }
  }
  if (!prec) return;
  if (lane == 0) b.tape_qact[row] = any ? 1 : 0;
  RowFrag<DCH> gHd, gq;
  zero_frag(gHd);
  zero_frag(gq);
  if (any) {
    RowFrag<DCH> hd = t, q, m[8], k[8], P[8];
    load_row(q, a.q + (size_t)row * d, d, lane);
    float w[8];
    // every operand of every active edge is requested before the first reduction waits
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      w[i] = 0.f;
      if (act[i]) {
        const int e = e0 + i;
        const size_t eo = ((size_t)vs * a.E + e) * d;
        w[i] = a.wgt[(size_t)vs * a.E + e];
        load_row(m[i], a.msg + eo, d, lane);
        load_row(k[i], a.keyv + eo, d, lane);
        load_row(P[i], a.raw_P + (size_t)a.widx[e] * d, d, lane);
      }
    }
    const float sqrt_d = sqrtf((float)d);
    // pass 1: g_w and the softmax inner product
    float gw[8], inner = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      gw[i] = 0.f;
      if (act[i]) {
        gw[i] = wave_sum(dot_frag(G, m[i]));
        inner += w[i] * gw[i];
      }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (!act[i]) continue;
      const size_t eo = ((size_t)vs * a.E + e0 + i) * d;
      const float g_lw = w[i] * (gw[i] - inner);
      // recompute the un-clamped mahalanobis term
      float part = 0.f;
      RowFrag<DCH> res, sp;
#pragma unroll
      for (int j = 0; j < DCH; ++j) {
        res.v[j] = make_float4(m[i].v[j].x - hd.v[j].x, m[i].v[j].y - hd.v[j].y, m[i].v[j].z - hd.v[j].z,
                               m[i].v[j].w - hd.v[j].w);
        sp.v[j] = make_float4(ic_softplus(P[i].v[j].x), ic_softplus(P[i].v[j].y), ic_softplus(P[i].v[j].z),
                              ic_softplus(P[i].v[j].w));
        const int c = 4 * lane + 256 * j;
        if (c < d)
          part += (res.v[j].x * res.v[j].x) * sp.v[j].x + (res.v[j].y * res.v[j].y) * sp.v[j].y +
                  (res.v[j].z * res.v[j].z) * sp.v[j].z + (res.v[j].w * res.v[j].w) * sp.v[j].w;
      }
      const float mahraw = wave_sum(part);
      const float g_mah = (mahraw <= IC_MAX_EXP / 0.5f) ? -0.5f * g_lw : 0.f;
      const float g_sc = g_lw / sqrt_d;
      RowFrag<DCH> gm, gk, gP;
#pragma unroll
      for (int j = 0; j < DCH; ++j) {
        float4 gr = make_float4(g_mah * 2.f * res.v[j].x * sp.v[j].x, g_mah * 2.f * res.v[j].y * sp.v[j].y,
                                g_mah * 2.f * res.v[j].z * sp.v[j].z, g_mah * 2.f * res.v[j].w * sp.v[j].w);
        gm.v[j] = make_float4(w[i] * G.v[j].x + gr.x, w[i] * G.v[j].y + gr.y, w[i] * G.v[j].z + gr.z,
                              w[i] * G.v[j].w + gr.w);
        gHd.v[j].x -= gr.x; gHd.v[j].y -= gr.y; gHd.v[j].z -= gr.z; gHd.v[j].w -= gr.w;
        gk.v[j] = make_float4(g_sc * q.v[j].x, g_sc * q.v[j].y, g_sc * q.v[j].z, g_sc * q.v[j].w);
        gq.v[j].x += g_sc * k[i].v[j].x; gq.v[j].y += g_sc * k[i].v[j].y;
        gq.v[j].z += g_sc * k[i].v[j].z; gq.v[j].w += g_sc * k[i].v[j].w;
        gP.v[j] = make_float4(g_mah * res.v[j].x * res.v[j].x * ic_softplus_grad(P[i].v[j].x),
                              g_mah * res.v[j].y * res.v[j].y * ic_softplus_grad(P[i].v[j].y),
                              g_mah * res.v[j].z * res.v[j].z * ic_softplus_grad(P[i].v[j].z),
                              g_mah * res.v[j].w * res.v[j].w * ic_softplus_grad(P[i].v[j].w));
      }
      store_row(gm, b.tape_gmsg + eo, d, lane);
      store_row(gk, b.tape_gkey + eo, d, lane);
      store_row(gP, b.g_rawP_rows + eo, d, lane);
    }
  }
  store_row(gHd, b.gHd + (size_t)row * d, d, lane);
  store_row(gq, b.tape_gq + (size_t)row * d, d, lane);
}

// ---------------------------------------------------------------- backward: transposed GEMVs
// blockIdx.y < E: W_e^T (ebias*gmsg) + Key_e^T gkey -> slab;  >= E: Q_r^T gq -> slab_q.
// Which (stream, edge | query) rows carry a gradient is read from the activity bytes k_router_comb_bwd has just
// written (tape_eact == the sender's previous mask under every stream addressing, tape_qact): ONE load that depends
// on nothing but the kernel arguments, where src[e] -> in_slot[s] -> mask was a chain of three -- most workgroups
// of a large map only run this test.
template <int DCH, int RPW>
__global__ __launch_bounds__(256) void k_router_mat_bwd(ic_router_bwd_args b) {
  const ic_router_args& a = b.f;
  __shared__ float red[4][DCH * 256 + 1];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int d = a.d;
  const int row0 = blockIdx.x * (4 * RPW) + wave * RPW;
  const size_t dd = (size_t)d * d;
  const bool prec = a.flags & IC_RF_PRECISION;
  const int y = blockIdx.y;
  const bool is_q = y >= a.E;
  const int e = y, r = y - a.E;
  const int src = is_q ? r : a.src[e];
  // block-uniform activity test
  const int sk = stream_step(b);
  bool any = false;  // streams are split over gridDim.z as in the region backward
  for (int sc = blockIdx.z; sc < a.S && !any; sc += 64 * gridDim.z) {
    const int s = sc + lane * gridDim.z;
    any = __ballot(s < a.S && (is_q ? b.tape_qact[(size_t)s * sk * a.R + r] : b.tape_eact[(size_t)s * sk * a.E + e]) != 0) != 0;
  }
  if (!any) return;
  const float* W = is_q ? a.query_W + (size_t)r * dd : a.W + (size_t)a.widx[e] * dd;
  const float* K = (!is_q && prec) ? a.key_W + (size_t)a.widx[e] * dd : nullptr;
  RowFrag<DCH> w[RPW], kf[RPW];
#pragma unroll
  for (int i = 0; i < RPW; ++i) {
    if (row0 + i < d) {
      load_row(w[i], W + (size_t)(row0 + i) * d, d, lane);
      if (K) load_row(kf[i], K + (size_t)(row0 + i) * d, d, lane);
      else zero_frag(kf[i]);
    } else {
      zero_frag(w[i]);
      zero_frag(kf[i]);
    }
  }
  const float bias = is_q ? 1.0f : a.ebias[e];
  const int SE = a.S * sk * a.E, SR = a.S * sk * a.R;
  for (int s = blockIdx.z; s < a.S; s += gridDim.z) {
    const int vs = s * sk;
    const bool act = (is_q ? b.tape_qact[(size_t)vs * a.R + r] : b.tape_eact[(size_t)vs * a.E + e]) != 0;
    if (!act) continue;  // block-uniform
    RowFrag<DCH> h, acc;
    zero_frag(acc);
    float part_b = 0.f;
    float* gvec = is_q ? b.tape_gq + ((size_t)vs * a.R + r) * d : b.tape_gmsg + ((size_t)vs * a.E + e) * d;
    const float* gkey = K ? b.tape_gkey + ((size_t)vs * a.E + e) * d : nullptr;
    if (!is_q) load_row(h, h_row(b, s, src), d, lane);
#pragma unroll
    for (int i = 0; i < RPW; ++i) {
      const int ch = row0 + i;
      if (ch < d) {  // wave-uniform
        const float g = gvec[ch];
        float gi = g;
        if (!is_q) {
          const float raw = wave_sum(dot_frag(w[i], h));
          part_b += g * raw;  // d/d ebias
          gi = g * bias;
          if (lane == 0) gvec[ch] = gi;  // fold ebias into the tape entry
        }
        axpy_frag(acc, gi, w[i]);
        if (gkey) axpy_frag(acc, gkey[ch], kf[i]);
      }
    }
#pragma unroll
    for (int j = 0; j < DCH; ++j) {
      float* p = &red[wave][256 * j + 4 * lane];
      p[0] = acc.v[j].x; p[1] = acc.v[j].y; p[2] = acc.v[j].z; p[3] = acc.v[j].w;
    }
    if (lane == 0) red[wave][DCH * 256] = part_b;
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
      if (is_q) {
        store_row(tot, b.slab_q + ((size_t)blockIdx.x * SR + vs * a.R + r) * d, d, lane);
      } else {
        store_row(tot, b.slab + ((size_t)blockIdx.x * SE + vs * a.E + e) * d, d, lane);
        if (lane == 0) {
          const int q = DCH * 256;
          b.g_ebias_part[(size_t)blockIdx.x * SE + vs * a.E + e] =
              (red[0][q] + red[1][q]) + (red[2][q] + red[3][q]);
        }
      }
    }
    __syncthreads();
  }
}

// The same products for MANY streams per edge (S / gridDim.z >= 32): the streams are tested 64 at a time with
// one ballot (k_router_mat_bwd asks for every stream's sender mask byte in turn: 64 dependent round trips
// per workgroup at S = 64), the active ones go through in batches of NSB -- operands of the whole batch
// requested together, one LDS hand-over (two barriers) per batch, wave u summing stream u's partials.
// Same sums as k_router_mat_bwd up to fp32 rounding (the edge bias is applied to the finished partial row).
__device__ __forceinline__ float lane_bcast(float v, int l) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l));
}

template <int DCH, int RPW, int NSB>
__global__ __launch_bounds__(256) void k_router_matT_bwd(ic_router_bwd_args b) {
  const ic_router_args& a = b.f;
  __shared__ float red[4][NSB][DCH * 256];
  __shared__ float red_b[4][NSB];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int d = a.d;
  const int row0 = blockIdx.x * (4 * RPW) + wave * RPW;
  const size_t dd = (size_t)d * d;
  const bool prec = a.flags & IC_RF_PRECISION;
  const int y = blockIdx.y;
  const bool is_q = y >= a.E;
  const int e = y, r = y - a.E;
  const int src = is_q ? r : a.src[e];
  const int sk = stream_step(b), nz = gridDim.z;
  const float* W = is_q ? a.query_W + (size_t)r * dd : a.W + (size_t)a.widx[e] * dd;
  const float* K = (!is_q && prec) ? a.key_W + (size_t)a.widx[e] * dd : nullptr;
  const float bias = is_q ? 1.0f : a.ebias[e];
  const int SE = a.S * sk * a.E, SR = a.S * sk * a.R;
  const int ch = row0 + (lane < RPW ? lane : 0);
  const bool gvalid = lane < RPW && ch < d;
  RowFrag<DCH> w[RPW], kf[RPW];
  bool loaded = false;
  for (int sc = blockIdx.z; sc < a.S; sc += 64 * nz) {
    const int sl = sc + lane * nz;
    unsigned long long act =
        __ballot(sl < a.S && (is_q ? b.tape_qact[(size_t)sl * sk * a.R + r] : b.tape_eact[(size_t)sl * sk * a.E + e]) != 0);
    if (!act) continue;
    if (!loaded) {  // (wave-uniform)
#pragma unroll
      for (int i = 0; i < RPW; ++i) {
        if (row0 + i < d) {
          load_row(w[i], W + (size_t)(row0 + i) * d, d, lane);
          if (K) load_row(kf[i], K + (size_t)(row0 + i) * d, d, lane);
          else zero_frag(kf[i]);
        } else {
          zero_frag(w[i]);
          zero_frag(kf[i]);
        }
      }
      loaded = true;
    }
    while (act) {
      int sv[NSB];
      float g[NSB], gk[NSB];
      RowFrag<DCH> h[NSB];
#pragma unroll
      for (int u = 0; u < NSB; ++u) {
        sv[u] = -1;
        g[u] = gk[u] = 0.f;
        zero_frag(h[u]);
        if (act) {
          const int s = sc + (__ffsll((long long)act) - 1) * nz;
          act &= act - 1;
          sv[u] = s;
          const int vs = s * sk;
          const float* gvec = is_q ? b.tape_gq + ((size_t)vs * a.R + r) * d : b.tape_gmsg + ((size_t)vs * a.E + e) * d;
          if (gvalid) {
            g[u] = gvec[ch];
            if (K) gk[u] = b.tape_gkey[((size_t)vs * a.E + e) * d + ch];
          }
          if (!is_q) load_row(h[u], h_row(b, s, src), d, lane);
        }
      }
#pragma unroll
      for (int u = 0; u < NSB; ++u) {
        if (sv[u] < 0) continue;  // wave-uniform
        // W^T g without the bias first: this workgroup's share of d/d ebias = sum_i g_i (W h)_i is the dot product of
        // that partial row with h -- ONE wave reduction per stream instead of one per weight row
        RowFrag<DCH> acc, acck;
        zero_frag(acc);
        zero_frag(acck);
#pragma unroll
        for (int i = 0; i < RPW; ++i) {
          if (row0 + i < d) {  // wave-uniform
            axpy_frag(acc, lane_bcast(g[u], i), w[i]);
            if (K) axpy_frag(acck, lane_bcast(gk[u], i), kf[i]);
          }
        }
        float part_b = 0.f;
        if (!is_q) {
          part_b = wave_sum(dot_frag(acc, h[u]));
#pragma unroll
          for (int j = 0; j < DCH; ++j) {
            acc.v[j].x = fmaf(bias, acc.v[j].x, acck.v[j].x); acc.v[j].y = fmaf(bias, acc.v[j].y, acck.v[j].y);
            acc.v[j].z = fmaf(bias, acc.v[j].z, acck.v[j].z); acc.v[j].w = fmaf(bias, acc.v[j].w, acck.v[j].w);
          }
        }
        if (!is_q && gvalid) b.tape_gmsg[((size_t)sv[u] * sk * a.E + e) * d + ch] = g[u] * bias;  // fold ebias into the tape entry
#pragma unroll
        for (int j = 0; j < DCH; ++j) *reinterpret_cast<float4*>(&red[wave][u][256 * j + 4 * lane]) = acc.v[j];
        if (lane == 0) red_b[wave][u] = part_b;
      }
      __syncthreads();
#pragma unroll
      for (int u = 0; u < NSB; ++u) {
        if (sv[u] < 0 || (u & 3) != wave) continue;
        RowFrag<DCH> tot;
#pragma unroll
        for (int j = 0; j < DCH; ++j) {
          const int c = 256 * j + 4 * lane;
          const float4 p0 = *reinterpret_cast<const float4*>(&red[0][u][c]), p1 = *reinterpret_cast<const float4*>(&red[1][u][c]);
          const float4 p2 = *reinterpret_cast<const float4*>(&red[2][u][c]), p3 = *reinterpret_cast<const float4*>(&red[3][u][c]);
          tot.v[j] = make_float4((p0.x + p1.x) + (p2.x + p3.x), (p0.y + p1.y) + (p2.y + p3.y),
                                 (p0.z + p1.z) + (p2.z + p3.z), (p0.w + p1.w) + (p2.w + p3.w));
        }
        const int vs = sv[u] * sk;
        if (is_q) {
          store_row(tot, b.slab_q + ((size_t)blockIdx.x * SR + vs * a.R + r) * d, d, lane);
        } else {
          store_row(tot, b.slab + ((size_t)blockIdx.x * SE + vs * a.E + e) * d, d, lane);
          if (lane == 0)
            b.g_ebias_part[(size_t)blockIdx.x * SE + vs * a.E + e] = (red_b[0][u] + red_b[1][u]) + (red_b[2][u] + red_b[3][u]);
        }
      }
      __syncthreads();
    }
  }
}

// ---------------------------------------------------------------- backward: gather per sender
template <int DCH>
__global__ __launch_bounds__(256) void k_router_gather_bwd(ic_router_bwd_args b, int nslab) {
  const ic_router_args& a = b.f;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int lrow = blockIdx.x * 4 + wave;
  if (lrow >= a.S * a.R) return;
  const int d = a.d, s = lrow / a.R, r = lrow % a.R;
  const int vs = s * stream_step(b), row = vs * a.R + r;
  const int SE = a.S * stream_step(b) * a.E, SR = a.S * stream_step(b) * a.R;
  const bool prec = a.flags & IC_RF_PRECISION;
  if (!stream_live(b, s)) return;
  RowFrag<DCH> g;
  zero_frag(g);
  if (prec) {
    load_row(g, b.gHd + (size_t)row * d, d, lane);
    if (b.tape_qact[row]) sum_slabs(g, b.slab_q + (size_t)row * d, (size_t)SR * d, nslab, d, lane);
  }
  if (prev_on(b, s, r)) {
    for (int i = b.src_off[r]; i < b.src_off[r + 1]; ++i) {
      const int e = b.src_edges[i];
      sum_slabs(g, b.slab + ((size_t)vs * a.E + e) * d, (size_t)SE * d, nslab, d, lane);
    }
  }
  float* out = b.in_slot ? b.gH + ((size_t)b.in_slot[s] * a.R + r) * d
                         : b.gH + (int64_t)s * (b.gH_sstride ? b.gH_sstride : (int64_t)a.R * d) + (size_t)r * d;
  store_row(g, out, d, lane);
}

// ---------------------------------------------------------------- deferred edge weight gradient
template <int DCH>
__global__ __launch_bounds__(256) void k_wgrad_edge(ic_wgrad_edge_args a) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int d = a.d;
  const int row0 = blockIdx.x * 8 + wave * 2;
  if (row0 >= d) return;
  const int y = blockIdx.y;  // edge id (which<2) or region id (which==2)
  const int n_g = a.which == 2 ? a.R : a.E;
  const int hrow = a.which == 2 ? y : a.src[y];
  const int m = a.which == 2 ? y : a.widx[y];
  RowFrag<DCH> acc0, acc1, op;
  zero_frag(acc0);
  zero_frag(acc1);
  // (slot, stream) pairs are tested 64 at a time (one lane each, one memory round trip) and the
  // contributing ones walked in order from the ballot, the next pair's operands in flight while
  // the current one is accumulated: same order of additions as a plain scan.
  bool any = false;
  const int TS = a.T * a.S;
  for (int base = 0; base < TS; base += 64) {
    const int ts = base + lane;
    unsigned long long act =
        __ballot(ts < TS && a.visited[ts / a.S] != 0 && a.tape_eact[(size_t)ts * n_g + y] != 0);
    if (!act) continue;
    any = true;
    RowFrag<DCH> nxt;
    float g0 = 0.f, g1 = 0.f, h0 = 0.f, h1 = 0.f;
    int cur = base + __ffsll((long long)act) - 1;
    act &= act - 1;
    {
      const float* g = a.tape_g + ((size_t)cur * n_g + y) * d;
      g0 = g[row0]; g1 = g[row0 + 1];
      load_row(op, a.tape_H + ((size_t)cur * a.R + hrow) * d, d, lane);
    }
    while (cur >= 0) {
      const int nx = act ? base + __ffsll((long long)act) - 1 : -1;
      act &= act - 1;
      if (nx >= 0) {
        const float* g = a.tape_g + ((size_t)nx * n_g + y) * d;
        h0 = g[row0]; h1 = g[row0 + 1];
        load_row(nxt, a.tape_H + ((size_t)nx * a.R + hrow) * d, d, lane);
      }
      axpy_frag(acc0, g0, op);
      axpy_frag(acc1, g1, op);
      op = nxt; g0 = h0; g1 = h1;
      cur = nx;
    }
  }
  if (!any) return;
  float* out = a.gW + ((size_t)m * d + row0) * d;
  {  // one workgroup per matrix slice (shared matrices go through k_wgrad_shared_mfma): plain read-modify-write
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
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) a.touched[m] = 1;
}

// W[widx[e]] += sum_b coef[b,e] * outer(H[b,dst_e], H[b,src_e])   (training.py:240-243)
template <int DCH>
__global__ __launch_bounds__(256) void k_hebbian(float* W, const int32_t* dst_of_edge,
                                                 const int32_t* src, const int32_t* widx,
                                                 const float* coef, const float* H, int B, int R,
                                                 int d, int E) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int row0 = blockIdx.x * 8 + wave * 2;
  if (row0 >= d) return;
  const int e = blockIdx.y;
  float* out = W + ((size_t)widx[e] * d + row0) * d;
  const int dst = dst_of_edge[e], sr = src[e];
  RowFrag<DCH> w0, w1, pre;
  bool loaded = false;
  for (int bb = 0; bb < B; ++bb) {
    const float c = coef[(size_t)bb * E + e];
    if (c == 0.f) continue;
    if (!loaded) {
      load_row(w0, out, d, lane);
      load_row(w1, out + d, d, lane);
      loaded = true;
    }
    const float* post = H + ((size_t)bb * R + dst) * d;
    load_row(pre, H + ((size_t)bb * R + sr) * d, d, lane);
    const float p0 = post[row0], p1 = post[row0 + 1];
#pragma unroll
    for (int j = 0; j < DCH; ++j) {  // W.add_(c * outer(post, pre)), same rounding order
      w0.v[j].x += c * (p0 * pre.v[j].x); w0.v[j].y += c * (p0 * pre.v[j].y);
      w0.v[j].z += c * (p0 * pre.v[j].z); w0.v[j].w += c * (p0 * pre.v[j].w);
      w1.v[j].x += c * (p1 * pre.v[j].x); w1.v[j].y += c * (p1 * pre.v[j].y);
      w1.v[j].z += c * (p1 * pre.v[j].z); w1.v[j].w += c * (p1 * pre.v[j].w);
    }
  }
  if (loaded) {
    store_row(w0, out, d, lane);
    store_row(w1, out + d, d, lane);
  }
}

// ---------------------------------------------------------------- factorized mode (IC_RF_FACTORIZED)
// msg_e = ebias_e * U (w_e * V h_src)  (gate.py:150-159,278-281); without precision routing the messages of a
// destination are summed, so M[dst] = U z[dst], z[dst] = sum_e ebias_e * w_e * t[src_e], t = V h: the two matrix
// products are shared-matrix GEMMs over the (stream, region) columns (router_mfma.hip, identity edge tables), the
// mix between them and its backward are the two elementwise kernels below.
template <int DCH>
__global__ __launch_bounds__(256) void k_fact_mix_fwd(ic_router_args a) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int row = blockIdx.x * 4 + wave;
  if (row >= a.S * a.R) return;
  const int d = a.d, s = row / a.R, r = row % a.R;
  RowFrag<DCH> acc;
  zero_frag(acc);
  bool any = false;
  for (int e = a.dst_off[r]; e < a.dst_off[r + 1]; ++e) {  // edge order: the reference's index_add_ order
    const int src = a.src[e];
    if (!prev_on(a, s, src)) continue;
    any = true;
    RowFrag<DCH> t, w;
    load_row(t, a.fact_t + ((size_t)s * a.R + src) * d, d, lane);
    load_row(w, a.fact_w + (size_t)e * d, d, lane);
    const float bias = a.ebias[e];
#pragma unroll
    for (int j = 0; j < DCH; ++j) {
      acc.v[j].x += bias * (w.v[j].x * t.v[j].x); acc.v[j].y += bias * (w.v[j].y * t.v[j].y);
      acc.v[j].z += bias * (w.v[j].z * t.v[j].z); acc.v[j].w += bias * (w.v[j].w * t.v[j].w);
    }
  }
  store_row(acc, a.fact_z + (size_t)row * d, d, lane);
  if (lane == 0) a.fact_dact[row] = any ? 1 : 0;
}

__device__ __forceinline__ int dst_of_edge(const ic_router_args& a, int e) {
  int lo = 0, hi = a.R;  // the r with dst_off[r] <= e < dst_off[r + 1]
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (a.dst_off[mid] <= e) lo = mid; else hi = mid;
  }
  return lo;
}

// rows [0, S*E): per (stream, edge) -- d/d w_e rows, d/d ebias_e, edge activity for the tape;
// rows [S*E, S*E + S*R): per (stream, sender) -- gt = sum over the sender's out-edges of ebias_e * w_e * gz[dst_e]
template <int DCH>
__global__ __launch_bounds__(256) void k_fact_mix_bwd(ic_router_bwd_args b) {
  const ic_router_args& a = b.f;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int id = blockIdx.x * 4 + wave;
  const int d = a.d, sk = stream_step(b);
  const int nE = a.S * a.E, nR = a.S * a.R;
  if (id < nE) {
    const int s = id / a.E, e = id % a.E, vs = s * sk;
    const int src = a.src[e];
    const bool act = prev_on(b, s, src);  // (false for a skipped stream)
    if (lane == 0) b.tape_eact[(size_t)vs * a.E + e] = act ? 1 : 0;
    if (!act) return;
    const int dst = dst_of_edge(a, e);
    RowFrag<DCH> gz, t, w, out;
    load_row(gz, b.fact_gz + ((size_t)vs * a.R + dst) * d, d, lane);
    load_row(t, a.fact_t + ((size_t)vs * a.R + src) * d, d, lane);
    load_row(w, a.fact_w + (size_t)e * d, d, lane);
    const float bias = a.ebias[e];
    float part = 0.f;
#pragma unroll
    for (int j = 0; j < DCH; ++j) {
      const int c = 4 * lane + 256 * j;
      out.v[j] = make_float4(bias * (gz.v[j].x * t.v[j].x), bias * (gz.v[j].y * t.v[j].y), bias * (gz.v[j].z * t.v[j].z),
                             bias * (gz.v[j].w * t.v[j].w));
      if (c < d)
        part += (gz.v[j].x * (w.v[j].x * t.v[j].x) + gz.v[j].y * (w.v[j].y * t.v[j].y)) +
                (gz.v[j].z * (w.v[j].z * t.v[j].z) + gz.v[j].w * (w.v[j].w * t.v[j].w));
    }
    store_row(out, b.g_fw_rows + ((size_t)vs * a.E + e) * d, d, lane);
    part = wave_sum(part);
    if (lane == 0) b.g_ebias_part[(size_t)vs * a.E + e] = part;  // (slab 0; the other slabs stay zero)
    return;
  }
  const int id2 = id - nE;
  if (id2 >= nR) return;
  const int s = id2 / a.R, r = id2 % a.R, vs = s * sk;
  RowFrag<DCH> acc;
  zero_frag(acc);
  if (prev_on(b, s, r)) {
    for (int i = b.src_off[r]; i < b.src_off[r + 1]; ++i) {
      const int e = b.src_edges[i];
      const int dst = dst_of_edge(a, e);
      RowFrag<DCH> gz, w;
      load_row(gz, b.fact_gz + ((size_t)vs * a.R + dst) * d, d, lane);
      load_row(w, a.fact_w + (size_t)e * d, d, lane);
      const float bias = a.ebias[e];
#pragma unroll
      for (int j = 0; j < DCH; ++j) {
        acc.v[j].x += bias * (w.v[j].x * gz.v[j].x); acc.v[j].y += bias * (w.v[j].y * gz.v[j].y);
        acc.v[j].z += bias * (w.v[j].z * gz.v[j].z); acc.v[j].w += bias * (w.v[j].w * gz.v[j].w);
      }
    }
  }
  store_row(acc, b.fact_gt + ((size_t)vs * a.R + r) * d, d, lane);
}

}  // namespace

static int check_router(const ic_router_args* a, const char* who) {
  IC_REQUIRE(a != nullptr, "%s: NULL args", who);
  IC_REQUIRE(a->S > 0 && a->R > 0 && a->E >= 0, "%s: S=%d R=%d E=%d", who, a->S, a->R, a->E);
  IC_REQUIRE(a->d >= 4 && a->d % 4 == 0 && a->d <= 1024, "%s: d=%d unsupported", who, a->d);
  IC_REQUIRE(a->dst_off && a->H && a->mask_prev && a->M, "%s: NULL pointer", who);
  IC_REQUIRE(a->E == 0 || (a->src && a->widx && (a->W || (a->flags & IC_RF_FACTORIZED)) && a->ebias), "%s: NULL edge pointer", who);
  if (a->flags & IC_RF_FACTORIZED) {
    IC_REQUIRE(!(a->flags & IC_RF_PRECISION), "%s: IC_RF_FACTORIZED with IC_RF_PRECISION is not supported", who);
    IC_REQUIRE(a->E == 0 || (a->fact_U && a->fact_V && a->fact_w && a->fact_iota && a->fact_zero && a->fact_ones && a->fact_t &&
                             a->fact_z && a->fact_dact),
               "%s: NULL factorized-mode pointer", who);
  }
  if (a->flags & IC_RF_PRECISION)
    IC_REQUIRE(a->key_W && a->query_W && a->raw_P && a->msg && a->keyv && a->q && a->wgt,
               "%s: precision-routing buffers are NULL", who);
  return IC_OK;
}

// precision routing: an edge's workgroup walks its streams one after the other (sender mask byte, sender row, four
// reductions each); with many streams they are split over gridDim.z, at most ~4-8 streams per workgroup
// Streams over gridDim.z for the precision edge kernel: only while the (slice, edge) grid alone leaves the chip
// short of workgroups.  Every z slice re-reads its edge's rows, and slices of one edge are dispatched far apart
// (no L2 reuse): at R=512 (110k workgroups without z) a split by 8 multiplied the weight traffic.
static int edge_nz(int S, int64_t base_wgs) {
  int nz = 1;
  while (nz * 8 <= S && nz < 8 && base_wgs * nz < 4096) nz *= 2;
  return nz;
}

// Shared matrices (by_direction) as ONE GEMM on the matrix cores, or through the per-edge kernels (which read W[widx[e]]
// in place)?  A GEMM tile runs its (up to) nW K loops back to back whatever the number of columns: a latency chain of
// ~50 us at d=256, against 9-45 us for the per-edge products of the few active senders of up to ~4096 (stream, region)
// columns (scripts/bydir_small_bench.py).  From kSharedGemmMinCols columns on the GEMM wins; IC_RF_SHARED_GEMM in the
// flags asks for it at any size, IC_SHARED_GEMM_MIN=<columns> in the environment moves the point.
static bool shared_as_gemm(const ic_router_args* a) {
  static const long long min_cols = [] { const char* e = getenv("IC_SHARED_GEMM_MIN"); return e && e[0] ? atoll(e) : (long long)kSharedGemmMinCols; }();
  return (a->flags & IC_RF_SHARED) && !(a->flags & IC_RF_PRECISION) && a->E > 0 &&
         ((a->flags & IC_RF_SHARED_GEMM) || (long long)a->S * a->R >= min_cols);
}

// by_direction WITH precision routing: the per-edge messages and keys as two shared-matrix GEMMs over the (stream, edge)
// pairs?  Same trade as above, counted in pairs -- of which the host only knows the total S * E, not how many carry
// an active sender.  Measured at R=512, d=256 (scripts/bydir_prec_bench.py, forward in us, GEMMs : edge kernels):
// 32 of 512 regions on, S = 8 / 16 / 64 (23 k / 47 k / 186 k pairs): 112 : 83, 134 : 119, 314 : 321;
// 128 on, S = 8 / 64: 143 : 153, 563 : 892.  They meet at ~10 k ACTIVE pairs; kSharedPrecGemmMinPairs is the total from
// which that holds at the sparse end.  Streams laid out plainly only (in_slot blocks are sized by the region count,
// which the pseudo-router of the GEMM replaces by the edge count).
constexpr long long kSharedPrecGemmMinPairs = 131072;
static bool shared_prec_as_gemm(const ic_router_args* a) {
  static const long long min_cols = [] { const char* e = getenv("IC_SHARED_GEMM_MIN"); return e && e[0] ? atoll(e) : kSharedPrecGemmMinPairs; }();
  return (a->flags & IC_RF_SHARED) && (a->flags & IC_RF_PRECISION) && a->E > 0 && !a->in_slot && a->d % 4 == 0 &&
         a->nW >= 1 && a->nW <= 8 && a->msg && a->keyv &&
         ((a->flags & IC_RF_SHARED_GEMM) || (long long)a->S * a->E >= min_cols);
}

// the identity-table router of one of the two row GEMMs of the factorized mode: R self-edges, one shared matrix
static ic_router_args fact_rows(const ic_router_args* a, const float* W) {
  ic_router_args p = {};
  p.S = a->S; p.R = a->R; p.d = a->d; p.E = a->R; p.flags = IC_RF_SHARED; p.nW = 1;
  p.dst_off = a->fact_iota; p.src = a->fact_iota; p.widx = a->fact_zero;
  p.W = W; p.ebias = a->fact_ones;
  p.shared_ws = a->shared_ws;  // (the column sort of the row GEMMs: active rows only)
  p.shared_ws_len = a->shared_ws_len;
  return p;
}

static int router_fact_fwd(const ic_router_args* a, hipStream_t st) {
  if (a->E == 0) return IC_OK;
  ic_router_args p = fact_rows(a, a->fact_V);  // t = V h for the active senders
  p.H = a->H; p.mask_prev = a->mask_prev; p.H_sstride = a->H_sstride; p.mask_sstride = a->mask_sstride; p.in_slot = a->in_slot;
  p.M = a->fact_t;
  int rc = ic_router_shared_fwd_launch(&p, st);
  if (rc) return rc;
  IC_DISPATCH_DCH(a->d, {
    IC_TIMED(IC_KID_router_combine, st, hipLaunchKernelGGL(k_fact_mix_fwd<DCH>, dim3(ic_ceil_div(a->S * a->R, 4)), dim3(256), 0, st, *a));
  });
  p = fact_rows(a, a->fact_U);  // M = U z for the destinations with an active in-edge
  p.H = a->fact_z; p.mask_prev = a->fact_dact; p.M = a->M;
  return ic_router_shared_fwd_launch(&p, st);
}

extern "C" int ic_router_bwd(const ic_router_bwd_args* b, ic_stream_t stream);

static int router_fact_bwd(const ic_router_bwd_args* b, hipStream_t st) {
  const ic_router_args* a = &b->f;
  IC_REQUIRE(b->g_fw_rows && b->fact_gz && b->fact_gt && b->fact_slab && b->fact_gb && b->tape_gU && b->tape_zU && b->tape_aU &&
                 b->tape_gV && b->tape_aV,
             "ic_router_bwd: NULL factorized-mode gradient pointer");
  if (a->E == 0) return IC_OK;
  const int64_t sk = b->sk > 1 ? b->sk : 0;
  ic_router_bwd_args q = {};
  q.f = fact_rows(a, a->fact_U);  // gz = U^T gM; tape of dU: rows gM, operand z
  q.f.H = a->fact_z; q.f.mask_prev = a->fact_dact; q.f.M = b->fact_gz;
  q.f.H_sstride = sk * a->R * a->d; q.f.mask_sstride = sk * a->R;
  q.src_off = a->fact_iota; q.src_edges = a->fact_iota;
  q.gM = b->gM; q.gH = b->fact_gz; q.gH_sstride = sk * a->R * a->d;
  q.g_ebias_part = b->fact_gb; q.slab = b->fact_slab;
  q.tape_gmsg = b->tape_gU; q.tape_H = b->tape_zU; q.tape_eact = b->tape_aU;
  q.sk = b->sk;
  int rc = ic_router_bwd(&q, (ic_stream_t)st);
  if (rc) return rc;
  IC_DISPATCH_DCH(a->d, {
    IC_TIMED(IC_KID_router_comb_bwd, st, hipLaunchKernelGGL(k_fact_mix_bwd<DCH>, dim3(ic_ceil_div(a->S * (a->E + a->R), 4)), dim3(256), 0, st, *b));
  });
  q.f = fact_rows(a, a->fact_V);  // gH = V^T gt; tape of dV: rows gt, operand H
  q.f.H = a->H; q.f.mask_prev = a->mask_prev; q.f.M = b->fact_gz;
  q.f.H_sstride = a->H_sstride; q.f.mask_sstride = a->mask_sstride;
  q.in_slot = b->in_slot;
  q.gM = b->fact_gt; q.gH = b->gH; q.gH_sstride = b->gH_sstride;
  q.tape_gmsg = b->tape_gV; q.tape_H = b->tape_H; q.tape_eact = b->tape_aV;
  return ic_router_bwd(&q, (ic_stream_t)st);
}

extern "C" int ic_router_fwd(const ic_router_args* a, ic_stream_t stream) {
  int rc = check_router(a, "ic_router_fwd");
  if (rc) return rc;
  hipStream_t st = (hipStream_t)stream;
  const int nsl = ic_ceil_div(a->d, kRowsPerBlock);
  if (a->flags & IC_RF_FACTORIZED) {
    int rc2 = router_fact_fwd(a, st);
    if (rc2) return rc2;
    IC_LAUNCH_CHECK("ic_router_fwd");
    return IC_OK;
  }
  if (a->flags & IC_RF_PRECISION) {
    // rows per workgroup: 8 while the (slice, edge) grid is small, 16 / 32 on large maps at d <= 256 (register budget;
    // IC_EDGE_ROWS=8/16/32 forces one: A/B timing, same results bit for bit)
    static const int rows_env = [] { const char* e = getenv("IC_EDGE_ROWS"); return e && e[0] ? atoi(e) : 0; }();
    int pp = 1;
    if (a->d <= 256) {
      const int64_t wgs8 = (int64_t)nsl * (a->E + a->R);
      pp = wgs8 >= 32768 ? 4 : (wgs8 >= 8192 ? 2 : 1);
      if (rows_env == 8 || rows_env == 16 || rows_env == 32) pp = rows_env / 8;
    }
    const int nslp = ic_ceil_div(a->d, kRowsPerBlock * pp);
    // Shared matrices (by_direction + precision): messages and keys of all (stream, edge) pairs as TWO GEMMs on the
    // matrix cores -- the shared-matrix forward over E pseudo-destinations with one in-edge each (identity edge table:
    // dst_off NULL), so its output rows are msg[s, e, :] / keyv[s, e, :]; the edge kernel then only forms the queries.
    int y0 = 0;
    if (shared_prec_as_gemm(a)) {
      ic_router_args p = {};
      p.S = a->S; p.R = a->E; p.d = a->d; p.E = a->E; p.flags = IC_RF_SHARED | (1 << 30) /* idle rows are not written */; p.nW = a->nW;
      p.src = a->src; p.widx = a->widx;  // (dst_off NULL: column (s, e) has the one in-edge e)
      p.H = a->H; p.mask_prev = a->mask_prev;
      p.H_sstride = a->H_sstride ? a->H_sstride : (int64_t)a->R * a->d;
      p.mask_sstride = a->mask_sstride ? a->mask_sstride : (int64_t)a->R;
      p.shared_ws = a->shared_ws; p.shared_ws_len = a->shared_ws_len;
      p.W = a->W; p.ebias = a->ebias; p.M = a->msg;
      int rc2 = ic_router_shared_fwd_launch(&p, st);
      if (rc2) return rc2;
      p.W = a->key_W; p.ebias = nullptr; p.M = a->keyv;  // (keys carry no edge bias)
      rc2 = ic_router_shared_fwd_launch(&p, st);
      if (rc2) return rc2;
      y0 = a->E;
    }
    const int ny = a->E + a->R - y0;
    const dim3 ge(nslp, ny, edge_nz(a->S, (int64_t)nslp * ny));
    IC_DISPATCH_DCH(a->d, {
      if (pp == 4) IC_TIMED(IC_KID_router_edge_fwd, st, hipLaunchKernelGGL((k_router_edge_fwd<DCH, 4>), ge, dim3(256), 0, st, *a, y0));
      else if (pp == 2) IC_TIMED(IC_KID_router_edge_fwd, st, hipLaunchKernelGGL((k_router_edge_fwd<DCH, 2>), ge, dim3(256), 0, st, *a, y0));
      else IC_TIMED(IC_KID_router_edge_fwd, st, hipLaunchKernelGGL((k_router_edge_fwd<DCH, 1>), ge, dim3(256), 0, st, *a, y0));
    });
    const dim3 gcomb(ic_ceil_div(a->S * a->R, 4));
    if (a->d <= 256) IC_TIMED(IC_KID_router_combine, st, hipLaunchKernelGGL(k_router_combine<1>, gcomb, dim3(256), 0, st, *a));
    else if (a->d <= 512) IC_TIMED(IC_KID_router_combine, st, hipLaunchKernelGGL(k_router_combine_serial<2>, gcomb, dim3(256), 0, st, *a));
    else IC_TIMED(IC_KID_router_combine, st, hipLaunchKernelGGL(k_router_combine_serial<4>, gcomb, dim3(256), 0, st, *a));
  } else if (shared_as_gemm(a)) {  // by_direction: one GEMM on the matrix cores
    IC_REQUIRE(a->nW >= 1 && a->nW <= 8, "ic_router_fwd: IC_RF_SHARED needs 1 <= nW <= 8 (nW=%d)", a->nW);
    int rc2 = ic_router_shared_fwd_launch(a, st);
    if (rc2) return rc2;
  } else {
    // rows per workgroup as for the precision edge kernel (IC_ROUTER_ROWS=8/16/32 forces one; same results)
    static const int rrows_env = [] { const char* e = getenv("IC_ROUTER_ROWS"); return e && e[0] ? atoi(e) : 0; }();
    int pp = 1;
    if (a->d <= 256) {
      const int64_t wgs8 = (int64_t)nsl * a->R;
      pp = wgs8 >= 16384 ? 4 : (wgs8 >= 4096 ? 2 : 1);
      if (rrows_env == 8 || rrows_env == 16 || rrows_env == 32) pp = rrows_env / 8;
    }
    const dim3 gr(ic_ceil_div(a->d, kRowsPerBlock * pp), a->R);
    IC_DISPATCH_DCH(a->d, {
      if (pp == 4) IC_TIMED(IC_KID_router_fwd, st, hipLaunchKernelGGL((k_router_fwd_ms<DCH, 4>), gr, dim3(256), 0, st, *a));
      else if (pp == 2) IC_TIMED(IC_KID_router_fwd, st, hipLaunchKernelGGL((k_router_fwd_ms<DCH, 2>), gr, dim3(256), 0, st, *a));
      else IC_TIMED(IC_KID_router_fwd, st, hipLaunchKernelGGL((k_router_fwd_ms<DCH, 1>), gr, dim3(256), 0, st, *a));
    });
  }
  IC_LAUNCH_CHECK("ic_router_fwd");
  return IC_OK;
}

// combine = false (ic_loop_step_fwd, precision mode): the per-destination combine is left to the
// routed region kernel that follows on the same stream.
int ic_front_launch(const ic_gate_args* g, const ic_router_args* a, ic_stream_t stream, bool combine) {
  IC_REQUIRE(g != nullptr && a != nullptr, "ic_front_fwd: NULL args");
  if (g->R > 64 || a->E == 0 || shared_as_gemm(a) || (a->flags & IC_RF_FACTORIZED)) {
    // big maps: the gate's scores want the whole chip (ic_gate_fwd); shared matrices: the router is an MFMA GEMM
    IC_REQUIRE(combine, "ic_front_fwd: no fused launch for R=%d, E=%d", g->R, a->E);
    int rc = ic_gate_fwd(g, stream);
    return rc ? rc : ic_router_fwd(a, stream);
  }
  int rc = check_router(a, "ic_front_fwd");
  if (rc) return rc;
  IC_REQUIRE(g->S == a->S && g->R == a->R && g->d == a->d, "ic_front_fwd: gate (S=%d,R=%d,d=%d) and router (S=%d,R=%d,d=%d) disagree",
             g->S, g->R, g->d, a->S, a->R, a->d);
  IC_REQUIRE(g->scores && g->H && g->gain_ema && g->homeo && g->mask && g->nbr_off && g->nbr, "ic_front_fwd: NULL gate pointer");
  IC_REQUIRE(!g->force_io || (g->sensor_idx >= 0 && g->sensor_idx < g->R && g->motor_idx >= 0 && g->motor_idx < g->R),
             "ic_front_fwd: io indices out of range");
  hipStream_t st = (hipStream_t)stream;
  const int nsl = ic_ceil_div(a->d, kRowsPerBlock);
  if (a->flags & IC_RF_PRECISION) {
    IC_DISPATCH_DCH(a->d, {
      IC_TIMED(IC_KID_router_edge_fwd, st, hipLaunchKernelGGL((k_front<DCH, true>), dim3(nsl, g->S + a->E + a->R, edge_nz(a->S, (int64_t)nsl * (a->E + a->R))), dim3(256), 0, st, *a, *g));
    });
    const dim3 gcomb(ic_ceil_div(a->S * a->R, 4));
    if (!combine) {
    } else if (a->d <= 256) IC_TIMED(IC_KID_router_combine, st, hipLaunchKernelGGL(k_router_combine<1>, gcomb, dim3(256), 0, st, *a));
    else if (a->d <= 512) IC_TIMED(IC_KID_router_combine, st, hipLaunchKernelGGL(k_router_combine_serial<2>, gcomb, dim3(256), 0, st, *a));
    else IC_TIMED(IC_KID_router_combine, st, hipLaunchKernelGGL(k_router_combine_serial<4>, gcomb, dim3(256), 0, st, *a));
  } else {
    IC_DISPATCH_DCH(a->d, { IC_TIMED(IC_KID_router_fwd, st, hipLaunchKernelGGL((k_front<DCH, false>), dim3(nsl, g->S + a->R), dim3(256), 0, st, *a, *g)); });
  }
  IC_LAUNCH_CHECK("ic_front_fwd");
  return IC_OK;
}

extern "C" int ic_front_fwd(const ic_gate_args* g, const ic_router_args* a, ic_stream_t stream) {
  return ic_front_launch(g, a, stream, true);
}

extern "C" int ic_router_bwd(const ic_router_bwd_args* b, ic_stream_t stream) {
  IC_REQUIRE(b != nullptr, "ic_router_bwd: NULL args");
  int rc = check_router(&b->f, "ic_router_bwd");
  if (rc) return rc;
  const ic_router_args* a = &b->f;
  IC_REQUIRE(b->gM && b->gH && b->tape_H && b->tape_eact && b->src_off,
             "ic_router_bwd: NULL gradient pointer");
  if (a->flags & IC_RF_FACTORIZED) {
    IC_REQUIRE(a->E == 0 || (b->src_edges && b->g_ebias_part), "ic_router_bwd: NULL edge gradient pointer");
    int rc2 = router_fact_bwd(b, (hipStream_t)stream);
    if (rc2) return rc2;
    IC_LAUNCH_CHECK("ic_router_bwd");
    return IC_OK;
  }
  IC_REQUIRE(a->E == 0 || (b->src_edges && b->g_ebias_part && b->tape_gmsg && b->slab),
             "ic_router_bwd: NULL edge gradient pointer");
  if (a->flags & IC_RF_PRECISION)
    IC_REQUIRE(b->g_rawP_rows && b->tape_gkey && b->tape_gq && b->tape_qact && b->gHd && b->slab_q,
               "ic_router_bwd: precision-routing gradient buffers are NULL");
  hipStream_t st = (hipStream_t)stream;
  const int nslab = ic_bwd_nslab(a->d);
  const dim3 blk(256), grow(ic_ceil_div(a->S * a->R, 4));
  const int ny = a->E + ((a->flags & IC_RF_PRECISION) ? a->R : 0);
  int nz = 1;
  while (nz < a->S && nz < 16 && nslab * ny * nz < 512) nz *= 2;
  // 8 streams or more: the batched variant of the transposed products (at S = 1 its 16 KB of LDS slow the drain of
  // the idle workgroups: 56 vs 29 us at R=512; 10M bench, S=16: 17 vs 33 us).  IC_BWD_SPLIT=0/1 forces one.
  static const int many_env = [] { const char* e = getenv("IC_BWD_SPLIT"); return (e && e[0]) ? (e[0] == '0' ? 0 : 1) : -1; }();
  const bool many = many_env >= 0 ? many_env == 1 : a->S >= 8;
  // shared matrices without precision routing: the input gradient is ONE MFMA GEMM (router_mfma.hip) in place of the
  // per-edge transposed GEMVs and the gather, whatever the stream addressing (in_slot / sk / gH_sstride)
  static const int dgrad_env = [] { const char* e = getenv("IC_DGRAD_MFMA"); return (e && e[0]) ? (e[0] == '0' ? 0 : 1) : -1; }();
  const bool plain = !b->in_slot && b->sk <= 1 && b->gH_sstride == 0;
  const bool dgrad_mfma = (a->flags & IC_RF_SHARED) && !(a->flags & IC_RF_PRECISION) && a->E > 0 && (plain || dgrad_env != 0) &&
                          a->d % 4 == 0 && a->nW >= 1 && a->nW <= 8 && ic_ceil_div(a->d, 128) <= nslab;
  if (a->d <= 256) {
    constexpr int DCH = 1;
    IC_TIMED(IC_KID_router_comb_bwd, st, hipLaunchKernelGGL(k_router_comb_bwd<DCH>, grow, blk, 0, st, *b, (int)dgrad_mfma));
    if (dgrad_mfma) {
      int rc2 = ic_router_shared_dgrad_launch(b, nslab, st);
      if (rc2) return rc2;
      IC_LAUNCH_CHECK("ic_router_bwd");
      return IC_OK;
    }
    // plain routing streams ONE matrix per edge: 64 rows per workgroup (as many bytes in flight as the precision
    // path's 32 rows of two matrices) and half the slabs; the unused slabs of g_ebias_part stay zero
    static const int rpw16_env = [] { const char* e = getenv("IC_MATT_RPW16"); return (e && e[0]) ? (e[0] == '0' ? 0 : 1) : -1; }();
    const bool rpw16 = many && !(a->flags & IC_RF_PRECISION) && nslab % 2 == 0 && nslab * ny >= 4096 && rpw16_env != 0;
    int ns_used = nslab;
    if (ny > 0 && rpw16) {
      ns_used = nslab / 2;
      IC_TIMED(IC_KID_router_mat_bwd, st, hipLaunchKernelGGL((k_router_matT_bwd<DCH, 16, 4>), dim3(ns_used, ny, nz), blk, 0, st, *b));
    } else if (ny > 0 && many) IC_TIMED(IC_KID_router_mat_bwd, st, hipLaunchKernelGGL((k_router_matT_bwd<DCH, 8, 4>), dim3(nslab, ny, nz), blk, 0, st, *b));
    else if (ny > 0) IC_TIMED(IC_KID_router_mat_bwd, st, hipLaunchKernelGGL((k_router_mat_bwd<DCH, 8>), dim3(nslab, ny, nz), blk, 0, st, *b));
    IC_TIMED(IC_KID_router_gather_bwd, st, hipLaunchKernelGGL(k_router_gather_bwd<DCH>, grow, blk, 0, st, *b, ns_used));
  } else if (a->d <= 512) {
    constexpr int DCH = 2;
    IC_TIMED(IC_KID_router_comb_bwd, st, hipLaunchKernelGGL(k_router_comb_bwd_serial<DCH>, grow, blk, 0, st, *b, (int)dgrad_mfma));
    if (dgrad_mfma) {
      int rc2 = ic_router_shared_dgrad_launch(b, nslab, st);
      if (rc2) return rc2;
      IC_LAUNCH_CHECK("ic_router_bwd");
      return IC_OK;
    }
    if (ny > 0 && many) IC_TIMED(IC_KID_router_mat_bwd, st, hipLaunchKernelGGL((k_router_matT_bwd<DCH, 2, 4>), dim3(nslab, ny, nz), blk, 0, st, *b));
    else if (ny > 0) IC_TIMED(IC_KID_router_mat_bwd, st, hipLaunchKernelGGL((k_router_mat_bwd<DCH, 2>), dim3(nslab, ny, nz), blk, 0, st, *b));
    IC_TIMED(IC_KID_router_gather_bwd, st, hipLaunchKernelGGL(k_router_gather_bwd<DCH>, grow, blk, 0, st, *b, nslab));
  } else {
    constexpr int DCH = 4;
    IC_TIMED(IC_KID_router_comb_bwd, st, hipLaunchKernelGGL(k_router_comb_bwd_serial<DCH>, grow, blk, 0, st, *b, (int)dgrad_mfma));
    if (dgrad_mfma) {
      int rc2 = ic_router_shared_dgrad_launch(b, nslab, st);
      if (rc2) return rc2;
      IC_LAUNCH_CHECK("ic_router_bwd");
      return IC_OK;
    }
    if (ny > 0 && many) IC_TIMED(IC_KID_router_mat_bwd, st, hipLaunchKernelGGL((k_router_matT_bwd<DCH, 2, 2>), dim3(nslab, ny, nz), blk, 0, st, *b));
    else if (ny > 0) IC_TIMED(IC_KID_router_mat_bwd, st, hipLaunchKernelGGL((k_router_mat_bwd<DCH, 2>), dim3(nslab, ny, nz), blk, 0, st, *b));
    IC_TIMED(IC_KID_router_gather_bwd, st, hipLaunchKernelGGL(k_router_gather_bwd<DCH>, grow, blk, 0, st, *b, nslab));
  }
  IC_LAUNCH_CHECK("ic_router_bwd");
  return IC_OK;
}

extern "C" int ic_router_wgrad(const ic_wgrad_edge_args* a, ic_stream_t stream) {
  IC_REQUIRE(a != nullptr, "ic_router_wgrad: NULL args");
  IC_REQUIRE(a->T >= 0 && a->S > 0 && a->R > 0 && a->E >= 0 && a->nW >= 0, "ic_router_wgrad: bad dims");
  IC_REQUIRE(a->d >= 4 && a->d % 4 == 0 && a->d <= 1024, "ic_router_wgrad: d=%d unsupported", a->d);
  IC_REQUIRE(a->which >= 0 && a->which <= 2, "ic_router_wgrad: which=%d", a->which);
  const int ny = a->which == 2 ? a->R : a->E;
  if (a->T == 0 || ny == 0) return IC_OK;
  IC_REQUIRE(a->tape_g && a->tape_H && a->tape_eact && a->visited && a->gW && a->touched,
             "ic_router_wgrad: NULL pointer");
  IC_REQUIRE(a->which == 2 || (a->src && a->widx), "ic_router_wgrad: NULL edge table");
  // shared matrices (by_direction): one deterministic MFMA GEMM per matrix.  A caller that passes the pair-list
  // workspace asks for the same GEMM for one matrix per edge too (dW_e = sum over the tape's slots g (x) h: with hundreds
  // of slots per edge -- the LoopEngine's pass over all loops of a train_step -- that is a [d,T] x [T,d] product per edge,
  // which the row-slice kernel below re-reads from L2 once per 8 rows: 88 us against ~20 at R=10, 1.1 ms at R=512)
  if (a->which != 2 && (a->nW != a->E || a->pair_ws != nullptr)) {
    IC_REQUIRE(a->nW >= 1 && a->nW <= 8192 && a->E <= 4096, "ic_router_wgrad: the GEMM path needs nW in [1,8192] and E <= 4096 (nW=%d, E=%d)",
               a->nW, a->E);
    int rc = ic_wgrad_shared_launch(a, (hipStream_t)stream);
    if (rc) return rc;
    IC_LAUNCH_CHECK("ic_router_wgrad");
    return IC_OK;
  }
  IC_DISPATCH_DCH(a->d, {
    IC_TIMED(IC_KID_wgrad_edge, (hipStream_t)stream, hipLaunchKernelGGL(k_wgrad_edge<DCH>, dim3(ic_ceil_div(a->d, 8), ny), dim3(256), 0,
                       (hipStream_t)stream, *a));
  });
  IC_LAUNCH_CHECK("ic_router_wgrad");
  return IC_OK;
}

extern "C" int ic_hebbian_rank1(float* W, const int32_t* dst_of_edge, const int32_t* src,
                                const int32_t* widx, const float* coef, const float* H, int32_t B,
                                int32_t R, int32_t d, int32_t E, ic_stream_t stream) {
  IC_REQUIRE(W && dst_of_edge && src && widx && coef && H, "ic_hebbian_rank1: NULL pointer");
  IC_REQUIRE(d >= 4 && d % 4 == 0 && d <= 1024, "ic_hebbian_rank1: d=%d unsupported", d);
  if (B <= 0 || E <= 0) return IC_OK;
  IC_DISPATCH_DCH(d, {
    IC_TIMED(IC_KID_hebbian, (hipStream_t)stream, hipLaunchKernelGGL(k_hebbian<DCH>, dim3(ic_ceil_div(d, 8), E), dim3(256), 0, (hipStream_t)stream,
                       W, dst_of_edge, src, widx, coef, H, B, R, d, E));
  });
  IC_LAUNCH_CHECK("ic_hebbian_rank1");
  return IC_OK;
}
