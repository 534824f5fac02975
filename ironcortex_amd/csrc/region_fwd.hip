// Region phase, forward: prologue -> rkv/state -> out -> finish.
// Replaces the Python loop over regions in CortexReasoner.forward_inner_step
// (reference model.py:271-294) and RWKVRegionCell.step/predict/skip (region.py:128-249).
//
// Parallelisation: the r/k/v/o GEMVs are weight-bandwidth bound (16 d^2 bytes per active
// region, 0.5 flop/byte), so each workgroup owns a ROW SLICE of one region's matrices, keeps
// it in registers, and serves every stream that has the region active.  A wave owns whole
// rows (lane l holds columns 4l..4l+3 of each 256-column chunk => one 1 KiB coalesced
// request per row per wave) and reduces dot products with wave64 xor-shuffles.
#include "ic_common.h"
#include "ic_route.h"
#include "ic_kwta.h"

namespace {


// ---------------------------------------------------------------- prologue (wave per row)
// x = RMSNorm_r( RMSNorm_in(M[r] (+sensor)) + pred )   (region.py:173-177, model.py:198-204)
// All operand rows are requested before the first reduction.
template <int DCH>
__device__ __forceinline__ void prologue_math(RowFrag<DCH>& u, const RowFrag<DCH>& sv, const RowFrag<DCH>& win,
                                              const RowFrag<DCH>& p, const RowFrag<DCH>& w, bool io, bool norm_in,
                                              bool pred, int d, RowFrag<DCH>& u1, RowFrag<DCH>& x, float& inv0,
                                              float& inv1) {
  const float inv_d = 1.0f / (float)d;
  inv0 = 1.0f;
  if (norm_in) {
    if (io) {
#pragma unroll
      for (int j = 0; j < DCH; ++j) {
        u.v[j].x += sv.v[j].x; u.v[j].y += sv.v[j].y; u.v[j].z += sv.v[j].z; u.v[j].w += sv.v[j].w;
      }
    }
    const float ss = wave_sum(dot_frag(u, u));
    const float s0 = sqrtf(ss * inv_d + IC_RMS_EPS);
    inv0 = 1.0f / s0;
#pragma unroll
    for (int j = 0; j < DCH; ++j) {
      u.v[j].x = win.v[j].x * (u.v[j].x / s0); u.v[j].y = win.v[j].y * (u.v[j].y / s0);
      u.v[j].z = win.v[j].z * (u.v[j].z / s0); u.v[j].w = win.v[j].w * (u.v[j].w / s0);
    }
  }
  if (pred) {
#pragma unroll
    for (int j = 0; j < DCH; ++j) {
      u.v[j].x += p.v[j].x; u.v[j].y += p.v[j].y; u.v[j].z += p.v[j].z; u.v[j].w += p.v[j].w;
    }
  }
  u1 = u;
  const float ss1 = wave_sum(dot_frag(u, u));
  const float s1 = sqrtf(ss1 * inv_d + IC_RMS_EPS);
  inv1 = 1.0f / s1;
#pragma unroll
  for (int j = 0; j < DCH; ++j) {
    x.v[j].x = w.v[j].x * (u.v[j].x / s1); x.v[j].y = w.v[j].y * (u.v[j].y / s1);
    x.v[j].z = w.v[j].z * (u.v[j].z / s1); x.v[j].w = w.v[j].w * (u.v[j].w / s1);
  }
}

template <int DCH>
__device__ __forceinline__ void prologue_row(const ic_region_args& a, int row, int lane, RowFrag<DCH>& u1,
                                             RowFrag<DCH>& x, float& inv0, float& inv1) {
  const int s = row / a.R, r = row % a.R;
  const int d = a.d;
  const bool norm_in = a.flags & IC_F_NORM_IN, pred = a.flags & IC_F_PRED;
  const bool io = norm_in && a.sensor && (r == a.sensor_idx || r == a.motor_idx);
  RowFrag<DCH> u, sv, win, p, w;
  load_row(u, a.M + (size_t)row * d, d, lane);
  if (io) load_row(sv, a.sensor + (size_t)s * d, d, lane);
  if (norm_in) load_row(win, a.norm_in_w, d, lane);
  if (pred) load_row(p, a.pred + (size_t)row * d, d, lane);
  load_row(w, a.norm_w + (size_t)r * d, d, lane);
  prologue_math(u, sv, win, p, w, io, norm_in, pred, d, u1, x, inv0, inv1);
}

// IC_F_CHAIN: the trace and dt of region r as they stand when stream s starts, given the last
// stream before s that had the region active (prev, -1: none).  Streams in between were
// inactive: each is one trace EMA step on its own message row (region.py:140-146) and dt+1.
// ROUTED (ic_loop_step_fwd): the message rows are not read from M but from the workgroup's LDS
// copy (mrows[s][d], combined from the edges' messages at the start of the kernel).
// The rows of the chain prologue that do not depend on the stream (the routed kernel requests them
// before its combines, so they arrive while the combines' dependent loads are in flight).
template <int DCH>
struct ChainRows {
  RowFrag<DCH> sv, win, w, p;
  int dt;
};
template <int DCH>
__device__ __forceinline__ void chain_rows(const ic_region_args& a, int r, int lane, ChainRows<DCH>& c) {
  const int d = a.d;
  const bool norm_in = a.flags & IC_F_NORM_IN, pred = a.flags & IC_F_PRED;
  const bool io = norm_in && a.sensor && (r == a.sensor_idx || r == a.motor_idx);
  zero_frag(c.p);
  if (io) load_row(c.sv, a.sensor, d, lane);
  if (norm_in) load_row(c.win, a.norm_in_w, d, lane);
  load_row(c.w, a.norm_w + (size_t)r * d, d, lane);
  if (pred) load_row(c.p, a.pred + (size_t)r * d, d, lane);
  c.dt = a.dt[r];
}

template <int DCH, bool ROUTED>
__device__ __forceinline__ void chain_prologue(const ic_region_args& a, int r, int s, int prev, int lane,
                                               float one_minus_alpha, RowFrag<DCH>& u1, RowFrag<DCH>& x,
                                               float& inv0, float& inv1, float& steps,
                                               const float* mrows, const ChainRows<DCH>& pre) {
  const int d = a.d;
  const bool norm_in = a.flags & IC_F_NORM_IN, pred = a.flags & IC_F_PRED;
  const bool io = norm_in && a.sensor && (r == a.sensor_idx || r == a.motor_idx);
  RowFrag<DCH> u, sv, win, p, w;
  int dtv = 0;
  zero_frag(p);
  if (ROUTED) {
    load_row(u, mrows + s * d, d, lane);
    sv = pre.sv; win = pre.win; w = pre.w;
    if (prev < 0) {
      p = pre.p;
      dtv = pre.dt;
    }
  } else {
    load_row(u, a.M + ((size_t)s * a.R + r) * d, d, lane);
    if (io) load_row(sv, a.sensor, d, lane);
    if (norm_in) load_row(win, a.norm_in_w, d, lane);
    load_row(w, a.norm_w + (size_t)r * d, d, lane);
    if (prev < 0) {
      if (pred) load_row(p, a.pred + (size_t)r * d, d, lane);
      dtv = a.dt[r];
    }
  }
  const float al = (float)a.pred_alpha;
  for (int q = prev + 1; q < s; ++q) {
    if (pred) {
      RowFrag<DCH> m;
      if (ROUTED) load_row(m, mrows + q * d, d, lane);
      else load_row(m, a.M + ((size_t)q * a.R + r) * d, d, lane);
#pragma unroll
      for (int j = 0; j < DCH; ++j) {
        p.v[j].x = al * p.v[j].x + one_minus_alpha * m.v[j].x;
        p.v[j].y = al * p.v[j].y + one_minus_alpha * m.v[j].y;
        p.v[j].z = al * p.v[j].z + one_minus_alpha * m.v[j].z;
        p.v[j].w = al * p.v[j].w + one_minus_alpha * m.v[j].w;
      }
    }
    dtv += 1;
  }
  steps = (float)(dtv + 1);
  prologue_math(u, sv, win, p, w, io, norm_in, pred, d, u1, x, inv0, inv1);
}

template <int DCH>
__global__ __launch_bounds__(256) void k_region_pro(ic_region_args a) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int row = blockIdx.x * 4 + wave;
  if (row >= a.S * a.R) return;
  if (!a.mask[row]) return;
  const int d = a.d;
  RowFrag<DCH> u1, x;
  float inv0, inv1;
  prologue_row(a, row, lane, u1, x, inv0, inv1);
  store_row(u1, a.u1 + (size_t)row * d, d, lane);
  store_row(x, a.x + (size_t)row * d, d, lane);
  if (lane == 0) {
    a.inv_s[row * 2 + 0] = inv0;
    a.inv_s[row * 2 + 1] = inv1;
  }
}

// Is region r active in any stream?  Every lane tests its own streams (64 per round trip) and
// the answer is a ballot: with hundreds of regions most workgroups of a launch are idle, and how
// fast they leave (one memory round trip, not one per stream) decides how soon the working ones
// get their slots.
// The slice kernels may split the streams over gridDim.z (workgroup z serves streams z, z + nz, ...): with many
// streams and few regions (the lockstep loops: S = 16, R = 10) a workgroup's serial walk over its active streams,
// not the weight rows it re-reads from L2, bounds the launch.  nz = 1 (always for IC_F_CHAIN) is the plain walk.
__device__ __forceinline__ bool region_has_work(const uint8_t* mask, int S, int R, int r) {
  const int lane = threadIdx.x & 63, nz = gridDim.z;
  for (int sc = blockIdx.z; sc < S; sc += 64 * nz) {
    const int s = sc + lane * nz;
    if (__ballot(s < S && mask[(size_t)s * R + r] != 0)) return true;
  }
  return false;
}

// ---------------------------------------------------------------- r,k,v GEMVs + state update
// PP = channel pairs per wave: a workgroup owns 8*PP rows of W_r, W_k, W_v (24*PP KiB at d=256),
// all loaded up front so every wave keeps 6*PP independent 1 KiB row requests in flight.
// FUSE: the prologue is recomputed by every wave instead of being a launch of its own (few
// streams: the extra row loads are L2 hits, the saved launch is ~6 us).
// CHAIN (IC_F_CHAIN, needs FUSE): the streams are consecutive steps of one state stream; the state
// a stream starts from is what the previous active stream of this launch computed (registers) or,
// before the first one, the input state [1,R,..].
// ROUTED (ic_loop_step_fwd, needs CHAIN, S <= kRoutedStreams, d % 8 == 0): the router's
// per-destination combine is part of this kernel instead of a launch of its own.  The waves of a
// workgroup share the streams' combines (wave w: streams w, w+4, ...; all in flight together), leave
// the message rows in LDS for the prologue, and workgroup 0 of every region also writes M and the
// edge weights (the trace EMA of k_region_fin_chain and the backward read them).
constexpr int kRoutedStreams = 16;

template <int DCH, int PP, bool FUSE, bool CHAIN, bool ROUTED>
__device__ __forceinline__ void region_rkv_body(const ic_region_args& a, float one_minus_alpha,
                                                const ic_router_args& ra) {
  __shared__ __align__(16) float mrows[ROUTED ? kRoutedStreams * 256 * DCH : 4];
  const int r = blockIdx.y;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const bool work = region_has_work(a.mask, a.S, a.R, r);
  auto combine_rows = [&]() {
    if (ROUTED) {
      // idle regions: only workgroup 0 (M and weights); working ones: every workgroup needs the rows
      // of its active streams, and of the sat-out ones when the trace EMA runs over them
      const bool wr = blockIdx.x == 0, pred = a.flags & IC_F_PRED;
      for (int s = wave; s < a.S; s += 4) {
        if (wr || (work && (pred || a.mask[s * a.R + r]))) {
          RowFrag<DCH> m;
          combine_row<DCH>(ra, s, r, lane, wr, m);
          if (work) store_row(m, mrows + s * a.d, a.d, lane);
        }
      }
    }
  };
  if (!work) {
    combine_rows();
    return;
  }
  const int d = a.d;
  const int row0 = blockIdx.x * (8 * PP) + wave * (2 * PP);  // even: rotation pairs stay in one wave
  if (!ROUTED && row0 >= d) return;  // (ROUTED: d % 8 == 0, and every wave reaches the barrier)
  const size_t dd = (size_t)d * d;
  const float* Wr = a.W + (size_t)r * 4 * dd;
  const float* Wk = Wr + dd;
  const float* Wv = Wk + dd;
  RowFrag<DCH> wr[2 * PP], wk[2 * PP], wv[2 * PP];
#pragma unroll
  for (int i = 0; i < 2 * PP; ++i) {
    if (row0 + i < d) {  // wave-uniform (d is even)
      load_row(wr[i], Wr + (size_t)(row0 + i) * d, d, lane);
      load_row(wk[i], Wk + (size_t)(row0 + i) * d, d, lane);
      load_row(wv[i], Wv + (size_t)(row0 + i) * d, d, lane);
    } else {
      zero_frag(wr[i]); zero_frag(wk[i]); zero_frag(wv[i]);
    }
  }
  const bool afd = a.flags & IC_F_AFD;
  // per-channel constants; lanes 0/1 finish one channel of each pair
  float rate[PP], pn[PP], on[PP], rc[PP], rs[PP];
  bool rot[PP], ok[PP];
#pragma unroll
  for (int p = 0; p < PP; ++p) {
    const int chr = row0 + 2 * p + (lane & 1);
    ok[p] = chr < d;
    const int ch = ok[p] ? chr : 0;
    rate[p] = -ic_softplus(a.raw_decay[(size_t)r * d + ch]);  // region.py:94-95
    pn[p] = on[p] = 0.f;
    if (afd) {
      const bool vec = a.flags & IC_F_NOISE_VEC;
      pn[p] = ic_noise(vec ? a.pn_param[(size_t)r * d + ch] : a.pn_param[r]);
      on[p] = ic_noise(vec ? a.on_param[(size_t)r * d + ch] : a.on_param[r]);
    }
    rc[p] = 1.f;
    rs[p] = 0.f;
    rot[p] = ch < 2 * a.m_time;
    if (rot[p]) {
      const float th = a.step_pos * a.W_time[ch >> 1];  // region.py:186
      rc[p] = cosf(th);
      rs[p] = sinf(th);
    }
  }
  ChainRows<DCH> pre;
  if (ROUTED) chain_rows<DCH>(a, r, lane, pre);  // in flight during the combines
  combine_rows();  // (after the weight rows were requested)
  if (ROUTED) __syncthreads();
  // Streams are walked 64 at a time from one ballot of the mask; the operands of the NEXT active
  // stream (x row, dt, and the state scalars lanes 0/1 update) are requested before the current
  // stream's dot products are reduced, so a stream costs one memory round trip, not three.
  struct Ops {
    RowFrag<DCH> xv;
    float steps;
    float num[PP], den[PP], var[PP];
  };
  auto request = [&](int s, int prev, Ops& o) {
    const int row = s * a.R + r;
    if (FUSE) {
      RowFrag<DCH> u1;
      float inv0, inv1;
      if (CHAIN) {
        chain_prologue<DCH, ROUTED>(a, r, s, prev, lane, one_minus_alpha, u1, o.xv, inv0, inv1, o.steps, mrows, pre);
      } else {
        prologue_row(a, row, lane, u1, o.xv, inv0, inv1);
      }
      if (blockIdx.x == 0 && wave == 0) {  // saved for the backward / the output projection
        store_row(u1, a.u1 + (size_t)row * d, d, lane);
        store_row(o.xv, a.x + (size_t)row * d, d, lane);
        if (lane == 0) {
          a.inv_s[row * 2 + 0] = inv0;
          a.inv_s[row * 2 + 1] = inv1;
        }
      }
    } else {
      load_row(o.xv, a.x + (size_t)row * d, d, lane);
    }
    if (!CHAIN) o.steps = (float)(a.dt[row] + 1);
    if (CHAIN && prev >= 0) return;  // state: handed over in registers after the previous stream
    const int srow = CHAIN ? r : row;
#pragma unroll
    for (int p = 0; p < PP; ++p) {
      o.num[p] = o.den[p] = o.var[p] = 0.f;
      if (lane < 2 && ok[p]) {
        const size_t idx = (size_t)srow * d + row0 + 2 * p + lane;
        o.num[p] = a.num[idx];
        o.den[p] = a.den[idx];
        if (afd) o.var[p] = a.var[idx];
      }
    }
  };
  const int nz = gridDim.z;
  for (int sc = blockIdx.z; sc < a.S; sc += 64 * nz) {
    const int sl = sc + lane * nz;
    unsigned long long act = __ballot(sl < a.S && a.mask[sl * a.R + r] != 0);
    if (!act) continue;
    Ops cur, nxt;
    int s_cur = sc + (__ffsll((long long)act) - 1) * nz;
    act &= act - 1;
    request(s_cur, -1, cur);
    while (s_cur >= 0) {
      const int s_nxt = act ? sc + (__ffsll((long long)act) - 1) * nz : -1;
      act &= act - 1;
      if (s_nxt >= 0) request(s_nxt, s_cur, nxt);
      const int row = s_cur * a.R + r;
      const float steps = cur.steps;
      float nn[PP], dn[PP], vn[PP];
#pragma unroll
      for (int p = 0; p < PP; ++p) {
        nn[p] = dn[p] = vn[p] = 0.f;
        const float pr0 = wave_sum(dot_frag(wr[2 * p], cur.xv)), pr1 = wave_sum(dot_frag(wr[2 * p + 1], cur.xv));
        const float pk0 = wave_sum(dot_frag(wk[2 * p], cur.xv)), pk1 = wave_sum(dot_frag(wk[2 * p + 1], cur.xv));
        const float pv0 = wave_sum(dot_frag(wv[2 * p], cur.xv)), pv1 = wave_sum(dot_frag(wv[2 * p + 1], cur.xv));
        if (lane < 2 && ok[p]) {
          const size_t idx = (size_t)row * d + row0 + 2 * p + lane;
          const float rr = ic_sigmoid(lane ? pr1 : pr0);
          const float kk = lane ? pk1 : pk0;
          float vv = lane ? pv1 : pv0;
          if (rot[p]) vv = lane ? (pv0 * rs[p] + pv1 * rc[p]) : (pv0 * rc[p] - pv1 * rs[p]);  // iron_rope.py:63-64
          const float lam = fmaxf(expf(rate[p] * steps), IC_EPS_DIV);  // region.py:97-98
          const float numd = cur.num[p] * lam, dend = cur.den[p] * lam;  // region.py:131-133
          const float w = expf(fminf(kk, IC_MAX_EXP));  // region.py:190
          float yv, num_n, den_n;
          if (afd) {  // region.py:192-212
            const float vard = cur.var[p] * (lam * lam) + pn[p] * steps;  // region.py:135
            const float prd = numd / (dend + IC_EPS_DIV);
            const float resid = w * vv - prd;
            const float obs = expf(fminf(-kk, IC_MAX_EXP)) * on[p] + IC_EPS_DIV;
            float gain = vard / (vard + obs);
            gain = fminf(fmaxf(gain, IC_EPS_DIV), 1.0f);  // fp32(1 - 1e-8) == 1
            const float st = prd + gain * resid;
            den_n = dend + w;
            num_n = st * den_n;
            vn[p] = (1.0f - gain) * vard;
            a.var_o[idx] = vn[p];
            a.sterm[idx] = (resid * resid) * (1.0f / (vard + obs));
            yv = rr * st;
          } else {  // region.py:214-216
            num_n = numd + w * vv;
            den_n = dend + w;
            yv = rr * (num_n / (den_n + IC_EPS_DIV));
          }
          a.num_o[idx] = num_n;
          a.den_o[idx] = den_n;
          a.y[idx] = yv;
          a.r_s[idx] = rr;
          a.k_s[idx] = kk;
          a.v_s[idx] = vv;
          nn[p] = num_n;
          dn[p] = den_n;
        }
      }
      if (CHAIN && s_nxt >= 0) {
#pragma unroll
        for (int p = 0; p < PP; ++p) {
          nxt.num[p] = nn[p];
          nxt.den[p] = dn[p];
          nxt.var[p] = vn[p];
        }
      }
      cur = nxt;
      s_cur = s_nxt;
    }
  }
}

// ---------------------------------------------------------------- r,k,v for MANY streams per region
// With >= 32 streams a workgroup of k_region_rkv walks several active streams of its region one after the
// other, each with its six reductions AND the state update on two lanes of every wave (R=512, k=32, S=64:
// 0.42 of the HBM peak against 0.53 at S=8).  Split as in the backward: k_region_rkvG streams the weights and
// leaves the raw r / k / v pre-activations of a batch of streams (input rows requested together), and
// k_region_state finishes every active row elementwise (float4 lanes: sigmoid, time rotation, decay
// fast-forward, WKV / AFD update).  Same dot products, same per-channel arithmetic: bit-identical.
template <int DCH, int PP, int NSB>
__global__ __launch_bounds__(256) void k_region_rkvG(ic_region_args a) {
  const int r = blockIdx.y;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int d = a.d;
  const int row0 = blockIdx.x * (8 * PP) + wave * (2 * PP);
  if (row0 >= d) return;
  const size_t dd = (size_t)d * d;
  const int nz = gridDim.z;
  RowFrag<DCH> wr[2 * PP], wk[2 * PP], wv[2 * PP];
  bool loaded = false;
  for (int sc = blockIdx.z; sc < a.S; sc += 64 * nz) {
    const int sl = sc + lane * nz;
    unsigned long long act = __ballot(sl < a.S && a.mask[sl * a.R + r] != 0);
    if (!act) continue;
    if (!loaded) {  // (wave-uniform)
      const float* Wr = a.W + (size_t)r * 4 * dd;
#pragma unroll
      for (int i = 0; i < 2 * PP; ++i) {
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
      RowFrag<DCH> xv[NSB];
#pragma unroll
      for (int u = 0; u < NSB; ++u) {
        srow[u] = -1;
        zero_frag(xv[u]);
        if (act) {
          const int s = sc + (__ffsll((long long)act) - 1) * nz;
          act &= act - 1;
          srow[u] = s * a.R + r;
          load_row(xv[u], a.x + (size_t)srow[u] * d, d, lane);
        }
      }
#pragma unroll
      for (int u = 0; u < NSB; ++u) {
        if (srow[u] < 0) continue;  // wave-uniform
#pragma unroll
        for (int p = 0; p < PP; ++p) {
          const float pr0 = wave_sum(dot_frag(wr[2 * p], xv[u])), pr1 = wave_sum(dot_frag(wr[2 * p + 1], xv[u]));
          const float pk0 = wave_sum(dot_frag(wk[2 * p], xv[u])), pk1 = wave_sum(dot_frag(wk[2 * p + 1], xv[u]));
          const float pv0 = wave_sum(dot_frag(wv[2 * p], xv[u])), pv1 = wave_sum(dot_frag(wv[2 * p + 1], xv[u]));
          const int ch = row0 + 2 * p + lane;
          if (lane < 2 && ch < d) {
            const size_t idx = (size_t)srow[u] * d + ch;
            a.r_s[idx] = lane ? pr1 : pr0;
            a.k_s[idx] = lane ? pk1 : pk0;
            a.v_s[idx] = lane ? pv1 : pv0;
          }
        }
      }
    }
  }
}

struct StateOut {
  float rr, vv, y, num, den, var, sterm;
};
// one channel of region.py:186-216 (the arithmetic of k_region_rkv's lanes 0/1); vv is already rotated
__device__ __forceinline__ StateOut state_one(bool afd, float steps, float rate, float pn, float on, float pr, float kk,
                                              float vv, float num_in, float den_in, float var_in) {
  StateOut o;
  o.rr = ic_sigmoid(pr);
  o.vv = vv;
  o.var = o.sterm = 0.f;
  const float lam = fmaxf(expf(rate * steps), IC_EPS_DIV);  // region.py:97-98
  const float numd = num_in * lam, dend = den_in * lam;     // region.py:131-133
  const float w = expf(fminf(kk, IC_MAX_EXP));              // region.py:190
  if (afd) {  // region.py:192-212
    const float vard = var_in * (lam * lam) + pn * steps;  // region.py:135
    const float prd = numd / (dend + IC_EPS_DIV);
    const float resid = w * vv - prd;
    const float obs = expf(fminf(-kk, IC_MAX_EXP)) * on + IC_EPS_DIV;
    float gain = vard / (vard + obs);
    gain = fminf(fmaxf(gain, IC_EPS_DIV), 1.0f);  // fp32(1 - 1e-8) == 1
    const float st = prd + gain * resid;
    o.den = dend + w;
    o.num = st * o.den;
    o.var = (1.0f - gain) * vard;
    o.sterm = (resid * resid) * (1.0f / (vard + obs));
    o.y = o.rr * st;
  } else {  // region.py:214-216
    o.num = numd + w * vv;
    o.den = dend + w;
    o.y = o.rr * (o.num / (o.den + IC_EPS_DIV));
  }
  return o;
}

template <int DCH>
__global__ __launch_bounds__(256) void k_region_state(ic_region_args a) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int row = blockIdx.x * 4 + wave;
  if (row >= a.S * a.R) return;
  if (!a.mask[row]) return;
  const int d = a.d, r = row % a.R;
  const size_t off = (size_t)row * d;
  const bool afd = a.flags & IC_F_AFD, vec = a.flags & IC_F_NOISE_VEC;
  RowFrag<DCH> pr, pk, pv, num, den, var, rdv, pnv, onv;
  load_row(pr, a.r_s + off, d, lane);
  load_row(pk, a.k_s + off, d, lane);
  load_row(pv, a.v_s + off, d, lane);
  load_row(num, a.num + off, d, lane);
  load_row(den, a.den + off, d, lane);
  load_row(rdv, a.raw_decay + (size_t)r * d, d, lane);
  zero_frag(var); zero_frag(pnv); zero_frag(onv);
  float pn_s = 0.f, on_s = 0.f;
  if (afd) {
    load_row(var, a.var + off, d, lane);
    if (vec) {
      load_row(pnv, a.pn_param + (size_t)r * d, d, lane);
      load_row(onv, a.on_param + (size_t)r * d, d, lane);
    } else {
      pn_s = ic_noise(a.pn_param[r]);
      on_s = ic_noise(a.on_param[r]);
    }
  }
  const float steps = (float)(a.dt[row] + 1);
  RowFrag<DCH> o_r, o_v, o_y, o_num, o_den, o_var, o_st;
#pragma unroll
  for (int j = 0; j < DCH; ++j) {
    const int ch0 = 256 * j + 4 * lane;
    // time rotation of v (iron_rope.py:63-64): channels (2p, 2p+1) rotate together
    float v0 = pv.v[j].x, v1 = pv.v[j].y, v2 = pv.v[j].z, v3 = pv.v[j].w;
    if (ch0 < 2 * a.m_time) {
      const float th = a.step_pos * a.W_time[ch0 >> 1];  // region.py:186
      const float c = cosf(th), sn = sinf(th);
      const float e = pv.v[j].x * c - pv.v[j].y * sn, o = pv.v[j].x * sn + pv.v[j].y * c;
      v0 = e; v1 = o;
    }
    if (ch0 + 2 < 2 * a.m_time) {
      const float th = a.step_pos * a.W_time[(ch0 >> 1) + 1];
      const float c = cosf(th), sn = sinf(th);
      const float e = pv.v[j].z * c - pv.v[j].w * sn, o = pv.v[j].z * sn + pv.v[j].w * c;
      v2 = e; v3 = o;
    }
#define IC_ST(c, vrot) state_one(afd, steps, -ic_softplus(rdv.v[j].c), vec ? ic_noise(pnv.v[j].c) : pn_s, \
                                 vec ? ic_noise(onv.v[j].c) : on_s, pr.v[j].c, pk.v[j].c, vrot, num.v[j].c, den.v[j].c, var.v[j].c)
    const StateOut s0 = IC_ST(x, v0), s1 = IC_ST(y, v1), s2 = IC_ST(z, v2), s3 = IC_ST(w, v3);
#undef IC_ST
    o_r.v[j] = make_float4(s0.rr, s1.rr, s2.rr, s3.rr);
    o_v.v[j] = make_float4(s0.vv, s1.vv, s2.vv, s3.vv);
    o_y.v[j] = make_float4(s0.y, s1.y, s2.y, s3.y);
    o_num.v[j] = make_float4(s0.num, s1.num, s2.num, s3.num);
    o_den.v[j] = make_float4(s0.den, s1.den, s2.den, s3.den);
    o_var.v[j] = make_float4(s0.var, s1.var, s2.var, s3.var);
    o_st.v[j] = make_float4(s0.sterm, s1.sterm, s2.sterm, s3.sterm);
  }
  store_row(o_r, a.r_s + off, d, lane);
  store_row(o_v, a.v_s + off, d, lane);
  store_row(o_y, a.y + off, d, lane);
  store_row(o_num, a.num_o + off, d, lane);
  store_row(o_den, a.den_o + off, d, lane);
  if (afd) {
    store_row(o_var, a.var_o + off, d, lane);
    store_row(o_st, a.sterm + off, d, lane);
  }
}

template <int DCH, int PP, bool FUSE, bool CHAIN>
__global__ __launch_bounds__(256) void k_region_rkv(ic_region_args a, float one_minus_alpha) {
  ic_router_args none;  // never read (ROUTED = false)
  region_rkv_body<DCH, PP, FUSE, CHAIN, false>(a, one_minus_alpha, none);
}

template <int DCH, int PP>
__global__ __launch_bounds__(256) void k_region_rkv_routed(ic_region_args a, float one_minus_alpha, ic_router_args ra) {
  region_rkv_body<DCH, PP, true, true, true>(a, one_minus_alpha, ra);
}

// ---------------------------------------------------------------- output projection (+ radial/tangential)
template <int DCH, int PP>
__global__ __launch_bounds__(256) void k_region_out(ic_region_args a) {
  const int r = blockIdx.y;
  if (!region_has_work(a.mask, a.S, a.R, r)) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int d = a.d;
  const int row0 = blockIdx.x * (8 * PP) + wave * (2 * PP);
  if (row0 >= d) return;
  const size_t dd = (size_t)d * d;
  const float* Wo = a.W + ((size_t)r * 4 + 3) * dd;
  RowFrag<DCH> wo[2 * PP];
#pragma unroll
  for (int i = 0; i < 2 * PP; ++i) {
    if (row0 + i < d) load_row(wo[i], Wo + (size_t)(row0 + i) * d, d, lane);
    else zero_frag(wo[i]);
  }
  const bool rt = a.flags & IC_F_RT, chain = a.flags & IC_F_CHAIN;
  const float rpn = rt ? ic_noise(a.rpn_param[r]) : 0.f, ron = rt ? ic_noise(a.ron_param[r]) : 0.f;
  // IC_F_CHAIN: the radius state is carried from one active stream to the next in registers
  float c_rad = (rt && chain) ? a.radius[r] : 0.f, c_rvar = (rt && chain) ? a.radius_var[r] : 0.f;
  struct Ops {
    RowFrag<DCH> yv;
    float xres[PP];
    float rad, rvar;
  };
  auto request = [&](int s, Ops& o) {
    const int row = s * a.R + r;
    load_row(o.yv, a.y + (size_t)row * d, d, lane);
    o.rad = o.rvar = 0.f;
    if (rt && !chain) {
      o.rad = a.radius[row];
      o.rvar = a.radius_var[row];
    }
#pragma unroll
    for (int p = 0; p < PP; ++p) {
      const int ch = row0 + 2 * p + lane;
      o.xres[p] = (lane < 2 && ch < d) ? a.x[(size_t)row * d + ch] : 0.f;
    }
  };
  const int nz = gridDim.z;
  for (int sc = blockIdx.z; sc < a.S; sc += 64 * nz) {
    const int sl = sc + lane * nz;
    unsigned long long act = __ballot(sl < a.S && a.mask[sl * a.R + r] != 0);
    if (!act) continue;
    Ops cur, nxt;
    int s_cur = sc + (__ffsll((long long)act) - 1) * nz;
    act &= act - 1;
    request(s_cur, cur);
    while (s_cur >= 0) {
      const int s_nxt = act ? sc + (__ffsll((long long)act) - 1) * nz : -1;
      act &= act - 1;
      if (s_nxt >= 0) request(s_nxt, nxt);
      const int row = s_cur * a.R + r;
      RowFrag<DCH>& yv = cur.yv;
      float scale = 1.0f;
      if (rt) {  // region.py:219-235
        const float nrm = fmaxf(sqrtf(wave_sum(dot_frag(yv, yv))), IC_EPS_DIV);
#pragma unroll
        for (int j = 0; j < DCH; ++j) {
          yv.v[j].x /= nrm; yv.v[j].y /= nrm; yv.v[j].z /= nrm; yv.v[j].w /= nrm;
        }
        const float rad = chain ? c_rad : cur.rad;
        const float pvar = (chain ? c_rvar : cur.rvar) + rpn;
        const float resid = nrm - rad;
        const float gain = pvar / (pvar + ron + IC_EPS_DIV);
        const float upd = rad + gain * resid;
        const float mu = (float)0.2, omu = (float)(1.0 - 0.2);
        scale = omu * rad + mu * upd;
        c_rad = scale;
        c_rvar = (1.0f - gain) * pvar;
        if (blockIdx.x == 0 && wave == 0) {
          const int trow = chain ? r : row;  // telemetry of the one state stream
          if (a.last_dir) store_row(yv, a.last_dir + (size_t)trow * d, d, lane);
          if (lane == 0) {
            a.radius_o[row] = scale;
            a.radius_var_o[row] = c_rvar;
            a.ynorm[row] = nrm;
            if (a.last_norm) a.last_norm[trow] = nrm;
          }
        }
      }
#pragma unroll
      for (int p = 0; p < PP; ++p) {
        const float h0 = wave_sum(dot_frag(wo[2 * p], yv)), h1 = wave_sum(dot_frag(wo[2 * p + 1], yv));
        const int ch = row0 + 2 * p + lane;
        if (lane < 2 && ch < d) {
          const float hd = lane ? h1 : h0;
          a.hpre[(size_t)row * d + ch] = cur.xres[p] + (rt ? scale * hd : hd);
        }
      }
      cur = nxt;
      s_cur = s_nxt;
    }
  }
}

}  // namespace

namespace {

// ---------------------------------------------------------------- finish (wave per row)
template <int DCH>
__global__ __launch_bounds__(256) void k_region_fin(ic_region_args a, float one_minus_alpha) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int row = blockIdx.x * 4 + wave;
  if (row >= a.S * a.R) return;
  const int d = a.d;
  const size_t off = (size_t)row * d;
  const bool afd = a.flags & IC_F_AFD, rt = a.flags & IC_F_RT, pred = a.flags & IC_F_PRED;
  RowFrag<DCH> z;
  zero_frag(z);
  const bool dead = a.dead && *a.dead != 0;  // pass-through launch (ic_region_args.dead)
  if (!dead && a.mask[row]) {
    RowFrag<DCH> h, out, g;
    load_row(h, a.hpre + off, d, lane);
    if (a.flags & IC_F_NOKWTA) {
      out = h;
#pragma unroll
      for (int j = 0; j < DCH; ++j) g.v[j] = make_float4(1.f, 1.f, 1.f, 1.f);
    } else {
      kwta_row(h, out, g, d, a.kwta_k, (a.flags & IC_F_KWTA_SOFT) != 0, a.kwta_temp);
    }
    store_row(out, a.H + off, d, lane);
    store_row(g, a.gate + off, d, lane);
    if (pred) store_row(z, a.pred_o + off, d, lane);  // region.py:175
    if (afd) {                                         // region.py:207-211
      RowFrag<DCH> st;
      load_row(st, a.sterm + off, d, lane);
      float sum = 0.f;
#pragma unroll
      for (int j = 0; j < DCH; ++j) sum += (st.v[j].x + st.v[j].y) + (st.v[j].z + st.v[j].w);
      sum = wave_sum(sum);
      if (lane == 0)
        a.surprise_o[row] = (float)0.99 * a.surprise[row] + (float)(1.0 - 0.99) * sum;
    }
    if (lane == 0) a.dt_o[row] = 0;  // region.py:179
  } else {
    store_row(z, a.H + off, d, lane);  // model.py:275
    RowFrag<DCH> t;
    if (pred) {  // region.py:146
      RowFrag<DCH> p, m;
      load_row(p, a.pred + off, d, lane);
      load_row(m, a.M + off, d, lane);
      const float al = (float)a.pred_alpha;
#pragma unroll
      for (int j = 0; j < DCH; ++j) {
        t.v[j].x = al * p.v[j].x + one_minus_alpha * m.v[j].x;
        t.v[j].y = al * p.v[j].y + one_minus_alpha * m.v[j].y;
        t.v[j].z = al * p.v[j].z + one_minus_alpha * m.v[j].z;
        t.v[j].w = al * p.v[j].w + one_minus_alpha * m.v[j].w;
      }
      store_row(dead ? p : t, a.pred_o + off, d, lane);
    }
    load_row(t, a.num + off, d, lane);
    store_row(t, a.num_o + off, d, lane);
    load_row(t, a.den + off, d, lane);
    store_row(t, a.den_o + off, d, lane);
    if (afd) {
      load_row(t, a.var + off, d, lane);
      store_row(t, a.var_o + off, d, lane);
    }
    if (lane == 0) {
      a.dt_o[row] = a.dt[row] + (dead ? 0 : 1);  // region.py:137-138
      if (afd) a.surprise_o[row] = a.surprise[row];
      if (rt) {
        a.radius_o[row] = a.radius[row];
        a.radius_var_o[row] = a.radius_var[row];
      }
    }
  }
}

// IC_F_CHAIN finish: one wave per REGION walks the streams in order and carries the trace, dt,
// surprise EMA and (for the pass-through copies of inactive streams) the source of the state rows.
template <int DCH>
__global__ __launch_bounds__(256) void k_region_fin_chain(ic_region_args a, float one_minus_alpha) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = blockIdx.x * 4 + wave;
  if (r >= a.R) return;
  const int d = a.d;
  const bool afd = a.flags & IC_F_AFD, rt = a.flags & IC_F_RT, pred = a.flags & IC_F_PRED;
  RowFrag<DCH> z, p;
  zero_frag(z);
  zero_frag(p);
  if (pred) load_row(p, a.pred + (size_t)r * d, d, lane);
  int dtv = a.dt[r];
  float surp = afd ? a.surprise[r] : 0.f;
  float rad = rt ? a.radius[r] : 0.f, rvar = rt ? a.radius_var[r] : 0.f;
  // rows written by an EARLIER launch that hold the current state (never this wave's own copies)
  const float* num_src = a.num + (size_t)r * d;
  const float* den_src = a.den + (size_t)r * d;
  const float* var_src = afd ? a.var + (size_t)r * d : nullptr;
  const float al = (float)a.pred_alpha;
  // the rows a stream needs that do not depend on the chain (hpre + surprise terms when active,
  // the message row when not) are requested one stream ahead
  struct Pre {
    RowFrag<DCH> h, st, m;
    bool on;
  };
  const bool dead = a.dead && *a.dead != 0;  // pass-through launch (ic_region_args.dead): nothing active, nothing moves
  auto fetch = [&](int s, Pre& p) {
    const int row = s * a.R + r;
    const size_t off = (size_t)row * d;
    p.on = !dead && a.mask[row] != 0;
    if (p.on) {
      load_row(p.h, a.hpre + off, d, lane);
      if (afd) load_row(p.st, a.sterm + off, d, lane);
    } else if (pred) {
      load_row(p.m, a.M + off, d, lane);
    }
  };
  Pre cur, nxt;
  fetch(0, cur);
  for (int s = 0; s < a.S; ++s) {
    const int row = s * a.R + r;
    const size_t off = (size_t)row * d;
    if (s + 1 < a.S) fetch(s + 1, nxt);
    if (cur.on) {
      RowFrag<DCH> h = cur.h, out, g;
      if (a.flags & IC_F_NOKWTA) {
        out = h;
#pragma unroll
        for (int j = 0; j < DCH; ++j) g.v[j] = make_float4(1.f, 1.f, 1.f, 1.f);
      } else {
        kwta_row(h, out, g, d, a.kwta_k, (a.flags & IC_F_KWTA_SOFT) != 0, a.kwta_temp);
      }
      store_row(out, a.H + off, d, lane);
      store_row(g, a.gate + off, d, lane);
      if (pred) {
        store_row(z, a.pred_o + off, d, lane);
        p = z;
      }
      if (afd) {
        const RowFrag<DCH>& st = cur.st;
        float sum = 0.f;
#pragma unroll
        for (int j = 0; j < DCH; ++j) sum += (st.v[j].x + st.v[j].y) + (st.v[j].z + st.v[j].w);
        sum = wave_sum(sum);
        surp = (float)0.99 * surp + (float)(1.0 - 0.99) * sum;
        if (lane == 0) a.surprise_o[row] = surp;
        var_src = a.var_o + off;
      }
      dtv = 0;
      if (lane == 0) a.dt_o[row] = 0;
      num_src = a.num_o + off;
      den_src = a.den_o + off;
      if (rt) {
        rad = a.radius_o[row];
        rvar = a.radius_var_o[row];
      }
    } else {
      store_row(z, a.H + off, d, lane);
      RowFrag<DCH> t;
      if (pred) {
        const RowFrag<DCH>& m = cur.m;
        if (!dead) {
#pragma unroll
          for (int j = 0; j < DCH; ++j) {
            p.v[j].x = al * p.v[j].x + one_minus_alpha * m.v[j].x;
            p.v[j].y = al * p.v[j].y + one_minus_alpha * m.v[j].y;
            p.v[j].z = al * p.v[j].z + one_minus_alpha * m.v[j].z;
            p.v[j].w = al * p.v[j].w + one_minus_alpha * m.v[j].w;
          }
        }
        store_row(p, a.pred_o + off, d, lane);
      }
      load_row(t, num_src, d, lane);
      store_row(t, a.num_o + off, d, lane);
      load_row(t, den_src, d, lane);
      store_row(t, a.den_o + off, d, lane);
      if (afd) {
        load_row(t, var_src, d, lane);
        store_row(t, a.var_o + off, d, lane);
      }
      if (!dead) dtv += 1;
      if (lane == 0) {
        a.dt_o[row] = dtv;
        if (afd) a.surprise_o[row] = surp;
        if (rt) {
          a.radius_o[row] = rad;
          a.radius_var_o[row] = rvar;
        }
      }
    }
    cur = nxt;
  }
}

template <int DCH>
__global__ __launch_bounds__(256) void k_kwta_rows(const float* x, float* y, float* gate, int rows,
                                                   int d, int k, int soft, float temp) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int row = blockIdx.x * 4 + wave;
  if (row >= rows) return;
  RowFrag<DCH> h, out, g;
  load_row(h, x + (size_t)row * d, d, lane);
  kwta_row(h, out, g, d, k, soft != 0, temp);
  store_row(out, y + (size_t)row * d, d, lane);
  if (gate) store_row(g, gate + (size_t)row * d, d, lane);
}

}  // namespace

static int check_region_args(const ic_region_args* a, const char* who) {
  IC_REQUIRE(a != nullptr, "%s: NULL args", who);
  IC_REQUIRE(a->S > 0 && a->R > 0, "%s: S=%d R=%d must be positive", who, a->S, a->R);
  IC_REQUIRE(a->d >= 4 && a->d % 4 == 0 && a->d <= 1024,
             "%s: d=%d unsupported (need multiple of 4 in [4,1024])", who, a->d);
  IC_REQUIRE(a->m_time >= 0 && 2 * a->m_time <= a->d, "%s: m_time=%d out of range", who, a->m_time);
  IC_REQUIRE(a->W && a->norm_w && a->raw_decay && a->M && a->mask && a->num && a->den && a->dt,
             "%s: NULL parameter/input pointer", who);
  IC_REQUIRE(a->m_time == 0 || a->W_time, "%s: W_time is NULL with m_time=%d", who, a->m_time);
  IC_REQUIRE(!(a->flags & IC_F_NORM_IN) || a->norm_in_w, "%s: norm_in_w is NULL", who);
  IC_REQUIRE(!(a->flags & IC_F_PRED) || (a->pred && a->pred_o), "%s: pred buffers are NULL", who);
  IC_REQUIRE(!(a->flags & IC_F_AFD) || (a->var && a->var_o && a->pn_param && a->on_param &&
                                        a->surprise && a->surprise_o && a->sterm),
             "%s: AFD buffers are NULL", who);
  IC_REQUIRE(!(a->flags & IC_F_RT) || (a->radius && a->radius_var && a->radius_o &&
                                       a->radius_var_o && a->rpn_param && a->ron_param && a->ynorm),
             "%s: radial-tangential buffers are NULL", who);
  IC_REQUIRE(a->x && a->u1 && a->inv_s && a->r_s && a->k_s && a->v_s && a->y && a->hpre && a->gate,
             "%s: NULL workspace pointer", who);
  return IC_OK;
}

// ra != nullptr: routed r/k/v kernel (the caller checked that the path applies)
static int region_phase_launch(const ic_region_args* a, const ic_router_args* ra, ic_stream_t stream) {
  int rc = check_region_args(a, "ic_region_phase_fwd");
  if (rc) return rc;
  IC_REQUIRE(a->H && a->num_o && a->den_o && a->dt_o, "ic_region_phase_fwd: NULL output pointer");
  hipStream_t st = (hipStream_t)stream;
  const int rows = a->S * a->R;
  const dim3 blk(256);
  const dim3 grow(ic_ceil_div(rows, 4));
  // rows per workgroup in the GEMV kernels: 8 (many small workgroups: best when few regions are
  // active and the launch is latency-bound), 16 or 32 (deeper per-wave load queues: best when the
  // launch streams hundreds of MiB).  rows_fwd = 0 lets the library choose from the problem size.
  // Measured on MI355X (profiles/, R=128/512, S=1/8/64): the r/k/v kernel is best at 8 rows at every size
  // (3 matrices x 8 rows already keep 24 KB per workgroup in flight); the output projection, one
  // matrix, wants 16 rows from R > 64 and 32 from R >= 256 (28 -> 22 us at R=512, S=8).
  int rpb = a->rows_fwd, rpb_out = a->rows_fwd;
  IC_REQUIRE(rpb == 0 || rpb == 8 || rpb == 16 || rpb == 32, "ic_region_phase_fwd: rows_fwd=%d (need 0, 8, 16 or 32)", rpb);
  if (rpb == 0) {
    rpb = 8;
    rpb_out = a->R >= 256 ? 32 : (a->R > 64 ? 16 : 8);
  }
  if (a->d > 256 && rpb > 16) rpb = 16;  // register budget at DCH >= 2
  if (a->d > 512 && rpb > 8) rpb = 8;    // DCH = 4: 16 rows of three matrices would spill
  if (a->d > 256 && rpb_out > 16) rpb_out = 16;
  // streams over gridDim.z when the slices x regions alone leave most of the chip idle (never for a state chain)
  int nz = 1;
  if (!(a->flags & IC_F_CHAIN) && !ra)
    while (nz < a->S && nz < 16 && ic_ceil_div(a->d, rpb) * a->R * nz < 1024) nz *= 2;
  const dim3 gmat(ic_ceil_div(a->d, rpb), a->R, nz), gout(ic_ceil_div(a->d, rpb_out), a->R, nz);
  const float oma = (float)(1.0 - a->pred_alpha);
  const bool chain = a->flags & IC_F_CHAIN;
  IC_REQUIRE(!chain || a->S <= 64, "ic_region_phase_fwd: IC_F_CHAIN needs S <= 64 (S=%d)", a->S);
  const bool fuse = chain || (a->S <= 2 && a->R <= 64);  // prologue inside rkv (measured a loss at R = 512)
#define IC_RKV_LAUNCH(PP)                                                                                   \
  IC_DISPATCH_DCH(a->d, {                                                                                   \
    if (chain) {                                                                                            \
      IC_TIMED(IC_KID_region_rkv, st, hipLaunchKernelGGL((k_region_rkv<DCH, PP, true, true>), gmat, blk, 0, st, *a, oma));  \
    } else if (fuse) {                                                                                      \
      IC_TIMED(IC_KID_region_rkv, st, hipLaunchKernelGGL((k_region_rkv<DCH, PP, true, false>), gmat, blk, 0, st, *a, oma));  \
    } else {                                                                                                \
      IC_TIMED(IC_KID_region_pro, st, hipLaunchKernelGGL(k_region_pro<DCH>, grow, blk, 0, st, *a));         \
      IC_TIMED(IC_KID_region_rkv, st, hipLaunchKernelGGL((k_region_rkv<DCH, PP, false, false>), gmat, blk, 0, st, *a, oma)); \
    }                                                                                                       \
  })
#define IC_OUT_LAUNCH(PP)                                                                                   \
  IC_DISPATCH_DCH(a->d, {                                                                                   \
    IC_TIMED(IC_KID_region_out, st, hipLaunchKernelGGL((k_region_out<DCH, PP>), gout, blk, 0, st, *a));     \
  })
  // IC_FWD_SPLIT=0/1 forces the fused / the split r,k,v path (A/B timing; same results bit for bit)
  static const int fsplit_env = [] { const char* e = getenv("IC_FWD_SPLIT"); return (e && e[0]) ? (e[0] == '0' ? 0 : 1) : -1; }();
  const bool fsplit = !chain && !fuse && !ra && rpb == 8 && (fsplit_env >= 0 ? fsplit_env == 1 : a->S / nz >= 32);
  if (ra) {
    IC_TIMED(IC_KID_region_rkv, st, hipLaunchKernelGGL((k_region_rkv_routed<1, 1>), gmat, blk, 0, st, *a, oma, *ra));
  } else if (fsplit) {
    IC_DISPATCH_DCH(a->d, {
      IC_TIMED(IC_KID_region_pro, st, hipLaunchKernelGGL(k_region_pro<DCH>, grow, blk, 0, st, *a));
      IC_TIMED(IC_KID_region_rkv, st, hipLaunchKernelGGL((k_region_rkvG<DCH, 1, (DCH <= 2 ? 4 : 2)>), gmat, blk, 0, st, *a));
      IC_TIMED(IC_KID_region_state, st, hipLaunchKernelGGL(k_region_state<DCH>, grow, blk, 0, st, *a));
    });
  } else if (rpb == 8) IC_RKV_LAUNCH(1);
  else if (rpb == 16) IC_RKV_LAUNCH(2);
  else IC_RKV_LAUNCH(4);
  if (rpb_out == 8) IC_OUT_LAUNCH(1);
  else if (rpb_out == 16) IC_OUT_LAUNCH(2);
  else IC_OUT_LAUNCH(4);
#undef IC_RKV_LAUNCH
#undef IC_OUT_LAUNCH
  IC_DISPATCH_DCH(a->d, {
    if (chain) {
      IC_TIMED(IC_KID_region_fin, st, hipLaunchKernelGGL(k_region_fin_chain<DCH>, dim3(ic_ceil_div(a->R, 4)), blk, 0, st, *a, oma));
    } else {
      IC_TIMED(IC_KID_region_fin, st, hipLaunchKernelGGL(k_region_fin<DCH>, grow, blk, 0, st, *a, oma));
    }
  });
  IC_LAUNCH_CHECK("ic_region_phase_fwd");
  return IC_OK;
}

extern "C" int ic_region_phase_fwd(const ic_region_args* a, ic_stream_t stream) {
  return region_phase_launch(a, nullptr, stream);
}

extern "C" int ic_loop_step_fwd(const ic_gate_args* g, const ic_router_args* ra, const ic_region_args* a,
                                ic_stream_t stream) {
  IC_REQUIRE(g != nullptr && ra != nullptr && a != nullptr, "ic_loop_step_fwd: NULL args");
  // the combine rides in the region kernel when it is the wave-per-destination one (precision
  // routing, d <= 256, in-degree <= 8 as for ic_router_fwd) and the streams are one state chain
  const bool routed = (ra->flags & IC_RF_PRECISION) && ra->E > 0 && g->R <= 64 && a->d <= 256 && a->d >= 32 &&
                      a->d % 8 == 0 && (a->flags & IC_F_CHAIN) && a->S <= kRoutedStreams &&
                      (a->rows_fwd == 0 || a->rows_fwd == 8);
  if (!routed) {
    int rc = ic_front_fwd(g, ra, stream);
    return rc ? rc : ic_region_phase_fwd(a, stream);
  }
  IC_REQUIRE(ra->S == a->S && ra->R == a->R && ra->d == a->d, "ic_loop_step_fwd: router (S=%d,R=%d,d=%d) and region (S=%d,R=%d,d=%d) disagree",
             ra->S, ra->R, ra->d, a->S, a->R, a->d);
  IC_REQUIRE(ra->M == a->M && g->mask == a->mask, "ic_loop_step_fwd: the region phase must read the M and mask this call's router and gate write");
  int rc = ic_front_launch(g, ra, stream, false);
  return rc ? rc : region_phase_launch(a, ra, stream);
}

extern "C" int ic_kwta_fwd(const float* x, float* y, float* gate, int32_t rows, int32_t d, int32_t k,
                           int32_t soft, float temp, ic_stream_t stream) {
  IC_REQUIRE(x && y, "ic_kwta_fwd: NULL pointer");
  IC_REQUIRE(rows >= 0, "ic_kwta_fwd: rows=%d", rows);
  IC_REQUIRE(d >= 4 && d % 4 == 0 && d <= 1024, "ic_kwta_fwd: d=%d unsupported", d);
  if (rows == 0) return IC_OK;
  IC_DISPATCH_DCH(d, {
    IC_TIMED(IC_KID_kwta_rows, (hipStream_t)stream, hipLaunchKernelGGL(k_kwta_rows<DCH>, dim3(ic_ceil_div(rows, 4)), dim3(256), 0,
                       (hipStream_t)stream, x, y, gate, rows, d, k, soft, temp));
  });
  IC_LAUNCH_CHECK("ic_kwta_fwd");
  return IC_OK;
}
