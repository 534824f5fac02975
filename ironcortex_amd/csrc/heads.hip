// Loop heads, forward only: workspace write, token logits, steering statistics, vector MLPs.
// Replaces the per-inner-step ATen calls of reference model.py:296-321 (heads), :371-386
// (planner / critic) and :425-438 (verifier energy + branch score) when no gradient is needed.
// All of it is latency-bound vector work (<= 2.5 MB of weights per step): the kernels are
// written for few launches and coalesced row reads, not for bandwidth.
#include <mutex>

#include "ic_common.h"
#include "ic_kwta.h"

namespace {

// maps of up to this many regions walk all R rows where a sum over the ACTIVE regions is formed (two round trips at
// R = 10); larger ones find the active rows by ballot first (R = 512: 64 dependent round trips became 8 + n_active / 8)
constexpr int kSmallR = 16;

constexpr int kMaxCols = 4096;  // longest input vector a GEMV stages in LDS

__device__ __forceinline__ float block_sum(float v, float* red) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  v = wave_sum(v);
  __syncthreads();
  if (lane == 0) red[wave] = v;
  __syncthreads();
  return (red[0] + red[1]) + (red[2] + red[3]);
}
__device__ __forceinline__ float block_max(float v, float* red) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  v = wave_max(v);
  __syncthreads();
  if (lane == 0) red[wave] = v;
  __syncthreads();
  return fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

__device__ __forceinline__ float apply_act(float v, int act) {
  if (act == IC_ACT_GELU) return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
  if (act == IC_ACT_SILU) return v / (1.0f + expf(-v));
  return v;
}

// Descriptor of batch b of a batched GEMV (ic_gemv_desc.nbatch > 1).
__device__ __forceinline__ ic_gemv_desc gemv_batch(ic_gemv_desc g, int b) {
  if (b == 0) return g;
  if (g.x0) g.x0 += (int64_t)b * g.x0_bs;
  if (g.x1) g.x1 += (int64_t)b * g.x1_bs;
  if (g.x2) g.x2 += (int64_t)b * g.x2_bs;
  g.y += (int64_t)b * g.y_bs;
  return g;
}

// One wave per output row, 8 rows per workgroup; the input vector is staged (and optionally
// layer-normalised) in LDS once per workgroup.
__global__ __launch_bounds__(256) void k_gemv(ic_gemv_desc g0, const float* stats, float* score) {
  const ic_gemv_desc g = gemv_batch(g0, blockIdx.y);
  if (score) {
    stats += 4 * blockIdx.y;
    score += blockIdx.y;
  }
  __shared__ float xs[kMaxCols];
  __shared__ float red[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int cols = g.n0 + g.n1 + g.n2;
  for (int c = threadIdx.x; c < cols; c += 256) {
    float v;
    if (c < g.n0) v = g.x0[c];
    else if (c < g.n0 + g.n1) v = g.x1[c - g.n0];
    else v = g.x2[c - g.n0 - g.n1];
    xs[c] = v;
  }
  __syncthreads();
  if (g.ln_g) {  // torch.nn.LayerNorm: biased variance
    float s = 0.f;
    for (int c = threadIdx.x; c < cols; c += 256) s += xs[c];
    const float mean = block_sum(s, red) / (float)cols;
    float q = 0.f;
    for (int c = threadIdx.x; c < cols; c += 256) {
      const float t = xs[c] - mean;
      q += t * t;
    }
    const float rstd = 1.0f / sqrtf(block_sum(q, red) / (float)cols + g.ln_eps);
    for (int c = threadIdx.x; c < cols; c += 256) xs[c] = (xs[c] - mean) * rstd * g.ln_g[c] + g.ln_b[c];
    __syncthreads();
  }
  for (int i = 0; i < 2; ++i) {
    const int row = blockIdx.x * 8 + wave * 2 + i;
    if (row >= g.rows) break;
    const float* w = g.W + (size_t)row * cols;
    float acc = 0.f;
    for (int c = lane; c < cols; c += 64) acc = fmaf(w[c], xs[c], acc);
    acc = wave_sum(acc);
    if (lane == 0) {
      const float out = apply_act(acc + (g.b ? g.b[row] : 0.f), g.act);
      g.y[row] = out;
      // the verifier's last layer: energy -> branch score (model.py:431-438)
      if (score && row == 0) score[0] = 0.6f * stats[2] + 0.2f * (-out) + 0.2f * stats[1];
    }
  }
}

// Register variant for inputs of up to 64*NX columns: every wave loads the input vector itself
// (lane-strided, L2 hits) TOGETHER with its two weight rows, and computes the LayerNorm
// statistics itself -- no LDS staging, no workgroup barrier, one memory round trip.
template <int NX>
__device__ __forceinline__ void gemv_reg_body(const ic_gemv_desc& g0, const float* stats, float* score) {
  const ic_gemv_desc g = gemv_batch(g0, blockIdx.y);
  if (score) {
    stats += 4 * blockIdx.y;
    score += blockIdx.y;
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int cols = g.n0 + g.n1 + g.n2;
  const int row0 = blockIdx.x * 8 + wave * 2;
  if (row0 >= g.rows) return;
  const bool two = row0 + 1 < g.rows;
  float xr[NX], w0[NX], w1[NX], lg[NX], lb[NX];
#pragma unroll
  for (int i = 0; i < NX; ++i) {
    const int c = lane + 64 * i;
    float v = 0.f;
    if (c < cols) {
      if (c < g.n0) v = g.x0[c];
      else if (c < g.n0 + g.n1) v = g.x1[c - g.n0];
      else v = g.x2[c - g.n0 - g.n1];
    }
    xr[i] = v;
    w0[i] = c < cols ? g.W[(size_t)row0 * cols + c] : 0.f;
    w1[i] = (two && c < cols) ? g.W[(size_t)(row0 + 1) * cols + c] : 0.f;
    lg[i] = (g.ln_g && c < cols) ? g.ln_g[c] : 0.f;
    lb[i] = (g.ln_g && c < cols) ? g.ln_b[c] : 0.f;
  }
  const float b0 = g.b ? g.b[row0] : 0.f, b1 = (g.b && two) ? g.b[row0 + 1] : 0.f;
  if (g.ln_g) {  // torch.nn.LayerNorm: biased variance
    float sm = 0.f;
#pragma unroll
    for (int i = 0; i < NX; ++i) sm += xr[i];
    const float mean = wave_sum(sm) / (float)cols;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      const float t = (lane + 64 * i < cols) ? xr[i] - mean : 0.f;
      q += t * t;
    }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)cols + g.ln_eps);
#pragma unroll
    for (int i = 0; i < NX; ++i) xr[i] = (lane + 64 * i < cols) ? (xr[i] - mean) * rstd * lg[i] + lb[i] : 0.f;
  }
  float a0 = 0.f, a1 = 0.f;
#pragma unroll
  for (int i = 0; i < NX; ++i) {
    a0 = fmaf(w0[i], xr[i], a0);
    a1 = fmaf(w1[i], xr[i], a1);
  }
  a0 = wave_sum(a0);
  a1 = wave_sum(a1);
  if (lane == 0) {
    const float o0 = apply_act(a0 + b0, g.act);
    g.y[row0] = o0;
    if (two) g.y[row0 + 1] = apply_act(a1 + b1, g.act);
    // the verifier's last layer: energy -> branch score (model.py:431-438)
    if (score && row0 == 0) score[0] = 0.6f * stats[2] + 0.2f * (-o0) + 0.2f * stats[1];
  }
}

template <int NX>
__global__ __launch_bounds__(256) void k_gemv_reg(ic_gemv_desc g0, const float* stats, float* score) {
  gemv_reg_body<NX>(g0, stats, score);
}

// Up to 4 independent GEMVs in one launch: blockIdx.z selects the descriptor.
struct MultiArgs {
  ic_gemv_desc d[4];
};
template <int NX>
__global__ __launch_bounds__(256) void k_gemv_multi(MultiArgs m, const float* stats, float* score) {
  const ic_gemv_desc& g0 = m.d[blockIdx.z];
  if ((int)blockIdx.x * 8 >= g0.rows || (int)blockIdx.y >= (g0.nbatch > 1 ? g0.nbatch : 1)) return;
  gemv_reg_body<NX>(g0, blockIdx.z == 0 ? stats : nullptr, blockIdx.z == 0 ? score : nullptr);
}

// A whole chain of small GEMVs in ONE workgroup (16 waves): layer i+1 reads layer i's output
// after a workgroup barrier, so a 3-6 layer MLP costs one launch instead of one per layer.
// At <= a few MB of (cache-resident) weights the launch gaps dominate, not the single CU's
// bandwidth.  Optional epilogue: the branch score of model.py:431-438 from the last output.
struct ChainArgs {
  ic_gemv_desc d[6];
  int n;
  const float* stats;
  float* score;
};

__device__ __forceinline__ float block_sum16(float v, float* red) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  v = wave_sum(v);
  __syncthreads();
  if (lane == 0) red[wave] = v;
  __syncthreads();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += red[i];
  return s;
}

__global__ __launch_bounds__(1024) void k_gemv_chain1(ChainArgs c) {
  __shared__ __align__(16) float xs[kMaxCols];
  __shared__ float red[16];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  constexpr int RPI = 8;  // rows in flight per wave
  for (int L = 0; L < c.n; ++L) {
    const ic_gemv_desc& g = c.d[L];
    const int cols = g.n0 + g.n1 + g.n2;
    for (int i = threadIdx.x; i < cols; i += 1024) {
      float v;
      if (i < g.n0) v = g.x0[i];
      else if (i < g.n0 + g.n1) v = g.x1[i - g.n0];
      else v = g.x2[i - g.n0 - g.n1];
      xs[i] = v;
    }
    __syncthreads();
    if (g.ln_g) {
      float s = 0.f;
      for (int i = threadIdx.x; i < cols; i += 1024) s += xs[i];
      const float mean = block_sum16(s, red) / (float)cols;
      float q = 0.f;
      for (int i = threadIdx.x; i < cols; i += 1024) {
        const float t = xs[i] - mean;
        q += t * t;
      }
      const float rstd = 1.0f / sqrtf(block_sum16(q, red) / (float)cols + g.ln_eps);
      for (int i = threadIdx.x; i < cols; i += 1024) xs[i] = (xs[i] - mean) * rstd * g.ln_g[i] + g.ln_b[i];
      __syncthreads();
    }
    const bool vec4 = (cols & 3) == 0 && ((reinterpret_cast<uintptr_t>(g.W) & 15) == 0);
    for (int r0 = wave * RPI; r0 < g.rows; r0 += 16 * RPI) {
      float acc[RPI];
#pragma unroll
      for (int i = 0; i < RPI; ++i) acc[i] = 0.f;
      if (vec4) {
        for (int c4 = lane; c4 < cols / 4; c4 += 64) {
          const float4 xv = *reinterpret_cast<const float4*>(xs + 4 * c4);
#pragma unroll
          for (int i = 0; i < RPI; ++i) {
            if (r0 + i < g.rows) {
              const float4 w = *reinterpret_cast<const float4*>(g.W + (size_t)(r0 + i) * cols + 4 * c4);
              acc[i] = fmaf(w.x, xv.x, acc[i]);
              acc[i] = fmaf(w.y, xv.y, acc[i]);
              acc[i] = fmaf(w.z, xv.z, acc[i]);
              acc[i] = fmaf(w.w, xv.w, acc[i]);
            }
          }
        }
      } else {
        for (int cc = lane; cc < cols; cc += 64) {
          const float xv = xs[cc];
#pragma unroll
          for (int i = 0; i < RPI; ++i)
            if (r0 + i < g.rows) acc[i] = fmaf(g.W[(size_t)(r0 + i) * cols + cc], xv, acc[i]);
        }
      }
#pragma unroll
      for (int i = 0; i < RPI; ++i) {
        const float t = wave_sum(acc[i]);
        if (lane == 0 && r0 + i < g.rows) g.y[r0 + i] = apply_act(t + (g.b ? g.b[r0 + i] : 0.f), g.act);
      }
    }
    __threadfence_block();
    __syncthreads();
  }
  if (c.score && threadIdx.x == 0) {
    const float energy = c.d[c.n - 1].y[0];
    c.score[0] = 0.6f * c.stats[2] + 0.2f * (-energy) + 0.2f * c.stats[1];
  }
}

// ONE launch for the heads of an inner step.  blockIdx.y selects the work:
//   0            z = [gate; facets] . H[motor] + bias                    (heads.py:96-119)
//   1 + r        region head r on H[r], active regions other than motor  (model.py:311-318)
//   1 + R        slots_out[j] = KWTA(slots_in[j] + H[j-th active region]) (model.py:300-306, heads.py:34-36)
// and the last workgroup to finish combines: logits, softmax, steering statistics.
template <int DCH>
__device__ __forceinline__ void ws_write_rows(const ic_heads_args& a, int lane, int wave, const int xb) {
  const int j = xb * 4 + wave;
  if (j >= a.n_slots) return;
  int found = -1, seen = 0;  // index of the (j+1)-th set bit of mask
  for (int base = 0; base < a.R && found < 0; base += 64) {
    const int r = base + lane;
    const bool on = r < a.R && a.mask[r] != 0;
    const unsigned long long bal = __ballot(on);
    const int cnt = __popcll(bal);
    if (seen + cnt > j) {
      unsigned long long m = bal;
      for (int t = 0; t < j - seen; ++t) m &= m - 1;  // drop the lowest set bits
      found = base + __ffsll((long long)m) - 1;
    }
    seen += cnt;
  }
  const int d = a.d;
  RowFrag<DCH> s, h, out, g;
  load_row(s, a.slots_in + (size_t)j * d, d, lane);
  if (found >= 0) {
    load_row(h, a.H + (size_t)found * d, d, lane);
#pragma unroll
    for (int c = 0; c < DCH; ++c) {
      s.v[c].x += h.v[c].x; s.v[c].y += h.v[c].y; s.v[c].z += h.v[c].z; s.v[c].w += h.v[c].w;
    }
  }
  kwta_row(s, out, g, d, a.kwta_k, false, 1.0f);
  store_row(out, a.slots_out + (size_t)j * d, d, lane);
}

template <int DCH>
__device__ __forceinline__ void token_rows(const ic_heads_args& a, int lane, int wave, const int grp, const int xb) {
  const int d = a.d;
  int rows;
  const float *W, *b, *x;
  float* y;
  if (grp == 0) {
    rows = a.Kf + a.Kf * a.V;
    W = a.lm_W - (size_t)a.Kf * d; b = a.lm_b - a.Kf;  // facet row i sits at z[Kf + i]
    x = a.H + (size_t)a.motor_idx * d; y = a.z;
  } else {
    const int r = grp - 1;
    if (r == a.motor_idx || !a.mask[r]) {
      // not part of the mean (model.py:314-317): zeros, so the combine can add all R rows
      // without a mask test in its load chain (x + 0 is exact)
      const int v = xb * 8 + (int)threadIdx.x;  // this block's 8 rows (threads 0..7)
      if (threadIdx.x >= 8) return;
      if (v < a.V) a.per_region[(size_t)r * a.V + v] = 0.f;
      return;
    }
    rows = a.V;
    W = a.rh_W + (size_t)r * a.V * d; b = a.rh_b + (size_t)r * a.V; x = a.H + (size_t)r * d;
    y = a.per_region + (size_t)r * a.V;
  }
  const int row0 = xb * 8 + wave * 2;
  if (row0 >= rows) return;
  RowFrag<DCH> xv, w[2];
  float bias[2] = {0.f, 0.f};
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = row0 + i;
    if (row < rows) {
      const float* wr = W + (size_t)row * d;
      const float* br = b + row;
      if (grp == 0 && row < a.Kf) {  // the MFS gate rows come first in z
        wr = a.lm_gate_W + (size_t)row * d;
        br = a.lm_gate_b + row;
      }
      load_row(w[i], wr, d, lane);
      bias[i] = *br;
    } else {
      zero_frag(w[i]);
    }
  }
  load_row(xv, x, d, lane);
  const float s0 = wave_sum(dot_frag(w[0], xv)), s1 = wave_sum(dot_frag(w[1], xv));
  if (lane == 0) {
    y[row0] = s0 + bias[0];
    if (row0 + 1 < rows) y[row0 + 1] = s1 + bias[1];
  }
}

// mixture-of-facets logits + region heads -> logits, softmax, steering statistics (one workgroup)
// The activity ballots of a stream's regions, 64 regions per ballot, in ONE memory round trip: every lane requests its
// byte of all 16 groups (R <= 1024) before the first ballot is formed -- a loop of "load, ballot, use" is one dependent
// round trip per 64 regions (8 at R = 512, and k_logits_combine went through it once per 256 columns of V).
// lb: this wave's 16 words in LDS (written by lane 0, read back by the same wave: LDS operations of a wave complete
// in order).  Bit l of lb[b]: region 64 b + l is on and is not `skip`.  Returns the number of set bits.
__device__ __forceinline__ int mask_ballots(const uint8_t* mask, int R, int skip, unsigned long long* lb) {
  const int lane = threadIdx.x & 63;
  uint8_t m[16];
#pragma unroll
  for (int b = 0; b < 16; ++b) {
    const int r = 64 * b + lane;
    m[b] = (r < R && r != skip) ? mask[r] : (uint8_t)0;
  }
  int cnt = 0;
#pragma unroll
  for (int b = 0; b < 16; ++b) {
    const unsigned long long x = __ballot(m[b] != 0);
    cnt += __popcll(x);
    if (lane == 0) lb[b] = x;
  }
  return cnt;
}
__device__ __forceinline__ unsigned long long wave_uniform64(unsigned long long x) {
  const unsigned int lo = __builtin_amdgcn_readfirstlane((unsigned int)x), hi = __builtin_amdgcn_readfirstlane((unsigned int)(x >> 32));
  return ((unsigned long long)hi << 32) | lo;
}

__device__ __forceinline__ void logits_combine(const ic_heads_args& a, float* red) {
  constexpr int VPT = 4;  // V <= 1024 (checked on the host)
  const int V = a.V, Kf = a.Kf, R = a.R, tid = threadIdx.x;
  // every global read is unconditional and independent of the others: one round trip
  constexpr int KF = 16;  // Kf <= 16 (checked on the host); fixed trip counts keep zg in registers
  float zg[KF];
#pragma unroll
  for (int f = 0; f < KF; ++f) zg[f] = f < Kf ? a.z[f] : -INFINITY;
  // number of region heads in the mean: every lane tests its own regions (one round trip for
  // R <= 64), the count comes from ballots -- an exact small integer either way
  __shared__ unsigned long long lbal[4][16];
  unsigned long long* lb = lbal[tid >> 6];
  const int nwi = mask_ballots(a.mask, R, a.motor_idx, lb);
  const float nw = (float)nwi;
  const int n = R * a.d;
  float g1 = 0.f, g0 = 0.f, sp = 0.f;
  for (int r = tid; r < R; r += 256) {  // row sums of squares from k_heads
    g1 += a.gsq[r];
    g0 += a.gsq[R + r];
    if (a.surprise) sp += a.surprise[r];
  }
  float gmx = -INFINITY;
#pragma unroll
  for (int f = 0; f < KF; ++f) gmx = fmaxf(gmx, zg[f]);
  float gsum = 0.f;
#pragma unroll
  for (int f = 0; f < KF; ++f) {
    zg[f] = f < Kf ? expf(zg[f] - gmx) : 0.f;
    gsum += zg[f];
  }
  const float denom = 1.0f + nw;
  float lg[VPT];
  float mx = -INFINITY;
#pragma unroll
  for (int j = 0; j < VPT; ++j) {
    const int v = tid + 256 * j;
    lg[j] = -INFINITY;
    float reg = 0.f;
    if (R > kSmallR) {
      // large maps: only the rows of the regions in the mean, found 64 at a time by a ballot (every lane of the wave
      // takes part, also those beyond V) and added in the same (row) order -- the rows skipped hold zeros, so the sum
      // is the small-map loop's bit for bit, in R/64 + n_active/8 memory round trips instead of R/8
      for (int base = 0; base < R && (tid & ~63) + 256 * j < V; base += 64) {  // (a wave wholly beyond V has nothing to add)
        unsigned long long bal = wave_uniform64(lb[base >> 6]);
        while (bal) {
          float pr[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            pr[u] = 0.f;
            if (bal) {  // wave-uniform
              const int row = base + __ffsll((long long)bal) - 1;
              bal &= bal - 1;
              if (v < V) pr[u] = a.per_region[(size_t)row * V + v];
            }
          }
#pragma unroll
          for (int u = 0; u < 8; ++u) reg += pr[u];
        }
      }
    }
    if (v < V) {
      float zf[KF];
#pragma unroll
      for (int f = 0; f < KF; ++f) zf[f] = f < Kf ? a.z[Kf + f * V + v] : 0.f;
      if (R <= kSmallR) {
        for (int r0 = 0; r0 < R; r0 += 8) {  // inactive rows hold zeros: no mask test in the load chain
          float pr[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) pr[u] = r0 + u < R ? a.per_region[(size_t)(r0 + u) * V + v] : 0.f;
#pragma unroll
          for (int u = 0; u < 8; ++u) reg += pr[u];
        }
      }
      float acc = 0.f;
#pragma unroll
      for (int f = 0; f < KF; ++f)
        if (f < Kf) acc += (zg[f] / gsum) * zf[f];
      lg[j] = (acc + reg) / denom;
      a.logits[v] = lg[j];
      mx = fmaxf(mx, lg[j]);
    }
  }
  mx = block_max(mx, red);
  float se = 0.f;
#pragma unroll
  for (int j = 0; j < VPT; ++j)
    if (tid + 256 * j < V) se += expf(lg[j] - mx);
  se = block_sum(se, red);
  float clp = 0.f, pmax = 0.f;
#pragma unroll
  for (int j = 0; j < VPT; ++j) {
    const int v = tid + 256 * j;
    if (v < V) {
      const float p = expf(lg[j] - mx) / se;
      a.probs[v] = p;
      clp += p * logf(p + 1e-12f);
      pmax = fmaxf(pmax, p);
    }
  }
  {  // the five closing reductions share one pair of barriers (same summation order as block_sum)
    __shared__ float red5[5][4];
    const int lane = tid & 63, wave = tid >> 6;
    const float v0 = wave_sum(clp), v1 = wave_max(pmax), v2 = wave_sum(g1), v3 = wave_sum(g0), v4 = wave_sum(sp);
    __syncthreads();
    if (lane == 0) {
      red5[0][wave] = v0; red5[1][wave] = v1; red5[2][wave] = v2; red5[3][wave] = v3; red5[4][wave] = v4;
    }
    __syncthreads();
    clp = (red5[0][0] + red5[0][1]) + (red5[0][2] + red5[0][3]);
    pmax = fmaxf(fmaxf(red5[1][0], red5[1][1]), fmaxf(red5[1][2], red5[1][3]));
    g1 = (red5[2][0] + red5[2][1]) + (red5[2][2] + red5[2][3]);
    g0 = (red5[3][0] + red5[3][1]) + (red5[3][2] + red5[3][3]);
    sp = ((red5[4][0] + red5[4][1]) + (red5[4][2] + red5[4][3])) / (float)R;
  }
  if (tid == 0) {
    a.stats[0] = 1.0f - pmax;
    a.stats[1] = clp;
    a.stats[2] = g1 / (float)n - g0 / (float)n;
    a.stats[3] = sp;
    if (a.aux_surprise) *a.aux_surprise = sp;
  }
}

// Measured on MI355X: electing a last workgroup to run the combine in the same launch
// (device-scope fence + ticket per workgroup) cost 190 us for this 3000-workgroup grid -- the
// fences write back / invalidate the per-XCD L2s -- against ~6 us for a second launch.  So the
// combine stays a launch of its own.
// sum of squares of row r of H and of H_par (goodness terms, utils.py:159-162), wave per row
template <int DCH>
__device__ __forceinline__ void sq_rows(const ic_heads_args& a, int lane, int wave, const int xb) {
  const int r = xb * 4 + wave;
  if (r >= a.R) return;
  RowFrag<DCH> h1, h0;
  load_row(h1, a.H + (size_t)r * a.d, a.d, lane);
  load_row(h0, a.H_par + (size_t)r * a.d, a.d, lane);
  const float s1 = wave_sum(dot_frag(h1, h1)), s0 = wave_sum(dot_frag(h0, h0));
  if (lane == 0) {
    a.gsq[r] = s1;
    a.gsq[a.R + r] = s0;
  }
}

// Arguments of stream s of a batched call (ic_heads_args.S > 1): every pointer moved to that
// stream's block.
__device__ __forceinline__ ic_heads_args heads_stream(ic_heads_args a, int s) {
  const size_t rd = (size_t)a.R * a.d, sd = (size_t)a.n_slots * a.d;
  if (a.par_slot) a.H_par += (int64_t)a.par_slot[s] * (int64_t)rd;  // multi-stream: parents by block index
  else a.H_par += (int64_t)s * (a.Hpar_sstride ? a.Hpar_sstride : (int64_t)rd);
  if (a.slots_in) {
    if (a.slots_slot) a.slots_in += (int64_t)a.slots_slot[s] * (int64_t)sd;
    else a.slots_in += (int64_t)s * (a.slots_in_sstride ? a.slots_in_sstride : (int64_t)sd);
  }
  if (s == 0) return a;
  a.H += s * rd;
  a.mask += (size_t)s * a.R;
  if (a.slots_out) a.slots_out += s * sd;
  if (a.surprise) a.surprise += (size_t)s * a.R;
  a.z += (size_t)s * (a.Kf + a.Kf * a.V);
  a.per_region += (size_t)s * a.R * a.V;
  a.gsq += (size_t)s * 2 * a.R;
  a.logits += (size_t)s * a.V;
  a.probs += (size_t)s * a.V;
  a.stats += 4 * s;
  if (a.aux_surprise) a.aux_surprise += 4 * s;
  if (a.ws_mean) a.ws_mean += (size_t)s * a.d;
  return a;
}

// token_rows for a batch of streams [s0, s1) (at most kHeadsSB): the two weight rows of a wave are read ONCE for the
// batch and the streams' input rows are all requested before the first reduction -- with 32 lockstep streams
// the per-stream grid re-read the 4.7 MB of head weights 32 times from L2 (30 us); same sums per stream.
constexpr int kHeadsSB = 8;
template <int DCH>
__device__ __forceinline__ void token_rows_multi(const ic_heads_args& a0, int lane, int wave, const int grp, const int xb0,
                                                 const int nxb, const int s0, const int s1) {
  const int d = a0.d;
  const int rows = grp == 0 ? a0.Kf + a0.Kf * a0.V : a0.V;
  const int r = grp - 1;
  RowFrag<DCH> xv[kHeadsSB];
  bool on[kHeadsSB];
  bool any = false;
#pragma unroll
  for (int u = 0; u < kHeadsSB; ++u) {
    on[u] = false;
    zero_frag(xv[u]);
    if (s0 + u < s1) {
      const ic_heads_args a = heads_stream(a0, s0 + u);
      on[u] = grp == 0 || (r != a.motor_idx && a.mask[r] != 0);
      if (on[u] && xb0 * 8 + wave * 2 < rows) load_row(xv[u], a.H + (size_t)(grp == 0 ? a.motor_idx : r) * d, d, lane);
      any |= on[u];
    }
  }
  // maps above kSmallR regions: the combine only reads the rows of the regions in the mean, nothing to zero here
  if (!any && a0.R > kSmallR) return;
  for (int xb = xb0; xb < xb0 + nxb; ++xb) {
    const int row0 = xb * 8 + wave * 2;
    RowFrag<DCH> w[2];
    float bias[2] = {0.f, 0.f};
    zero_frag(w[0]);
    zero_frag(w[1]);
    if (any && row0 < rows) {
      const float* W = grp == 0 ? a0.lm_W - (size_t)a0.Kf * d : a0.rh_W + (size_t)r * a0.V * d;  // facet row i sits at z[Kf + i]
      const float* b = grp == 0 ? a0.lm_b - a0.Kf : a0.rh_b + (size_t)r * a0.V;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int row = row0 + i;
        if (row < rows) {
          const float* wr = W + (size_t)row * d;
          const float* br = b + row;
          if (grp == 0 && row < a0.Kf) {  // the MFS gate rows come first in z
            wr = a0.lm_gate_W + (size_t)row * d;
            br = a0.lm_gate_b + row;
          }
          load_row(w[i], wr, d, lane);
          bias[i] = *br;
        }
      }
    }
#pragma unroll
    for (int u = 0; u < kHeadsSB; ++u) {
      if (s0 + u >= s1) break;
      const ic_heads_args a = heads_stream(a0, s0 + u);
      if (!on[u]) {  // not part of the mean (model.py:314-317): zeros, so the small-map combine can add all R rows
        const int v = xb * 8 + (int)threadIdx.x;  // this block's 8 rows (threads 0..7)
        if (threadIdx.x < 8 && v < a.V && a0.R <= kSmallR) a.per_region[(size_t)r * a.V + v] = 0.f;
        continue;
      }
      if (row0 >= rows) continue;
      float* y = grp == 0 ? a.z : a.per_region + (size_t)r * a.V;
      const float t0 = wave_sum(dot_frag(w[0], xv[u])), t1 = wave_sum(dot_frag(w[1], xv[u]));
      if (lane == 0) {
        y[row0] = t0 + bias[0];
        if (row0 + 1 < rows) y[row0 + 1] = t1 + bias[1];
      }
    }
  }
}

// mean of the active rows of H (same arithmetic as k_steer_select): one workgroup per stream
__device__ __forceinline__ void ws_mean_rows(const ic_heads_args& a) {
  if (a.R > kSmallR) {
    // large maps: the active rows found 64 at a time by a ballot, 8 of them in flight, added in row order: the same
    // sums as below (which walks all R rows), in R/64 + n_active/8 round trips instead of R/8
    __shared__ unsigned long long lbal[4][16];
    unsigned long long* lb = lbal[threadIdx.x >> 6];
    const int cnt = mask_ballots(a.mask, a.R, -1, lb);
    for (int c = threadIdx.x; c < a.d + 255 - (a.d + 255) % 256; c += 256) {  // (whole waves: the row walk is wave-uniform)
      float s = 0.f;
      for (int base = 0; base < a.R; base += 64) {
        unsigned long long bal = wave_uniform64(lb[base >> 6]);
        while (bal) {
          float v[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            v[u] = 0.f;
            if (bal) {  // wave-uniform
              const int row = base + __ffsll((long long)bal) - 1;
              bal &= bal - 1;
              if (c < a.d) v[u] = a.H[(size_t)row * a.d + c];
            } else {
              v[u] = -0.f;  // (x + -0.f == x for every x: the padding of the last batch changes nothing)
            }
          }
#pragma unroll
          for (int u = 0; u < 8; ++u) s += v[u];
        }
      }
      if (c < a.d) a.ws_mean[c] = s / fmaxf((float)cnt, 1.0f);
    }
    return;
  }
  float n_on = 0.f;
  for (int r = 0; r < a.R; ++r) n_on += a.mask[r] ? 1.0f : 0.0f;
  const float den = fmaxf(n_on, 1.0f);
  for (int c = threadIdx.x; c < a.d; c += 256) {
    float s = 0.f;
    for (int r0 = 0; r0 < a.R; r0 += 8) {  // 8 rows in flight, added in row order if active
      float v[8];
      bool on[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) on[u] = r0 + u < a.R && a.mask[r0 + u] != 0;
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = r0 + u < a.R ? a.H[(size_t)(r0 + u) * a.d + c] : 0.f;
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (on[u]) s += v[u];
    }
    a.ws_mean[c] = s / den;
  }
}

// A GEMV whose input IS the mean of the active rows of H (the first planner layer, heads.py:57 on
// model.py:344-349): every wave forms the mean in its own lanes' columns -- same order of
// additions as ws_mean_rows, same dot-product order as k_gemv_reg, so the result is bit-identical
// to ws_mean_rows followed by k_gemv_reg -- and needs no launch after the one that wrote H.
template <int NX>
__device__ __forceinline__ void gemv_on_ws_mean(const ic_heads_args& a, const ic_gemv_desc& g, const int xb) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int cols = a.d;
  const int row0 = xb * 8 + wave * 2;
  if (row0 >= g.rows) return;
  const bool two = row0 + 1 < g.rows;
  float xr[NX], w0[NX], w1[NX];
#pragma unroll
  for (int i = 0; i < NX; ++i) {
    const int c = lane + 64 * i;
    w0[i] = c < cols ? g.W[(size_t)row0 * cols + c] : 0.f;
    w1[i] = (two && c < cols) ? g.W[(size_t)(row0 + 1) * cols + c] : 0.f;
    xr[i] = 0.f;
  }
  const float b0 = g.b ? g.b[row0] : 0.f, b1 = (g.b && two) ? g.b[row0 + 1] : 0.f;
  constexpr int UB = NX <= 4 ? 8 : (NX <= 8 ? 4 : 2);
  float n_on = 0.f;
  if (a.R > kSmallR) {  // large maps: active rows by ballot, as in ws_mean_rows (same sums)
    __shared__ unsigned long long lbal[4][16];
    unsigned long long* lb = lbal[wave];
    const int cnt = mask_ballots(a.mask, a.R, -1, lb);
    for (int base = 0; base < a.R; base += 64) {
      unsigned long long bal = wave_uniform64(lb[base >> 6]);
      while (bal) {
        float v[UB][NX];
        bool on[UB];
#pragma unroll
        for (int u = 0; u < UB; ++u) {
          on[u] = bal != 0;  // wave-uniform
          int row = 0;
          if (bal) {
            row = base + __ffsll((long long)bal) - 1;
            bal &= bal - 1;
          }
#pragma unroll
          for (int i = 0; i < NX; ++i) {
            const int c = lane + 64 * i;
            v[u][i] = (on[u] && c < cols) ? a.H[(size_t)row * a.d + c] : 0.f;
          }
        }
#pragma unroll
        for (int u = 0; u < UB; ++u)
          if (on[u]) {
#pragma unroll
            for (int i = 0; i < NX; ++i) xr[i] += v[u][i];
          }
      }
    }
    n_on = (float)cnt;
  } else {
    for (int r = 0; r < a.R; ++r) n_on += a.mask[r] ? 1.0f : 0.0f;
  }
  const float den = fmaxf(n_on, 1.0f);
  // rows are requested UB at a time whether active or not (L2 hits), then added in row order if
  // active: one memory round trip per UB rows instead of two per active row
  for (int r0 = 0; r0 < (a.R > kSmallR ? 0 : a.R); r0 += UB) {
    float v[UB][NX];
    bool on[UB];
#pragma unroll
    for (int u = 0; u < UB; ++u) on[u] = r0 + u < a.R && a.mask[r0 + u] != 0;
#pragma unroll
    for (int u = 0; u < UB; ++u)
#pragma unroll
      for (int i = 0; i < NX; ++i) {
        const int c = lane + 64 * i;
        v[u][i] = (r0 + u < a.R && c < cols) ? a.H[(size_t)(r0 + u) * a.d + c] : 0.f;
      }
#pragma unroll
    for (int u = 0; u < UB; ++u)
      if (on[u]) {
#pragma unroll
        for (int i = 0; i < NX; ++i) xr[i] += v[u][i];
      }
  }
  float a0 = 0.f, a1 = 0.f;
#pragma unroll
  for (int i = 0; i < NX; ++i) {
    const float v = xr[i] / den;
    a0 = fmaf(w0[i], v, a0);
    a1 = fmaf(w1[i], v, a1);
  }
  a0 = wave_sum(a0);
  a1 = wave_sum(a1);
  if (lane == 0) {
    g.y[row0] = apply_act(a0 + b0, g.act);
    if (two) g.y[row0 + 1] = apply_act(a1 + b1, g.act);
  }
}

// 1-D grid, no idle workgroups: [token head: ceil(lm_rows/8)] [region heads: R x ceil(V/8)]
// [workspace write: ceil(n_slots/4)] [row sums of squares: ceil(R/4)] [ws_mean: 1 if asked for]
// [a GEMV on that mean: ceil(rows/8) if asked for]
template <int DCH>
__global__ __launch_bounds__(256) void k_heads(ic_heads_args a0, ic_gemv_desc g0, int n_g, int sb, int rp8) {
  // blockIdx.y = a batch of up to kHeadsSB streams: the head rows serve the whole batch from one read of their
  // weights, the small per-stream pieces run stream after stream
  const int S = a0.S > 1 ? a0.S : 1;
  const int s0 = blockIdx.y * sb, s1 = min(S, s0 + sb);  // sb = 1 (few streams: one workgroup row per stream) or kHeadsSB
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int b = blockIdx.x;
  // region heads: rp8 groups of 8 rows per workgroup (1 on small maps; 8 on large ones, where a grid of R x V/8
  // workgroups of which a few per cent find an active region was the launch: 16 384 at R = 512)
  const int n_tok = (a0.Kf + a0.Kf * a0.V + 7) / 8, n_rh = (a0.V + 8 * rp8 - 1) / (8 * rp8);
  if (b < n_tok) return token_rows_multi<DCH>(a0, lane, wave, 0, b, 1, s0, s1);
  b -= n_tok;
  if (b < a0.R * n_rh) return token_rows_multi<DCH>(a0, lane, wave, 1 + b / n_rh, (b % n_rh) * rp8, rp8, s0, s1);
  b -= a0.R * n_rh;
  // the small per-stream pieces keep a workgroup of their own per stream: n_small blocks for each stream of the batch
  const int n_ws = (a0.n_slots + 3) / 4, n_sq = (a0.R + 3) / 4;
  const int n_small = n_ws + n_sq + (a0.ws_mean ? 1 : 0) + n_g;
  const int s = s0 + b / n_small;
  if (s >= s1) return;
  const ic_heads_args a = heads_stream(a0, s);
  int c = b % n_small;
  if (c < n_ws) return ws_write_rows<DCH>(a, lane, wave, c);
  c -= n_ws;
  if (c < n_sq) return sq_rows<DCH>(a, lane, wave, c);
  c -= n_sq;
  if (a.ws_mean) {
    if (c == 0) return ws_mean_rows(a);
    c -= 1;
  }
  if (c < n_g) gemv_on_ws_mean<4 * DCH>(a, gemv_batch(g0, s), c);
}

__global__ __launch_bounds__(256) void k_logits_combine(ic_heads_args a0) {
  __shared__ float red[4];
  logits_combine(heads_stream(a0, blockIdx.x), red);
}

// The combine of S streams next to ONE independent GEMV (the first planner layer, whose input
// ic_heads_fwd's first launch wrote): blockIdx.z == 0 combines stream blockIdx.y, z == 1 is the GEMV.
template <int NX>
__global__ __launch_bounds__(256) void k_combine_gemv(ic_heads_args a0, ic_gemv_desc g, int S) {
  if (blockIdx.z == 0) {
    if (blockIdx.x != 0 || (int)blockIdx.y >= S) return;
    __shared__ float red[4];
    logits_combine(heads_stream(a0, blockIdx.y), red);
    return;
  }
  if ((int)blockIdx.x * 8 >= g.rows || (int)blockIdx.y >= (g.nbatch > 1 ? g.nbatch : 1)) return;
  gemv_reg_body<NX>(g, nullptr, nullptr);
}

__global__ void k_branch_score(const float* stats, const float* energy, float* score) {
  if (threadIdx.x == 0) score[0] = 0.6f * stats[2] + 0.2f * (-energy[0]) + 0.2f * stats[1];
}

__global__ __launch_bounds__(256) void k_steer_select(ic_steer_args a) {
  int best = 0;
  if (a.scores) {
    float bs = a.scores[0];
    for (int c = 1; c < a.n_cand; ++c)
      if (a.scores[c] > bs) {
        bs = a.scores[c];
        best = c;
      }
  }
  const float* H = a.H_cand[0];
  const uint8_t* mk = a.mask_cand[0];
  for (int c = 1; c < a.n_cand; ++c)
    if (c == best) {
      H = a.H_cand[c];
      mk = a.mask_cand[c];
    }
  float n_on = 0.f;
  for (int r = 0; r < a.R; ++r) n_on += mk[r] ? 1.0f : 0.0f;
  const float den = fmaxf(n_on, 1.0f);
  for (int c = threadIdx.x; c < a.d; c += 256) {
    float s = 0.f;
    for (int r = 0; r < a.R; ++r)
      if (mk[r]) s += H[(size_t)r * a.d + c];
    a.ws_prev[c] = s / den;
    a.motor[c] = H[(size_t)a.motor_idx * a.d + c];
  }
}

// ---------------------------------------------------------------- device-side steering (ic_loop_steer)
// One workgroup.  Wave c < C runs the critic's scalar output layer for candidate c with the arithmetic of
// gemv_reg_body (lane-strided columns, one fmaf chain per lane, wave_sum), then thread 0 takes the decisions of
// reference model.py:371-392,440-452 for the next outer iteration and writes its control block.
__device__ __forceinline__ void ctl_zero_state(const ic_loop_steer_args& a, int32_t* nx, int ws_slot) {
  const float v = a.vhat0 ? a.vhat0[0] : 0.f;
  const int burst = ((double)v > 0.0) ? (a.R / 8 > 1 ? a.R / 8 : 1) : 0;  // utils.py:140-144 with u_mean = 0
  const int kt = a.k_active + burst > 0 ? a.k_active + burst : 0;
  const float t5 = (float)(a.w5 * (double)v), t6 = (float)(a.w6 * 0.0);
  for (int c = 0; c < 4; ++c) {
    nx[IC_CTL_IN_SLOT + c] = a.zero_slot;
    nx[IC_CTL_SLOTS_SLOT + c] = ws_slot;
    nx[IC_CTL_K_TOTAL + c] = kt;
    nx[IC_CTL_T5 + c] = __float_as_int(t5);
    nx[IC_CTL_T6 + c] = __float_as_int(t6);
  }
  nx[IC_CTL_DEAD] = 0;
  nx[IC_CTL_HALT] = 0;  // utils.py:147-151: k = 0 < max(1, K_inner / 2)
  nx[IC_CTL_TOP] = -1;
  nx[IC_CTL_U] = __float_as_int(0.f);
  nx[IC_CTL_VHAT] = __float_as_int(v);
}

__global__ __launch_bounds__(256) void k_loop_steer(ic_loop_steer_args a) {
  __shared__ float vh[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int C = a.C;
  if (a.mode != IC_STEER_INIT && wave < C) {
    float out = 0.f;
    if (a.crit_W) {
      constexpr int NX = 16;  // crit_cols <= 1024 (checked on the host); same lane / column mapping as k_gemv_reg
      const float* x = a.crit_x + (size_t)wave * a.crit_cols;
      float acc = 0.f;
#pragma unroll
      for (int i = 0; i < NX; ++i) {
        const int c = lane + 64 * i;
        const float xv = c < a.crit_cols ? x[c] : 0.f;
        const float wv = c < a.crit_cols ? a.crit_W[c] : 0.f;
        acc = fmaf(wv, xv, acc);
      }
      acc = wave_sum(acc);
      out = acc + (a.crit_b ? a.crit_b[0] : 0.f);
    }
    if (lane == 0) {
      vh[wave] = out;
      if (a.vhat) a.vhat[wave] = out;
    }
  }
  __syncthreads();
  if (threadIdx.x != 0) return;
  int32_t* nx = a.ctl_next;
  if (a.mode == IC_STEER_INIT) {
    if (nx) ctl_zero_state(a, nx, a.zero_slot);
    return;
  }
  const int32_t* cur = a.ctl;
  const bool dead = cur[IC_CTL_DEAD] != 0;
  int top0 = 0;
  if (!dead) {
    top0 = (C == 2 && a.scores[1] > a.scores[0]) ? 1 : 0;  // score descending, first on ties
    a.rec[0] = a.slot0 + top0;
    a.rec[1] = a.slot0 + C - 1;
    a.rec[2] = (a.k + 1) * C;
    a.rec[3] = a.k + 1;
  }
  if (!nx) return;
  if (a.mode == IC_STEER_LAST) {  // the next loop starts from the zero state and the winner's workspace
    ctl_zero_state(a, nx, a.rec[0]);
    return;
  }
  if (dead || cur[IC_CTL_HALT] != 0) {  // the loop is over: what is queued behind it passes the state through
    for (int c = 0; c < 4; ++c) {
      nx[IC_CTL_IN_SLOT + c] = cur[IC_CTL_IN_SLOT + c];
      nx[IC_CTL_SLOTS_SLOT + c] = cur[IC_CTL_SLOTS_SLOT + c];
      nx[IC_CTL_K_TOTAL + c] = IC_GATE_OFF;
      nx[IC_CTL_T5 + c] = 0;
      nx[IC_CTL_T6 + c] = 0;
    }
    nx[IC_CTL_DEAD] = 1;
    nx[IC_CTL_HALT] = 0;
    nx[IC_CTL_TOP] = dead ? -1 : top0;
    nx[IC_CTL_U] = cur[IC_CTL_U];
    nx[IC_CTL_VHAT] = cur[IC_CTL_VHAT];
    return;
  }
  const float u = a.stats[(C - 1) * 4];  // uncertainty of the iteration's LAST logits (model.py:378,398)
  const float v = vh[top0];              // the winner's critic value (model.py:384-386)
  const int burst = ((double)u > 0.30 || (double)v > 0.0) ? (a.R / 8 > 1 ? a.R / 8 : 1) : 0;  // utils.py:140-144
  const int kt = a.k_active + burst > 0 ? a.k_active + burst : 0;
  const float t5 = (float)(a.w5 * (double)v), t6 = (float)(a.w6 * (double)u);
  const int half = a.K_inner / 2 > 1 ? a.K_inner / 2 : 1;
  for (int c = 0; c < 4; ++c) {
    const int src = a.slot0 + (c == 0 ? top0 : (C == 2 ? 1 - top0 : top0));
    nx[IC_CTL_IN_SLOT + c] = src;
    nx[IC_CTL_SLOTS_SLOT + c] = src;
    nx[IC_CTL_K_TOTAL + c] = kt;
    nx[IC_CTL_T5 + c] = __float_as_int(t5);
    nx[IC_CTL_T6 + c] = __float_as_int(t6);
  }
  nx[IC_CTL_DEAD] = 0;
  nx[IC_CTL_HALT] = ((double)u < 0.05 && a.k + 1 >= half) ? 1 : 0;  // utils.py:147-151 for iteration k + 1
  nx[IC_CTL_TOP] = top0;
  nx[IC_CTL_U] = __float_as_int(u);
  nx[IC_CTL_VHAT] = __float_as_int(v);
}

}  // namespace

// weights once (the batches of a launch share them), input and output vectors per batch
static double gemv_bytes(const ic_gemv_desc& g) {
  const double cols = g.n0 + g.n1 + g.n2, nb = g.nbatch > 1 ? g.nbatch : 1;
  return 4.0 * ((double)g.rows * cols + nb * (cols + g.rows) + (g.b ? g.rows : 0) + (g.ln_g ? 2 * cols : 0));
}

static int check_gemv(const ic_gemv_desc& g, int i, const char* who) {
  const int cols = g.n0 + g.n1 + g.n2;
  IC_REQUIRE(g.W && g.y && g.rows > 0, "%s[%d]: NULL W/y or rows=%d", who, i, g.rows);
  IC_REQUIRE(g.n0 >= 0 && g.n1 >= 0 && g.n2 >= 0 && cols > 0 && cols <= kMaxCols,
             "%s[%d]: %d input columns (need 1..%d)", who, i, cols, kMaxCols);
  IC_REQUIRE((g.n0 == 0 || g.x0) && (g.n1 == 0 || g.x1) && (g.n2 == 0 || g.x2), "%s[%d]: NULL input segment", who, i);
  IC_REQUIRE(!g.ln_g || g.ln_b, "%s[%d]: LayerNorm weight without bias", who, i);
  IC_REQUIRE(g.act >= IC_ACT_NONE && g.act <= IC_ACT_SILU, "%s[%d]: act=%d", who, i, g.act);
  return IC_OK;
}

// Chains of up to 6 layers and <= 128 KB of weights run as one single-workgroup launch (one CU
// streams ~25 GB/s: measured 79 us for 2 MB); anything larger goes layer by layer over the chip,
// the last layer folding in the branch score.
static int run_chain(const ic_gemv_desc* descs, int n, const float* stats, float* score, hipStream_t st,
                     const char* who) {
  size_t bytes = 0;
  for (int i = 0; i < n; ++i) {
    int rc = check_gemv(descs[i], i, who);
    if (rc) return rc;
    bytes += (size_t)descs[i].rows * (descs[i].n0 + descs[i].n1 + descs[i].n2) * 4;
  }
  bool batched = false;
  for (int i = 0; i < n; ++i) batched |= descs[i].nbatch > 1;
  if (n >= 1 && n <= 6 && bytes <= (128u << 10) && !batched) {
    ChainArgs c;
    for (int i = 0; i < n; ++i) c.d[i] = descs[i];
    c.n = n;
    c.stats = stats;
    c.score = score;
    IC_BYTES(IC_KID_gemv, bytes + [&] { double v = 0; for (int i = 0; i < n; ++i) v += 4.0 * (descs[i].rows + descs[i].n0 + descs[i].n1 + descs[i].n2); return v; }());
    IC_TIMED(IC_KID_gemv, st, hipLaunchKernelGGL(k_gemv_chain1, dim3(1), dim3(1024), 0, st, c));
    return IC_OK;
  }
  for (int i = 0; i < n; ++i) {
    float* sc = (i == n - 1) ? score : nullptr;
    const int cols = descs[i].n0 + descs[i].n1 + descs[i].n2;
    const dim3 grid(ic_ceil_div(descs[i].rows, 8), descs[i].nbatch > 1 ? descs[i].nbatch : 1), blk(256);
    IC_BYTES(IC_KID_gemv, gemv_bytes(descs[i]));
    if (cols <= 256) IC_TIMED(IC_KID_gemv, st, hipLaunchKernelGGL(k_gemv_reg<4>, grid, blk, 0, st, descs[i], stats, sc));
    else if (cols <= 512) IC_TIMED(IC_KID_gemv, st, hipLaunchKernelGGL(k_gemv_reg<8>, grid, blk, 0, st, descs[i], stats, sc));
    else if (cols <= 1024) IC_TIMED(IC_KID_gemv, st, hipLaunchKernelGGL(k_gemv_reg<16>, grid, blk, 0, st, descs[i], stats, sc));
    else IC_TIMED(IC_KID_gemv, st, hipLaunchKernelGGL(k_gemv, grid, blk, 0, st, descs[i], stats, sc));
  }
  return IC_OK;
}

extern "C" int ic_gemv_chain(const ic_gemv_desc* descs, int32_t n, ic_stream_t stream) {
  IC_REQUIRE(descs != nullptr && n >= 0, "ic_gemv_chain: NULL descriptors");
  int rc = run_chain(descs, n, nullptr, nullptr, (hipStream_t)stream, "ic_gemv_chain");
  if (rc) return rc;
  IC_LAUNCH_CHECK("ic_gemv_chain");
  return IC_OK;
}

extern "C" int ic_gemv_multi(const ic_gemv_desc* descs, int32_t n, const float* stats, float* score,
                             ic_stream_t stream) {
  IC_REQUIRE(descs != nullptr && n >= 1 && n <= 4, "ic_gemv_multi: n=%d (need 1..4 descriptors)", n);
  IC_REQUIRE(!score || (stats && descs[0].rows == 1), "ic_gemv_multi: score needs stats and a scalar descs[0]");
  MultiArgs m;
  int rows = 0, cols = 0, nb = 1;
  for (int i = 0; i < n; ++i) {
    int rc = check_gemv(descs[i], i, "ic_gemv_multi");
    if (rc) return rc;
    m.d[i] = descs[i];
    rows = descs[i].rows > rows ? descs[i].rows : rows;
    const int c = descs[i].n0 + descs[i].n1 + descs[i].n2;
    cols = c > cols ? c : cols;
    nb = descs[i].nbatch > nb ? descs[i].nbatch : nb;
  }
  IC_REQUIRE(cols <= 1024, "ic_gemv_multi: %d input columns (need <= 1024)", cols);
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid(ic_ceil_div(rows, 8), nb, n), blk(256);
  IC_BYTES(IC_KID_gemv, [&] { double v = 0; for (int i = 0; i < n; ++i) v += gemv_bytes(descs[i]); return v; }());
  if (cols <= 256) IC_TIMED(IC_KID_gemv, st, hipLaunchKernelGGL(k_gemv_multi<4>, grid, blk, 0, st, m, stats, score));
  else if (cols <= 512) IC_TIMED(IC_KID_gemv, st, hipLaunchKernelGGL(k_gemv_multi<8>, grid, blk, 0, st, m, stats, score));
  else IC_TIMED(IC_KID_gemv, st, hipLaunchKernelGGL(k_gemv_multi<16>, grid, blk, 0, st, m, stats, score));
  IC_LAUNCH_CHECK("ic_gemv_multi");
  return IC_OK;
}

extern "C" int ic_verify_score(const ic_gemv_desc* descs, int32_t n, const float* stats, float* score,
                               ic_stream_t stream) {
  IC_REQUIRE(descs != nullptr && n >= 1 && stats && score, "ic_verify_score: NULL argument");
  IC_REQUIRE(descs[n - 1].rows == 1, "ic_verify_score: the last layer must produce the scalar energy");
  int rc = run_chain(descs, n, stats, score, (hipStream_t)stream, "ic_verify_score");
  if (rc) return rc;
  IC_LAUNCH_CHECK("ic_verify_score");
  return IC_OK;
}

static int heads_fwd(const ic_heads_args* a, const ic_gemv_desc* gm, const ic_gemv_desc* g, ic_stream_t stream,
                     bool timed);

extern "C" int ic_heads_fwd(const ic_heads_args* a, ic_stream_t stream) {
  return heads_fwd(a, nullptr, nullptr, stream, true);
}

extern "C" int ic_heads_gemv_fwd(const ic_heads_args* a, const ic_gemv_desc* g_mean, const ic_gemv_desc* g_next,
                                 ic_stream_t stream) {
  IC_REQUIRE(a != nullptr, "ic_heads_gemv_fwd: NULL args");
  if (g_mean) {
    IC_REQUIRE(g_mean->W && g_mean->y && g_mean->rows > 0, "ic_heads_gemv_fwd: g_mean has NULL W/y or no rows");
    IC_REQUIRE(g_mean->n0 == a->d && g_mean->n1 == 0 && g_mean->n2 == 0 && !g_mean->ln_g,
               "ic_heads_gemv_fwd: g_mean must take exactly the d=%d columns of the mean (n0=%d n1=%d n2=%d)", a->d,
               g_mean->n0, g_mean->n1, g_mean->n2);
    IC_REQUIRE((g_mean->nbatch > 1 ? g_mean->nbatch : 1) == (a->S > 1 ? a->S : 1),
               "ic_heads_gemv_fwd: g_mean.nbatch=%d must equal the number of streams %d", g_mean->nbatch, a->S);
    IC_REQUIRE(g_mean->act >= IC_ACT_NONE && g_mean->act <= IC_ACT_SILU, "ic_heads_gemv_fwd: act=%d", g_mean->act);
  }
  if (g_next) {
    int rc = check_gemv(*g_next, 0, "ic_heads_gemv_fwd");
    if (rc) return rc;
    IC_REQUIRE(g_next->n0 + g_next->n1 + g_next->n2 <= 1024, "ic_heads_gemv_fwd: %d input columns (need <= 1024)",
               g_next->n0 + g_next->n1 + g_next->n2);
  }
  return heads_fwd(a, g_mean, g_next, stream, true);
}

// timed = false: the launches are never bracketed by profiling events (ic_prof_begin may create
// events, i.e. allocate), for the entry point that must not allocate.
#define IC_MAYBE_TIMED(timed, kid, st, ...) \
  do {                                      \
    if (timed) IC_TIMED(kid, st, __VA_ARGS__); \
    else { __VA_ARGS__; }                   \
  } while (0)

static int heads_fwd(const ic_heads_args* a, const ic_gemv_desc* gm, const ic_gemv_desc* g, ic_stream_t stream,
                     bool timed) {
  IC_REQUIRE(a != nullptr, "ic_heads_fwd: NULL args");
  IC_REQUIRE(a->R > 0 && a->V > 0 && a->Kf > 0 && a->Kf <= 16 && a->n_slots >= 0,
             "ic_heads_fwd: R=%d V=%d Kf=%d n_slots=%d", a->R, a->V, a->Kf, a->n_slots);
  IC_REQUIRE(a->d >= 4 && a->d % 4 == 0 && a->d <= 1024, "ic_heads_fwd: d=%d unsupported", a->d);
  IC_REQUIRE(a->motor_idx >= 0 && a->motor_idx < a->R, "ic_heads_fwd: motor_idx=%d", a->motor_idx);
  IC_REQUIRE(a->H && a->H_par && a->mask && a->lm_gate_W && a->lm_gate_b && a->lm_W && a->lm_b && a->rh_W && a->rh_b && a->z &&
                 a->per_region && a->logits && a->probs && a->stats && a->gsq,
             "ic_heads_fwd: NULL pointer");
  IC_REQUIRE(a->n_slots == 0 || (a->slots_in && a->slots_out), "ic_heads_fwd: NULL workspace slots");
  hipStream_t st = (hipStream_t)stream;
  IC_REQUIRE(a->V <= 1024, "ic_heads_fwd: V=%d (need <= 1024)", a->V);
  const int S = a->S > 1 ? a->S : 1;
  IC_REQUIRE(S <= 64, "ic_heads_fwd: S=%d (need <= 64)", S);
  const int n_g = gm ? ic_ceil_div(gm->rows, 8) : 0;
  const int sb = S > 4 ? kHeadsSB : 1;  // (two branches of one loop: 9 us per stream row against 14 us batched)
  const int rp8 = a->R > 64 ? 8 : 1;  // 8-row groups per region-head workgroup (k_heads)
  const int nblk = ic_ceil_div(a->Kf + a->Kf * a->V, 8) + a->R * ic_ceil_div(a->V, 8 * rp8) +
                   sb * (ic_ceil_div(a->n_slots, 4) + ic_ceil_div(a->R, 4) + (a->ws_mean ? 1 : 0) + n_g);
  ic_gemv_desc g_none = {};
  if (timed) {
    // shape-determined part: MFS gate + facets, the two state blocks (goodness terms), workspace rows, the riding
    // GEMV; the region-head rows of the ACTIVE regions (V x d each) are added by the caller from its unit counts
    IC_BYTES(IC_KID_token_rows, 4.0 * ((double)(a->Kf + a->Kf * a->V) * (a->d + 1) + (double)S * (2.0 * a->R + 3.0 * a->n_slots) * a->d) +
                                    (gm ? gemv_bytes(*gm) : 0.0));
    IC_BYTES(IC_KID_logits_combine, 4.0 * (double)S * (a->Kf + (double)a->Kf * a->V + (double)a->R * a->V + 4.0 * a->V) + (g ? gemv_bytes(*g) : 0.0));
  }
  IC_DISPATCH_DCH(a->d, {
    IC_MAYBE_TIMED(timed, IC_KID_token_rows, st, hipLaunchKernelGGL(k_heads<DCH>, dim3(nblk, ic_ceil_div(S, sb)), dim3(256), 0, st, *a, gm ? *gm : g_none, n_g, sb, rp8));
  });
  if (g) {
    const int cols = g->n0 + g->n1 + g->n2, nb = g->nbatch > 1 ? g->nbatch : 1;
    const dim3 grid(ic_ceil_div(g->rows, 8), nb > S ? nb : S, 2);
    if (cols <= 256) IC_TIMED(IC_KID_logits_combine, st, hipLaunchKernelGGL(k_combine_gemv<4>, grid, dim3(256), 0, st, *a, *g, S));
    else if (cols <= 512) IC_TIMED(IC_KID_logits_combine, st, hipLaunchKernelGGL(k_combine_gemv<8>, grid, dim3(256), 0, st, *a, *g, S));
    else IC_TIMED(IC_KID_logits_combine, st, hipLaunchKernelGGL(k_combine_gemv<16>, grid, dim3(256), 0, st, *a, *g, S));
  } else {
    IC_MAYBE_TIMED(timed, IC_KID_logits_combine, st, hipLaunchKernelGGL(k_logits_combine, dim3(S), dim3(256), 0, st, *a));
  }
  IC_LAUNCH_CHECK("ic_heads_fwd");
  return IC_OK;
}

// ------------------------------------------------------------------ ic_heads_fwd without any allocation
// The three inter-launch buffers of ic_heads_fwd (z, per_region, gsq: written by its first launch,
// read by its second) live in a file-scope device array instead of caller-provided memory.  Each
// GPU holds its own copy of the array at its own address: the address is looked up once per device
// (hipGetSymbolAddress with that device current) and only that pointer is ever handed to kernels.
// Nothing in the array has to start at zero: every word the second launch reads is written by the
// first launch of the same call.  Calls that overlap on one GPU (two streams) share the array and race.
static __device__ __align__(256) float scratch[IC_HEADS_STATIC_FLOATS];

namespace {
constexpr int kMaxDevices = 64;
float* g_scratch_ptr[kMaxDevices] = {};
std::mutex g_scratch_mu;
}  // namespace

// Resolves the scratch array's address on the CURRENT device (idempotent).  Called from
// ic_device_check() -- the library's set-up call -- and, failing that, from the entry point's first
// call on a device.
__attribute__((visibility("hidden"))) int ic_heads_static_setup(float** out) {
  int dev = -1;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) {
    ic_set_error("ic_heads_fwd_static: no current HIP device (or device index >= %d)", kMaxDevices);
    return IC_EDEVICE;
  }
  std::lock_guard<std::mutex> lk(g_scratch_mu);
  if (!g_scratch_ptr[dev]) {
    void* p = nullptr;
    const hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
    if (e != hipSuccess || p == nullptr) {
      ic_set_error("ic_heads_fwd_static: hipGetSymbolAddress(scratch) on device %d: %s", dev, hipGetErrorString(e));
      return IC_EDEVICE;
    }
    g_scratch_ptr[dev] = static_cast<float*>(p);
  }
  if (out) *out = g_scratch_ptr[dev];
  return IC_OK;
}

extern "C" int ic_heads_fwd_static(const ic_heads_args* a, ic_stream_t stream) {
  IC_REQUIRE(a != nullptr, "ic_heads_fwd_static: NULL args");
  IC_REQUIRE(a->R > 0 && a->V > 0 && a->Kf > 0, "ic_heads_fwd_static: R=%d V=%d Kf=%d", a->R, a->V, a->Kf);
  const size_t S = a->S > 1 ? (size_t)a->S : 1;
  // per-stream blocks, each padded to 64 floats (256 bytes)
  auto pad = [](size_t n) { return (n + 63) / 64 * 64; };
  const size_t nz = pad(S * ((size_t)a->Kf + (size_t)a->Kf * a->V)), npr = pad(S * (size_t)a->R * a->V),
               ng = pad(S * 2 * (size_t)a->R);
  IC_REQUIRE(nz + npr + ng <= (size_t)IC_HEADS_STATIC_FLOATS,
             "ic_heads_fwd_static: S=%zu R=%d V=%d Kf=%d needs %zu scratch floats, the static buffer holds %d "
             "(use ic_heads_fwd with caller-provided z / per_region / gsq)",
             S, a->R, a->V, a->Kf, nz + npr + ng, IC_HEADS_STATIC_FLOATS);
  float* base = nullptr;
  const int rc = ic_heads_static_setup(&base);
  if (rc) return rc;
  ic_heads_args b = *a;
  b.z = base;
  b.per_region = base + nz;
  b.gsq = base + nz + npr;
  return heads_fwd(&b, nullptr, nullptr, stream, /*timed=*/false);
}

extern "C" int ic_branch_score(const float* stats, const float* energy, float* score, ic_stream_t stream) {
  IC_REQUIRE(stats && energy && score, "ic_branch_score: NULL pointer");
  hipStream_t st = (hipStream_t)stream;
  IC_TIMED(IC_KID_branch_score, st, hipLaunchKernelGGL(k_branch_score, dim3(1), dim3(64), 0, st, stats, energy, score));
  IC_LAUNCH_CHECK("ic_branch_score");
  return IC_OK;
}

extern "C" int ic_steer_select(const ic_steer_args* a, ic_stream_t stream) {
  IC_REQUIRE(a != nullptr, "ic_steer_select: NULL args");
  IC_REQUIRE(a->R > 0 && a->d > 0 && a->n_cand >= 1 && a->n_cand <= 8, "ic_steer_select: R=%d d=%d n_cand=%d",
             a->R, a->d, a->n_cand);
  IC_REQUIRE(a->motor_idx >= 0 && a->motor_idx < a->R && a->ws_prev && a->motor, "ic_steer_select: bad motor/outputs");
  for (int c = 0; c < a->n_cand; ++c)
    IC_REQUIRE(a->H_cand[c] && a->mask_cand[c], "ic_steer_select: NULL candidate %d", c);
  hipStream_t st = (hipStream_t)stream;
  IC_BYTES(IC_KID_steer_select, 4.0 * (a->R + 2.0) * a->d + a->R);
  IC_TIMED(IC_KID_steer_select, st, hipLaunchKernelGGL(k_steer_select, dim3(1), dim3(256), 0, st, *a));
  IC_LAUNCH_CHECK("ic_steer_select");
  return IC_OK;
}

extern "C" int ic_loop_steer(const ic_loop_steer_args* a, ic_stream_t stream) {
  IC_REQUIRE(a != nullptr, "ic_loop_steer: NULL args");
  IC_REQUIRE(a->mode >= IC_STEER_INIT && a->mode <= IC_STEER_LAST, "ic_loop_steer: mode=%d", a->mode);
  IC_REQUIRE(a->C >= 1 && a->C <= 2, "ic_loop_steer: C=%d branches (need 1 or 2)", a->C);
  IC_REQUIRE(a->R > 0 && a->K_inner > 0 && a->k >= 0 && a->k < a->K_inner, "ic_loop_steer: R=%d K_inner=%d k=%d", a->R,
             a->K_inner, a->k);
  IC_REQUIRE(a->zero_slot >= 0 && a->slot0 >= 0, "ic_loop_steer: negative slot");
  if (a->mode == IC_STEER_INIT) {
    IC_REQUIRE(a->ctl_next, "ic_loop_steer: IC_STEER_INIT needs ctl_next");
  } else {
    IC_REQUIRE(a->ctl && a->rec && a->scores && a->stats, "ic_loop_steer: NULL ctl / rec / scores / stats");
    IC_REQUIRE((a->mode == IC_STEER_LAST) == (a->k == a->K_inner - 1), "ic_loop_steer: mode=%d at k=%d of %d", a->mode, a->k,
               a->K_inner);
    IC_REQUIRE(!a->crit_W || (a->crit_x && a->crit_cols > 0 && a->crit_cols <= 1024),
               "ic_loop_steer: critic layer with %d columns (need 1..1024) or NULL input", a->crit_cols);
  }
  hipStream_t st = (hipStream_t)stream;
  IC_BYTES(IC_KID_loop_steer, a->crit_W ? 4.0 * (a->crit_cols * (1.0 + a->C) + 1) + 4.0 * IC_CTL_WORDS * 2 : 4.0 * IC_CTL_WORDS * 2);
  IC_TIMED(IC_KID_loop_steer, st, hipLaunchKernelGGL(k_loop_steer, dim3(1), dim3(256), 0, st, *a));
  IC_LAUNCH_CHECK("ic_loop_steer");
  return IC_OK;
}
