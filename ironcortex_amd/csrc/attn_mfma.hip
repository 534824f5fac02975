// Token-mixer attention on the matrix cores: the bidirectional softmax attention of
// reference iron_rope.py:100-251 (IronRoPESelfAttention; rotary q/k and the Fourier relative bias arrive as
// extra q/k feature columns, see ironcortex_amd/mixer.py) and of attention/adaptive_filter_attention.py:8-127
// (scores times exp(-alpha |i-j| dt)), without the [B,H,T,T] score / probability / log-probability tensors.
//
// fp32 in, fp32 accumulate: v_mfma_f32_32x32x2_f32.  Operand layout (wave64): lane l supplies A[l % 32][l / 32]
// and B[l / 32][l % 32]; accumulator register v of lane l is C[8 * (v / 4) + 4 * (l / 32) + v % 4][l % 32].
//
// Forward: a workgroup owns 128 queries of one (batch, head), a wave 32 of them; key tiles of 32 pass through
// LDS.  Scores are formed TRANSPOSED (S^T = K Q^T: keys down the accumulator registers, the lane's query
// across), so a query's running max / sum / statistics stay inside one lane (plus one exchange between the two
// lane halves at the end), and P^T in the accumulator registers IS the B operand of O^T += V^T P^T: the
// probabilities never touch LDS.  Two passes over the keys (log-sum-exp first): the reference's energy
// statistic sum log(p + eps) needs the normalised probabilities.
// Backward: two kernels with the same tiling, so that every output row has one owner (no float atomics,
// run-to-run identical): dK / dV by key blocks (scores formed untransposed), dQ (+ d alpha partial sums) by
// query blocks.
#include "ic_common.h"

namespace {

typedef float v16f __attribute__((ext_vector_type(16)));
constexpr int kTile = 32;    // keys (queries) per LDS tile
constexpr int kOwn = 128;    // rows a workgroup owns (32 per wave)

__device__ __forceinline__ v16f zero16() {
  v16f z;
#pragma unroll
  for (int i = 0; i < 16; ++i) z[i] = 0.f;
  return z;
}
// the row of accumulator register v in a 32x32 tile
__device__ __forceinline__ int acc_row(int v, int half) { return 8 * (v >> 2) + 4 * half + (v & 3); }

// rows [r0, r0+32) of a [T, W] row-major block -> LDS tile with padded rows; rows >= T read as zero
template <int W>
__device__ __forceinline__ void stage_tile(float (*dst)[W + 1], const float* src, int r0, int T, int tid) {
  constexpr int per_row = W / 4;
  for (int i = tid; i < kTile * per_row; i += 256) {
    const int r = i / per_row, c = (i - r * per_row) * 4;
    float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r0 + r < T) x = *reinterpret_cast<const float4*>(src + (size_t)(r0 + r) * W + c);
    dst[r][c] = x.x; dst[r][c + 1] = x.y; dst[r][c + 2] = x.z; dst[r][c + 3] = x.w;
  }
}

// the same in two halves, so the next tile's global loads are in flight while the current one is multiplied:
// fetch_tile (global -> registers), put_tile (registers -> LDS)
template <int W>
struct TileRegs {
  float4 x[W / 32];
};
template <int W>
__device__ __forceinline__ void fetch_tile(TileRegs<W>& t, const float* src, int r0, int T, int tid) {
  constexpr int per_row = W / 4;
#pragma unroll
  for (int j = 0; j < W / 32; ++j) {
    const int i = tid + 256 * j, r = i / per_row, c = (i - r * per_row) * 4;
    t.x[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r0 + r < T) t.x[j] = *reinterpret_cast<const float4*>(src + (size_t)(r0 + r) * W + c);
  }
}
template <int W>
__device__ __forceinline__ void put_tile(float (*dst)[W + 1], const TileRegs<W>& t, int tid) {
  constexpr int per_row = W / 4;
#pragma unroll
  for (int j = 0; j < W / 32; ++j) {
    const int i = tid + 256 * j, r = i / per_row, c = (i - r * per_row) * 4;
    dst[r][c] = t.x[j].x; dst[r][c + 1] = t.x[j].y; dst[r][c + 2] = t.x[j].z; dst[r][c + 3] = t.x[j].w;
  }
}

// acc += X[rows 0..31][.] * own^T: A operand from an LDS tile (row = lane % 32), B operand from registers
// (own[i] = element 2i + half of this lane's own row)
template <int W>
__device__ __forceinline__ v16f tile_times_own(const float (*Xs)[W + 1], const float (&own)[W / 2], int lane) {
  v16f acc = zero16();
  const int r = lane & 31, h = lane >> 5;
#pragma unroll
  for (int i = 0; i < W / 2; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(Xs[r][2 * i + h], own[i], acc, 0, 0, 0);
  return acc;
}

// out[t] += Y^T[32t.., rows] * P: A operand Y[row = acc_row(v)][32t + lane % 32] from LDS, B operand = the 16
// accumulator registers of a score-shaped tile (register v <-> tile row acc_row(v, half))
template <int W>
__device__ __forceinline__ void acc_outer(v16f (&out)[W / 32], const float (*Ys)[W + 1], const v16f& p, int lane) {
  const int c = lane & 31, h = lane >> 5;
#pragma unroll
  for (int t = 0; t < W / 32; ++t) {
#pragma unroll
    for (int v = 0; v < 16; ++v)
      out[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(Ys[acc_row(v, h)][32 * t + c], p[v], out[t], 0, 0, 0);
  }
}

// this lane's own row of a [T, W] block into registers (element 2i + half), through an LDS tile
template <int W>
__device__ __forceinline__ void load_own(float (&own)[W / 2], float (*tmp)[W + 1], const float* src, int blk0, int T,
                                         int tid) {
  const int lane = tid & 63, wave = tid >> 6;
  for (int c = 0; c < 4; ++c) {
    stage_tile<W>(tmp, src, blk0 + 32 * c, T, tid);
    __syncthreads();
    if (wave == c) {
#pragma unroll
      for (int i = 0; i < W / 2; ++i) own[i] = tmp[lane & 31][2 * i + (lane >> 5)];
    }
    __syncthreads();
  }
}

__device__ __forceinline__ float xhalf(float v) { return __shfl_xor(v, 32, 64); }

// ---------------------------------------------------------------- forward
template <int DQK, int DV>
__global__ __launch_bounds__(256) void k_attn_fwd(ic_attn_args a) {
  __shared__ float Ks[kTile][DQK + 1];
  __shared__ float Vs[kTile][DV + 1];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5;
  const int bh = blockIdx.y, T = a.T;
  const int Tv = a.valid_len ? a.valid_len[bh / a.H] : T;  // keys >= Tv are padding
  const float* Q = a.q + (size_t)bh * T * DQK;
  const float* K = a.k + (size_t)bh * T * DQK;
  const float* V = a.v + (size_t)bh * T * DV;
  const int q = blockIdx.x * kOwn + wave * 32 + (lane & 31);
  float qreg[DQK / 2];
  load_own<DQK>(qreg, Ks, Q, blockIdx.x * kOwn, T, tid);
  const bool afa = a.alpha != nullptr;
  const float rate = afa ? -a.alpha[0] * a.lag_dt : 0.f;
  const int Tk = Tv < T ? Tv : T;
  // pass 1: log-sum-exp of this lane's keys (rows acc_row(v, half) of every tile)
  float m = -INFINITY, l = 0.f;
  TileRegs<DQK> kt;
  TileRegs<DV> vt;
  fetch_tile<DQK>(kt, K, 0, T, tid);
  for (int k0 = 0; k0 < Tk; k0 += kTile) {
    put_tile<DQK>(Ks, kt, tid);
    __syncthreads();
    if (k0 + kTile < Tk) fetch_tile<DQK>(kt, K, k0 + kTile, T, tid);
    v16f s = tile_times_own<DQK>(Ks, qreg, lane);
    float tm = -INFINITY;
#pragma unroll
    for (int v = 0; v < 16; ++v) {
      const int key = k0 + acc_row(v, half);
      float x = s[v];
      if (afa) x *= expf(rate * fabsf((float)(key - q)));
      x = key < Tk ? x : -INFINITY;
      s[v] = x;
      tm = fmaxf(tm, x);
    }
    if (tm > -INFINITY) {
      const float mn = fmaxf(m, tm);
      float add = 0.f;
#pragma unroll
      for (int v = 0; v < 16; ++v) add += expf(s[v] - mn);  // exp(-inf) = 0 for the padding keys
      l = l * expf(m - mn) + add;
      m = mn;
    }
    __syncthreads();
  }
  {
    const float m2 = xhalf(m), l2 = xhalf(l);
    const float M = fmaxf(m, m2);
    const float L = (m > -INFINITY ? l * expf(m - M) : 0.f) + (m2 > -INFINITY ? l2 * expf(m2 - M) : 0.f);
    m = M + logf(L);  // the query's log-sum-exp (both halves hold it)
  }
  const float lse = m;
  // pass 2: probabilities, statistics, O^T += V^T P^T
  v16f o[DV / 32];
#pragma unroll
  for (int t = 0; t < DV / 32; ++t) o[t] = zero16();
  float st0 = 0.f, st1 = 0.f;
  fetch_tile<DQK>(kt, K, 0, T, tid);
  fetch_tile<DV>(vt, V, 0, T, tid);
  for (int k0 = 0; k0 < Tk; k0 += kTile) {
    put_tile<DQK>(Ks, kt, tid);
    put_tile<DV>(Vs, vt, tid);
    __syncthreads();
    if (k0 + kTile < Tk) {
      fetch_tile<DQK>(kt, K, k0 + kTile, T, tid);
      fetch_tile<DV>(vt, V, k0 + kTile, T, tid);
    }
    v16f s = tile_times_own<DQK>(Ks, qreg, lane);
#pragma unroll
    for (int v = 0; v < 16; ++v) {
      const int key = k0 + acc_row(v, half);
      float x = s[v];
      if (afa) x *= expf(rate * fabsf((float)(key - q)));
      float p = 0.f;
      if (key < Tk) {
        p = expf(x - lse);
        const float lg = logf(p + a.eps_log);
        st0 += lg;
        st1 += p * lg;
      }
      s[v] = p;
    }
    acc_outer<DV>(o, Vs, s, lane);
    __syncthreads();
  }
  st0 += xhalf(st0);
  st1 += xhalf(st1);
  if (q < T) {
    if (half == 0) {
      a.lse[(size_t)bh * T + q] = lse;
      a.stats[((size_t)bh * T + q) * 2 + 0] = st0;
      a.stats[((size_t)bh * T + q) * 2 + 1] = st1;
    }
    float* orow = a.out + ((size_t)bh * T + q) * DV;
#pragma unroll
    for (int t = 0; t < DV / 32; ++t) {
#pragma unroll
      for (int v = 0; v < 16; ++v) orow[32 * t + acc_row(v, half)] = o[t][v];
    }
  }
}

// ---------------------------------------------------------------- backward: dK, dV (a wave owns 32 keys)
// Scores untransposed (S = Q K^T: queries down the registers, the lane's key across): P and dS in the
// accumulator registers are the B operands of dV^T += dO^T P and dK^T += Q^T dS.
template <int DQK, int DV>
__global__ __launch_bounds__(256) void k_attn_bwd_kv(ic_attn_bwd_args b) {
  __shared__ float Qs[kTile][DQK + 1];
  __shared__ float Gs[kTile][DV + 1];  // dO tile
  __shared__ float lse_s[kTile], dr_s[kTile];
  const ic_attn_args& a = b.f;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5;
  const int bh = blockIdx.y, T = a.T;
  const int Tv = a.valid_len ? a.valid_len[bh / a.H] : T;
  const int Tk = Tv < T ? Tv : T;
  const float* Q = a.q + (size_t)bh * T * DQK;
  const float* K = a.k + (size_t)bh * T * DQK;
  const float* V = a.v + (size_t)bh * T * DV;
  const float* G = b.dout + (size_t)bh * T * DV;
  const int key = blockIdx.x * kOwn + wave * 32 + (lane & 31);
  float kreg[DQK / 2], vreg[DV / 2];
  load_own<DQK>(kreg, Qs, K, blockIdx.x * kOwn, T, tid);
  load_own<DV>(vreg, Gs, V, blockIdx.x * kOwn, T, tid);
  const bool afa = a.alpha != nullptr;
  const float rate = afa ? -a.alpha[0] * a.lag_dt : 0.f;
  const bool key_ok = key < Tk;
  v16f dk[DQK / 32], dv[DV / 32];
#pragma unroll
  for (int t = 0; t < DQK / 32; ++t) dk[t] = zero16();
#pragma unroll
  for (int t = 0; t < DV / 32; ++t) dv[t] = zero16();
  TileRegs<DQK> qt;
  TileRegs<DV> gt;
  fetch_tile<DQK>(qt, Q, 0, T, tid);
  fetch_tile<DV>(gt, G, 0, T, tid);
  for (int q0 = 0; q0 < T; q0 += kTile) {
    put_tile<DQK>(Qs, qt, tid);
    put_tile<DV>(Gs, gt, tid);
    if (tid < kTile) {
      const bool ok = q0 + tid < T;
      lse_s[tid] = ok ? a.lse[(size_t)bh * T + q0 + tid] : 0.f;
      dr_s[tid] = ok ? b.drow[(size_t)bh * T + q0 + tid] : 0.f;
    }
    __syncthreads();
    if (q0 + kTile < T) {
      fetch_tile<DQK>(qt, Q, q0 + kTile, T, tid);
      fetch_tile<DV>(gt, G, q0 + kTile, T, tid);
    }
    v16f s = tile_times_own<DQK>(Qs, kreg, lane);   // S[q][key]
    v16f dp = tile_times_own<DV>(Gs, vreg, lane);   // dP[q][key] = dO[q] . V[key]
#pragma unroll
    for (int v = 0; v < 16; ++v) {
      const int qi = acc_row(v, half), q = q0 + qi;
      float dec = 1.f;
      if (afa) dec = expf(rate * fabsf((float)(key - q)));
      const bool ok = key_ok && q < T;
      const float p = ok ? expf(s[v] * dec - lse_s[qi]) : 0.f;
      s[v] = p;
      dp[v] = p * (dp[v] - dr_s[qi]) * dec;  // d loss / d (q.k)
    }
    acc_outer<DV>(dv, Gs, s, lane);
    acc_outer<DQK>(dk, Qs, dp, lane);
    __syncthreads();
  }
  if (key < T) {
    float* krow = b.dk + ((size_t)bh * T + key) * DQK;
    float* vrow = b.dv + ((size_t)bh * T + key) * DV;
#pragma unroll
    for (int t = 0; t < DQK / 32; ++t) {
#pragma unroll
      for (int v = 0; v < 16; ++v) krow[32 * t + acc_row(v, half)] = dk[t][v];
    }
#pragma unroll
    for (int t = 0; t < DV / 32; ++t) {
#pragma unroll
      for (int v = 0; v < 16; ++v) vrow[32 * t + acc_row(v, half)] = dv[t][v];
    }
  }
}

// ---------------------------------------------------------------- backward: dQ (+ d alpha) (a wave owns 32 queries)
template <int DQK, int DV>
__global__ __launch_bounds__(256) void k_attn_bwd_q(ic_attn_bwd_args b) {
  __shared__ float Ks[kTile][DQK + 1];
  __shared__ float Vs[kTile][DV + 1];
  __shared__ float wsum[4];
  const ic_attn_args& a = b.f;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5;
  const int bh = blockIdx.y, T = a.T;
  const int Tv = a.valid_len ? a.valid_len[bh / a.H] : T;
  const int Tk = Tv < T ? Tv : T;
  const float* Q = a.q + (size_t)bh * T * DQK;
  const float* K = a.k + (size_t)bh * T * DQK;
  const float* V = a.v + (size_t)bh * T * DV;
  const float* G = b.dout + (size_t)bh * T * DV;
  const int q = blockIdx.x * kOwn + wave * 32 + (lane & 31);
  float qreg[DQK / 2], greg[DV / 2];
  load_own<DQK>(qreg, Ks, Q, blockIdx.x * kOwn, T, tid);
  load_own<DV>(greg, Vs, G, blockIdx.x * kOwn, T, tid);
  const bool q_ok = q < T;
  const float lse = q_ok ? a.lse[(size_t)bh * T + q] : 0.f;
  const float dr = q_ok ? b.drow[(size_t)bh * T + q] : 0.f;
  const bool afa = a.alpha != nullptr;
  const float rate = afa ? -a.alpha[0] * a.lag_dt : 0.f;
  v16f dq[DQK / 32];
#pragma unroll
  for (int t = 0; t < DQK / 32; ++t) dq[t] = zero16();
  float dal = 0.f;
  TileRegs<DQK> kt;
  TileRegs<DV> vt;
  fetch_tile<DQK>(kt, K, 0, T, tid);
  fetch_tile<DV>(vt, V, 0, T, tid);
  for (int k0 = 0; k0 < Tk; k0 += kTile) {
    put_tile<DQK>(Ks, kt, tid);
    put_tile<DV>(Vs, vt, tid);
    __syncthreads();
    if (k0 + kTile < Tk) {
      fetch_tile<DQK>(kt, K, k0 + kTile, T, tid);
      fetch_tile<DV>(vt, V, k0 + kTile, T, tid);
    }
    v16f s = tile_times_own<DQK>(Ks, qreg, lane);   // S^T[key][q]
    v16f dp = tile_times_own<DV>(Vs, greg, lane);   // dP^T[key][q]
#pragma unroll
    for (int v = 0; v < 16; ++v) {
      const int key = k0 + acc_row(v, half);
      const float lag = fabsf((float)(key - q));
      float dec = 1.f;
      if (afa) dec = expf(rate * lag);
      const bool ok = q_ok && key < Tk;
      const float sc = s[v] * dec;
      const float p = ok ? expf(sc - lse) : 0.f;
      const float dsc = p * (dp[v] - dr);  // d loss / d score
      if (afa) dal += dsc * sc * (-lag * a.lag_dt);
      s[v] = dsc * dec;
    }
    acc_outer<DQK>(dq, Ks, s, lane);
    __syncthreads();
  }
  if (q_ok) {
    float* qrow = b.dq + ((size_t)bh * T + q) * DQK;
#pragma unroll
    for (int t = 0; t < DQK / 32; ++t) {
#pragma unroll
      for (int v = 0; v < 16; ++v) qrow[32 * t + acc_row(v, half)] = dq[t][v];
    }
  }
  if (b.dalpha_part) {  // one partial sum per workgroup, added up in a fixed order by the caller
    const float w = wave_sum(dal);
    if (lane == 0) wsum[wave] = w;
    __syncthreads();
    if (tid == 0) b.dalpha_part[(size_t)bh * gridDim.x + blockIdx.x] = (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]);
  }
}

int check_attn(const ic_attn_args* a, const char* who) {
  IC_REQUIRE(a != nullptr, "%s: NULL args", who);
  IC_REQUIRE(a->B >= 1 && a->H >= 1 && a->T >= 1, "%s: B=%d H=%d T=%d", who, a->B, a->H, a->T);
  IC_REQUIRE((int64_t)a->B * a->H <= 65535, "%s: B*H=%lld exceeds the grid's y extent", who, (long long)a->B * a->H);
  IC_REQUIRE(a->dqk % 32 == 0 && a->dqk >= 32 && a->dqk <= 256, "%s: dqk=%d (need a multiple of 32 in [32, 256])", who, a->dqk);
  IC_REQUIRE(a->dv % 32 == 0 && a->dv >= 32 && a->dv <= 128, "%s: dv=%d (need a multiple of 32 in [32, 128])", who, a->dv);
  IC_REQUIRE(a->q && a->k && a->v && a->lse, "%s: NULL tensor pointer", who);
  return IC_OK;
}

}  // namespace

// (DQK, DV) pairs the mixer produces: head width 32 / 64 / 128, alone (AFA) or with the 64 Fourier-bias columns
#define IC_ATTN_DISPATCH(a, BODY)                                                              \
  do {                                                                                         \
    const int key_ = (a)->dqk * 1000 + (a)->dv;                                                \
    switch (key_) {                                                                            \
      case 32032: { constexpr int DQK = 32, DV = 32; BODY; } break;                            \
      case 64064: { constexpr int DQK = 64, DV = 64; BODY; } break;                            \
      case 128128: { constexpr int DQK = 128, DV = 128; BODY; } break;                         \
      case 96032: { constexpr int DQK = 96, DV = 32; BODY; } break;                            \
      case 128064: { constexpr int DQK = 128, DV = 64; BODY; } break;                          \
      case 192128: { constexpr int DQK = 192, DV = 128; BODY; } break;                         \
      default:                                                                                 \
        IC_REQUIRE(false, "ic_attn: no kernel for dqk=%d, dv=%d", (a)->dqk, (a)->dv);          \
    }                                                                                          \
  } while (0)

extern "C" int ic_attn_fwd(const ic_attn_args* a, ic_stream_t stream) {
  int rc = check_attn(a, "ic_attn_fwd");
  if (rc) return rc;
  IC_REQUIRE(a->out && a->stats, "ic_attn_fwd: NULL output pointer");
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid(ic_ceil_div(a->T, kOwn), a->B * a->H);
  IC_ATTN_DISPATCH(a, { IC_TIMED(IC_KID_attn_fwd, st, hipLaunchKernelGGL((k_attn_fwd<DQK, DV>), grid, dim3(256), 0, st, *a)); });
  IC_LAUNCH_CHECK("ic_attn_fwd");
  return IC_OK;
}

extern "C" int ic_attn_bwd(const ic_attn_bwd_args* b, ic_stream_t stream) {
  IC_REQUIRE(b != nullptr, "ic_attn_bwd: NULL args");
  const ic_attn_args* a = &b->f;
  int rc = check_attn(a, "ic_attn_bwd");
  if (rc) return rc;
  IC_REQUIRE(b->dout && b->drow && b->dq && b->dk && b->dv, "ic_attn_bwd: NULL tensor pointer");
  IC_REQUIRE(a->alpha == nullptr || b->dalpha_part != nullptr, "ic_attn_bwd: alpha without dalpha_part");
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid(ic_ceil_div(a->T, kOwn), a->B * a->H);
  IC_ATTN_DISPATCH(a, {
    IC_TIMED(IC_KID_attn_bwd_kv, st, hipLaunchKernelGGL((k_attn_bwd_kv<DQK, DV>), grid, dim3(256), 0, st, *b));
    IC_TIMED(IC_KID_attn_bwd_q, st, hipLaunchKernelGGL((k_attn_bwd_q<DQK, DV>), grid, dim3(256), 0, st, *b));
  });
  IC_LAUNCH_CHECK("ic_attn_bwd");
  return IC_OK;
}
