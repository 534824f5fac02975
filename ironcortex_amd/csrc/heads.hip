// Loop heads, forward only: workspace write, token logits, steering statistics, vector MLPs.
// Replaces the per-inner-step ATen calls of reference model.py:296-321 (heads), :371-386
// (planner / critic) and :425-438 (verifier energy + branch score) when no gradient is needed.
// All of it is latency-bound vector work (<= 2.5 MB of weights per step): the kernels are
// written for few launches and coalesced row reads, not for bandwidth.
#include "ic_common.h"
#include "ic_kwta.h"

namespace {

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

// One wave per output row, 8 rows per workgroup; the input vector is staged (and optionally
// layer-normalised) in LDS once per workgroup.
__global__ __launch_bounds__(256) void k_gemv(ic_gemv_desc g) {
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
    if (lane == 0) g.y[row] = apply_act(acc + (g.b ? g.b[row] : 0.f), g.act);
  }
}

// slots_out[j] = KWTA(slots_in[j] + H[j-th active region], k)     (model.py:300-306, heads.py:34-36)
template <int DCH>
__global__ __launch_bounds__(256) void k_ws_write(ic_heads_args a) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = blockIdx.x * 4 + wave;
  if (j >= a.n_slots) return;
  // index of the (j+1)-th set bit of mask
  int found = -1, seen = 0;
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

// blockIdx.y == 0: z = lm_W . H[motor] + lm_b (gate + facets); blockIdx.y == 1 + r: the head of
// active region r != motor on H[r].  One wave per two rows.
template <int DCH>
__global__ __launch_bounds__(256) void k_token_rows(ic_heads_args a) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int d = a.d;
  const int grp = blockIdx.y;
  int rows;
  const float *W, *b, *x;
  float* y;
  if (grp == 0) {
    rows = a.Kf + a.Kf * a.V;
    W = a.lm_W - (size_t)a.Kf * d; b = a.lm_b - a.Kf;  // facet row i sits at z[Kf + i]
    x = a.H + (size_t)a.motor_idx * d; y = a.z;
  } else {
    const int r = grp - 1;
    if (r == a.motor_idx || !a.mask[r]) return;
    rows = a.V;
    W = a.rh_W + (size_t)r * a.V * d; b = a.rh_b + (size_t)r * a.V; x = a.H + (size_t)r * d;
    y = a.per_region + (size_t)r * a.V;
  }
  const int row0 = blockIdx.x * 8 + wave * 2;
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

// One workgroup: mixture-of-facets logits + region heads -> logits, softmax, steering statistics.
__global__ __launch_bounds__(256) void k_logits_combine(ic_heads_args a) {
  __shared__ float red[4];
  const int V = a.V, Kf = a.Kf;
  // facet gate: softmax over Kf (every thread, Kf is tiny)
  float gmx = -INFINITY;
  for (int f = 0; f < Kf; ++f) gmx = fmaxf(gmx, a.z[f]);
  float gsum = 0.f;
  for (int f = 0; f < Kf; ++f) gsum += expf(a.z[f] - gmx);
  float nw = 0.f;
  for (int r = 0; r < a.R; ++r) nw += (r != a.motor_idx && a.mask[r]) ? 1.0f : 0.0f;
  const float denom = 1.0f + nw;
  float mx = -INFINITY;
  for (int v = threadIdx.x; v < V; v += 256) {
    float acc = 0.f;
    for (int f = 0; f < Kf; ++f) acc += (expf(a.z[f] - gmx) / gsum) * a.z[Kf + f * V + v];
    float reg = 0.f;
    for (int r = 0; r < a.R; ++r)
      if (r != a.motor_idx && a.mask[r]) reg += a.per_region[(size_t)r * V + v];
    const float lg = (acc + reg) / denom;
    a.logits[v] = lg;
    mx = fmaxf(mx, lg);
  }
  mx = block_max(mx, red);
  float se = 0.f;
  for (int v = threadIdx.x; v < V; v += 256) se += expf(a.logits[v] - mx);
  se = block_sum(se, red);
  float clp = 0.f, pmax = 0.f;
  for (int v = threadIdx.x; v < V; v += 256) {
    const float p = expf(a.logits[v] - mx) / se;
    a.probs[v] = p;
    clp += p * logf(p + 1e-12f);
    pmax = fmaxf(pmax, p);
  }
  clp = block_sum(clp, red);
  pmax = block_max(pmax, red);
  // goodness(H) - goodness(H_par): mean over regions of mean_d h^2   (utils.py:159-162)
  const int n = a.R * a.d;
  float g1 = 0.f, g0 = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) {
    const float h1 = a.H[i], h0 = a.H_par[i];
    g1 = fmaf(h1, h1, g1);
    g0 = fmaf(h0, h0, g0);
  }
  g1 = block_sum(g1, red);
  g0 = block_sum(g0, red);
  float sp = 0.f;
  if (a.surprise)
    for (int r = threadIdx.x; r < a.R; r += 256) sp += a.surprise[r];
  sp = block_sum(sp, red) / (float)a.R;
  if (threadIdx.x == 0) {
    a.stats[0] = 1.0f - pmax;
    a.stats[1] = clp;
    a.stats[2] = g1 / (float)n - g0 / (float)n;
    a.stats[3] = sp;
    if (a.aux_surprise) *a.aux_surprise = sp;
  }
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

}  // namespace

extern "C" int ic_gemv_chain(const ic_gemv_desc* descs, int32_t n, ic_stream_t stream) {
  IC_REQUIRE(descs != nullptr && n >= 0, "ic_gemv_chain: NULL descriptors");
  hipStream_t st = (hipStream_t)stream;
  for (int i = 0; i < n; ++i) {
    const ic_gemv_desc& g = descs[i];
    const int cols = g.n0 + g.n1 + g.n2;
    IC_REQUIRE(g.W && g.y && g.rows > 0, "ic_gemv_chain[%d]: NULL W/y or rows=%d", i, g.rows);
    IC_REQUIRE(g.n0 >= 0 && g.n1 >= 0 && g.n2 >= 0 && cols > 0 && cols <= kMaxCols,
               "ic_gemv_chain[%d]: %d input columns (need 1..%d)", i, cols, kMaxCols);
    IC_REQUIRE((g.n0 == 0 || g.x0) && (g.n1 == 0 || g.x1) && (g.n2 == 0 || g.x2),
               "ic_gemv_chain[%d]: NULL input segment", i);
    IC_REQUIRE(!g.ln_g || g.ln_b, "ic_gemv_chain[%d]: LayerNorm weight without bias", i);
    IC_REQUIRE(g.act >= IC_ACT_NONE && g.act <= IC_ACT_SILU, "ic_gemv_chain[%d]: act=%d", i, g.act);
    IC_TIMED(IC_KID_gemv, st, hipLaunchKernelGGL(k_gemv, dim3(ic_ceil_div(g.rows, 8)), dim3(256), 0, st, g));
  }
  IC_LAUNCH_CHECK("ic_gemv_chain");
  return IC_OK;
}

extern "C" int ic_heads_fwd(const ic_heads_args* a, ic_stream_t stream) {
  IC_REQUIRE(a != nullptr, "ic_heads_fwd: NULL args");
  IC_REQUIRE(a->R > 0 && a->V > 0 && a->Kf > 0 && a->Kf <= 64 && a->n_slots >= 0,
             "ic_heads_fwd: R=%d V=%d Kf=%d n_slots=%d", a->R, a->V, a->Kf, a->n_slots);
  IC_REQUIRE(a->d >= 4 && a->d % 4 == 0 && a->d <= 1024, "ic_heads_fwd: d=%d unsupported", a->d);
  IC_REQUIRE(a->motor_idx >= 0 && a->motor_idx < a->R, "ic_heads_fwd: motor_idx=%d", a->motor_idx);
  IC_REQUIRE(a->H && a->H_par && a->mask && a->lm_gate_W && a->lm_gate_b && a->lm_W && a->lm_b && a->rh_W && a->rh_b && a->z &&
                 a->per_region && a->logits && a->probs && a->stats,
             "ic_heads_fwd: NULL pointer");
  IC_REQUIRE(a->n_slots == 0 || (a->slots_in && a->slots_out), "ic_heads_fwd: NULL workspace slots");
  hipStream_t st = (hipStream_t)stream;
  const int lm_rows = a->Kf + a->Kf * a->V;
  const int max_rows = lm_rows > a->V ? lm_rows : a->V;
  IC_DISPATCH_DCH(a->d, {
    if (a->n_slots > 0)
      IC_TIMED(IC_KID_ws_write, st, hipLaunchKernelGGL(k_ws_write<DCH>, dim3(ic_ceil_div(a->n_slots, 4)), dim3(256), 0, st, *a));
    IC_TIMED(IC_KID_token_rows, st, hipLaunchKernelGGL(k_token_rows<DCH>, dim3(ic_ceil_div(max_rows, 8), 1 + a->R), dim3(256), 0, st, *a));
  });
  IC_TIMED(IC_KID_logits_combine, st, hipLaunchKernelGGL(k_logits_combine, dim3(1), dim3(256), 0, st, *a));
  IC_LAUNCH_CHECK("ic_heads_fwd");
  return IC_OK;
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
  IC_TIMED(IC_KID_steer_select, st, hipLaunchKernelGGL(k_steer_select, dim3(1), dim3(256), 0, st, *a));
  IC_LAUNCH_CHECK("ic_steer_select");
  return IC_OK;
}
