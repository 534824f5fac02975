// Backward of y = act(W x + b) for a small batch of rows (the planner / critic / verifier / RTD
// heads that train_step differentiates through, reference heads.py:39-91, energy.py:12-30).  The forward is
// ic_gemv_chain with nbatch rows; here the pre-activation is recomputed (the weights are re-read anyway for the
// input gradient), so nothing but x, W, b is saved.
//   gz[b,o] = gy[b,o] * act'(W[o,:] . x[b,:] + bias[o])
//   gW[o,:] = sum_b gz[b,o] x[b,:]      gb[o] = sum_b gz[b,o]      gx[b,:] = sum_o gz[b,o] W[o,:]
// Two launches that fill the chip at these sizes (rows, n <= a few hundred; round 2 ran ceil(rows/32) = 8
// workgroups with a serial row loop: 97 us per call):
//   k_linear_bwd_gz   workgroup = 8 output rows (a wave 2 of them): z, gz, the gW / gb rows (one owner each)
//   k_linear_bwd_gx   workgroup = (32 input columns, 4 batch rows): gx = gz^T W with the rows of W split over 32
//                     thread groups, partial sums added in group order.  No float atomics, run-to-run identical.
#include "ic_common.h"

namespace {

constexpr int kMaxB = 16;
constexpr int kRowsWG = 8;   // output rows per workgroup of k_linear_bwd_gz
constexpr int kColsWG = 32;  // input columns per workgroup of k_linear_bwd_gx (8 column quads)
constexpr int kNB = 4;       // batch rows per workgroup of k_linear_bwd_gx
constexpr int kRG = 32;      // row groups (threads along the reduction) of k_linear_bwd_gx
constexpr int kGxMaxRows = 2048;

__device__ __forceinline__ float act_grad(float z, int act) {
  if (act == IC_ACT_GELU) {  // d/dz [z Phi(z)] = Phi(z) + z phi(z)
    const float cdf = 0.5f * (1.0f + erff(z * 0.70710678118654752440f));
    const float pdf = 0.39894228040143267794f * expf(-0.5f * z * z);
    return cdf + z * pdf;
  }
  if (act == IC_ACT_SILU) {
    const float s = 1.0f / (1.0f + expf(-z));
    return s * (1.0f + z * (1.0f - s));
  }
  return 1.0f;
}

// XS: the B input rows are staged in LDS once per workgroup (B * n <= kXsFloats); otherwise they are read from
// global memory with every row's load issued before the first use (a load inside an `if (b < B)` block makes the
// compiler wait for each in turn: 19 us for ONE wave at B = 16 -- measured on the first version of this kernel).
constexpr int kXsFloats = 12288;  // 48 KB

template <bool XS>
__global__ __launch_bounds__(256) void k_linear_bwd_gz(ic_linear_bwd_args a, float* gzw) {
  __shared__ float xs[XS ? kXsFloats : 4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n = a.n, B = a.B;
  if (XS) {
    for (int t = threadIdx.x; t < B * n; t += 256) xs[t] = a.x[t];
    __syncthreads();
  }
  const int o0 = blockIdx.x * kRowsWG + wave * 2;
  if (o0 >= a.rows) return;
  const bool two = o0 + 1 < a.rows;
  const float* W0 = a.W + (size_t)o0 * n;
  const float* W1 = W0 + (two ? n : 0);
  const float* xb = XS ? xs : a.x;
  float acc0[kMaxB], acc1[kMaxB];
#pragma unroll
  for (int b = 0; b < kMaxB; ++b) acc0[b] = acc1[b] = 0.f;
  for (int i = lane; i < n; i += 64) {
    const float w0 = W0[i], w1 = W1[i];
    float xv[kMaxB];
#pragma unroll
    for (int b = 0; b < kMaxB; ++b) xv[b] = xb[(size_t)(b < B ? b : 0) * n + i];  // unconditional: all in flight together
#pragma unroll
    for (int b = 0; b < kMaxB; ++b) {
      acc0[b] = fmaf(w0, xv[b], acc0[b]);
      acc1[b] = fmaf(w1, xv[b], acc1[b]);
    }
  }
  const float bias0 = a.b ? a.b[o0] : 0.f, bias1 = (a.b && two) ? a.b[o0 + 1] : 0.f;
  float gy0[kMaxB], gy1[kMaxB];
#pragma unroll
  for (int b = 0; b < kMaxB; ++b) {
    const int bc = b < B ? b : 0;
    gy0[b] = a.gy[(size_t)bc * a.rows + o0];
    gy1[b] = a.gy[(size_t)bc * a.rows + o0 + (two ? 1 : 0)];
  }
  float gz0[kMaxB], gz1[kMaxB];
  float gb0 = 0.f, gb1 = 0.f;
#pragma unroll
  for (int b = 0; b < kMaxB; ++b) {
    gz0[b] = gz1[b] = 0.f;
    if (b < B) {  // (wave-uniform; rows b >= B keep gz = 0 and drop out of every sum below)
      const float z0 = wave_sum(acc0[b]) + bias0, z1 = wave_sum(acc1[b]) + bias1;
      gz0[b] = gy0[b] * act_grad(z0, a.act);
      gz1[b] = two ? gy1[b] * act_grad(z1, a.act) : 0.f;
      gb0 += gz0[b];
      gb1 += gz1[b];
    }
  }
  if (a.gW) {
    float* g0 = a.gW + (size_t)o0 * n;
    float* g1 = g0 + n;
    for (int i = lane; i < n; i += 64) {
      float xv[kMaxB];
#pragma unroll
      for (int b = 0; b < kMaxB; ++b) xv[b] = xb[(size_t)(b < B ? b : 0) * n + i];
      float s0 = 0.f, s1 = 0.f;
#pragma unroll
      for (int b = 0; b < kMaxB; ++b) {  // gz = 0 for b >= B: fmaf(0, x, s) == s exactly (x is finite input data)
        s0 = fmaf(gz0[b], xv[b], s0);
        s1 = fmaf(gz1[b], xv[b], s1);
      }
      g0[i] = s0;
      if (two) g1[i] = s1;
    }
  }
  if (lane == 0 && a.gb) {
    a.gb[o0] = gb0;
    if (two) a.gb[o0 + 1] = gb1;
  }
  if (gzw && lane < B) {  // gz[b, o] for the input-gradient kernel
    float v0 = 0.f, v1 = 0.f;
#pragma unroll
    for (int b = 0; b < kMaxB; ++b) {
      v0 = lane == b ? gz0[b] : v0;
      v1 = lane == b ? gz1[b] : v1;
    }
    gzw[(size_t)lane * a.rows + o0] = v0;
    if (two) gzw[(size_t)lane * a.rows + o0 + 1] = v1;
  }
}

// gx[b, i] = sum_o gz[b,o] W[o,i]: thread (cq, rg) owns the column quad cq of the tile and the rows rg, rg + kRG, ...
__global__ __launch_bounds__(256) void k_linear_bwd_gx(ic_linear_bwd_args a, const float* gzw) {
  __shared__ float gz_s[kNB][kGxMaxRows];
  __shared__ float4 red[kRG][kNB][kColsWG / 4];
  const int tid = threadIdx.x;
  constexpr int NQ = kColsWG / 4;
  const int cq = tid % NQ, rg = tid / NQ;
  const int n = a.n, rows = a.rows;
  const int b0 = blockIdx.y * kNB;
  const int nb = a.B - b0 < kNB ? a.B - b0 : kNB;
  for (int t = tid; t < kNB * rows; t += 256) {
    const int bb = t / rows, o = t - bb * rows;
    gz_s[bb][o] = bb < nb ? gzw[(size_t)(b0 + bb) * rows + o] : 0.f;
  }
  __syncthreads();
  const int c0 = blockIdx.x * kColsWG + cq * 4;
  float4 acc[kNB];
#pragma unroll
  for (int bb = 0; bb < kNB; ++bb) acc[bb] = make_float4(0.f, 0.f, 0.f, 0.f);
  const bool vec = (n & 3) == 0 && c0 + 3 < n;
  auto load_w = [&](int o) {
    float4 w = make_float4(0.f, 0.f, 0.f, 0.f);
    if (o < rows) {
      const float* Wr = a.W + (size_t)o * n;
      if (vec) {
        w = *reinterpret_cast<const float4*>(Wr + c0);
      } else {
        if (c0 < n) w.x = Wr[c0];
        if (c0 + 1 < n) w.y = Wr[c0 + 1];
        if (c0 + 2 < n) w.z = Wr[c0 + 2];
        if (c0 + 3 < n) w.w = Wr[c0 + 3];
      }
    }
    return w;
  };
  constexpr int UR = 8;  // rows in flight per thread (32 KB per workgroup: a CU's stream rate is bytes in flight / latency)
  for (int o = rg; o < rows; o += kRG * UR) {
    float4 w[UR];
#pragma unroll
    for (int u = 0; u < UR; ++u) w[u] = load_w(o + u * kRG);
#pragma unroll
    for (int u = 0; u < UR; ++u) {  // rows past the end carry w = 0 and gz_s is never read for them
      const int ou = o + u * kRG;
      if (ou < rows) {
#pragma unroll
        for (int bb = 0; bb < kNB; ++bb) {
          const float g = gz_s[bb][ou];
          acc[bb].x = fmaf(g, w[u].x, acc[bb].x);
          acc[bb].y = fmaf(g, w[u].y, acc[bb].y);
          acc[bb].z = fmaf(g, w[u].z, acc[bb].z);
          acc[bb].w = fmaf(g, w[u].w, acc[bb].w);
        }
      }
    }
  }
#pragma unroll
  for (int bb = 0; bb < kNB; ++bb) red[rg][bb][cq] = acc[bb];
  __syncthreads();
  // NQ column quads x kNB batch rows outputs of 4 floats: one thread each, partial sums added in row-group order
  if (tid < kNB * NQ) {
    const int bb = tid / NQ, q = tid % NQ;
    float4 s = red[0][bb][q];
    for (int g = 1; g < kRG; ++g) {
      const float4 t = red[g][bb][q];
      s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
    }
    const int c = blockIdx.x * kColsWG + q * 4;
    if (bb < nb) {
      float* out = a.gx + (size_t)(b0 + bb) * n;
      if (c < n) out[c] = s.x;
      if (c + 1 < n) out[c + 1] = s.y;
      if (c + 2 < n) out[c + 2] = s.z;
      if (c + 3 < n) out[c + 3] = s.w;
    }
  }
}

}  // namespace

extern "C" int ic_linear_bwd(const ic_linear_bwd_args* a, ic_stream_t stream) {
  IC_REQUIRE(a != nullptr, "ic_linear_bwd: NULL args");
  IC_REQUIRE(a->B >= 1 && a->B <= kMaxB, "ic_linear_bwd: B=%d (need 1..%d rows)", a->B, kMaxB);
  IC_REQUIRE(a->n >= 1 && a->rows >= 1, "ic_linear_bwd: n=%d rows=%d", a->n, a->rows);
  IC_REQUIRE(a->act >= IC_ACT_NONE && a->act <= IC_ACT_SILU, "ic_linear_bwd: act=%d", a->act);
  IC_REQUIRE(a->W && a->x && a->gy, "ic_linear_bwd: NULL W / x / gy");
  IC_REQUIRE(!a->gx || a->slab, "ic_linear_bwd: gx needs the slab workspace (B * rows floats)");
  IC_REQUIRE(!a->gx || a->rows <= kGxMaxRows, "ic_linear_bwd: gx with rows=%d (need <= %d)", a->rows, kGxMaxRows);
  hipStream_t st = (hipStream_t)stream;
  // algorithmic bytes: W read once per launch that needs it, gW written, x / gy / gx rows
  IC_BYTES(IC_KID_linear_bwd, 4.0 * ((double)a->rows * a->n * (1.0 + (a->gW ? 1.0 : 0.0) + (a->gx ? 1.0 : 0.0)) +
                                     (double)a->B * (2.0 * a->n + 2.0 * a->rows)));
  if (a->B * a->n <= kXsFloats)
    IC_TIMED(IC_KID_linear_bwd, st, hipLaunchKernelGGL(k_linear_bwd_gz<true>, dim3(ic_ceil_div(a->rows, kRowsWG)), dim3(256), 0, st, *a,
                                                        a->gx ? a->slab : nullptr));
  else
    IC_TIMED(IC_KID_linear_bwd, st, hipLaunchKernelGGL(k_linear_bwd_gz<false>, dim3(ic_ceil_div(a->rows, kRowsWG)), dim3(256), 0, st, *a,
                                                        a->gx ? a->slab : nullptr));
  if (a->gx) {
    IC_TIMED(IC_KID_linear_bwd, st, hipLaunchKernelGGL(k_linear_bwd_gx, dim3(ic_ceil_div(a->n, kColsWG), ic_ceil_div(a->B, kNB)),
                                                        dim3(256), 0, st, *a, (const float*)a->slab));
  }
  IC_LAUNCH_CHECK("ic_linear_bwd");
  return IC_OK;
}
