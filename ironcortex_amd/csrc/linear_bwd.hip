// Backward of y = act(W x + b) for a small batch of rows (the planner / critic / verifier / RTD
// heads that train_step differentiates through, reference heads.py:39-91, energy.py:12-30).  The forward is
// ic_gemv_chain with nbatch rows; here the pre-activation is recomputed (the weights are re-read anyway for the
// input gradient), so nothing but x, W, b is saved.
//   gz[b,o] = gy[b,o] * act'(W[o,:] . x[b,:] + bias[o])
//   gW[o,:] = sum_b gz[b,o] x[b,:]      gb[o] = sum_b gz[b,o]      gx[b,:] = sum_o gz[b,o] W[o,:]
// Workgroup = 32 output rows (a wave 8 of them): every gW / gb row has one owner; the input gradient leaves as one
// partial row per workgroup and batch row, summed in workgroup order by k_linear_bwd_sum.  No float atomics.
#include "ic_common.h"

namespace {

constexpr int kMaxB = 16;
constexpr int kRowsWG = 32;

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

__global__ __launch_bounds__(256) void k_linear_bwd_rows(ic_linear_bwd_args a, float* part) {
  __shared__ float gz_s[kMaxB][kRowsWG];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = a.n, B = a.B;
  for (int r = 0; r < 8; ++r) {
    const int ol = wave * 8 + r, o = blockIdx.x * kRowsWG + ol;
    float gz[kMaxB];
#pragma unroll
    for (int b = 0; b < kMaxB; ++b) gz[b] = 0.f;
    if (o < a.rows) {  // wave-uniform
      const float* Wr = a.W + (size_t)o * n;
      float acc[kMaxB];
#pragma unroll
      for (int b = 0; b < kMaxB; ++b) acc[b] = 0.f;
      for (int i = lane; i < n; i += 64) {
        const float w = Wr[i];
#pragma unroll
        for (int b = 0; b < kMaxB; ++b)
          if (b < B) acc[b] = fmaf(w, a.x[(size_t)b * n + i], acc[b]);
      }
      const float bias = a.b ? a.b[o] : 0.f;
      float gbs = 0.f;
#pragma unroll
      for (int b = 0; b < kMaxB; ++b) {
        if (b < B) {
          const float z = wave_sum(acc[b]) + bias;
          gz[b] = a.gy[(size_t)b * a.rows + o] * act_grad(z, a.act);
          gbs += gz[b];
        }
      }
      if (a.gW) {
        float* gWr = a.gW + (size_t)o * n;
        for (int i = lane; i < n; i += 64) {
          float s = 0.f;
#pragma unroll
          for (int b = 0; b < kMaxB; ++b)
            if (b < B) s = fmaf(gz[b], a.x[(size_t)b * n + i], s);
          gWr[i] = s;
        }
      }
      if (lane == 0 && a.gb) a.gb[o] = gbs;
    }
    if (lane < kMaxB) {
      float v = 0.f;
#pragma unroll
      for (int b = 0; b < kMaxB; ++b) v = lane == b ? gz[b] : v;
      gz_s[lane][ol] = v;
    }
  }
  __syncthreads();
  if (!part) return;
  const int o0 = blockIdx.x * kRowsWG;
  const int no = a.rows - o0 < kRowsWG ? a.rows - o0 : kRowsWG;
  for (int i = tid; i < n; i += 256) {
    float acc[kMaxB];
#pragma unroll
    for (int b = 0; b < kMaxB; ++b) acc[b] = 0.f;
    for (int ol = 0; ol < no; ++ol) {
      const float w = a.W[(size_t)(o0 + ol) * n + i];
#pragma unroll
      for (int b = 0; b < kMaxB; ++b)
        if (b < B) acc[b] = fmaf(w, gz_s[b][ol], acc[b]);
    }
#pragma unroll
    for (int b = 0; b < kMaxB; ++b)
      if (b < B) part[((size_t)blockIdx.x * B + b) * n + i] = acc[b];
  }
}

__global__ __launch_bounds__(256) void k_linear_bwd_sum(const float* part, float* gx, int nblk, int total) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  float s = 0.f;
  for (int k = 0; k < nblk; ++k) s += part[(size_t)k * total + i];
  gx[i] = s;
}

}  // namespace

extern "C" int ic_linear_bwd(const ic_linear_bwd_args* a, ic_stream_t stream) {
  IC_REQUIRE(a != nullptr, "ic_linear_bwd: NULL args");
  IC_REQUIRE(a->B >= 1 && a->B <= kMaxB, "ic_linear_bwd: B=%d (need 1..%d rows)", a->B, kMaxB);
  IC_REQUIRE(a->n >= 1 && a->rows >= 1, "ic_linear_bwd: n=%d rows=%d", a->n, a->rows);
  IC_REQUIRE(a->act >= IC_ACT_NONE && a->act <= IC_ACT_SILU, "ic_linear_bwd: act=%d", a->act);
  IC_REQUIRE(a->W && a->x && a->gy, "ic_linear_bwd: NULL W / x / gy");
  const int nblk = ic_ceil_div(a->rows, kRowsWG);
  IC_REQUIRE(!a->gx || nblk == 1 || a->slab, "ic_linear_bwd: gx with %d row blocks needs the slab workspace", nblk);
  hipStream_t st = (hipStream_t)stream;
  float* part = !a->gx ? nullptr : (nblk == 1 ? a->gx : a->slab);  // one block: its partial rows ARE gx
  IC_TIMED(IC_KID_linear_bwd, st, hipLaunchKernelGGL(k_linear_bwd_rows, dim3(nblk), dim3(256), 0, st, *a, part));
  if (a->gx && nblk > 1) {
    const int total = a->B * a->n;
    IC_TIMED(IC_KID_linear_bwd, st, hipLaunchKernelGGL(k_linear_bwd_sum, dim3(ic_ceil_div(total, 256)), dim3(256), 0, st,
                                                        (const float*)a->slab, a->gx, nblk, total));
  }
  IC_LAUNCH_CHECK("ic_linear_bwd");
  return IC_OK;
}
