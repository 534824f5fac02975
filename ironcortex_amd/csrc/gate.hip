// Gate scoring + neighbour-suppressed top-k (reference gate.py:54-87, utils.py:107-124),
// RMSNorm / goodness row primitives (utils.py:20-35,159-162).
#include <stdarg.h>
#include <string.h>

#include "ic_common.h"
#include "ic_gate.h"

// ---------------------------------------------------------------- error plumbing
static thread_local char g_err[512] = "";
void ic_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
extern "C" const char* ic_last_error(void) { return g_err; }
extern "C" int ic_version(void) { return 1; }
extern "C" int ic_device_check(char* name, int cap) {
  int dev = 0;
  hipDeviceProp_t prop;
  if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) {
    ic_set_error("ic_device_check: no HIP device");
    return IC_EDEVICE;
  }
  if (name && cap > 0) snprintf(name, cap, "%s", prop.gcnArchName);
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
    ic_set_error("ic_device_check: built for gfx950, device is %s", prop.gcnArchName);
    return IC_EDEVICE;
  }
  return ic_heads_static_setup(nullptr);  // per-device address of ic_heads_fwd_static's scratch array
}

namespace {

// scores[s,r] = w1*mean|H| + w2*gain + w3*focus - w4*homeo + t5 + t6, left to right in fp32
template <int DCH>
__global__ __launch_bounds__(256) void k_gate_scores(ic_gate_args a) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int row = blockIdx.x * 4 + wave;
  if (row >= a.S * a.R) return;
  const int s = row / a.R, r = row % a.R;
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
    float sc = a.w1 * content;
    sc = sc + a.w2 * a.gain_ema[r];
    sc = sc + a.w3 * focus;
    sc = sc - a.w4 * a.homeo[r];
    sc = sc + gate_t5(a, s);
    sc = sc + gate_t6(a, s);
    a.scores[row] = sc;
  }
}

// One workgroup per stream.
constexpr int kGateLdsNbr = 6144;  // a hex map of 1024 regions
// (256 threads, or 1024 on maps above 256 regions: one region per thread in the selection rounds of ic_gate.h)
__global__ __launch_bounds__(1024) void k_gate_select(ic_gate_args a) {
  __shared__ __align__(16) float sc[1024];
  __shared__ int order[1024];
  __shared__ uint8_t supp[1024];
  // the adjacency in LDS while it fits (hex maps: <= 6 neighbours per region)
  __shared__ uint16_t l_off[1025];
  __shared__ uint16_t l_nbr[kGateLdsNbr];
  const int s = blockIdx.x;
  for (int i = threadIdx.x; i < a.R; i += blockDim.x) sc[i] = a.scores[s * a.R + i];
  __syncthreads();
  if (a.nbr_off[a.R] <= kGateLdsNbr) gate_select_body(a, s, sc, order, supp, l_off, l_nbr);  // (workgroup-uniform)
  else gate_select_body(a, s, sc, order, supp);
}

// Scores and selection in ONE launch for small region counts (body: ic_gate.h).
template <int DCH>
__global__ __launch_bounds__(256) void k_gate_fused(ic_gate_args a) {
  __shared__ __align__(16) float sc[1024];
  __shared__ int order[1024];
  __shared__ uint8_t supp[1024];
  gate_fused_body<DCH>(a, blockIdx.x, sc, order, supp);
}

template <int DCH>
__global__ __launch_bounds__(256) void k_rmsnorm_fwd(const float* x, const float* w, float* y,
                                                     float* inv_s, int rows, int d, float eps) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int row = blockIdx.x * 4 + wave;
  if (row >= rows) return;
  RowFrag<DCH> u, wf;
  load_row(u, x + (size_t)row * d, d, lane);
  load_row(wf, w, d, lane);
  const float sdev = sqrtf(wave_sum(dot_frag(u, u)) / (float)d + eps);
#pragma unroll
  for (int j = 0; j < DCH; ++j)
    u.v[j] = make_float4(wf.v[j].x * (u.v[j].x / sdev), wf.v[j].y * (u.v[j].y / sdev),
                         wf.v[j].z * (u.v[j].z / sdev), wf.v[j].w * (u.v[j].w / sdev));
  store_row(u, y + (size_t)row * d, d, lane);
  if (inv_s && lane == 0) inv_s[row] = 1.0f / sdev;
}

template <int DCH>
__global__ __launch_bounds__(256) void k_rmsnorm_bwd(const float* x, const float* w,
                                                     const float* inv_s, const float* gy, float* gx,
                                                     float* gw_rows, int rows, int d) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int row = blockIdx.x * 4 + wave;
  if (row >= rows) return;
  RowFrag<DCH> u, wf, g, gw;
  load_row(u, x + (size_t)row * d, d, lane);
  load_row(wf, w, d, lane);
  load_row(g, gy + (size_t)row * d, d, lane);
  const float inv = inv_s[row];
#pragma unroll
  for (int j = 0; j < DCH; ++j) {
    gw.v[j] = make_float4(g.v[j].x * u.v[j].x * inv, g.v[j].y * u.v[j].y * inv,
                          g.v[j].z * u.v[j].z * inv, g.v[j].w * u.v[j].w * inv);
    g.v[j] = make_float4(g.v[j].x * wf.v[j].x, g.v[j].y * wf.v[j].y, g.v[j].z * wf.v[j].z,
                         g.v[j].w * wf.v[j].w);
  }
  store_row(gw, gw_rows + (size_t)row * d, d, lane);
  const float k3 = wave_sum(dot_frag(g, u)) * inv * inv * inv / (float)d;
#pragma unroll
  for (int j = 0; j < DCH; ++j)
    g.v[j] = make_float4(g.v[j].x * inv - u.v[j].x * k3, g.v[j].y * inv - u.v[j].y * k3,
                         g.v[j].z * inv - u.v[j].z * k3, g.v[j].w * inv - u.v[j].w * k3);
  store_row(g, gx + (size_t)row * d, d, lane);
}

template <int DCH>
__global__ __launch_bounds__(256) void k_goodness_fwd(const float* H, float* g, int rows, int d) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int row = blockIdx.x * 4 + wave;
  if (row >= rows) return;
  RowFrag<DCH> h;
  load_row(h, H + (size_t)row * d, d, lane);
  const float v = wave_sum(dot_frag(h, h)) / (float)d;
  if (lane == 0) g[row] = v;
}

template <int DCH>
__global__ __launch_bounds__(256) void k_goodness_bwd(const float* H, const float* gg, float* gH,
                                                      int rows, int d) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int row = blockIdx.x * 4 + wave;
  if (row >= rows) return;
  RowFrag<DCH> h;
  load_row(h, H + (size_t)row * d, d, lane);
  const float c = gg[row] * 2.0f / (float)d;
#pragma unroll
  for (int j = 0; j < DCH; ++j)
    h.v[j] = make_float4(c * h.v[j].x, c * h.v[j].y, c * h.v[j].z, c * h.v[j].w);
  store_row(h, gH + (size_t)row * d, d, lane);
}

// gx = gy * gate for hard KWTA (utils.py:59-60); soft mode handled inside the region phase only.
__global__ void k_mul(const float* a, const float* b, float* o, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) o[i] = a[i] * b[i];
}

}  // namespace

static int check_gate(const ic_gate_args* a, const char* who) {
  IC_REQUIRE(a != nullptr, "%s: NULL args", who);
  IC_REQUIRE(a->S > 0 && a->R > 0 && a->R <= 1024, "%s: S=%d R=%d (R must be in [1,1024])", who, a->S,
             a->R);
  IC_REQUIRE(a->scores, "%s: scores is NULL", who);
  return IC_OK;
}

extern "C" int ic_gate_scores(const ic_gate_args* a, ic_stream_t stream) {
  int rc = check_gate(a, "ic_gate_scores");
  if (rc) return rc;
  IC_REQUIRE(a->d >= 4 && a->d % 4 == 0 && a->d <= 1024, "ic_gate_scores: d=%d unsupported", a->d);
  IC_REQUIRE(a->H && a->gain_ema && a->homeo, "ic_gate_scores: NULL pointer");
  IC_DISPATCH_DCH(a->d, {
    IC_TIMED(IC_KID_gate_scores, (hipStream_t)stream, hipLaunchKernelGGL(k_gate_scores<DCH>, dim3(ic_ceil_div(a->S * a->R, 4)), dim3(256), 0,
                       (hipStream_t)stream, *a));
  });
  IC_LAUNCH_CHECK("ic_gate_scores");
  return IC_OK;
}

extern "C" int ic_gate_select(const ic_gate_args* a, ic_stream_t stream) {
  int rc = check_gate(a, "ic_gate_select");
  if (rc) return rc;
  IC_REQUIRE(a->mask && a->nbr_off && a->nbr, "ic_gate_select: NULL pointer");
  IC_REQUIRE(!a->force_io || (a->sensor_idx >= 0 && a->sensor_idx < a->R && a->motor_idx >= 0 &&
                              a->motor_idx < a->R),
             "ic_gate_select: io indices out of range");
  IC_TIMED(IC_KID_gate_select, (hipStream_t)stream, hipLaunchKernelGGL(k_gate_select, dim3(a->S), dim3(a->R > 512 ? 1024 : (a->R > 256 ? 512 : 256)), 0, (hipStream_t)stream, *a));
  IC_LAUNCH_CHECK("ic_gate_select");
  return IC_OK;
}

extern "C" int ic_gate_fwd(const ic_gate_args* a, ic_stream_t stream) {
  IC_REQUIRE(a != nullptr, "ic_gate_fwd: NULL args");
  if (a->R > 64) {  // big maps: scores over the whole chip, then the walk
    int rc = ic_gate_scores(a, stream);
    return rc ? rc : ic_gate_select(a, stream);
  }
  int rc = check_gate(a, "ic_gate_fwd");
  if (rc) return rc;
  IC_REQUIRE(a->d >= 4 && a->d % 4 == 0 && a->d <= 1024, "ic_gate_fwd: d=%d unsupported", a->d);
  IC_REQUIRE(a->H && a->gain_ema && a->homeo && a->mask && a->nbr_off && a->nbr, "ic_gate_fwd: NULL pointer");
  IC_REQUIRE(!a->force_io || (a->sensor_idx >= 0 && a->sensor_idx < a->R && a->motor_idx >= 0 &&
                              a->motor_idx < a->R),
             "ic_gate_fwd: io indices out of range");
  IC_DISPATCH_DCH(a->d, {
    IC_TIMED(IC_KID_gate_select, (hipStream_t)stream, hipLaunchKernelGGL(k_gate_fused<DCH>, dim3(a->S), dim3(256), 0, (hipStream_t)stream, *a));
  });
  IC_LAUNCH_CHECK("ic_gate_fwd");
  return IC_OK;
}

extern "C" int ic_kwta_bwd(const float* x, const float* gate, const float* gy, float* gx, int32_t rows,
                           int32_t d, int32_t k, int32_t soft, float temp, ic_stream_t stream) {
  (void)x; (void)k; (void)temp;
  IC_REQUIRE(gate && gy && gx, "ic_kwta_bwd: NULL pointer");
  IC_REQUIRE(!soft, "ic_kwta_bwd: soft mode is only differentiable inside ic_region_phase_bwd");
  const size_t n = (size_t)rows * d;
  if (n == 0) return IC_OK;
  IC_TIMED(IC_KID_mul, (hipStream_t)stream, hipLaunchKernelGGL(k_mul, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, gy,
                     gate, gx, n));
  IC_LAUNCH_CHECK("ic_kwta_bwd");
  return IC_OK;
}

#define IC_ROWS_CHECK(who)                                                                   \
  IC_REQUIRE(rows >= 0, who ": rows=%d", rows);                                              \
  IC_REQUIRE(d >= 4 && d % 4 == 0 && d <= 1024, who ": d=%d unsupported", d);                \
  if (rows == 0) return IC_OK;

extern "C" int ic_rmsnorm_fwd(const float* x, const float* w, float* y, float* inv_s, int32_t rows,
                              int32_t d, float eps, ic_stream_t stream) {
  IC_REQUIRE(x && w && y, "ic_rmsnorm_fwd: NULL pointer");
  IC_ROWS_CHECK("ic_rmsnorm_fwd");
  IC_DISPATCH_DCH(d, {
    IC_TIMED(IC_KID_rmsnorm_fwd, (hipStream_t)stream, hipLaunchKernelGGL(k_rmsnorm_fwd<DCH>, dim3(ic_ceil_div(rows, 4)), dim3(256), 0,
                       (hipStream_t)stream, x, w, y, inv_s, rows, d, eps));
  });
  IC_LAUNCH_CHECK("ic_rmsnorm_fwd");
  return IC_OK;
}

extern "C" int ic_rmsnorm_bwd(const float* x, const float* w, const float* inv_s, const float* gy,
                              float* gx, float* gw_rows, int32_t rows, int32_t d, ic_stream_t stream) {
  IC_REQUIRE(x && w && inv_s && gy && gx && gw_rows, "ic_rmsnorm_bwd: NULL pointer");
  IC_ROWS_CHECK("ic_rmsnorm_bwd");
  IC_DISPATCH_DCH(d, {
    IC_TIMED(IC_KID_rmsnorm_bwd, (hipStream_t)stream, hipLaunchKernelGGL(k_rmsnorm_bwd<DCH>, dim3(ic_ceil_div(rows, 4)), dim3(256), 0,
                       (hipStream_t)stream, x, w, inv_s, gy, gx, gw_rows, rows, d));
  });
  IC_LAUNCH_CHECK("ic_rmsnorm_bwd");
  return IC_OK;
}

extern "C" int ic_goodness_fwd(const float* H, float* g, int32_t rows, int32_t d, ic_stream_t stream) {
  IC_REQUIRE(H && g, "ic_goodness_fwd: NULL pointer");
  IC_ROWS_CHECK("ic_goodness_fwd");
  IC_DISPATCH_DCH(d, {
    IC_TIMED(IC_KID_goodness_fwd, (hipStream_t)stream, hipLaunchKernelGGL(k_goodness_fwd<DCH>, dim3(ic_ceil_div(rows, 4)), dim3(256), 0,
                       (hipStream_t)stream, H, g, rows, d));
  });
  IC_LAUNCH_CHECK("ic_goodness_fwd");
  return IC_OK;
}

extern "C" int ic_goodness_bwd(const float* H, const float* gg, float* gH, int32_t rows, int32_t d,
                               ic_stream_t stream) {
  IC_REQUIRE(H && gg && gH, "ic_goodness_bwd: NULL pointer");
  IC_ROWS_CHECK("ic_goodness_bwd");
  IC_DISPATCH_DCH(d, {
    IC_TIMED(IC_KID_goodness_bwd, (hipStream_t)stream, hipLaunchKernelGGL(k_goodness_bwd<DCH>, dim3(ic_ceil_div(rows, 4)), dim3(256), 0,
                       (hipStream_t)stream, H, gg, gH, rows, d));
  });
  IC_LAUNCH_CHECK("ic_goodness_bwd");
  return IC_OK;
}
