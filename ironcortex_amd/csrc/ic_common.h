// Shared device helpers for the IronCortex gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/ironcortex_hip.h"

#define IC_EPS_DIV 1e-8f   // constants.py:1
#define IC_MAX_EXP 40.0f   // constants.py:3
#define IC_MAX_NOISE 1.0f  // constants.py:4
#define IC_RMS_EPS 1e-5f   // utils.py:28

void ic_set_error(const char* fmt, ...);

#define IC_REQUIRE(cond, ...)        \
  do {                               \
    if (!(cond)) {                   \
      ic_set_error(__VA_ARGS__);     \
      return IC_EINVAL;              \
    }                                \
  } while (0)

#define IC_LAUNCH_CHECK(name)                                              \
  do {                                                                     \
    hipError_t e__ = hipGetLastError();                                    \
    if (e__ != hipSuccess) {                                               \
      ic_set_error("%s: launch failed: %s", name, hipGetErrorString(e__)); \
      return IC_ELAUNCH;                                                   \
    }                                                                      \
  } while (0)

// ---- wave64 reductions (all lanes receive the result) ----
// DPP data movement inside the VALU (quad permutes, row mirrors, row broadcasts) instead of six
// ds_bpermute round trips through the LDS crossbar per reduction: the sum lands in lane 63 and is
// read back as a scalar.  Summation order: pairs, quads, half rows, rows of 16, rows 0+1 / 2+3, total.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float ic_dpp(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xF, false));
}
__device__ __forceinline__ float wave_sum(float v) {
  v += ic_dpp<0xB1, 0xF>(v);   // quad_perm [1,0,3,2]
  v += ic_dpp<0x4E, 0xF>(v);   // quad_perm [2,3,0,1]
  v += ic_dpp<0x141, 0xF>(v);  // row_half_mirror
  v += ic_dpp<0x140, 0xF>(v);  // row_mirror: every lane of a row of 16 holds the row's sum
  v += ic_dpp<0x142, 0xA>(v);  // row_bcast:15 into rows 1 and 3
  v += ic_dpp<0x143, 0xC>(v);  // row_bcast:31 into rows 2 and 3: lane 63 holds the total
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ float wave_max(float v) {
  v = fmaxf(v, ic_dpp<0xB1, 0xF>(v));
  v = fmaxf(v, ic_dpp<0x4E, 0xF>(v));
  v = fmaxf(v, ic_dpp<0x141, 0xF>(v));
  v = fmaxf(v, ic_dpp<0x140, 0xF>(v));
  // (rows that a row broadcast does not write read back 0 from ic_dpp: combine through a select, not fmaxf(v, 0))
  const float b15 = ic_dpp<0x142, 0xA>(v);
  const int row = (threadIdx.x & 63) >> 4;
  v = (row & 1) ? fmaxf(v, b15) : v;
  const float b31 = ic_dpp<0x143, 0xC>(v);
  v = (row >= 2) ? fmaxf(v, b31) : v;
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

// torch.nn.functional.softplus (beta=1, threshold=20)
__device__ __forceinline__ float ic_softplus(float x) { return x > 20.0f ? x : log1pf(expf(x)); }
// d softplus / dx as torch's backward computes it
__device__ __forceinline__ float ic_softplus_grad(float x) {
  if (x > 20.0f) return 1.0f;
  float z = expf(x);
  return z / (z + 1.0f);
}
__device__ __forceinline__ float ic_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }

// softplus(p).clamp_min(EPS).clamp_max(MAX_NOISE)  (region.py:100-126)
__device__ __forceinline__ float ic_noise(float p) {
  return fminf(fmaxf(ic_softplus(p), IC_EPS_DIV), IC_MAX_NOISE);
}
// derivative of ic_noise w.r.t. p (clamps pass gradient inside [min,max] inclusive)
__device__ __forceinline__ float ic_noise_grad(float p) {
  float s = ic_softplus(p);
  return (s >= IC_EPS_DIV && s <= IC_MAX_NOISE) ? ic_softplus_grad(p) : 0.0f;
}

// Column ownership inside a wave for a row of width d: lane l owns float4 chunks at
// columns 4*l + 256*j, j < DCH.  d must be a multiple of 4.
template <int DCH>
struct RowFrag {
  float4 v[DCH];
};

template <int DCH>
__device__ __forceinline__ void load_row(RowFrag<DCH>& f, const float* __restrict__ row, int d,
                                         int lane) {
#pragma unroll
  for (int j = 0; j < DCH; ++j) {
    int c = 4 * lane + 256 * j;
    f.v[j] = (c < d) ? *reinterpret_cast<const float4*>(row + c) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
}
template <int DCH>
__device__ __forceinline__ void store_row(const RowFrag<DCH>& f, float* __restrict__ row, int d,
                                          int lane) {
#pragma unroll
  for (int j = 0; j < DCH; ++j) {
    int c = 4 * lane + 256 * j;
    if (c < d) *reinterpret_cast<float4*>(row + c) = f.v[j];
  }
}
template <int DCH>
__device__ __forceinline__ float dot_frag(const RowFrag<DCH>& a, const RowFrag<DCH>& b) {
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < DCH; ++j) {
    s = fmaf(a.v[j].x, b.v[j].x, s);
    s = fmaf(a.v[j].y, b.v[j].y, s);
    s = fmaf(a.v[j].z, b.v[j].z, s);
    s = fmaf(a.v[j].w, b.v[j].w, s);
  }
  return s;
}
template <int DCH>
__device__ __forceinline__ void axpy_frag(RowFrag<DCH>& acc, float a, const RowFrag<DCH>& x) {
#pragma unroll
  for (int j = 0; j < DCH; ++j) {
    acc.v[j].x = fmaf(a, x.v[j].x, acc.v[j].x);
    acc.v[j].y = fmaf(a, x.v[j].y, acc.v[j].y);
    acc.v[j].z = fmaf(a, x.v[j].z, acc.v[j].z);
    acc.v[j].w = fmaf(a, x.v[j].w, acc.v[j].w);
  }
}
template <int DCH>
__device__ __forceinline__ void scale_frag(RowFrag<DCH>& f, float c) {
#pragma unroll
  for (int j = 0; j < DCH; ++j) {
    f.v[j].x *= c; f.v[j].y *= c; f.v[j].z *= c; f.v[j].w *= c;
  }
}
template <int DCH>
__device__ __forceinline__ void zero_frag(RowFrag<DCH>& f) {
#pragma unroll
  for (int j = 0; j < DCH; ++j) f.v[j] = make_float4(0.f, 0.f, 0.f, 0.f);
}

// g += slab[0] + slab[1] + ... (rows `stride` floats apart), added in slab order, with up to
// eight row loads in flight at a time (a serial load-add chain costs a memory round trip each).
template <int DCH>
__device__ __forceinline__ void sum_slabs(RowFrag<DCH>& g, const float* __restrict__ first, size_t stride,
                                          int nslab, int d, int lane) {
  for (int s0 = 0; s0 < nslab; s0 += 8) {
    RowFrag<DCH> t[8];
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (s0 + u < nslab) load_row(t[u], first + (size_t)(s0 + u) * stride, d, lane);
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (s0 + u < nslab) {
#pragma unroll
        for (int j = 0; j < DCH; ++j) {
          g.v[j].x += t[u].v[j].x; g.v[j].y += t[u].v[j].y; g.v[j].z += t[u].v[j].z; g.v[j].w += t[u].v[j].w;
        }
      }
    }
  }
}

// Dispatch on the number of 256-column chunks a row needs (d <= 1024).
#define IC_DISPATCH_DCH(d, ...)                    \
  do {                                             \
    if ((d) <= 256) {                              \
      constexpr int DCH = 1;                       \
      __VA_ARGS__;                                 \
    } else if ((d) <= 512) {                       \
      constexpr int DCH = 2;                       \
      __VA_ARGS__;                                 \
    } else {                                       \
      constexpr int DCH = 4;                       \
      __VA_ARGS__;                                 \
    }                                              \
  } while (0)

enum IcKid {
  IC_KID_region_pro,
  IC_KID_region_rkv,
  IC_KID_region_out,
  IC_KID_region_fin,
  IC_KID_kwta_rows,
  IC_KID_region_fin_bwd,
  IC_KID_region_out_bwd,
  IC_KID_region_mid_bwd,
  IC_KID_region_rkv_bwd,
  IC_KID_region_pro_bwd,
  IC_KID_wgrad_region,
  IC_KID_router_edge_fwd,
  IC_KID_router_combine,
  IC_KID_router_fwd,
  IC_KID_router_comb_bwd,
  IC_KID_router_mat_bwd,
  IC_KID_router_gather_bwd,
  IC_KID_wgrad_edge,
  IC_KID_hebbian,
  IC_KID_gate_scores,
  IC_KID_gate_select,
  IC_KID_mul,
  IC_KID_rmsnorm_fwd,
  IC_KID_rmsnorm_bwd,
  IC_KID_goodness_fwd,
  IC_KID_goodness_bwd,
  IC_KID_gemv,
  IC_KID_ws_write,
  IC_KID_token_rows,
  IC_KID_logits_combine,
  IC_KID_branch_score,
  IC_KID_steer_select,
  IC_KID_router_mfma,
  IC_KID_wgrad_mfma,
  IC_KID_region_state,
  IC_KID_attn_fwd,
  IC_KID_attn_bwd_kv,
  IC_KID_attn_bwd_q,
  IC_KID_linear_bwd,
  IC_KID_loop_steer,
  IC_KID_COUNT
};
#define IC_KID_NAMES {"region_pro", "region_rkv", "region_out", "region_fin", "kwta_rows", "region_fin_bwd", "region_out_bwd", "region_mid_bwd", "region_rkv_bwd", "region_pro_bwd", "wgrad_region", "router_edge_fwd", "router_combine", "router_fwd", "router_comb_bwd", "router_mat_bwd", "router_gather_bwd", "wgrad_edge", "hebbian", "gate_scores", "gate_select", "mul", "rmsnorm_fwd", "rmsnorm_bwd", "goodness_fwd", "goodness_bwd", "gemv", "ws_write", "token_rows", "logits_combine", "branch_score", "steer_select", "router_mfma", "wgrad_mfma", "region_state", "attn_fwd", "attn_bwd_kv", "attn_bwd_q", "linear_bwd", "loop_steer"}

// Optional live kernel timing with HIP events on the launch stream (ic_prof_* in the header).
extern bool g_ic_prof_on;
extern int g_ic_prof_only;
bool ic_prof_sampled();  // every launch, or 1 in ic_prof_stride() of the selected kernel's
void ic_prof_begin(int kid, hipStream_t st);
void ic_prof_end(int kid, hipStream_t st);
#define IC_TIMED(kid, st, ...)                                  \
  do {                                                          \
    const bool t__ = g_ic_prof_on && (g_ic_prof_only < 0 || g_ic_prof_only == (kid)) && ic_prof_sampled(); \
    if (t__) ic_prof_begin(kid, (hipStream_t)(st));             \
    __VA_ARGS__;                                                \
    if (t__) ic_prof_end(kid, (hipStream_t)(st));               \
  } while (0)

// shape-determined algorithmic bytes of a launch (see prof.hip); costs one predictable branch when off
void ic_prof_count(int kid, double bytes);
#define IC_BYTES(kid, ...)                                       \
  do {                                                           \
    if (g_ic_prof_on) ic_prof_count(kid, (double)(__VA_ARGS__)); \
  } while (0)

static inline int ic_ceil_div(int a, int b) { return (a + b - 1) / b; }

// library-internal (heads.hip): looks up, once per device, the address of the static scratch array of
// ic_heads_fwd_static on the current device
__attribute__((visibility("hidden"))) int ic_heads_static_setup(float** out);

// library-internal (router_mfma.hip): the shared-matrix (by_direction) forward / weight-gradient GEMMs on MFMA
__attribute__((visibility("hidden"))) int ic_router_shared_fwd_launch(const ic_router_args* a, hipStream_t st);
__attribute__((visibility("hidden"))) int ic_wgrad_shared_launch(const ic_wgrad_edge_args* a, hipStream_t st);
__attribute__((visibility("hidden"))) int ic_router_shared_dgrad_launch(const ic_router_bwd_args* b, int nslab, hipStream_t st);

// library-internal: gate + router edge launch of ic_front_fwd, with or without the combine launch
__attribute__((visibility("hidden"))) int ic_front_launch(const ic_gate_args* g, const ic_router_args* a,
                                                          ic_stream_t stream, bool combine);
