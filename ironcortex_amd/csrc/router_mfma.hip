// Shared-matrix edge transforms on the matrix cores (reference gate.py:142-159,274-281: the
// by_direction mode keeps 6 d x d matrices, one per hex direction, shared by every edge of that
// direction).  With a shared matrix the per-edge GEMVs of the per_edge mode become true GEMMs:
//
//   forward   M^T[d, N] = sum_k  W_k [d,d]  x  X_k [d,N]      N = S*R (stream, destination) columns,
//             X_k[:, n] = ebias_e * H[s, src_e] for the in-edge e of n's destination with widx[e] == k
//             whose sender was active (a zero column otherwise).  Output-stationary: one workgroup
//             owns a 128 x 32 tile of M^T and walks K = nW*d; no per-edge message buffer, no atomics.
//   wgrad     dW_k[d,d] += G_k [d,P] x X_k^T [P,d]            P = (tape slot, stream, edge of direction k)
//             pairs; one workgroup owns a 128 x 32 tile of dW_k and walks all pairs in a fixed
//             order: deterministic (replaces the float atomics of the GEMV kernel for shared matrices).
//
// fp32 in, fp32 accumulate: v_mfma_f32_32x32x2_f32 (64 MACs per lane and instruction).  A wave owns a
// 32 x 32 accumulator tile; A (weights / gradient columns) and B (gathered sender rows) are staged through
// LDS in 32-wide K chunks, rows padded to 33 floats so the MFMA operand reads are bank-conflict free.
#include "ic_common.h"
#include "ic_route.h"

namespace {

typedef float v16f __attribute__((ext_vector_type(16)));
constexpr int kTM = 128, kTN = 32, kTK = 32, kMaxShared = 8;

// MFMA operand layout (wave64, 32x32x2): lane l supplies A[l % 32][l / 32] and B[l / 32][l % 32];
// accumulator register v of lane l is C[8 * (v / 4) + 4 * (l / 32) + v % 4][l % 32].
__device__ __forceinline__ void mfma_chunk(const float (*As)[kTK + 1], const float (*Bs)[kTN + 1], int wave, int lane,
                                           v16f& acc) {
  const int ai = wave * 32 + (lane & 31), kh = lane >> 5, bj = lane & 31;
#pragma unroll
  for (int kk = 0; kk < kTK; kk += 2)
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(As[ai][kk + kh], Bs[kk + kh][bj], acc, 0, 0, 0);
}

// ---- columns sorted by activity pattern (ic_router_args.shared_ws: [0,256) counts, [256,512) cursors, [512,512+N) the
// patterns, [512+N,512+2N) the permutation).  Forward (mode 0): pattern of column (s, r) = bit k set iff r has an
// in-edge of direction k whose sender was active in stream s.  Backward (mode 1): 1 iff (s, r) itself was an active
// sender.  k_shared_pattern counts the columns per pattern (LDS histogram, one global integer atomic per non-empty bin
// and workgroup); k_shared_scatter gives every column a place inside its pattern's range: a rank among its workgroup's
// columns of that pattern (LDS atomic) on top of a range the workgroup reserves with one global atomic per non-empty
// bin.  The order INSIDE a pattern is whatever the atomics make it: it does not matter, every column's result is its
// own and is written to the column's own place.
// Edge tables of the GEMM's columns.  dst_off == nullptr: column r has exactly ONE in-edge, edge r (the per-edge
// products of precision routing: columns are (stream, edge) pairs, router.hip shared_prec_fwd); ebias == nullptr: 1.
__device__ __forceinline__ int col_edges_lo(const ic_router_args& a, int r) { return a.dst_off ? a.dst_off[r] : r; }
__device__ __forceinline__ int col_edges_hi(const ic_router_args& a, int r) { return a.dst_off ? a.dst_off[r + 1] : r + 1; }
__device__ __forceinline__ float col_edge_bias(const ic_router_args& a, int e) { return a.ebias ? a.ebias[e] : 1.f; }

// library-internal flag of the pseudo-routers built in router.hip: columns without an active in-edge are NOT written
// (per-edge message / key rows that nobody reads), instead of written as zeros (the router's output proper)
constexpr int kSkipIdleCols = 1 << 30;

constexpr int kPatBins = 256, kPermHdr = 512, kSortMinCols = 64;  // (below two tiles there is nothing to sort; the caller decides from where it pays)
__global__ __launch_bounds__(256) void k_shared_pattern(ic_router_args a, int mode, const int32_t* bwd_slot) {
  __shared__ int hist[kPatBins];
  const int N = a.S * a.R, n = blockIdx.x * 256 + threadIdx.x;
  hist[threadIdx.x] = 0;
  __syncthreads();
  if (n < N) {
    const int s = n / a.R, r = n % a.R;
    int pat = 0;
    if (mode == 0) {
      for (int e = col_edges_lo(a, r); e < col_edges_hi(a, r); ++e)
        if (prev_on(a, s, a.src[e])) pat |= 1 << a.widx[e];
      pat &= kPatBins - 1;
    } else {
      // (ic_router_bwd_args.in_slot: the stream's block of H / mask_prev, negative = the stream takes no part)
      if (bwd_slot) pat = bwd_slot[s] >= 0 && a.mask_prev[(size_t)bwd_slot[s] * a.R + r] != 0;
      else pat = prev_on(a, s, r) ? 1 : 0;
    }
    pat = ~pat & (kPatBins - 1);  // sort key: the complement, so that all-inactive columns come LAST (their tiles only write zeros)
    a.shared_ws[kPermHdr + n] = pat;
    atomicAdd(&hist[pat], 1);
  }
  __syncthreads();
  if (hist[threadIdx.x]) atomicAdd(&a.shared_ws[threadIdx.x], hist[threadIdx.x]);
}

__global__ __launch_bounds__(256) void k_shared_scatter(ic_router_args a) {
  __shared__ int scan[2][kPatBins], mine[kPatBins], start[kPatBins];
  const int N = a.S * a.R, t = threadIdx.x, n = blockIdx.x * 256 + t;
  const int cnt = a.shared_ws[t];
  scan[0][t] = cnt;
  mine[t] = 0;
  __syncthreads();
  int cur = 0;
  for (int off = 1; off < kPatBins; off <<= 1, cur ^= 1) {  // inclusive prefix over the 256 pattern counts
    scan[cur ^ 1][t] = scan[cur][t] + (t >= off ? scan[cur][t - off] : 0);
    __syncthreads();
  }
  const int base_t = scan[cur][t] - cnt;  // exclusive prefix: where key t's range starts
  int pat = 0, rank = 0;
  if (n < N) {
    pat = a.shared_ws[kPermHdr + n];
    rank = atomicAdd(&mine[pat], 1);
  }
  __syncthreads();
  if (mine[t]) start[t] = base_t + atomicAdd(&a.shared_ws[kPatBins + t], mine[t]);
  __syncthreads();
  if (n < N) {
    const int pos = start[pat] + rank;
    if (pos >= 0 && pos < N) a.shared_ws[kPermHdr + N + pos] = n;
  }
}

// NB 32-column blocks per workgroup: the A chunk (rows of W_k) staged once in LDS and its fragment read once from LDS
// serves NB MFMAs, one per column block.  Same sums per column whatever NB.
template <int NB>
__global__ __launch_bounds__(256) void k_router_shared_fwd_mfma(ic_router_args a, const int32_t* perm) {
  constexpr int TN = kTN * NB;
  __shared__ float As[kTM][kTK + 1];
  __shared__ float Bs[kTK][TN + 1];
  __shared__ int tsrc[kMaxShared][TN];
  __shared__ float tscale[kMaxShared][TN];
  __shared__ int any_k[kMaxShared];
  __shared__ int tcol[TN];
  __shared__ int cact[TN];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int d = a.d, N = a.S * a.R;
  const int row0 = blockIdx.x * kTM, col0 = blockIdx.y * TN;
  if (tid < kMaxShared) any_k[tid] = 0;
  if (tid < TN) {  // this tile's columns: in (stream, region) order, or through the sorted permutation
    int n = col0 + tid;
    if (perm && n < N) n = perm[n];
    tcol[tid] = (unsigned)n < (unsigned)N ? n : -1;
    cact[tid] = 0;
  }
  for (int idx = tid; idx < kMaxShared * TN; idx += 256) {
    tsrc[idx / TN][idx % TN] = -1;
    tscale[idx / TN][idx % TN] = 0.f;
  }
  __syncthreads();
  // per column and shared matrix: the sender row and bias of the (unique) active in-edge, or -1.  One thread per
  // (column, in-edge slot): the tables of all of a column's edges are read in ONE round of dependent loads
  for (int idx = tid; idx < kMaxShared * TN; idx += 256) {
    const int c = idx % TN, i = idx / TN, n = tcol[c];
    if (n >= 0) {
      const int s = n / a.R, r = n % a.R;
      for (int e = col_edges_lo(a, r) + i; e < col_edges_hi(a, r); e += kMaxShared) {
        const int k = a.widx[e], sr = a.src[e];
        if (k >= 0 && k < a.nW && prev_on(a, s, sr)) {
          tsrc[k][c] = sr;
          tscale[k][c] = col_edge_bias(a, e);
          any_k[k] = 1;  // (same value from every writer)
          cact[c] = 1;
        }
      }
    }
  }
  __syncthreads();
  v16f acc[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb)
#pragma unroll
    for (int v = 0; v < 16; ++v) acc[nb][v] = 0.f;
  for (int k = 0; k < a.nW; ++k) {
    if (!any_k[k]) continue;  // workgroup-uniform
    const float* W = a.W + (size_t)k * d * d;
    // A: rows row0.. of W_k, columns kk0..kk0+31; thread -> (row tid / 2, 16 columns)
    // B: the gathered, bias-scaled sender rows; thread -> (column tid / 8 of every block, 4 k's)
    // The next chunk's operands are requested into registers before the current chunk's MFMAs run.
    const int ar = tid >> 1, ah = (tid & 1) * 16, agrow = row0 + ar;
    const int bc = tid >> 3, bks = (tid & 7) * 4;
    float bsc[NB];
    const float* brow[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      const int c = nb * kTN + bc, bsrow = tsrc[k][c];
      bsc[nb] = tscale[k][c];
      brow[nb] = bsrow >= 0 ? h_row(a, tcol[c] / a.R, bsrow) : nullptr;
    }
    float4 av[4], bv[NB];
    auto fetch = [&](int kk0) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int kk = kk0 + ah + 4 * q;
        av[q] = (agrow < d && kk < d) ? *reinterpret_cast<const float4*>(W + (size_t)agrow * d + kk) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int nb = 0; nb < NB; ++nb)
        bv[nb] = (brow[nb] && kk0 + bks < d) ? *reinterpret_cast<const float4*>(brow[nb] + kk0 + bks) : make_float4(0.f, 0.f, 0.f, 0.f);
    };
    fetch(0);
    for (int kk0 = 0; kk0 < d; kk0 += kTK) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        As[ar][ah + 4 * q + 0] = av[q].x; As[ar][ah + 4 * q + 1] = av[q].y; As[ar][ah + 4 * q + 2] = av[q].z; As[ar][ah + 4 * q + 3] = av[q].w;
      }
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        const int c = nb * kTN + bc;
        Bs[bks + 0][c] = bv[nb].x * bsc[nb]; Bs[bks + 1][c] = bv[nb].y * bsc[nb]; Bs[bks + 2][c] = bv[nb].z * bsc[nb]; Bs[bks + 3][c] = bv[nb].w * bsc[nb];
      }
      __syncthreads();
      if (kk0 + kTK < d) fetch(kk0 + kTK);
      {
        const int ai = wave * 32 + (lane & 31), kh = lane >> 5, bj = lane & 31;
#pragma unroll
        for (int kk = 0; kk < kTK; kk += 2) {
          const float af = As[ai][kk + kh];
#pragma unroll
          for (int nb = 0; nb < NB; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(af, Bs[kk + kh][nb * kTN + bj], acc[nb], 0, 0, 0);
        }
      }
      __syncthreads();
    }
  }
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    const int n = tcol[nb * kTN + (lane & 31)];
    if (n >= 0 && (!(a.flags & kSkipIdleCols) || cact[nb * kTN + (lane & 31)])) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int i = row0 + wave * 32 + 8 * q + 4 * (lane >> 5);
        if (i < d) *reinterpret_cast<float4*>(a.M + (size_t)n * d + i) = make_float4(acc[nb][4 * q], acc[nb][4 * q + 1], acc[nb][4 * q + 2], acc[nb][4 * q + 3]);
      }
    }
  }
}

// dW_k += sum over (slot t, stream s, edge e with widx[e] == k, active) g[t,s,e,:] (x) H[t,s,src_e,:]
__global__ __launch_bounds__(256) void k_wgrad_shared_mfma(ic_wgrad_edge_args a) {
  __shared__ float As[kTM][kTK + 1];
  __shared__ float Bs[kTK][kTN + 1];
  __shared__ uint16_t elist[4096];
  __shared__ int ecnt;
  __shared__ uint8_t pact[kTK];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int d = a.d, k = blockIdx.z;
  const int row0 = blockIdx.x * kTM, col0 = blockIdx.y * kTN;
  if (wave == 0) {  // the edges of this matrix, in edge order (a fixed order of additions)
    int cnt = 0;
    for (int base = 0; base < a.E; base += 64) {
      const int e = base + lane;
      const bool on = e < a.E && a.widx[e] == k;
      const unsigned long long b = __ballot(on);
      if (on) elist[cnt + __popcll(b & ((1ull << lane) - 1ull))] = (uint16_t)e;
      cnt += __popcll(b);
    }
    if (lane == 0) ecnt = cnt;
  }
  __syncthreads();
  const int cnt = ecnt;
  const long long P = (long long)a.T * a.S * cnt;
  v16f acc;
#pragma unroll
  for (int v = 0; v < 16; ++v) acc[v] = 0.f;
  int seen = 0;
  for (long long p0 = 0; p0 < P; p0 += kTK) {
    if (tid < kTK) {
      const long long p = p0 + tid;
      uint8_t on = 0;
      if (p < P) {
        const int ts = (int)(p / cnt), e = elist[p % cnt];
        on = a.visited[ts / a.S] != 0 && a.tape_eact[(size_t)ts * a.E + e] != 0;
      }
      pact[tid] = on;
    }
    const int any = __syncthreads_or(tid < kTK ? (int)pact[tid] : 0);  // (pact written before the barrier inside)
    if (!any) continue;  // workgroup-uniform
    seen = 1;
    {  // A[i][p] = g[pair p][row0 + i]; thread -> (pair tid % 32, 16 rows)
      const int pp = tid & 31, i0 = (tid >> 5) * 16;
      const long long p = p0 + pp;
      const float* g = nullptr;
      if (pact[pp]) {
        const int ts = (int)(p / cnt), e = elist[p % cnt];
        g = a.tape_g + ((size_t)ts * a.E + e) * d;
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int i = row0 + i0 + 4 * q;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (g && i < d) v = *reinterpret_cast<const float4*>(g + i);
        As[i0 + 4 * q + 0][pp] = v.x; As[i0 + 4 * q + 1][pp] = v.y; As[i0 + 4 * q + 2][pp] = v.z; As[i0 + 4 * q + 3][pp] = v.w;
      }
    }
    {  // B[p][j] = H[pair p's sender][col0 + j]; thread -> (pair tid / 8, 4 columns)
      const int pp = tid >> 3, js = (tid & 7) * 4;
      const long long p = p0 + pp;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (pact[pp] && col0 + js < d) {
        const int ts = (int)(p / cnt), e = elist[p % cnt];
        v = *reinterpret_cast<const float4*>(a.tape_H + ((size_t)ts * a.R + a.src[e]) * d + col0 + js);
      }
      Bs[pp][js + 0] = v.x; Bs[pp][js + 1] = v.y; Bs[pp][js + 2] = v.z; Bs[pp][js + 3] = v.w;
    }
    __syncthreads();
    mfma_chunk(As, Bs, wave, lane, acc);
    __syncthreads();
  }
  if (!seen) return;
  const int j = col0 + (lane & 31);
  if (j < d) {
    float* out = a.gW + (size_t)k * d * d;
#pragma unroll
    for (int v = 0; v < 16; ++v) {
      const int i = row0 + wave * 32 + 8 * (v / 4) + 4 * (lane >> 5) + (v % 4);
      if (i < d) out[(size_t)i * d + j] += acc[v];
    }
  }
  if (blockIdx.x == 0 && blockIdx.y == 0 && tid == 0) a.touched[k] = 1;
}

// ---- the same GEMM over a COMPACTED pair list (ic_wgrad_edge_args.pair_ws)
// k_wgrad_pairs: one workgroup per matrix k writes, in (slot, stream, edge) order, the ids ts*E + e of the pairs
// that contribute (slot visited, edge active) to ws + 2*nW + T*S*(edges with widx < k); ws[k] = their number,
// ws[nW + k] = where the list starts.  Two passes over the activity bytes (count, scan, fill), 16 waves.
constexpr int kMaxTS = 4096;
__global__ __launch_bounds__(1024) void k_wgrad_pairs(ic_wgrad_edge_args a) {
  __shared__ uint16_t elist[4096];
  __shared__ int cnt_ts[kMaxTS];
  __shared__ int ecnt, ebelow;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int k = blockIdx.x, TS = a.T * a.S;
  if (wave == 0) {
    int cnt = 0, below = 0;
    for (int base = 0; base < a.E; base += 64) {
      const int e = base + lane;
      const int w = e < a.E ? a.widx[e] : -1;
      const unsigned long long b = __ballot(w == k);
      if (w == k) elist[cnt + __popcll(b & ((1ull << lane) - 1ull))] = (uint16_t)e;
      cnt += __popcll(b);
      below += __popcll(__ballot(w >= 0 && w < k));
    }
    if (lane == 0) { ecnt = cnt; ebelow = below; }
  }
  __syncthreads();
  const int cnt = ecnt;
  for (int ts = wave; ts < TS; ts += 16) {  // pass 1: contributing pairs per (slot, stream)
    int n = 0;
    if (a.visited[ts / a.S]) {
      for (int i0 = 0; i0 < cnt; i0 += 64) {
        const int i = i0 + lane;
        n += __popcll(__ballot(i < cnt && a.tape_eact[(size_t)ts * a.E + elist[i < cnt ? i : 0]] != 0));
      }
    }
    if (lane == 0) cnt_ts[ts] = n;
  }
  __syncthreads();
  if (wave == 0) {  // exclusive scan over ts (wave-wide shuffles, 64 at a time)
    int carry = 0;
    for (int t0 = 0; t0 < TS; t0 += 64) {
      const int t = t0 + lane;
      const int v = t < TS ? cnt_ts[t] : 0;
      int inc = v;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const int u = __shfl_up(inc, o, 64);
        if (lane >= o) inc += u;
      }
      if (t < TS) cnt_ts[t] = carry + inc - v;
      carry += __shfl(inc, 63, 64);
    }
    if (lane == 0) {
      a.pair_ws[k] = carry;
      a.pair_ws[a.nW + k] = 2 * a.nW + TS * ebelow;
    }
  }
  __syncthreads();
  int* list = a.pair_ws + 2 * a.nW + (size_t)TS * ebelow;
  for (int ts = wave; ts < TS; ts += 16) {  // pass 2: fill
    if (!a.visited[ts / a.S]) continue;
    int off = cnt_ts[ts];
    for (int i0 = 0; i0 < cnt; i0 += 64) {
      const int i = i0 + lane;
      const int e = elist[i < cnt ? i : 0];
      const bool on = i < cnt && a.tape_eact[(size_t)ts * a.E + e] != 0;
      const unsigned long long b = __ballot(on);
      if (on) list[off + __popcll(b & ((1ull << lane) - 1ull))] = ts * a.E + e;
      off += __popcll(b);
    }
  }
}

// ks > 1: split-K -- workgroup (k, sp) walks the sp-th share of the list and stores its tile of partial sums
// to part[k][sp] (plain stores, zeros for an empty share); k_wgrad_reduce adds the shares in order.
__global__ __launch_bounds__(256) void k_wgrad_shared_mfma_list(ic_wgrad_edge_args a, int ks, float* part) {
  __shared__ float As[kTM][kTK + 1];
  __shared__ float Bs[kTK][kTN + 1];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int d = a.d, k = blockIdx.z / ks, sp = blockIdx.z % ks;
  const int row0 = blockIdx.x * kTM, col0 = blockIdx.y * kTN;
  const int Pall = a.pair_ws[k];
  if (Pall == 0) return;
  const int* list = a.pair_ws + a.pair_ws[a.nW + k];
  const int cps = ((Pall + kTK - 1) / kTK + ks - 1) / ks;  // chunks per share
  const int pb = sp * cps * kTK, P = min(Pall, pb + cps * kTK);
  v16f acc;
#pragma unroll
  for (int v = 0; v < 16; ++v) acc[v] = 0.f;
  // A[i][p] = g[pair p][row0 + i]; thread -> (pair tid % 32, 16 rows)
  // B[p][j] = H[pair p's sender][col0 + j]; thread -> (pair tid / 8, 4 columns); next chunk prefetched into registers
  const int app = tid & 31, ai0 = (tid >> 5) * 16, bpp = tid >> 3, bjs = (tid & 7) * 4;
  float4 av[4], bv;
  auto fetch = [&](int p0) {
    const float* g = p0 + app < P ? a.tape_g + (size_t)list[p0 + app] * d : nullptr;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int i = row0 + ai0 + 4 * q;
      av[q] = (g && i < d) ? *reinterpret_cast<const float4*>(g + i) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    bv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (p0 + bpp < P && col0 + bjs < d) {
      const int id = list[p0 + bpp], ts = id / a.E, e = id - ts * a.E;
      bv = *reinterpret_cast<const float4*>(a.tape_H + ((size_t)ts * a.R + a.src[e]) * d + col0 + bjs);
    }
  };
  if (pb < P) fetch(pb);
  for (int p0 = pb; p0 < P; p0 += kTK) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      As[ai0 + 4 * q + 0][app] = av[q].x; As[ai0 + 4 * q + 1][app] = av[q].y; As[ai0 + 4 * q + 2][app] = av[q].z; As[ai0 + 4 * q + 3][app] = av[q].w;
    }
    Bs[bpp][bjs + 0] = bv.x; Bs[bpp][bjs + 1] = bv.y; Bs[bpp][bjs + 2] = bv.z; Bs[bpp][bjs + 3] = bv.w;
    __syncthreads();
    if (p0 + kTK < P) fetch(p0 + kTK);
    mfma_chunk(As, Bs, wave, lane, acc);
    __syncthreads();
  }
  const int j = col0 + (lane & 31);
  if (j < d) {
    float* out = ks > 1 ? part + ((size_t)k * ks + sp) * d * d : a.gW + (size_t)k * d * d;
#pragma unroll
    for (int v = 0; v < 16; ++v) {
      const int i = row0 + wave * 32 + 8 * (v / 4) + 4 * (lane >> 5) + (v % 4);
      if (i < d) {
        if (ks > 1) out[(size_t)i * d + j] = acc[v];
        else out[(size_t)i * d + j] += acc[v];
      }
    }
  }
  if (ks == 1 && blockIdx.x == 0 && blockIdx.y == 0 && tid == 0) a.touched[k] = 1;
}

// gW[k] += part[k][0] + part[k][1] + ... (in that order); float4 per thread
__global__ __launch_bounds__(256) void k_wgrad_reduce(ic_wgrad_edge_args a, int ks, const float* part) {
  const int k = blockIdx.y;
  if (a.pair_ws[k] == 0) return;
  const size_t dd = (size_t)a.d * a.d, i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i >= dd) return;
  const float* p = part + (size_t)k * ks * dd + i;
  float4 t = *reinterpret_cast<const float4*>(p);
  for (int sp = 1; sp < ks; ++sp) {
    const float4 u = *reinterpret_cast<const float4*>(p + (size_t)sp * dd);
    t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w;
  }
  float4* out = reinterpret_cast<float4*>(a.gW + (size_t)k * dd + i);
  float4 o = *out;
  o.x += t.x; o.y += t.y; o.z += t.z; o.w += t.w;
  *out = o;
  if (blockIdx.x == 0 && threadIdx.x == 0) a.touched[k] = 1;
}

// ---- input gradient of the shared-matrix forward (no precision routing):
//   gH^T[d, N] = sum_k  W_k^T [d,d]  x  G_k [d,N]        N = S*R (stream, SENDER) columns,
//   G_k[:, n] = gM[s, dst_e] for the out-edge e of n's sender with widx[e] == k (a hex region has at most one
//   neighbour per direction), a zero column when the sender was inactive.  Per k the tile of W_k^T g is finished
//   before the edge bias is applied, so d/d ebias_e = (W_k^T g) . h_sender comes out of the same accumulators
//   (this workgroup's 128 rows: partial blockIdx.x of g_ebias_part).  Output-stationary like the forward: every
//   (stream, sender) row of gH is written once, zeros for inactive senders; no slabs, no gather launch.
__global__ __launch_bounds__(256) void k_router_shared_dgrad_mfma(ic_router_bwd_args b, const int32_t* perm) {
  const ic_router_args& a = b.f;
  __shared__ float As[kTM][kTK + 1];
  __shared__ float Bs[kTK][kTN + 1];
  __shared__ int tdst[kMaxShared][kTN], tedge[kMaxShared][kTN];
  __shared__ float tscale[kMaxShared][kTN];
  __shared__ int any_k[kMaxShared];
  __shared__ float pb[4][kTN];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int d = a.d, N = a.S * a.R;
  const int row0 = blockIdx.x * kTM, col0 = blockIdx.y * kTN;
  __shared__ int tcol[kTN];
  // stream addressing of the backward (router.hip): stream s lives at index s * sk of the per-stream arrays except
  // f.H / f.mask_prev / gH, which sit at block in_slot[s] when in_slot is set (negative: the stream takes no part)
  const int sk = b.sk > 1 ? b.sk : 1;
  auto sender_on = [&](int s, int r) {
    if (b.in_slot) return b.in_slot[s] >= 0 && a.mask_prev[(size_t)b.in_slot[s] * a.R + r] != 0;
    return prev_on(a, s, r);
  };
  if (tid < kMaxShared) any_k[tid] = 0;
  if (tid < kTN) {  // this tile's columns: in (stream, region) order, or active senders first (sorted permutation)
    int n = col0 + tid;
    if (perm && n < N) n = perm[n];
    const bool ok = (unsigned)n < (unsigned)N && (!b.in_slot || b.in_slot[n / a.R] >= 0);
    tcol[tid] = ok ? n : -1;
  }
  __syncthreads();
  {  // per column and matrix: the out-edge of the column's sender, its destination and bias, or -1.  One thread per
     // (column, out-edge slot), so that the tables of a sender's edges are read in one round of dependent loads
    const int c = tid & 31, i = tid >> 5, n = tcol[c];
    tdst[i][c] = -1;
    tedge[i][c] = -1;
    tscale[i][c] = 0.f;
    __syncthreads();
    if (n >= 0) {
      const int s = n / a.R, r = n % a.R;
      if (sender_on(s, r)) {
        for (int q = b.src_off[r] + i; q < b.src_off[r + 1]; q += kMaxShared) {
          const int e = b.src_edges[q], k = a.widx[e];
          if (k < 0 || k >= a.nW) continue;
          int lo = 0, hi = a.R;  // destination of e: the r' with dst_off[r'] <= e < dst_off[r'+1]
          while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (a.dst_off[mid] <= e) lo = mid; else hi = mid;
          }
          tdst[k][c] = lo;
          tedge[k][c] = e;
          tscale[k][c] = a.ebias[e];
          any_k[k] = 1;
        }
      }
    }
  }
  __syncthreads();
  const int j = lane & 31, n = tcol[j];
  const int s_col = n >= 0 ? n / a.R : 0, r_col = n >= 0 ? n % a.R : 0;
  // this lane's 16 rows of its column's sender state (for d/d ebias)
  float hv[16];
#pragma unroll
  for (int v = 0; v < 16; ++v) hv[v] = 0.f;
  if (n >= 0) {
    const float* hrow = b.in_slot ? a.H + ((size_t)b.in_slot[s_col] * a.R + r_col) * d : h_row(a, s_col, r_col);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int i = row0 + wave * 32 + 8 * q + 4 * (lane >> 5);
      if (i < d) {
        const float4 t = *reinterpret_cast<const float4*>(hrow + i);
        hv[4 * q] = t.x; hv[4 * q + 1] = t.y; hv[4 * q + 2] = t.z; hv[4 * q + 3] = t.w;
      }
    }
  }
  v16f tot;
#pragma unroll
  for (int v = 0; v < 16; ++v) tot[v] = 0.f;
  const int SE = a.S * sk * a.E;
  for (int k = 0; k < a.nW; ++k) {
    if (!any_k[k]) continue;  // workgroup-uniform
    const float* W = a.W + (size_t)k * d * d;
    v16f acc;
#pragma unroll
    for (int v = 0; v < 16; ++v) acc[v] = 0.f;
    // A[i][kk] = W_k[kk0 + kk][row0 + i]; thread -> (kk = tid / 8, 16 i's), rows of W read coalesced
    // B[kk][c] = gM[s, dst][kk0 + kk]; thread -> (column tid / 8, 4 k's); next chunk prefetched into registers
    const int akk = tid >> 3, ai0 = (tid & 7) * 16;
    const int bc = tid >> 3, bks = (tid & 7) * 4, bdst = tdst[k][bc];
    const float* brow = bdst >= 0 ? b.gM + ((size_t)(tcol[bc] / a.R) * sk * a.R + bdst) * d : nullptr;
    float4 av[4], bv;
    auto fetch = [&](int kk0) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int i = row0 + ai0 + 4 * q;
        av[q] = (kk0 + akk < d && i < d) ? *reinterpret_cast<const float4*>(W + (size_t)(kk0 + akk) * d + i) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
      bv = (brow && kk0 + bks < d) ? *reinterpret_cast<const float4*>(brow + kk0 + bks) : make_float4(0.f, 0.f, 0.f, 0.f);
    };
    fetch(0);
    for (int kk0 = 0; kk0 < d; kk0 += kTK) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        As[ai0 + 4 * q + 0][akk] = av[q].x; As[ai0 + 4 * q + 1][akk] = av[q].y; As[ai0 + 4 * q + 2][akk] = av[q].z; As[ai0 + 4 * q + 3][akk] = av[q].w;
      }
      Bs[bks + 0][bc] = bv.x; Bs[bks + 1][bc] = bv.y; Bs[bks + 2][bc] = bv.z; Bs[bks + 3][bc] = bv.w;
      __syncthreads();
      if (kk0 + kTK < d) fetch(kk0 + kTK);
      mfma_chunk(As, Bs, wave, lane, acc);
      __syncthreads();
    }
    // d/d ebias of this column's edge: (W_k^T g) . h over this workgroup's rows, in a fixed order
    float part = 0.f;
#pragma unroll
    for (int v = 0; v < 16; ++v) part = fmaf(acc[v], hv[v], part);
    part += __shfl_xor(part, 32, 64);
    if (lane < 32) pb[wave][j] = part;
    const float sc = tscale[k][j];
#pragma unroll
    for (int v = 0; v < 16; ++v) tot[v] = fmaf(sc, acc[v], tot[v]);
    __syncthreads();
    if (tid < kTN) {
      const int e = tedge[k][tid], nn = tcol[tid];
      if (e >= 0 && nn >= 0)
        b.g_ebias_part[(size_t)blockIdx.x * SE + (size_t)(nn / a.R) * sk * a.E + e] = (pb[0][tid] + pb[1][tid]) + (pb[2][tid] + pb[3][tid]);
    }
    __syncthreads();
  }
  if (n >= 0) {
    float* out = b.in_slot ? b.gH + ((size_t)b.in_slot[s_col] * a.R + r_col) * d
                           : b.gH + (int64_t)s_col * (b.gH_sstride ? b.gH_sstride : (int64_t)a.R * d) + (size_t)r_col * d;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int i = row0 + wave * 32 + 8 * q + 4 * (lane >> 5);
      if (i < d) *reinterpret_cast<float4*>(out + i) = make_float4(tot[4 * q], tot[4 * q + 1], tot[4 * q + 2], tot[4 * q + 3]);
    }
  }
}

}  // namespace

// The sorted permutation of the columns (mode 0: forward patterns, 1: active senders first) in a->shared_ws, when the
// caller brought one and the map is large enough for the two small launches to pay; *perm stays null otherwise.
static int shared_sort(const ic_router_args* a, int mode, const int32_t* bwd_slot, hipStream_t st, const int32_t** perm) {
  const int N = a->S * a->R;
  if (!a->shared_ws || N < kSortMinCols || a->nW > 8 || a->shared_ws_len < (long long)kPermHdr + 2ll * N) return IC_OK;
  if (hipMemsetAsync(a->shared_ws, 0, kPermHdr * sizeof(int32_t), st) != hipSuccess) {
    ic_set_error("ic_router: hipMemsetAsync of the column-sort counters failed");
    return IC_ELAUNCH;
  }
  IC_TIMED(IC_KID_router_mfma, st, hipLaunchKernelGGL(k_shared_pattern, dim3(ic_ceil_div(N, 256)), dim3(256), 0, st, *a, mode, bwd_slot));
  IC_TIMED(IC_KID_router_mfma, st, hipLaunchKernelGGL(k_shared_scatter, dim3(ic_ceil_div(N, 256)), dim3(256), 0, st, *a));
  *perm = a->shared_ws + kPermHdr + N;
  return IC_OK;
}

// library-internal launchers (router.hip: ic_router_fwd / ic_router_wgrad)
int ic_router_shared_fwd_launch(const ic_router_args* a, hipStream_t st) {
  IC_BYTES(IC_KID_router_mfma, 4.0 * ((double)a->nW * a->d * a->d + 2.0 * a->S * a->R * a->d));
  const int32_t* perm = nullptr;
  if (const int rc = shared_sort(a, 0, nullptr, st, &perm)) return rc;
  static const int nb_env = [] { const char* e = getenv("IC_SHARED_NB"); return e && e[0] ? atoi(e) : 0; }();
  const int nb = nb_env == 1 || nb_env == 2 ? nb_env : 1;
  const dim3 grid(ic_ceil_div(a->d, kTM), ic_ceil_div(a->S * a->R, kTN * nb));
  if (nb == 2) IC_TIMED(IC_KID_router_mfma, st, hipLaunchKernelGGL(k_router_shared_fwd_mfma<2>, grid, dim3(256), 0, st, *a, perm));
  else IC_TIMED(IC_KID_router_mfma, st, hipLaunchKernelGGL(k_router_shared_fwd_mfma<1>, grid, dim3(256), 0, st, *a, perm));
  return IC_OK;
}

int ic_wgrad_shared_launch(const ic_wgrad_edge_args* a, hipStream_t st) {
  const dim3 grid(ic_ceil_div(a->d, kTM), ic_ceil_div(a->d, kTN), a->nW);
  const long long TS = (long long)a->T * a->S;
  if (a->pair_ws && TS <= kMaxTS && TS * a->E < (1ll << 31) && a->pair_ws_len >= 2ll * a->nW + TS * a->E) {
    // split-K when the lists can be long and the workspace has room for the partial tiles (float, 256-byte aligned)
    const long long list_len = (2ll * a->nW + TS * a->E + 63) / 64 * 64, dd = (long long)a->d * a->d;
    int ks = 1;
    if (TS * a->E / a->nW >= 1024 && a->d % 4 == 0)
      for (ks = 8; ks > 1 && a->pair_ws_len < list_len + (long long)a->nW * ks * dd; ks /= 2) {}
    float* part = reinterpret_cast<float*>(a->pair_ws + list_len);
    IC_TIMED(IC_KID_wgrad_mfma, st, hipLaunchKernelGGL(k_wgrad_pairs, dim3(a->nW), dim3(1024), 0, st, *a));
    IC_TIMED(IC_KID_wgrad_mfma, st, hipLaunchKernelGGL(k_wgrad_shared_mfma_list, dim3(grid.x, grid.y, a->nW * ks), dim3(256), 0, st, *a, ks, part));
    if (ks > 1)
      IC_TIMED(IC_KID_wgrad_mfma, st, hipLaunchKernelGGL(k_wgrad_reduce, dim3((unsigned)((dd / 4 + 255) / 256), a->nW), dim3(256), 0, st, *a, ks, part));
    return IC_OK;
  }
  IC_TIMED(IC_KID_wgrad_mfma, st, hipLaunchKernelGGL(k_wgrad_shared_mfma, grid, dim3(256), 0, st, *a));
  return IC_OK;
}

int ic_router_shared_dgrad_launch(const ic_router_bwd_args* b, int nslab, hipStream_t st) {
  const ic_router_args* a = &b->f;
  (void)nslab;
  const dim3 grid(ic_ceil_div(a->d, kTM), ic_ceil_div(a->S * a->R, kTN));
  IC_BYTES(IC_KID_router_mfma, 4.0 * ((double)a->nW * a->d * a->d + 3.0 * a->S * a->R * a->d));
  const int32_t* perm = nullptr;
  if (const int rc = shared_sort(a, 1, b->in_slot, st, &perm)) return rc;
  IC_TIMED(IC_KID_router_mfma, st, hipLaunchKernelGGL(k_router_shared_dgrad_mfma, grid, dim3(256), 0, st, *b, perm));
  return IC_OK;
}
