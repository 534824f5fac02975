/*
 * ironcortex_hip.h — C ABI of libironcortex_hip.so (MI355X / gfx950 only).
 *
 * Drop-in boundary for the IronCortex inner-step hot path (gate -> router ->
 * region cells -> KWTA).  The reference has no FFI of its own: the path sits
 * behind plain Python methods on nn.Modules.  Every entry
 * point below names the reference method(s) it replaces (file:line into the
 * reference tree) and is what a ctypes binding on the reference side would
 * call.
 *
 * Conventions
 *   - all tensors are dense, row-major, fp32 unless stated; pointers are
 *     DEVICE pointers; NULL is allowed only where a field says so.
 *   - S = independent streams (sequences), R = regions, d = width, E = edges.
 *   - the library never allocates, never synchronises the stream and never
 *     throws: a call returns 0 or a negative IC_E* code and ic_last_error()
 *     gives the text for the calling thread.
 *   - masks are uint8 (0/1), dt counters int32.
 */
#ifndef IRONCORTEX_HIP_H
#define IRONCORTEX_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* ic_stream_t; /* a hipStream_t (0 = default stream) */

enum {
  IC_OK = 0,
  IC_EINVAL = -1,  /* bad shape / NULL pointer / unsupported size */
  IC_ELAUNCH = -2, /* hipGetLastError() after a launch */
  IC_EDEVICE = -3  /* no gfx950 device / wrong arch */
};

/* region-phase flags */
enum {
  IC_F_AFD = 1 << 0,        /* enable_adaptive_filter_dynamics (region.py:192-212) */
  IC_F_RT = 1 << 1,         /* enable_radial_tangential_updates (region.py:218-235) */
  IC_F_PRED = 1 << 2,       /* use_predictive_trace (region.py:140-146,173-175) */
  IC_F_NOKWTA = 1 << 3,     /* disable_kwta (region.py:238) */
  IC_F_KWTA_SOFT = 1 << 4,  /* soft KWTA during warm-up (region.py:244-246) */
  IC_F_NOISE_VEC = 1 << 5,  /* afd_noise_mode == "vector" (region.py:63-68) */
  IC_F_NORM_IN = 1 << 6,    /* apply shared norm_in + sensor add (model.py:198-204) */
  IC_F_NEED_GRAD = 1 << 7,  /* write the saved-for-backward buffers */
  IC_F_CHAIN = 1 << 8       /* forward only: the S "streams" are S consecutive steps of ONE state stream
                               (the branches of an outer iteration, model.py:388-392, which the
                               reference runs one after another on the same cells): the input state
                               arrays and `sensor` hold one stream ([1,R,..]), stream s > 0 starts
                               from the state stream s-1 left, outputs are [S,R,..].  S <= 64. */
};

/* router flags */
enum {
  IC_RF_PRECISION = 1 << 0, /* enable_precision_routed_messages (gate.py:292-318) */
  IC_RF_SHARED = 1 << 1,    /* widx selects one of nW <= 8 matrices SHARED by many edges (edge_transform_mode
                               "by_direction", gate.py:142-149,274-277), no destination has two incoming edges
                               and no sender two outgoing edges with the same widx: without IC_RF_PRECISION the forward then runs as ONE GEMM
                               [d, nW*d] x [nW*d, S*R] on the matrix cores (fp32 MFMA) instead of per-edge GEMVs --
                               from 8192 (stream, region) columns on: below, a tile's nW back-to-back K loops are a
                               latency chain (~50 us at d=256) that the per-edge kernels, reading W[widx[e]] in place
                               for the handful of active senders, undercut (9-45 us) */
  IC_RF_SHARED_GEMM = 1 << 3, /* with IC_RF_SHARED: the GEMM whatever the number of columns.  With IC_RF_PRECISION too
                               (by_direction + precision routing, gate.py:274-277 with :292-318) the messages and the
                               keys of all (stream, edge) pairs are two such GEMMs (from 8192 pairs on, streams laid
                               out plainly), the queries and the per-destination softmax combine as in per_edge mode */
  IC_RF_FACTORIZED = 1 << 2 /* edge_transform_mode "factorized" (gate.py:150-159,278-281): msg_e = ebias_e * U (w_e * V h_src)
                               with two matrices shared by ALL edges.  Without precision routing the messages of a
                               destination are summed, so M[dst] = U (sum_e ebias_e * w_e * (V h_src_e)): V once per
                               active sender and U once per destination with an active in-edge, both as shared-matrix
                               GEMMs over the (stream, region) columns (fp32 MFMA forward), an elementwise mix in
                               between.  Not combined with IC_RF_PRECISION.  See the fact_* fields. */
};

const char* ic_last_error(void);
/* Fills name (<= cap bytes) with the device's gcnArchName; returns IC_OK only on gfx950. */
int ic_device_check(char* name, int cap);
int ic_version(void);

/* ------------------------------------------------------------------ region phase
 * One inner step's region work for S streams x R regions in 4 launches:
 *   prologue   x = RMSNorm_r( RMSNorm_in(M[r] (+sensor)) + pred )     region.py:173-177, model.py:198-204
 *   rkv        r,k,v GEMVs, time rotation of v, decay fast-forward,
 *              WKV (or AFD/Kalman) state update, y                     region.py:178-216, 94-135
 *   out        h_pre = x + W_o y   (or x + radius * W_o dir)           region.py:218-237
 *   finish     active: h = KWTA(h_pre), pred <- 0, dt <- 0
 *              inactive: H = 0, pred <- a*pred + (1-a)*M, dt += 1      region.py:137-146,238-249, model.py:275-290
 * Regions are gathered by mask[S,R]; weights stay in place (no stacking copy).
 */
typedef struct ic_region_args {
  int32_t S, R, d;
  int32_t m_time;     /* rotated channel pairs of v (region.py:87) */
  int32_t kwta_k;     /* winners; <=0 -> zeros; clamped to d */
  int32_t flags;      /* IC_F_* */
  int32_t sensor_idx; /* regions that add the sensor vector (-1: none) */
  int32_t motor_idx;
  int32_t rows_fwd;   /* rows of W per workgroup in the forward GEMVs (multiple of 8; 0 -> 8) */
  float step_pos;     /* inner-step position in [0,1] (model.py:289) */
  float kwta_temp;    /* soft-KWTA temperature */
  double pred_alpha;  /* 0.95 (region.py:140); a double so that fp32(1-alpha) matches Python */
  /* parameters */
  const float* W;         /* [R,4,d,d]  r,k,v,o  (region.py:56-59) */
  const float* norm_w;    /* [R,d]      region.norm.weight */
  const float* raw_decay; /* [R,d]      region.py:62 */
  const float* norm_in_w; /* [d] or NULL (model.norm_in.weight) */
  const float* W_time;    /* [m_time]   region.py:88-91 */
  const float* pn_param;  /* [R] or [R,d]  AFD only */
  const float* on_param;  /* [R] or [R,d]  AFD only */
  const float* rpn_param; /* [R]  RT only */
  const float* ron_param; /* [R]  RT only */
  /* inputs */
  const float* M;      /* [S,R,d] router messages (or x_in when !IC_F_NORM_IN) */
  const float* sensor; /* [S,d] or NULL */
  const uint8_t* mask; /* [S,R] active regions */
  const float *num, *den, *var, *pred; /* [S,R,d]; var AFD only; pred if IC_F_PRED */
  const float *radius, *radius_var;    /* [S,R] RT only */
  const float* surprise;               /* [S,R] AFD only */
  const int32_t* dt;                   /* [S,R] */
  /* outputs */
  float *H, *num_o, *den_o, *var_o, *pred_o; /* [S,R,d] */
  float *radius_o, *radius_var_o, *surprise_o; /* [S,R] */
  float *last_dir, *last_norm; /* [S,R,d],[S,R] RT telemetry, may be NULL */
  int32_t* dt_o;
  /* saved for backward / inter-kernel workspace (always required) */
  float* x;     /* [S,R,d] normalised input */
  float* u1;    /* [S,R,d] x_in + pred */
  float* inv_s; /* [S,R,2] 1/rms of the two norms */
  float *r_s, *k_s, *v_s; /* [S,R,d] sigmoid(r), k, rotated v */
  float* y;     /* [S,R,d] */
  float* hpre;  /* [S,R,d] pre-KWTA activations */
  float* gate;  /* [S,R,d] KWTA multiplier (0/1 hard) */
  float* ynorm; /* [S,R]  RT only */
  float* sterm; /* [S,R,d] AFD surprise terms */
  const int32_t* dead; /* NULL, or one device word read at run time: non-zero turns the launch into a pass-through
                          (every region inactive, state / trace / dt copied unchanged, H = 0) -- an outer iteration
                          queued behind a reasoning loop that has already halted (device-steered loops, ic_loop_steer:
                          the host launches every iteration without knowing whether should_halt has fired) */
} ic_region_args;

int ic_region_phase_fwd(const ic_region_args* a, ic_stream_t stream);

/* Backward of the same phase (autograd of region.py:168-249 + model.py:198-204,275-290).
 * Weight gradients are NOT formed here: the per-step gradient vectors are written to a
 * tape slot and ic_region_wgrad() forms dW once per backward pass. */
typedef struct ic_region_bwd_args {
  ic_region_args f; /* same dims/params/inputs/saved as the forward call (outputs unused) */
  /* incoming gradients, NULL = zero */
  const float *gH, *gnum_o, *gden_o, *gvar_o, *gpred_o; /* [S,R,d] */
  const float *gradius_o, *gradius_var_o;               /* [S,R] */
  /* outgoing gradients */
  float *gM, *gnum, *gden, *gvar, *gpred; /* [S,R,d] */
  float *gradius, *gradius_var;           /* [S,R] */
  float* gsens;                           /* [S,2,d] sensor/motor contributions, or NULL */
  /* per-row parameter gradients (host reduces over S and accumulates) */
  float *g_norm_w_rows, *g_raw_decay_rows, *g_norm_in_rows; /* [S,R,d] */
  float *g_pn_rows, *g_on_rows; /* [S,R,d] AFD */
  float* g_rnoise_rows;         /* [S,R,2] RT: d/d rpn_param, d/d ron_param */
  /* tape slot for the deferred weight gradient */
  float* tape_g; /* [S,R,4,d] gradient vectors of the r,k,v,o pre-activations */
  float* tape_x; /* [S,R,d]   operand of r,k,v (the normalised input x) */
  float* tape_o; /* [S,R,d]   operand of o (y, or dir under RT) */
  /* workspace */
  float* ghpre;   /* [S,R,d] gradient w.r.t. the pre-KWTA activations (written only under soft KWTA;
                     with hard KWTA it is gH * gate, recomputed where it is needed) */
  float* slab;    /* [nslab,S,R,d] partial W_o^T g; nslab = ic_bwd_nslab(d) */
  float* slab2;   /* [nslab,S,R,d] partial W_{r,k,v}^T g (must not alias slab: the kernel that
                     writes slab2 is still reading slab) */
  float* slab_rt; /* [nslab,S,R] RT only: partial d/d radius' */
  int32_t sk;     /* spread-out streams: stream s lives at stream index s*sk of EVERY [S,R,..] / [nslab,S,R,..]
                     array of this call (f.sensor and gsens rows excepted: gsens is spread, f.sensor is
                     [S,d]), which are then sized for S*sk streams.  0/1 = contiguous.  Lets the same
                     step of several independent loops, recorded sk step slots apart, run as the
                     streams of one launch (the reference detaches the state between the samples of a
                     batch, model.py:482-487, so their backward passes are independent). */
} ic_region_bwd_args;

int ic_region_phase_bwd(const ic_region_bwd_args* a, ic_stream_t stream);
/* number of row slabs the backward kernels use for width d (also used by the router backward) */
int ic_bwd_nslab(int32_t d);

/* dW[r,j] (+)= sum over tape slots t, streams s with mask: g[t,s,r,j,:] (x) operand[t,s,r,:]
 * (operand = x for j<3, tape_o for j==3).  One launch per backward pass.
 * visited[t]==0 slots are skipped.  touched[r] is set to 1 for every region that received
 * a contribution.  replaces the autograd accumulation into r_lin/k_lin/v_lin/o_lin.weight.grad. */
typedef struct ic_wgrad_region_args {
  int32_t T, S, R, d;
  const float* tape_g;    /* [T,S,R,4,d] */
  const float* tape_x;    /* [T,S,R,d] */
  const float* tape_o;    /* [T,S,R,d] */
  const uint8_t* mask;    /* [T,S,R] */
  const uint8_t* visited; /* [T] */
  float* gW;              /* [R,4,d,d] accumulated in place */
  uint8_t* touched;       /* [R] */
} ic_wgrad_region_args;
int ic_region_wgrad(const ic_wgrad_region_args* a, ic_stream_t stream);

/* ------------------------------------------------------------------ router
 * Router.aggregate, per-edge transforms in place (gate.py:251-335).  Edges are stored
 * dst-major exactly as Router.__init__ builds them (gate.py:124-133): edge ids
 * dst_off[r]..dst_off[r+1]-1 arrive at r, src[e] is the sender.  widx[e] selects the
 * weight matrix (e for per_edge, direction 0..5 for by_direction, gate.py:274-277).
 * ebias[e] = 1 + fb_alpha * b_e (gate.py:284-290) is computed by the caller.
 */
typedef struct ic_router_args {
  int32_t S, R, d, E, flags;
  const int32_t* dst_off; /* [R+1] */
  const int32_t* src;     /* [E] */
  const int32_t* widx;    /* [E] */
  const float* W;         /* [nW,d,d] */
  const float* ebias;     /* [E] */
  const float* key_W;     /* [nW,d,d]  precision only */
  const float* query_W;   /* [R,d,d]   precision only */
  const float* raw_P;     /* [nW,d]    precision only */
  const float* H;         /* [S,R,d] */
  const uint8_t* mask_prev; /* [S,R] */
  float* M;               /* [S,R,d] out */
  /* saved / workspace (precision only; may be NULL otherwise) */
  float* msg;  /* [S,E,d] biased messages */
  float* keyv; /* [S,E,d] key vectors */
  float* q;    /* [S,R,d] queries */
  float* wgt;  /* [S,E]  routing weights (0 for inactive edges) */
  int64_t H_sstride;    /* elements between consecutive streams' H blocks; 0 = contiguous (R*d) */
  int64_t mask_sstride; /* bytes between consecutive streams' mask_prev rows; 0 = contiguous (R) */
  int32_t nW;           /* number of weight matrices (needed with IC_RF_SHARED; 0 = E) */
  int32_t pad_;
  const int32_t* in_slot; /* [S] or NULL (forward).  When set, stream s reads its input states / previous mask from
                           block in_slot[s] of H / mask_prev (R*d floats / R bytes per block; the strides above are
                           ignored): the streams of a multi-stream launch (independent reasoning loops run in
                           lockstep) each continue from their own winner of the previous outer iteration. */
  /* IC_RF_FACTORIZED only (W, key_W, query_W, raw_P unused): */
  const float* fact_U;      /* [d,d]  Router.U.weight */
  const float* fact_V;      /* [d,d]  Router.V.weight */
  const float* fact_w;      /* [E,d]  Router.w_edge, edge order */
  const int32_t* fact_iota; /* [R+1]  0, 1, ..., R: identity edge tables of the two row GEMMs */
  const int32_t* fact_zero; /* [R]    zeros (their widx) */
  const float* fact_ones;   /* [R]    ones (their edge bias) */
  float* fact_t;            /* [S,R,d] saved: V h of the active senders, zeros elsewhere */
  float* fact_z;            /* [S,R,d] saved: sum over the active in-edges of ebias_e * w_e * t[src_e] */
  uint8_t* fact_dact;       /* [S,R]   saved: 1 if the region has an active in-edge */
  /* IC_RF_SHARED, optional: workspace of shared_ws_len >= 512 + 2*S*R 32-bit words (512 + 2*S*E with IC_RF_PRECISION,
   * whose GEMM columns are the (stream, edge) pairs).  With it (worth it from ~8192 columns on) the forward GEMM walks
   * the (stream, destination) columns in an order SORTED by which directions have an active in-edge, so that a
   * 32-column tile skips the matrices none of its columns uses, and the data-gradient GEMM of ic_router_bwd (through
   * .f) walks the active senders first: the matrix cores only multiply active rows (the results are the same bit for
   * bit: every column's sums are its own).  Calls sharing one workspace must be ordered on one stream.
   * NULL (or too short for the call): columns in (stream, region) order. */
  int32_t* shared_ws;
  int64_t shared_ws_len;
} ic_router_args;
int ic_router_fwd(const ic_router_args* a, ic_stream_t stream);

typedef struct ic_router_bwd_args {
  ic_router_args f;
  const int32_t* src_off;   /* [R+1] CSR by sender: out-edges of each region */
  const int32_t* src_edges; /* [E] edge ids grouped by sender */
  const float* gM; /* [S,R,d] */
  float* gH;       /* [S,R,d] */
  float* g_ebias_part; /* [nslab,S,E] partial sums, host reduces over nslab and S */
  float* g_rawP_rows;  /* [S,E,d] precision only */
  /* tape slot for deferred weight gradients */
  float* tape_gmsg; /* [S,E,d] gradient w.r.t. the raw message W_e h_src (ebias folded in) */
  float* tape_gkey; /* [S,E,d] precision only */
  float* tape_gq;   /* [S,R,d] precision only */
  float* tape_H;    /* [S,R,d] copy of H */
  uint8_t* tape_eact; /* [S,E] 1 if edge was active */
  uint8_t* tape_qact; /* [S,R] precision only: 1 if the region had an active incoming edge */
  /* workspace */
  float* gHd;    /* [S,R,d] precision only: direct d/dH_dst */
  float* slab;   /* [nslab,S,E,d], nslab = ic_bwd_nslab(d) */
  float* slab_q; /* [nslab,S,R,d] precision only */
  int64_t gH_sstride; /* elements between consecutive streams' gH blocks; 0 = contiguous (R*d) */
  int32_t sk;         /* spread-out streams as in ic_region_bwd_args.sk: applies to every per-stream array
                         of this call except f.H, f.mask_prev and gH */
  const int32_t* in_slot; /* [S] or NULL.  When set, stream s reads its input states / previous mask
                         from block in_slot[s] of f.H / f.mask_prev (R*d floats / R bytes per block) and
                         writes gH to block in_slot[s] of gH (the strides above are ignored);
                         in_slot[s] < 0: the stream is skipped (tape_eact / tape_qact rows cleared). */
  /* IC_RF_FACTORIZED only.  Per-stream arrays follow sk like the others.  tape_gmsg is unused; tape_H (copy of H) and
   * tape_eact (edge activity, decides which w_e hold a gradient) are written as usual. */
  float* g_fw_rows;   /* [S,E,d] rows of d/d w_e (written for active edges; host reduces over S and steps) */
  float* fact_gz;     /* [S,R,d] workspace: U^T gM */
  float* fact_gt;     /* [S,R,d] workspace: gradient w.r.t. t */
  float* fact_slab;   /* [nslab,S,R,d] workspace of the two row GEMMs' input gradients */
  float* fact_gb;     /* [nslab,S,R] workspace (their unused edge-bias partials) */
  float* tape_gU;     /* [S,R,d] tape slot of dU = sum gM (x) z:  gradient rows */
  float* tape_zU;     /* [S,R,d]                                   operand rows (copy of fact_z) */
  uint8_t* tape_aU;   /* [S,R]                                     row took part */
  float* tape_gV;     /* [S,R,d] tape slot of dV = sum gt (x) H:  gradient rows (operand: tape_H) */
  uint8_t* tape_aV;   /* [S,R]                                     row took part */
} ic_router_bwd_args;
int ic_router_bwd(const ic_router_bwd_args* a, ic_stream_t stream);

typedef struct ic_wgrad_edge_args {
  int32_t T, S, R, d, E, nW, which; /* which: 0 = W_edge, 1 = key_W, 2 = query_W */
  const int32_t* dst_off;
  const int32_t* src;
  const int32_t* widx;
  const float* tape_g;      /* [T,S,E,d]  (which==2: [T,S,R,d]) */
  const float* tape_H;      /* [T,S,R,d] */
  const uint8_t* tape_eact; /* [T,S,E]    (which==2: [T,S,R] row-active) */
  const uint8_t* visited;   /* [T] */
  float* gW;                /* [nW,d,d] (which==2: [R,d,d]) accumulated in place */
  uint8_t* touched;         /* [nW] or [R] */
  int32_t* pair_ws;         /* optional (which < 2): workspace of pair_ws_len >= 2*nW + T*S*E 32-bit words for the
                               compacted list of contributing (slot, stream, edge) pairs per matrix; with it the MFMA
                               GEMM walks only those pairs.  ceil64(that) + ks*nW*d*d words (ks = 2, 4 or 8) more let
                               it split K over ks workgroups per tile (partial tiles summed in order: deterministic).
                               Shared matrices (nW != E) without it: the GEMM tests every pair itself.  One matrix per
                               edge (nW == E): passing it SELECTS the GEMM (worth it from ~64 (slot, stream) pairs on
                               the tape; widx[e] < 0 leaves edge e out), NULL the row-slice kernel. */
  int64_t pair_ws_len;
} ic_wgrad_edge_args;
/* (which < 2 and nW != E or pair_ws set: one deterministic MFMA GEMM per matrix, dW_k += G_k X_k^T, no float atomics;
 * nW <= 8192, E <= 4096) */
int ic_router_wgrad(const ic_wgrad_edge_args* a, ic_stream_t stream);

/* ------------------------------------------------------------------ gate
 * Gate.score_regions (gate.py:54-68): scores = w1*mean|H| + w2*gain_ema + w3*focus
 *   - w4*homeo + t5 + t6 with t5 = fp32(w5*value_bias), t6 = fp32(w6*u_mean) from the host. */
typedef struct ic_gate_args {
  int32_t S, R, d;
  int32_t sensor_idx, motor_idx;
  int32_t k_total;   /* k_active + burst (before the forced pair is subtracted) */
  int32_t force_io;  /* io_force_on */
  float w1, w2, w3, w4;
  float t5, t6;
  const float* H;          /* [S,R,d] */
  const float* gain_ema;   /* [R] */
  const float* homeo;      /* [R] */
  const uint8_t* focus_any; /* [S] 1 if any focus bit set (gate.py:42-52) */
  const int32_t* nbr_off;  /* [R+1] adjacency CSR */
  const int32_t* nbr;      /* [nnz] */
  float* scores;           /* [S,R] out */
  uint8_t* mask;           /* [S,R] out */
  int64_t H_sstride;       /* elements between the H blocks of consecutive streams; 0 = contiguous (R*d).
                              Lets the branches of one outer iteration, whose input states sit in
                              arbitrary step slots, go through ONE launch. */
  /* multi-stream launches (independent reasoning loops in lockstep): per-stream values instead of the scalars
   * above; each may be NULL.  Device-readable memory (pinned host memory is fine: a few words per launch). */
  const int32_t* in_slot;   /* [S] stream s reads H block in_slot[s] (R*d floats per block; H_sstride ignored) */
  const float* t5_s;        /* [S] replaces t5 (each loop has its own critic value) */
  const float* t6_s;        /* [S] replaces t6 (its own uncertainty) */
  const int32_t* k_total_s; /* [S] replaces k_total (its own burst); IC_GATE_OFF: the stream selects NOTHING (empty
                               mask, the forced pair included): an iteration behind a loop that has halted */
} ic_gate_args;
enum { IC_GATE_OFF = -(1 << 30) };
int ic_gate_scores(const ic_gate_args* a, ic_stream_t stream);
/* Gate.select_k + nms_topk (gate.py:70-87, utils.py:107-124).  Total order: score
 * descending, index ascending (stable); quirks of the greedy walk are reproduced. */
int ic_gate_select(const ic_gate_args* a, ic_stream_t stream);
/* ic_gate_scores followed by ic_gate_select; ONE launch when R <= 64 (the selecting workgroup
 * computes the scores of its stream itself). */
int ic_gate_fwd(const ic_gate_args* a, ic_stream_t stream);

/* Gate (new mask) and router (messages from the previous states and mask) of the same inner step as
 * ONE launch sequence: the two read the same inputs and are independent (model.py:254-262 calls
 * them back to back).  Same results as ic_gate_fwd followed by ic_router_fwd; g and a must agree
 * in S, R, d.  Falls back to the two calls for R > 64 or E == 0. */
int ic_front_fwd(const ic_gate_args* g, const ic_router_args* a, ic_stream_t stream);

/* One whole inner step (reference model.py:254-290, forward_inner_step: gate -> router -> region
 * bank): ic_front_fwd(g, ra) followed by ic_region_phase_fwd(a) on the same stream, same results
 * bit for bit.  ra->M must be a->M and g->mask must be a->mask.  With precision routing, d <= 256,
 * R <= 64 and IC_F_CHAIN the router's per-destination combine has no launch of its own: the
 * region's r/k/v kernel forms each message row from the edge messages in its prologue (and writes
 * M and the edge weights for the backward); otherwise exactly the two calls. */
int ic_loop_step_fwd(const ic_gate_args* g, const ic_router_args* ra, const ic_region_args* a,
                     ic_stream_t stream);

/* ------------------------------------------------------------------ primitives
 * KWTA (utils.py:38-60) on `rows` rows of width d: threshold = k-th largest |x|,
 * ties keep extra units.  gate receives the multiplier (for the backward). */
int ic_kwta_fwd(const float* x, float* y, float* gate, int32_t rows, int32_t d, int32_t k,
                int32_t soft, float temp, ic_stream_t stream);
/* gx = gy * gate (+ soft-mode threshold terms) */
int ic_kwta_bwd(const float* x, const float* gate, const float* gy, float* gx, int32_t rows,
                int32_t d, int32_t k, int32_t soft, float temp, ic_stream_t stream);
/* RMSNorm (utils.py:20-35): y = w * x / sqrt(mean(x^2)+eps) */
int ic_rmsnorm_fwd(const float* x, const float* w, float* y, float* inv_s, int32_t rows, int32_t d,
                   float eps, ic_stream_t stream);
int ic_rmsnorm_bwd(const float* x, const float* w, const float* inv_s, const float* gy, float* gx,
                   float* gw_rows, int32_t rows, int32_t d, ic_stream_t stream);

/* Per-region forward-forward goodness g[b,r] = mean_d H[b,r,:]^2 (training.py:127-128,
 * utils.py:159-162) and its backward gH = gg * 2 H / d. */
int ic_goodness_fwd(const float* H, float* g, int32_t rows, int32_t d, ic_stream_t stream);
int ic_goodness_bwd(const float* H, const float* gg, float* gH, int32_t rows, int32_t d,
                    ic_stream_t stream);

/* Hebbian edge nudges (training.py:147-156,240-243):
 * W[widx[e]] += sum_b coef[b,e] * outer(H[b,dst_e], H[b,src_e]),
 * coef = 1e-3*gain[dst] when both ends were active in sample b and gain>0, else 0. */
int ic_hebbian_rank1(float* W, const int32_t* dst_of_edge, const int32_t* src, const int32_t* widx,
                     const float* coef, const float* H, int32_t B, int32_t R, int32_t d, int32_t E,
                     ic_stream_t stream);

/* ------------------------------------------------------------------ loop heads (forward only)
 * The small heads that steer the reasoning loop but never carry a gradient in train_step or
 * generation (reference model.py:296-321,376-438).  One C call per inner step instead of ~40
 * ATen dispatches.  Callers that need d(logits)/d(params) use the PyTorch modules instead. */
enum { IC_ACT_NONE = 0, IC_ACT_GELU = 1, IC_ACT_SILU = 2 };

/* y = act(W . LN(cat(x0,x1,x2)) + b): nn.Linear on a vector (heads.py:39-46 MLP, energy.py:12-30
 * EnergyVerifierHead incl. its leading LayerNorm).  GELU is the exact erf form, LayerNorm the
 * biased-variance form of torch.nn.LayerNorm. */
typedef struct ic_gemv_desc {
  const float* W;    /* [rows, n0+n1+n2] row-major (nn.Linear.weight) */
  const float* b;    /* [rows] or NULL */
  const float* ln_g; /* [n0+n1+n2] LayerNorm weight/bias applied to the input first, or NULL */
  const float* ln_b;
  const float *x0, *x1, *x2; /* input segments (torch.cat order); unused ones have n = 0 */
  float* y;          /* [rows] */
  int32_t n0, n1, n2;
  int32_t rows, act; /* IC_ACT_* */
  float ln_eps;
  int32_t nbatch;    /* 0 or 1 = one vector; else that many independent input vectors ... */
  int64_t x0_bs, x1_bs, x2_bs, y_bs; /* ... whose segments / outputs are this many floats apart */
} ic_gemv_desc;
/* descs is a HOST array; the n GEMVs run in order on the stream (each may read the previous y). */
int ic_gemv_chain(const ic_gemv_desc* descs, int32_t n, ic_stream_t stream);
/* n <= 4 INDEPENDENT GEMVs as one launch (e.g. layer i of the verifier and layer i of the planner
 * chain, which do not read each other).  With score != NULL, descs[0] is the verifier's last
 * layer and gets the branch-score epilogue of ic_verify_score. */
int ic_gemv_multi(const ic_gemv_desc* descs, int32_t n, const float* stats, float* score, ic_stream_t stream);

/* Workspace write + token logits of one inner step (model.py:296-318, heads.py:26-29,96-119):
 *   slots_out = KWTA(slots_in + delta, d/4), delta[j] = H of the j-th active region (index order);
 *   logits = (MFS(H[motor]) + sum_{r active, r != motor} region_head_r(H[r])) / (1 + #such r);
 * plus the loop's steering statistics of those logits and states. */
typedef struct ic_heads_args {
  int32_t R, d, V, Kf, n_slots, motor_idx, kwta_k;
  const float* H;        /* [R,d] this step's region states */
  const float* H_par;    /* [R,d] the states the step started from */
  const uint8_t* mask;   /* [R] active regions of this step */
  const float* slots_in; /* [n_slots,d] */
  float* slots_out;      /* [n_slots,d] */
  const float* lm_gate_W; /* [Kf,d]  TokenHead_MFS.gate (heads.py:99) */
  const float* lm_gate_b; /* [Kf] */
  const float* lm_W;     /* [Kf,V,d] the Kf facets (heads.py:100) */
  const float* lm_b;     /* [Kf,V] */
  const float* rh_W;     /* [R,V,d] region_heads */
  const float* rh_b;     /* [R,V] */
  const float* surprise; /* [R] surprise_ema after this step, or NULL */
  float* z;              /* [Kf + Kf*V] scratch: gate pre-activations, then facet outputs */
  float* per_region;     /* [R,V] scratch */
  float* gsq;            /* [2R] scratch: per-row sums of squares of H and H_par */
  float* logits;         /* [V] */
  float* probs;          /* [V] softmax(logits) */
  float* stats;          /* [4] u = 1 - max p (utils.py:127-130); sum p log(p+1e-12) (utils.py:133-137);
                                goodness(H) - goodness(H_par) (model.py:430); mean surprise */
  float* aux_surprise;   /* where the verifier's aux vector wants mean surprise, or NULL */
  /* S > 1: the same for S steps at once (the branches of an outer iteration).  Per-stream blocks
   * are contiguous (H [S,R,d], mask [S,R], slots_out, logits, surprise, z, per_region, gsq, probs;
   * stats 4 floats apart, aux_surprise 4 floats apart) except the two inputs that come from the
   * steps' parents, which have their own stream strides (0 = contiguous). */
  int32_t S;                /* 0 or 1 = one step */
  int64_t Hpar_sstride;     /* elements between consecutive streams' H_par blocks */
  int64_t slots_in_sstride; /* elements between consecutive streams' slots_in blocks */
  float* ws_mean;           /* [S,d] or NULL: mean of the active rows of H (zeros when none) -- the planner's
                               input for the NEXT outer iteration should this step win (model.py:344-349) */
  const int32_t* par_slot;   /* [S] or NULL: stream s's parent states are block par_slot[s] of H_par (R*d floats per block) */
  const int32_t* slots_slot; /* [S] or NULL: its incoming workspace is block slots_slot[s] of slots_in (n_slots*d floats) */
} ic_heads_args;
int ic_heads_fwd(const ic_heads_args* a, ic_stream_t stream);
/* ic_heads_fwd for callers that cannot allocate device memory (code under HIP graph capture, tight
 * serving loops, harnesses that forbid hipMalloc).  NO COUNTERPART IN THE REFERENCE.  Same inputs,
 * outputs and results (bit for bit) as ic_heads_fwd, but a->z, a->per_region and a->gsq are IGNORED
 * (may be NULL): those inter-launch buffers live in a static device array of
 * IC_HEADS_STATIC_FLOATS floats (4 MiB per GPU the library runs on, held until the process ends).
 * Returns IC_EINVAL when S*(Kf + Kf*V) + S*R*V + S*2R (each term padded to 64) exceeds it.
 * Calls no hipMalloc / hipFree / event creation (its launches are never profiled by ic_prof_*).
 * All calls on one GPU share the array: calls that overlap on one GPU (different streams) race. */
enum { IC_HEADS_STATIC_FLOATS = 1 << 20 };
int ic_heads_fwd_static(const ic_heads_args* a, ic_stream_t stream);
/* The same as ic_heads_fwd, with up to two GEMVs riding in its two launches (either may be NULL):
 *   g_mean  takes the mean of each stream's active rows of H as its d-column input (the first planner
 *           layer, heads.py:57 on model.py:344-349; nbatch must equal S) and rides in the FIRST launch;
 *           bit-identical to ws_mean followed by ic_gemv_chain(g_mean, 1);
 *   g_next  any GEMV that only depends on the first launch, rides in the second (the combine's).
 */
int ic_heads_gemv_fwd(const ic_heads_args* a, const ic_gemv_desc* g_mean, const ic_gemv_desc* g_next,
                      ic_stream_t stream);

/* The energy verifier chain (energy.py:12-30; last layer -> scalar energy) followed by the branch
 * score below, in the same launch when the chain is small. */
int ic_verify_score(const ic_gemv_desc* descs, int32_t n, const float* stats, float* score,
                    ic_stream_t stream);
/* (batched descriptors: batch b reads stats + 4*b and writes score + b) */
/* score = 0.6*stats[2] + 0.2*(-energy) + 0.2*stats[1]   (model.py:431-438) */
int ic_branch_score(const float* stats, const float* energy, float* score, ic_stream_t stream);

/* Start of an outer iteration (model.py:371-386): among n_cand candidate states pick the one
 * with the highest score (first on ties; scores NULL -> candidate 0) WITHOUT a host round trip,
 * and form ws_prev = mean of its previously active rows (zeros when none) and its motor row,
 * the inputs of the planner / critic chain. */
typedef struct ic_steer_args {
  int32_t R, d, n_cand, motor_idx;
  const float* H_cand[8];       /* device pointers [R,d] */
  const uint8_t* mask_cand[8];  /* device pointers [R] */
  const float* scores;          /* [n_cand] device, or NULL */
  float* ws_prev;               /* [d] */
  float* motor;                 /* [d] */
} ic_steer_args;
int ic_steer_select(const ic_steer_args* a, ic_stream_t stream);

/* ------------------------------------------------------------------ device-side steering of the branching loop
 * The control flow of reference model.py:371-452 between two outer iterations, decided ON THE DEVICE so that the
 * host can queue every launch of every loop of a batch without waiting for a result (and capture them in a HIP
 * graph): rank the C <= 2 branches of the iteration that just ran by their scores (descending, first on ties:
 * the order the host code takes), hand the winner's states to the next iteration as parents, take ITS critic
 * value v_hat and the uncertainty u of the iteration's last logits (utils.py:127-130), derive the burst
 * (utils.py:140-144), the gate's two scalar terms (gate.py:58-67: t5 = fp32(w5 * v_hat), t6 = fp32(w6 * u),
 * products formed in double as the host does) and the halting test (utils.py:147-151).  The launches of the next
 * iteration read all of that from a CONTROL BLOCK through the per-stream indirections that exist for multi-stream
 * launches (ic_gate_args.in_slot / k_total_s / t5_s / t6_s, ic_router_args.in_slot, ic_heads_args.par_slot /
 * slots_slot, ic_region_args.dead).  The kernel also runs the critic's scalar output layer (heads.py:69-70) for
 * the C candidates -- same arithmetic as ic_gemv_chain -- so it REPLACES that launch instead of adding one.
 *
 * Control block = IC_CTL_WORDS 32-bit words:
 *   [IC_CTL_IN_SLOT + c]    arena slot whose H / mask branch c starts from       (int32)
 *   [IC_CTL_SLOTS_SLOT + c] arena slot whose workspace it starts from            (int32)
 *   [IC_CTL_K_TOTAL + c]    k_active + burst, or IC_GATE_OFF when the loop has halted (int32)
 *   [IC_CTL_T5 + c], [IC_CTL_T6 + c]  the gate's scalar terms                    (float)
 *   [IC_CTL_DEAD]           1: the loop halted before this iteration (its launches are pass-throughs)
 *   [IC_CTL_HALT]           1: this iteration is the loop's last (should_halt fires after it)
 *   [IC_CTL_TOP]            winning branch of the PREVIOUS iteration (-1: none), for the host's record
 *   [IC_CTL_U], [IC_CTL_VHAT]  u_mean and v_hat the iteration runs with (float), for the host's record
 */
enum {
  IC_CTL_WORDS = 32,
  IC_CTL_IN_SLOT = 0, IC_CTL_SLOTS_SLOT = 4, IC_CTL_K_TOTAL = 8, IC_CTL_T5 = 12, IC_CTL_T6 = 16,
  IC_CTL_DEAD = 20, IC_CTL_HALT = 21, IC_CTL_TOP = 22, IC_CTL_U = 23, IC_CTL_VHAT = 24
};
enum {
  IC_STEER_INIT = 0, /* write the first block of the first loop of a batch (no iteration has run) */
  IC_STEER_STEP = 1, /* after iteration k < K_inner - 1: ctl_next is the block of iteration k + 1 */
  IC_STEER_LAST = 2  /* after iteration K_inner - 1: ctl_next is the first block of the NEXT loop (or NULL) */
};
typedef struct ic_loop_steer_args {
  int32_t mode;          /* IC_STEER_* */
  int32_t C;             /* branches per outer iteration, 1 or 2 */
  int32_t R, k_active;   /* burst = max(1, R / 8); k_total = max(0, k_active + burst) */
  int32_t k, K_inner;    /* index of the iteration that just ran; iterations per loop */
  int32_t slot0;         /* arena slot of its branch 0 (branch c: slot0 + c) */
  int32_t zero_slot;     /* slot of the all-zero input state (H = 0, mask = 0) every loop starts from; the first
                            loop of a batch also takes its workspace from it (IC_STEER_INIT) */
  double w5, w6;         /* Gate.w5, Gate.w6 */
  const float* crit_W;   /* [crit_cols] the critic's output layer (rows = 1); NULL: no critic value (v_hat = 0) */
  const float* crit_b;   /* [1] or NULL */
  const float* crit_x;   /* [C, crit_cols] its input for every candidate */
  int32_t crit_cols;     /* <= 1024 */
  int32_t pad_;
  float* vhat;           /* [C] out: the candidates' critic values */
  const float* scores;   /* [C] branch scores of iteration k (model.py:431-438) */
  const float* stats;    /* [C,4] ic_heads_args.stats of its branches */
  const float* vhat0;    /* [1] critic value of the all-zero state (first iteration of every loop) */
  const int32_t* ctl;    /* control block iteration k ran with (unused by IC_STEER_INIT) */
  int32_t* ctl_next;     /* block to fill, see mode; NULL: nothing follows */
  int32_t* rec;          /* [4] record of the loop, kept up to date while it is live: slot of the winner so far,
                            last executed slot, executed steps, executed iterations */
} ic_loop_steer_args;
int ic_loop_steer(const ic_loop_steer_args* a, ic_stream_t stream);

/* ------------------------------------------------------------------ launch sequences as HIP graphs
 * No counterpart in the reference (its loop takes a host decision per outer iteration, model.py:371-452).  With
 * ic_loop_steer no launch of a batch's reasoning loops depends on a host value, so the whole sequence (~1300
 * launches per train_step of the 10M config) is captured once and replayed with one call per step.
 *   ic_graph_begin(stream)            start capturing the library calls made on `stream` (a stream of the caller's own,
 *                                     not the default stream); nothing runs until the graph is launched
 *   ic_graph_end(stream, &exec, &n)   stop capturing, instantiate; n (may be NULL) receives the number of nodes
 *   ic_graph_launch(exec, stream)     run the captured sequence on `stream` (any stream, stream-ordered as usual)
 *   ic_graph_destroy(exec)
 * The kernel arguments (device pointers, shapes) are frozen at capture: the caller replays a graph only while
 * every pointer it handed over is unchanged.  Profiling (ic_prof_enable) must be off while capturing. */
int ic_graph_begin(ic_stream_t stream);
int ic_graph_end(ic_stream_t stream, void** exec_out, int32_t* n_nodes);
int ic_graph_launch(void* exec, ic_stream_t stream);
int ic_graph_destroy(void* exec);

/* ------------------------------------------------------------------ measurement hooks
 * No counterpart in the reference (its cfg.profile timers read perf_counter without a device
 * sync, model.py:219-294).  When enabled every kernel launch of this library is bracketed by
 * HIP events on its own stream; ic_prof_read() synchronises on them and returns the number of
 * launches and their summed duration since the previous read.  Used by bench.py's roofline. */
int ic_prof_enable(int on);
/* Bracket only kernel id `kid` (-1 = all): two event records per launch cost more than some of
 * these kernels run, so a throughput measurement times just the kernel it reports on. */
int ic_prof_select(int kid);
/* Bracket 1 launch in n of those that qualify (n = 1: every one): a sample of a kernel's launches
 * gives the same average at a fraction of the event cost. */
int ic_prof_stride(int n);
int ic_prof_kernel_count(void);
const char* ic_prof_kernel_name(int kid);
int ic_prof_read(int kid, int64_t* launches, double* total_ms);
/* Launches and summed ALGORITHMIC bytes since the previous read, for the kernels whose traffic follows
 * from their shapes alone (GEMV chains: rows x cols weights + vectors; heads; gate), counted on the host
 * at launch time while profiling is enabled.  Mask-gathered kernels are priced by the caller from
 * device-side unit counts (bench.py). */
int ic_prof_bytes_read(int kid, int64_t* launches, double* bytes);

/* ---------------------------------------------------------------- heads that carry a gradient
 * Backward of y = act(W x + b) over B <= 16 rows: the planner / critic / RTD heads and the energy verifier's
 * linear layers that train_step differentiates through (reference heads.py:39-91 MLP / PlannerHead / CriticHead /
 * RTDHead, energy.py:12-30; their autograd in training.py:158-231).  The forward is ic_gemv_chain with
 * nbatch = B; the pre-activation is recomputed here.  gW / gb are WRITTEN (not accumulated); every output row
 * has one owner and the input gradient is summed in a fixed order: no float atomics. */
typedef struct ic_linear_bwd_args {
  int32_t B, n, rows, act; /* batch rows, input columns, output rows, IC_ACT_* */
  const float* W;          /* [rows,n] */
  const float* b;          /* [rows] or NULL */
  const float* x;          /* [B,n] */
  const float* gy;         /* [B,rows] */
  float* gx;               /* [B,n] or NULL */
  float* gW;               /* [rows,n] or NULL */
  float* gb;               /* [rows] or NULL */
  float* slab;             /* [B,rows] workspace (gz), needed when gx != NULL; rows <= 2048 then */
} ic_linear_bwd_args;
int ic_linear_bwd(const ic_linear_bwd_args* a, ic_stream_t stream);

/* ---------------------------------------------------------------- token-mixer attention
 * Replaces the score / softmax / statistics / value product of reference iron_rope.py:201-235
 * (IronRoPESelfAttention.forward: q k^T / sqrt(hd) + relative Fourier bias, key padding, softmax, att @ v and the
 * attention-energy statistic -mean log(att + eps)) and of attention/adaptive_filter_attention.py:60-88,100-117
 * (AdaptiveFilterAttention._exact_path without a mask: scores * exp(-alpha |i-j| dt), softmax, energy and entropy
 * statistics).  The caller folds 1/sqrt(hd), the rotary rotation and the rank-2m Fourier features of the bias
 * into the q / k rows (ironcortex_amd/mixer.py), so a score is a plain dot product of dqk columns.
 * fp32 MFMA, two passes over the keys, no [T,T] tensor in memory. */
typedef struct ic_attn_args {
  int B, H, T;              /* batch, heads, positions */
  int dqk, dv;              /* columns of a q / k row and of a v row; (dqk, dv) in {(32,32), (64,64), (128,128),
                               (96,32), (128,64), (192,128)}: head width alone or + 64 bias columns (zero-pad others) */
  const float* q;           /* [B,H,T,dqk] */
  const float* k;           /* [B,H,T,dqk] */
  const float* v;           /* [B,H,T,dv] */
  const int32_t* valid_len; /* [B] or NULL: keys >= valid_len[b] are padding (probability 0, left out of stats) */
  const float* alpha;       /* NULL, or one device scalar: score(i,j) = q_i.k_j * exp(-alpha * |i-j| * lag_dt) */
  float lag_dt;
  float eps_log;            /* the eps of log(p + eps) in the statistics (constants.py EPS_LOG) */
  float* out;               /* [B,H,T,dv]  softmax(scores) @ v */
  float* lse;               /* [B,H,T]     log-sum-exp of each query's scores (saved for the backward) */
  float* stats;             /* [B,H,T,2]   sum_k log(p+eps), sum_k p*log(p+eps) over the valid keys of each query */
} ic_attn_args;
int ic_attn_fwd(const ic_attn_args* a, ic_stream_t stream);
/* Backward of ic_attn_fwd's `out` (the statistics carry no gradient, as in the reference: computed under
 * no_grad / detached).  Every dq / dk / dv row is written by one workgroup, d alpha leaves as per-workgroup
 * partial sums: no float atomics, run-to-run identical. */
typedef struct ic_attn_bwd_args {
  ic_attn_args f;           /* the forward's arguments (out, lse as it wrote them; stats unused) */
  const float* dout;        /* [B,H,T,dv] */
  const float* drow;        /* [B,H,T] = sum over dv of dout * out */
  float* dq;                /* [B,H,T,dqk] */
  float* dk;                /* [B,H,T,dqk] */
  float* dv;                /* [B,H,T,dv] */
  float* dalpha_part;       /* [B*H*ceil(T/128)] partial sums of d loss / d alpha (needed iff f.alpha) */
} ic_attn_bwd_args;
int ic_attn_bwd(const ic_attn_bwd_args* b, ic_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* IRONCORTEX_HIP_H */
