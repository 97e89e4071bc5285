/* gemnet_hip.h — C ABI of the MI355X-native (gfx950) GemNet hot path.
 *
 * The reference (TUM-DAML/gemnet_pytorch) is 100 % Python and has no FFI: the boundary its
 * callers see is the Python class gemnet.model.gemnet.GemNet (gemnet/model/gemnet.py:21).
 * This header is the seam UNDER that class: one exported symbol family per hot-path row of
 * SURVEY.md §8(a).  Each declaration cites the reference code it replaces (file:line under
 * /root/reference).  INTEGRATION.md shows the ctypes stub a maintainer of the reference
 * would add to bind these.
 *
 * Conventions
 *   - plain C ABI, no torch / C++ types; device pointers are raw `void*`/typed pointers,
 *     sizes are explicit `int`/`int64_t`, indices are int32 on the device.
 *   - caller-owned buffers: no entry point allocates, frees or synchronises.
 *   - every launch goes to the explicit `stream` (a hipStream_t passed as void*).
 *   - return value: 0 on success, otherwise a hipError_t code (gn_error_string()).
 *   - re-entrant, no global mutable state (since ABI 13 every arithmetic / layout switch is an ARGUMENT of the launch it
 *     applies to; the only statics are set-once std::atomic flags of idempotent kernel attributes); safe to call
 *     concurrently from the caller's thread and the autograd worker thread(s).
 *   - all floating tensors are fp32, row-major, contiguous unless a leading dimension is given.
 */
#ifndef GEMNET_HIP_H
#define GEMNET_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

int gn_abi_version(void);
const char* gn_error_string(int code);

/* ---- dense contractions (P11: base_layers.py:44-48 Dense.forward = nn.Linear + ScaledSiLU,
 *      :84-89 ResidualLayer; embedding_block.py:60-75 concat-Dense) on f32 MFMA ------------
 *   acc[m][n] = sum_k opA(A)[m][k] * opB(B)[k][n]
 *     opA(A)[m][k] = trans_a ? A[k*lda+m] : A[m*lda+k]
 *     opB(B)[k][n] = trans_b ? B[k*ldb+n] : B[n*ldb+k]       (trans_b=0: B is a torch Linear weight (N,K))
 *     if a_dact_pre: opA(A)[m][k] *= d/dz ssilu(a_dact_pre[same index as A])   (backward through the activation)
 *   z = acc + (gadd1 ? gadd1[gidx1[m]*ldg+n] : 0) + (gadd2 ? gadd2[gidx2[m]*ldg+n] : 0)
 *   if pre_out: pre_out[m*ldc+n] = z
 *   y = act ? ssilu(z) : z ;  ssilu(z) = z*sigmoid(z)/0.6   (base_layers.py:51-58)
 *   if mul: y *= mul[m*ldmul+n]
 *   y *= alpha
 *   if res: y = (y + res[(ridx ? ridx[m] : m)*ldres+n]) * beta
 *   if res2: y = (y + res2[m*ldres2+n]) * beta2
 *   C[m*ldc+n] = y
 */
typedef struct {
  const float* A; const float* B; float* C;
  int M, N, K;
  int lda, ldb, ldc;
  int trans_a, trans_b;
  const float* a_dact_pre;
  int act;
  float* pre_out;
  const float* mul; int ldmul;
  float alpha;
  const float* res; int ldres;
  float beta;
  const float* gadd1; const int32_t* gidx1;
  const float* gadd2; const int32_t* gidx2;
  int ldg;
  const int32_t* ridx;
  const float* res2; int ldres2;
  float beta2;
  /* split-K (weight-gradient GEMMs: tiny M x N, K = number of edges): `splitk` > 1 slices of K are
   * accumulated by separate workgroups into splitk_ws (splitk * M * N floats, caller-owned) and summed
   * in a fixed order by a second kernel; only `alpha` of the epilogue is applied. */
  float* splitk_ws; int splitk;
} gn_gemm_args;
int gn_gemm_f32(const gn_gemm_args* args, void* stream);
/* Same contraction with an explicit tile variant (cfg >= 0; tuning and tests).  cfg < 0 = automatic. */
int gn_gemm_f32_cfg(const gn_gemm_args* args, int cfg, void* stream);

/* ---- LDS-resident layer chains ----------------------------------------------------------------
 * One launch runs a whole stack of Dense / ResidualLayer (base_layers.py:44-89) — or its adjoint —
 * for a tile of 16..80 rows (chosen so that the launch is one round of <= 256 workgroups) whose activations
 * stay in LDS between layers; only the weights stream (from L2, straight into MFMA fragment registers) and
 * only pre-activations / the final result go back to memory.  A chain is a short program of ops over two
 * LDS "slots" 0..2 (tile rows x up to 128 columns each):
 *   GN_OP_LOAD   slot <- src[(rows ? rows[m] : m), 0:width]                       (global -> LDS)
 *   GN_OP_SCALE  dst_slot <- a_slot * alpha * (src ? phi(src[m, :]) : 1); optional copy to `out`;
 *                phi selected by `act`: 0 ssilu'(x) (adjoint of an activation), 1 x (Hadamard), 2 ssilu(x)
 *   GN_OP_GEMM   z = slot[a_slot] (rows x K) @ W^T (W is (N,K), k-contiguous) + gadd1[gidx1[m]] + gadd2[gidx2[m]]
 *                pre_out <- z;  a = act ? ssilu(z) : z;  y = a * phi_mul(mul);  y *= alpha;
 *                y = (y + res) * beta;  y = (y + res2) * beta2      (mul/res/res2: an LDS slot or a global (M,N);
 *                phi_mul by `mul_mode` for a GLOBAL mul: 1 identity (default 0 means identity too), 2 ssilu', 3 ssilu)
 *                slot[y_slot] <- y (may alias a_slot / res slots);  out <- y
 *                second output (y2_slot >= 0 or out2): y2 = (y2_src ? a : y) * alpha2 * phi2(Z2[m, :]) with phi2 by
 *                `mode2` as for SCALE (Z2 NULL: plain scale);  slot[y2_slot] <- y2;  out2 <- y2.   The adjoint of a
 *                residual layer needs G = dz1 W1 + c G_old AND dz2' = G c' ssilu'(z2') of the next layer: one op.
 *   GN_OP_STORE  out[m, 0:width] <- slot                                           (LDS -> global)
 * Constraints: M <= 2^24, N, K <= 128, K % 16 == 0, W contiguous and 16-byte aligned, every global matrix row-major with row length N (resp. `ld` for
 * LOAD/SCALE/STORE), W rows 16-byte aligned. */
enum { GN_OP_LOAD = 0, GN_OP_SCALE = 1, GN_OP_GEMM = 2, GN_OP_STORE = 3 };
#define GN_CHAIN_MAX_OPS 20
typedef struct {
  int kind;
  int slot;                 /* LOAD/STORE: slot; SCALE: dst slot; GEMM: y_slot (-1 none) */
  int a_slot;               /* SCALE: source slot; GEMM: A operand slot */
  int width;                /* LOAD/SCALE/STORE: columns */
  int ld;                   /* LOAD/SCALE/STORE: leading dimension of src/out */
  const float* src;         /* LOAD: source; SCALE: Z for ssilu'(Z) (NULL: plain scale) */
  const int32_t* rows;      /* LOAD: optional row gather */
  const float* W; int N; int K;
  int act;                  /* GEMM: bit 0 = ScaledSiLU on z; bit 1 (gn_chain_split_f32 only, needs bit 0 and pre_out) =
                               pre_out receives ssilu'(z) instead of z — the factor a first-order adjoint multiplies by */
  float alpha;
  const float* gadd1; const int32_t* gidx1;
  const float* gadd2; const int32_t* gidx2;
  float* pre_out;
  int mul_slot; const float* mul_g;
  int res_slot; const float* res_g; const int32_t* res_rows; float beta;
  int res2_slot; const float* res2_g; float beta2;
  float* out;
  int mul_mode;             /* GEMM: 0/1 mul as is, 2 ssilu'(mul_g), 3 ssilu(mul_g) (global mul only) */
  int y2_slot;              /* GEMM: slot of the second output (-1 none) */
  int y2_src;               /* 0: derived from the final y, 1: from a (after the activation, before mul) */
  int mode2;                /* phi2: 0 ssilu'(Z2), 1 Z2, 2 ssilu(Z2) */
  float alpha2;
  const float* Z2;          /* (M,N) global or NULL */
  float* out2;              /* (M,N) global or NULL */
  /* Source term of the second-order sweeps of force training (trainer.py:346: loss.backward() through dE/dR): one
   * extra summand  s = src_alpha * phis(Zs) * srcP * srcQ  added to
   *   src_stage 1: the value of a SCALE op (Zs = its `src`), or a GEMM's y right after the mul / alpha stages
   *                (y = a * phi_mul(mul) * alpha + s; Zs = the GLOBAL mul operand), before res / res2;
   *   src_stage 2: the second output of a GEMM or LOAD (y2 = ... * alpha2 * phi2(Z2) + s; Zs = Z2).
   * phis by `src_mode`: 0 -> 1, 1 -> ssilu''(Zs).  The adjoint of y = ssilu(z) under a tangent dz is
   * zbar = ybar ssilu'(z) + lambda_y ssilu''(z) dz: the second summand is this term (P = lambda_y, Q = dz). */
  int src_stage;            /* 0: none */
  int src_mode;
  float src_alpha;
  const float* srcP;        /* (M,N) global */
  const float* srcQ;        /* (M,N) global or NULL (then s = src_alpha * phis(Zs) * srcP) */
} gn_chain_op;
typedef struct {
  int M;
  int n_ops;
  gn_chain_op ops[GN_CHAIN_MAX_OPS];
} gn_chain_args;
int gn_chain_f32(const gn_chain_args* args, void* stream);

/* The same chain programs on the bf16 matrix pipe with SPLIT operands (csrc/chain2.hip): every fp32 operand is the
 * exact sum of three bf16 planes (hi + mid + lo); `nprod` = 6 keeps the six largest cross products per element
 * (fp32-equivalent: dropped terms are below 2^-24 of the product; 6/16 of the f32-MFMA pipe time), 3 keeps
 * hh + hm + mh, 1 is plain bf16 operands — always fp32 accumulation, fp32 epilogue, exact fp32 residual stream.
 * Differences from gn_chain_f32: every GEMM op's `W` points at the PACKED weight produced by gn_pack_weight_split
 * (not at the fp32 matrix); N % 16 == 0, K % 4 == 0; slots 0/1 live in LDS, slot 2 is a register-resident parking
 * slot (written by a plain SCALE or as a GEMM's y_slot; readable as mul/res/res2 only).
 * Replaces: Dense/ResidualLayer stacks (base_layers.py:44-89) exactly like gn_chain_f32. */
/* nprod = GN_CHAIN_F16X2: two fp16 planes instead, x ~= hi + 2^-11 lo (22 significand bits), three products
 * hh + 2^-11 (hl + lh) — the error-corrected half-precision form; half the matrix-pipe time and two thirds of the LDS
 * traffic of nprod = 6 at ~4x the operand rounding of fp32 (force MAE 1e-6 eV/A on the 4-block models).  Operands must
 * stay below 65504 in magnitude (beyond: inf, which propagates) and lose relative accuracy below 2^-14 times the scale of
 * their dot product: meant for activations and first-order adjoints, whose scale the model fixes — not for sweeps whose
 * scale follows the loss.  `W` must have been packed with fmt = GN_SPLIT_F16X2. */
#define GN_CHAIN_F16X2 2
/* nprod = GN_CHAIN_F16X2 | GN_CHAIN_WIDE: the same arithmetic in the "wide" kernel layout (csrc/chain3.hip): workgroups of 4
 * waves x 32 output columns over row tiles of gn_chain_wide_tile_rows(M) rows (a multiple of 8, at most 48), two workgroups
 * per CU — one workgroup's epilogue / LOAD / weight wait runs under the other's MFMA phase.  Same results bit for bit. */
#define GN_CHAIN_WIDE 0x100
/* Per-launch tuning of the wide layout, OR-ed into `nprod` (ABI 13: until ABI 12 two process-global setters):
 * GN_CHAIN_WIDE_ROWS(r), r a multiple of 8 in 8..48, fixes the tile height (0 = gn_chain_wide_tile_rows(M));
 * GN_CHAIN_WIDE_STAGGER(u) starts the second workgroup of a CU u x 64 cycles late (0 = none). */
/* nprod = GN_CHAIN_F16X2 | GN_CHAIN_ROW (ABI 15; csrc/chain4.hip): the same programs with a wave owning 16 ROWS and all
 * columns — the slots are fp32 register arrays (exact residual stream), the fp16 planes of a GEMM's operand are formed per
 * GEMM under a fresh power-of-two row scale (no range limit on activations, no restriction on programs that add global
 * tensors into resident values), only the weights pass through LDS (one loader wave per workgroup, double-buffered).
 * `W` must have been packed with fmt = GN_SPLIT_F16X2_ROW; additionally K % 16 == 0 and width % 16 == 0.
 * Same results as GN_CHAIN_F16X2 to fp32 rounding (not bit for bit: the K order inside a dot product differs). */
#define GN_CHAIN_ROW 0x200
#define GN_CHAIN_WIDE_ROWS(r) ((((r) / 8) & 0xf) << 12)
#define GN_CHAIN_WIDE_STAGGER(u) (((u) & 0xffff) << 16)
/* A program must start with a GN_OP_LOAD (the kernel's prologue relies on that op's barrier); else hipErrorInvalidValue. */
int gn_chain_split_f32(const gn_chain_args* args, int nprod, void* stream);
/* the automatic tile height of the wide layout for M rows (a pure function) */
int gn_chain_wide_tile_rows(int M);
/* W (N,K) fp32 with row pitch ldw — or, trans != 0, the (K,N) matrix whose transpose is the weight — -> three bf16
 * planes in MFMA-fragment order; `out` holds gn_pack_weight_split_bytes(N,K) bytes (16-byte aligned). */
int gn_pack_weight_split(const float* W, int N, int K, int ldw, int trans, void* out, void* stream);
/* fmt: GN_SPLIT_BF16X3 (the above) or GN_SPLIT_F16X2 (two fp16 planes, for nprod = GN_CHAIN_F16X2; uses the first two
 * thirds of the same buffer size). */
#define GN_SPLIT_BF16X3 0
#define GN_SPLIT_F16X2 1
/* GN_SPLIT_F16X2_ROW (ABI 15): two fp16 planes as GN_SPLIT_F16X2 with the K order of the row-resident layout
 * (nprod = GN_CHAIN_F16X2 | GN_CHAIN_ROW): element i of lane group g in k-chunk c is k = 16 (2 c + i / 4) + 4 g + i % 4. */
#define GN_SPLIT_F16X2_ROW 2
int gn_pack_weight_split_fmt(const float* W, int N, int K, int ldw, int trans, int fmt, void* out, void* stream);
int64_t gn_pack_weight_split_bytes(int N, int K);
/* The same for ALL weights of a training step in one launch (the weights change every optimizer step,
 * trainer.py:353-358): `jobs` is a DEVICE table sorted by unit_begin (first entry 0); job j owns the units
 * [unit_begin, unit_begin + ceil(N/16) * ceil(K/32) * 64); total_units = the end of the last job. */
typedef struct {
  const float* W; void* out;
  int N, K, ldw, trans, unit_begin, fmt;   /* fmt: GN_SPLIT_* */
} gn_pack_job;
int gn_pack_weight_split_grouped(const gn_pack_job* jobs, int n_jobs, int total_units, void* stream);

/* C (M,N) = alpha * A^T B with A (K,M), B (K,N) row-major: the weight-gradient product dW = dY^T X of every
 * Dense (base_layers.py:5-48; autograd of torch.nn.Linear in the reference) and the weight adjoints of the double
 * backward.  Split-K over `splitk` slices (gn_gemm_tn_splitk(M,N,K) gives the recommended count); `ws` is a
 * caller-owned workspace of splitk*M*N floats (may be NULL when splitk <= 1).  Deterministic (no atomics). */
int gn_gemm_tn_f32(const float* A, const float* B, float* C, int M, int N, int K, int lda, int ldb, int ldc,
                   float alpha, float* ws, int splitk, void* stream);
int gn_gemm_tn_splitk(int M, int N, int K);

/* Grouped form: ALL weight-gradient products of one training step (the ~300 dW = X^T Y leaves of the double
 * backward, trainer.py:346) as one launch + one fold launch.  Device-resident tables built by the caller:
 *   probs[p]    one product: out_p (M,N) = alpha X^T Y, X (K,M) row pitch ldx, Y (K,N) row pitch ldy, split over `splitk`
 *               slices of `kchunk` rows (multiple of 16); its workgroups are [wg_begin, wg_begin +
 *               ceil(M/64)*ceil(N/64)*splitk); partial z lands at ws + ws_off + z*M*N
 *   targets[t]  one accumulator (a parameter's .grad, n floats — contiguous, or a column slice): out[i] += sum of ws[slice_off[k] + i]
 *               for k in [slice_begin, slice_end) in list order (deterministic); workgroups [wg_begin, +ceil(n/64))
 * probs and targets are sorted by wg_begin (first entry 0). */
typedef struct {
  const float* X; const float* Y;
  int M, N, K, ldx, ldy, splitk, kchunk, wg_begin;
  int64_t ws_off;
  float alpha;              /* factor of this product (out_p = alpha X^T Y) */
  int pad_;
} gn_tn_problem;
typedef struct {
  float* out;
  int64_t n;
  int slice_begin, slice_end, wg_begin;
  int cols;                 /* 0: out is n contiguous floats; > 0: rows of `cols` floats with row pitch `ld` (a column
                             * slice of a wider parameter's .grad: the blocks of a concat-Dense weight) */
  int ld, pad_;
} gn_tn_target;
int gn_gemm_tn_grouped_f32(const gn_tn_problem* probs, int n_prob, int total_wg, const gn_tn_target* targets,
                           int n_target, int total_fold_wg, const int64_t* slice_off, float* ws, void* stream);

/* ---- radial-weighted edge -> atom aggregation (csrc/aggregate.hip) ---------------------------------------------------
 * AtomUpdateBlock / OutputBlock (atom_update_block.py:60-68,157-172):  out[a] = scale * sum_{e: id_a[e] = a} m[e] (.) (W rbf[e])
 * in ONE pass (the reference: Dense over the edges, Hadamard product, torch_scatter.scatter(add)).
 *   m (E,C) rbf (E,R) W (C,R) row-major; perm/seg_off = CSR of the edges by target atom (perm NULL: edges sorted);
 *   out (n_atoms, C).  C == 128 and R == 16 (every published configuration); other shapes: hipErrorInvalidValue.
 * Adjoint w.r.t. m and rbf (W constant: the force pass), one pass, deterministic:
 *   g_m[e] = scale * g_out[id_a[e]] (.) (W rbf[e]);  g_rbf[e] = scale * W^T (g_out[id_a[e]] (.) m[e]);  either may be NULL.
 *   accum bit 0: g_m += instead of =; bit 1: g_rbf += (running gradient of a tensor with several fused consumers). */
int gn_rbf_aggregate_fwd_f32(const float* m, const float* rbf, const float* W, const int32_t* perm, const int32_t* seg_off,
                             float* out, int64_t n_atoms, int C, int R, float scale, void* stream);
int gn_rbf_aggregate_bwd_f32(const float* g_out, const float* m, const float* rbf, const float* W, const int32_t* id_a,
                             float* g_m, float* g_rbf, int64_t E, int C, int R, float scale, int accum, void* stream);

/* ---- tensor basis in ANGLE form (csrc/geometry.hip, csrc/bilinear_ang.hip) ------------------------------------------
 * Instead of the (Q, S^2 = 49) harmonics of TensorBasisLayer (basis_layers.py:239-295) — 196 B per quadruplet, re-read
 * by every interaction block — the geometry kernel emits ang (Q,4) = (sin, cos) of Phi_cab and of Theta_cabd (16 B) and
 * the bilinear kernels rebuild Y_lm on the fly.  g_ang (Q,4) = (dE/dPhi_cab, dE/dTheta_cabd, 0, 0).
 * Shapes: S = 49, C = I = 32 (the published GemNet-Q configuration); anything else: hipErrorInvalidValue. */
int gn_quad_angles_fwd_f32(const float* R, const int32_t* qc, const int32_t* qa, const int32_t* qb, const int32_t* qd,
                           float* ang, int64_t Q, void* stream);
int gn_quad_angles_bwd_ld_f32(const float* g_ang, const float* R, const int32_t* qc, const int32_t* qa, const int32_t* qb,
                              const int32_t* qd, float* Gc, int ldc, float* Gb, int ldb, float* Gd, int ldd, int64_t Q,
                              void* stream);
/* `arith` of the three kernels below: GN_ANG_F16 = both MFMA operands split in registers into two fp16 planes, three
 * v_mfma_f32_16x16x32_f16 products, fp32 accumulation (cotangent blocks under one exact power-of-two scale per edge, lo planes
 * scaled by 2^11 with accumulators of their own); 0 = the f32-input MFMA (same results to fp32 rounding).  K1 takes x as it
 * is: with GN_ANG_F16 |x| must stay below 65 504 — the host passes 0 for models that left the fp16-plane Dense arithmetic and
 * for operands whose magnitude follows the caller's loss (kernels.bil_reduce_project).  Per call: no library state (ABI 13). */
#define GN_ANG_F16 1
/* K1 + K2 of the bilinear layer as gn_bil_reduce_project_f32, Y given as angles */
int gn_bil_reduce_project_ang_f32(const float* ang, const float* x, const int32_t* expand_idx, const int32_t* seg_off,
                                  const float* B, float* Sm, float* P, int64_t E, int S, int C, int I, int arith,
                                  void* stream);
/* per-quadruplet x-adjoint rows as gn_bil_expand_f32, Y given as angles.  row_pos (optional, ABI 14): dxt row that receives
 * quadruplet q — its position in the order of the EXPAND rows, so that gn_segsum_rows_f32 reads contiguous rows (perm = NULL)
 * instead of gathering 128-byte rows through the permutation; the same rows are summed in the same order. */
int gn_bil_expand_ang_f32(const float* ang, const float* dSm, const int32_t* seg_off, float* dxt, int64_t E, int S, int C,
                          int arith, const int32_t* row_pos, void* stream);
/* The x-adjoint of the quadruplet layer WITHOUT per-quadruplet rows in memory (ABI 15; csrc/bilinear_ang.hip,
 * bil_expand_rows_ang_kernel): dx[j] = sum_{q: g(q) = j} Y[q] dSm[r(q)] (interaction_block.py:517-566, backward through the
 * gather of x) with a wave owning 32 expand rows of one target atom and walking that atom's reduce edges, accumulators in
 * registers.  a_perm (or NULL) / a_seg (A+1): the reduce edges grouped by target atom; j_off (A+1): the expand rows of atom a
 * are [j_off[a], j_off[a+1]); qmap: per atom a dense grid [edges of a][rows of a] of quadruplet numbers (-1: none) starting at
 * g_off[a]; task t works on rows task_row0[t] .. + tile - 1 of atom task_atom[t], tile = 32 or 64 = bits 8-15 of `arith` (0: 32).
 * Replaces gn_bil_expand_ang_f32 + the segmented sum. */
int gn_bil_expand_rows_ang_f32(const float* ang, const float* dSm, const int32_t* a_perm, const int32_t* a_seg,
                               const int32_t* j_off, const int32_t* qmap, const int32_t* g_off, const int32_t* task_atom,
                               const int32_t* task_row0, int64_t n_tasks, float* dx, int S, int C, int arith, void* stream);
/* Second-order sweeps of GemNet-Q force training (trainer.py:338-346: loss.backward() through dE/dR) in angle form (ABI 13).
 * tang (Q,4) = (dPhi_cab, dTheta_cabd, 0, 0): the tangents of the two angles along the position tangent u = dL/dF
 * (gn_quad_angles_jvp_f32: the double backward of gn_quad_angles_bwd_ld_f32); the kernels rebuild
 * dY[q] = Y_theta dPhi + Y_phi dTheta with dual numbers next to Y[q].  f32-input MFMA (tangents carry the scale of the loss).
 *   gn_bil_reduce_project_ang_tan_f32   Smd[e] = sum_{q in seg(e)} ( dY[q] (x) x[g(q)] + Y[q] (x) tx[g(q)] )
 *                                        Pd[e]  = B[e]^T Smd[e] + tB[e]^T Sm[e]
 *       (tang / x NULL: no first term; tx NULL: no second; at least one; tB NULL: no tB^T Sm term; Pd NULL: K2 skipped) — the
 *       tangent sweep S3 of the quadruplet bilinear layer (interaction_block.py:517-566, efficient.py:159-189) in ONE launch
 *   gn_bil_expand_ang_tan_f32           dxt[q] = Y[q] D1[e] + dY[q] D2[e]    (D1 NULL: the second term only) — the per-quadruplet
 *       x-adjoint rows of the second adjoint S4 (summed over the expand rows by gn_segsum_rows_f32) */
int gn_quad_angles_jvp_f32(const float* R, const float* tR, const int32_t* qc, const int32_t* qa, const int32_t* qb,
                           const int32_t* qd, float* tang, int64_t Q, void* stream);
int gn_bil_reduce_project_ang_tan_f32(const float* ang, const float* tang, const float* x, const float* tx,
                                      const int32_t* expand_idx, const int32_t* seg_off, const float* B, const float* tB,
                                      const float* Sm, float* Smd, float* Pd, int64_t E, int S, int C, int I, void* stream);
int gn_bil_expand_ang_tan_f32(const float* ang, const float* tang, const float* D1, const float* D2, const int32_t* seg_off,
                              float* dxt, int64_t E, int S, int C, void* stream);
/* gn_bil_expand_ang_tan_f32 + the segmented sum over the expand rows in ONE pass without per-quadruplet rows in memory (ABI 15):
 * dx[j] = sum_{q: g(q) = j} (Y[q] D1[r(q)] + dY[q] D2[r(q)]), per-atom grid arguments as gn_bil_expand_rows_ang_f32, tile = 64. */
int gn_bil_expand_rows_ang_tan_f32(const float* ang, const float* tang, const float* D1, const float* D2, const int32_t* a_perm,
                                   const int32_t* a_seg, const int32_t* j_off, const int32_t* qmap, const int32_t* g_off,
                                   const int32_t* task_atom, const int32_t* task_row0, int64_t n_tasks, float* dx, int S, int C,
                                   int tile, void* stream);
/* The same x-adjoint SUMMED over the expand rows, without the per-quadruplet rows in memory: dx[j] = sum_{q: g(q) = j}
 * Y[q] dSm[r(q)].  Needs the quadruplet structure of GemNet (data_container.py:331-397): reduce edge and expand row of a
 * quadruplet end in the same target atom, and the expand rows (intermediate triplets) are sorted by that atom —
 * a_perm / a_seg: the reduce edges grouped by target atom (CSR over n_atoms; a_perm NULL = already grouped), j_off: first
 * expand row of every atom (n_atoms + 1).  One workgroup per atom keeps its rows in LDS: max_J >= the largest
 * j_off[a+1] - j_off[a], (max_J * 32 + 13 568) * 4 bytes <= 160 KB (max_J <= 848), else hipErrorInvalidValue (the caller
 * uses gn_bil_expand_ang_f32 + gn_segsum_rows_f32); an atom with more rows than max_J traps.  Deterministic (the edges of
 * an atom are summed in a_perm order).  Every row of dx that belongs to an atom is written; S = 49, C = 32. */
int gn_bil_expand_atoms_ang_f32(const float* ang, const float* dSm, const int32_t* seg_off, const int32_t* expand_idx,
                                const int32_t* a_perm, const int32_t* a_seg, const int32_t* j_off, float* dx,
                                int64_t n_atoms, int max_J, int S, int C, void* stream);
/* gradient w.r.t. the two angles of all nb <= 4 blocks sharing the basis (gn_bil_dy_multi_f32 contracted with dY/d angle) */
int gn_bil_dy_multi_ang_f32(const float* const* dSm_list, const float* const* x_list, int nb, const float* ang,
                            const int32_t* expand_idx, const int32_t* seg_off, float* g_ang, int64_t E, int S, int C,
                            int arith, void* stream);

/* batched small matmul C[b] = opA(A[b]) opB(B[b]), b < batch; row-major (m,k)/(k,n) blocks.
 * Replaces torch.matmul(rbf_W1, sum_k) and its adjoints (efficient.py:177-182). */
int gn_bmm_f32(const float* A, const float* B, float* C, int batch, int m, int n, int k,
               int trans_a, int trans_b, void* stream);

/* ---- index construction on the device (P12 of SURVEY.md §8 resident on the GPU, row N1) ---------------------------
 * What DataContainer.__getitem__ builds on the host per batch (data_container.py:244-408 edges / id_swap /
 * id_undir / batch_seg, :410-425 triplets, :427-489 quadruplets, :520-565 repeat_blocks / ragged_range) and
 * ase_calculator.py:155-158 rebuilds every MD step.  Same arrays, same canonical order and the same distance
 * rounding as the host builder (include/gemnet_index.h), as int32 device arrays.
 *   R         (A,3) float32 or float64 (r_is_f64) device positions
 *   mol_off   (B+1) int32 device: first atom of each molecule;  sq_off (B+1) int32 device: prefix sums of n_m^2
 *   nmax      largest molecule;  sum_n2 = sq_off[B]  (both known on the host from N)
 *   ws        caller-owned device workspace of gn_index_gpu_ws_bytes() bytes, shared by both stages
 * stage1 writes batch_seg (A), the edge arrays (capacity sum_n2 - A each: upper bound of E resp. Eint) and
 * sizes[6] = {E, T, Eint, Ica, Idb, Q} (host; one stream synchronisation).  With id_a == NULL only sizes[0] and
 * sizes[2] are produced.  stage2 fills the triplet / quadruplet arrays (caller allocates them from `sizes`).
 * Arrays with more than 2^31 - 1 entries are rejected. */
int64_t gn_index_gpu_ws_bytes(int A, int64_t sum_n2, int triplets_only);
int gn_index_gpu_stage1(const void* R, int r_is_f64, const int32_t* mol_off, const int32_t* sq_off, int B, int A,
                        int nmax, int64_t sum_n2, double cutoff, double int_cutoff, int triplets_only, void* ws,
                        int32_t* batch_seg, int32_t* id_a, int32_t* id_c, int32_t* id_undir, int32_t* id_swap,
                        int32_t* id4_int_a, int32_t* id4_int_b, int64_t* sizes, void* stream);
int gn_index_gpu_stage2(const int32_t* mol_off, const int32_t* sq_off, int B, int A, int64_t sum_n2,
                        int triplets_only, void* ws, const int32_t* id_a, const int32_t* id_c,
                        const int32_t* id4_int_a, const int32_t* id4_int_b, int64_t E, int64_t Eint,
                        int32_t* id3_reduce_ca, int32_t* id3_expand_ba, int32_t* Kidx3, int32_t* id4_reduce_ca,
                        int32_t* id4_expand_db, int32_t* id4_reduce_cab, int32_t* id4_expand_abd, int32_t* Kidx4,
                        int32_t* id4_reduce_intm_ca, int32_t* id4_expand_intm_db, int32_t* id4_reduce_intm_ab,
                        int32_t* id4_expand_intm_ab, void* stream);

/* Capacity form of the triplets-only build (ABI 14): the whole MD step of ase_calculator.py:148-170 — neighbour list, index
 * arrays, model — as ONE capturable graph.  No read-back: the counts stay on the device.  The edge / triplet arrays of the
 * batch are built into `staging` (4 e_cap + 2 t_cap int32) and then committed into the arrays the model reads — e_cap /
 * t_cap rows each, the rows behind the batch filled with the pad rows of gemnet_pytorch_amd/padded.py (triplets-only
 * layout: n_groups groups of 3 dummy atoms behind atom a_cap) — by one kernel.  A batch that does not fit, or whose padding
 * would break the rules of that scheme, leaves the arrays as they were (valid indices of the previous step) and reports
 *   state[0] |= err (sticky)   state[1] = E   state[2] = T   state[3] = err of this call
 *   err bits: 1 E > e_cap, 2 T > t_cap, 4 pad triplets without a complete quad of pad edges, 8 more than
 *             max(deg_bound, 2) pad edges into one dummy atom
 * (device int32[4], read by the caller when it synchronises anyway).  gn_index_poison_f32 overwrites an output of such a
 * step with NaN (state[3] != 0), so that the numbers computed from stale arrays cannot be mistaken for results.
 * e_cap, t_cap even; ws / mol_off / sq_off / nmax / sum_n2 as for stage1 with triplets_only = 1. */
int gn_index_gpu_padded_t(const void* R, int r_is_f64, const int32_t* mol_off, const int32_t* sq_off, int B, int A, int nmax,
                          int64_t sum_n2, double cutoff, void* ws, int e_cap, int t_cap, int a_cap, int n_groups,
                          int deg_bound, int32_t* staging, int32_t* id_c, int32_t* id_a, int32_t* id_swap, int32_t* id_undir,
                          int32_t* id3_reduce_ca, int32_t* id3_expand_ba, int32_t* state, void* stream);
int gn_index_poison_f32(float* x, int64_t n, const int32_t* state, void* stream);
/* The same for quadruplet models (data_container.py:427-489 as well): five families with their capacities e_cap, t_cap,
 * eint_cap, i_cap, q_cap; arrays (host array of 16 device pointers) = id_c, id_a, id_swap, id_undir, id3_reduce_ca,
 * id3_expand_ba, id4_int_a, id4_int_b, id4_reduce_intm_ca, id4_expand_intm_db, id4_reduce_intm_ab, id4_expand_intm_ab,
 * id4_reduce_ca, id4_expand_db, id4_reduce_cab, id4_expand_abd — each of its family's capacity; staging: 4 e_cap + 2 eint_cap
 * int32 (the two small families; the three large ones are written straight into `arrays` once a one-thread kernel has
 * compared ALL counts with the capacities — or not at all); dummy groups of 4 atoms behind a_cap, pad edges in units of six.
 *   state (device int32[8]): [0] |= err  [1] E  [2] T  [3] err of this call  [4] Eint  [5] I  [6] Q
 *   err bits as above plus 16 Eint > eint_cap, 32 I > i_cap, 64 Q > q_cap, 128 inconsistent intermediate-triplet lists
 * ws: gn_index_gpu_ws_bytes(A, sum_n2, triplets_only = 0). */
int gn_index_gpu_padded_q(const void* R, int r_is_f64, const int32_t* mol_off, const int32_t* sq_off, int B, int A, int nmax,
                          int64_t sum_n2, double cutoff, double int_cutoff, void* ws, int e_cap, int t_cap, int eint_cap,
                          int i_cap, int q_cap, int a_cap, int n_groups, int deg_bound, int32_t* staging, int32_t* const* arrays,
                          int32_t* state, void* stream);

/* ---- row gather / segmented sum (P2/P3/P10: `x[id3_expand_ba]` interaction_block.py:678,
 *      `x[id4_expand_*]` :543,:548, `x_ac[id_swap]` :693, h[id] embedding_block.py:70-71 and
 *      torch_scatter.scatter(..., reduce="add") atom_update_block.py:67,172, gemnet.py:580) -- */
int gn_gather_rows_f32(const float* x, const int32_t* idx, float* y, int64_t T, int C, void* stream);
/* y[t, :] = scale * x[idx[t], :] (.) m[t, :] — a row gather fused with a Hadamard product (C % 4 == 0, 16-byte aligned):
 * the operand q[e] = scale * g[id_a[e]] (.) m[e] of the weight gradient of dense_rbf in the training step
 * (atom_update_block.py:60-68 under autograd). */
int gn_gather_mul_f32(const float* x, const int32_t* idx, const float* m, float* y, int64_t T, int C, float scale,
                      void* stream);
/* x[n,:] = sum_{k in [seg_off[n], seg_off[n+1])} y[perm ? perm[k] : k, :]   (deterministic, no atomics) */
int gn_segsum_rows_f32(const float* y, const int32_t* perm, const int32_t* seg_off, float* x,
                       int64_t N, int C, void* stream);
/* x[n,:] = sum_{k < n_terms} sign[k] * sum_{i in [seg_off[k][n], seg_off[k][n+1])} y[k][perm[k] ? perm[k][i] : i, :]
 * — up to 4 CSR lists over the same N rows in one launch (y / perm / seg_off / sign: HOST arrays of n_terms device
 * pointers / floats); rows of C <= 4 floats.  The assembly of dE/dR from the per-triplet and per-edge terms of the
 * geometry adjoint (gemnet.py:420-451 differentiated; the reference: autograd through index_select).  Deterministic. */
int gn_segsum_multi_f32(int n_terms, const float* const* y, const int32_t* const* perm, const int32_t* const* seg_off,
                        const float* sign, float* x, int64_t N, int C, void* stream);

/* ---- CSR groupings of the index plans (csrc/csr.hip; ABI 13) ------------------------------------------------------------
 * The transposed grouping of a gather `y = x[idx]` (interaction_block.py:543,548,562,678,693; embedding_block.py:70-71;
 * atom_update_block.py:67): perm[k] = position of the k-th entry in row order (STABLE: entries of a row keep their input order,
 * the permutation of a stable argsort), seg_off[r] = number of entries with a key below r (n_rows + 1 values).  One rocPRIM
 * radix sort of (key, position) pairs over the significant key bits + one lower-bound launch; `ws`: caller-owned workspace of
 * gn_csr_ws_bytes(n, n_rows) bytes (256-byte aligned); no allocation, no synchronisation (capturable).  Keys int32 in
 * [0, n_rows), n < 2^31.  gn_seg_offsets_i32: the offsets alone for keys that are already sorted. */
int64_t gn_csr_ws_bytes(int64_t n, int64_t n_rows);
int gn_csr_build_i32(const int32_t* keys, int64_t n, int64_t n_rows, int32_t* perm, int32_t* seg_off, void* ws,
                     int64_t ws_bytes, void* stream);
int gn_seg_offsets_i32(const int32_t* sorted_keys, int64_t n, int64_t n_rows, int32_t* seg_off, void* stream);
/* CSR (perm (T), seg_out (n_rows + 1)) of T items that are sorted by their EDGE (item range of edge e: seg_off_of_edge[e] ..
 * seg_off_of_edge[e+1]) by the ROW of that edge, from the edge CSR (perm_e (E) or NULL = identity, seg_e (n_rows + 1)): the
 * edge lists expanded into item ranges — the permutation of the stable sort of the T item keys, without sorting them.
 * ws: E + 1 int32 (device).  Two launches, capturable. (ABI 14) */
int gn_expanded_csr_i32(const int32_t* perm_e, const int32_t* seg_e, int64_t n_rows, const int32_t* seg_off_of_edge, int64_t E,
                        int32_t* perm, int32_t* seg_out, int32_t* ws, void* stream);

/* ---- bilinear aggregation, CSR-segmented (P4: efficient.py:159-189 without the zero-padded
 *      (E,Kmax,C) tensors; SURVEY.md Appendix D kernels K1 and its two adjoints) -----------
 * r(t) = reduce edge of triplet/quadruplet t (sorted ascending, seg_off[e]..seg_off[e+1]),
 * g(t) = expand row.  S = num_spherical (7) or num_spherical^2 (49).  C = channels (<=128). */
/* Sm[e,s,c] = sum_{t in seg(e)} Y[t,s] * x[g(t),c] */
int gn_bil_reduce_f32(const float* Y, const float* x, const int32_t* expand_idx,
                      const int32_t* seg_off, float* Sm, int64_t E, int S, int C, void* stream);
/* dx[j,c] = sum_{k in segT(j)} sum_s Y[t,s] * dSm[r(t),s,c],  t = permT[k]   (j < J rows of x) */
int gn_bil_reduce_t_f32(const float* Y, const float* dSm, const int32_t* reduce_idx,
                        const int32_t* permT, const int32_t* segT_off, float* dx,
                        int64_t J, int S, int C, void* stream);
/* The same adjoint when r(t) and g(t) always lie in one GROUP of rows (triplets c->a<-b: both edges end in atom a;
 * data_container.py:262-300).  Group g owns rows grp_rows[grp_off[g]..grp_off[g+1]); grp_kseg[i] = {k0,k1} is the
 * transposed segment (range of permT) of row grp_rows[i]; rposT[k] = position of r(permT[k]) inside its group.
 * A workgroup parks the group's dSm blocks in LDS (max_rows*S*C*4 <= 160 KB, else hipErrorInvalidValue and the
 * caller uses gn_bil_reduce_t_f32) so dSm is read from HBM once, not once per t.  max_rows is a promise: a group with more
 * rows makes the kernel trap (the launch aborts) instead of writing past its LDS tile.
 * S = 7 and C = 64 only (else hipErrorInvalidValue).  Rows outside every group are not written. */
int gn_bil_reduce_t_grouped_f32(const float* Y, const float* dSm, const int32_t* grp_rows, const int32_t* grp_off,
                                const int32_t* grp_kseg, const int32_t* permT, const int32_t* rposT, float* dx,
                                int64_t G, int max_rows, int S, int C, void* stream);
/* dxt[t,c] = sum_s Y[t,s] * dSm[r(t),s,c]: the per-triplet/quadruplet rows of the adjoint above (dx[j] = sum of
 * dxt over the transposed segment of j: gn_segsum_rows_f32), grouped by reduce edge so that dSm[e] is read once
 * per edge instead of once per quadruplet (the S = 49 tensor basis: 56 GB -> 4 GB of traffic at B = 32). */
int gn_bil_expand_f32(const float* Y, const float* dSm, const int32_t* seg_off, float* dxt, int64_t E, int S, int C,
                      void* stream);
/* dY[t,s] = sum_c dSm[r(t),s,c] * x[g(t),c] */
int gn_bil_dot_f32(const float* dSm, const float* x, const int32_t* expand_idx,
                   const int32_t* seg_off, float* dY, int64_t E, int S, int C, void* stream);

/* Fused K1 + K2: Sm as above and P[e,i,c] = sum_s B[e,s,i] Sm[e,s,c]  (B = rbf_W1 (E,S,I); replaces the
 * per-edge bmm of efficient.py:180).  Both Sm (E,S,C) and P (E,I,C) are written. */
int gn_bil_reduce_project_f32(const float* Y, const float* x, const int32_t* expand_idx,
                              const int32_t* seg_off, const float* B, float* Sm, float* P, int64_t E, int S,
                              int C, int I, void* stream);
/* Extended K1 + K2 for the tangent sweep of the training step (trainer.py:346 through efficient.py:159-189), spherical
 * basis shapes only ((S, C, I) = (7, 64, 16); anything else: hipErrorInvalidValue):
 *   Sm[e] = Sm_init[e] + sum_{t in seg(e)} Y[t] (x) x[g(t)]        (Sm_init NULL: from zero)
 *   P[e]  = B[e]^T Sm[e] + B2[e]^T Sm2[e]                          (B2 / Sm2 NULL together: no second term; P NULL: K2 skipped)
 * so that dSm = K1(dY, x) + K1(Y, dx) and dP = B^T dSm + dB^T Sm take two launches and no elementwise adds. */
int gn_bil_reduce_project2_f32(const float* Y, const float* x, const int32_t* expand_idx, const int32_t* seg_off,
                               const float* B, const float* Sm_init, const float* B2, const float* Sm2, float* Sm, float* P,
                               int64_t E, int S, int C, int I, void* stream);
/* K1 + K2 + K3 in one launch (efficient.py:173-188 incl. the final `torch.matmul(..., self.weight)`), for callers that
 * do not need P afterwards (inference / frozen weights):
 *   Sm as above (written: the adjoint needs it);  out[e,o] = alpha * sum_{i,c} P[e,i,c] * W2T[o, i*C + c]
 * W2T = the (C,I,O) bilinear weight permuted to (O, I*C), k-contiguous, 16-byte aligned.
 * S = 7, C = 64, I = 16, O = 64 only (else hipErrorInvalidValue: use gn_bil_reduce_project_f32 + gn_gemm_f32). */
/* W2T_planes (optional, NULL = K3 on the f32-input MFMA): the same weight as two fp16 planes in fragment order,
 * gn_pack_weight_split_fmt(W2T, 64, 1024, 1024, 0, GN_SPLIT_F16X2, ...): K3 then runs as three v_mfma_f32_16x16x32_f16
 * products per fp32 product (P split into hi + 2^-11 lo planes in LDS), 5 x less matrix-pipe time, same result to fp32
 * rounding (22-bit operands: |P| < 65 504). */
int gn_bil_fused_fwd_f32(const float* Y, const float* x, const int32_t* expand_idx, const int32_t* seg_off,
                         const float* B, const float* W2T, const void* W2T_planes, float* Sm, float* out, int64_t E, int S,
                         int C, int I, int O, float alpha, void* stream);
/* Fused adjoint: gB[e,s,i] = sum_c Sm[e,s,c] dP[e,i,c]; dSm[e,s,c] = sum_i B[e,s,i] dP[e,i,c];
 * dY[t,s] = sum_c dSm[r(t),s,c] x[g(t),c].  x rows 16-byte aligned, C % 4 == 0. */
int gn_bil_project_bwd_f32(const float* dP, const float* Sm, const float* B, const float* x,
                           const int32_t* expand_idx, const int32_t* seg_off, float* gB, float* dSm, float* dY,
                           int64_t E, int S, int C, int I, void* stream);
/* Same with `accumulate` bit 0: dY += (the Y gradient summed over the interaction blocks that share one basis tensor
 * — saves a (T,S)-sized add per block: 1.8 GB for the quadruplet basis at B = 32); bit 1: gB += (the radial part of the
 * basis is shared by the blocks in the same way); bit 2: dSm += (the cross term dB mu_P of the training step's second
 * adjoint lands on B Pbar; spherical-basis shapes (7, 64, 16) only, else hipErrorInvalidValue). */
int gn_bil_project_bwd_acc_f32(const float* dP, const float* Sm, const float* B, const float* x,
                               const int32_t* expand_idx, const int32_t* seg_off, float* gB, float* dSm, float* dY,
                               int64_t E, int S, int C, int I, int accumulate, void* stream);
/* The adjoint of the whole bilinear tail in one launch (K3^T then the gB / dSm part of the fused adjoint above; dP never
 * leaves LDS):  dP[e,(i,c)] = alpha * sum_o g[e,o] W2[(i,c),o];  gB[e] = Sm[e] dP[e]^T;  dSm[e] = B[e] dP[e].
 * W2 = the bilinear weight as (I*C, O) row-major (efficient.py:159-189 differentiated).  (S, C, I, O) = (7, 64, 16, 64)
 * only (else hipErrorInvalidValue); `accumulate` bit 1: gB += (as gn_bil_project_bwd_acc_f32).  Bit-identical to
 * gn_gemm_f32 + gn_bil_project_bwd_f32(dY = NULL) when W2_planes is NULL.  W2_planes (optional): W2 as two fp16 planes in
 * fragment order, gn_pack_weight_split_fmt(W2, 1024, 64, 64, 0, GN_SPLIT_F16X2, ...): the first product then runs on
 * v_mfma_f32_16x16x32_f16 with split operands (g under one exact power-of-two scale per edge row: any magnitude), same result
 * to fp32 rounding. */
int gn_bil_fused_bwd_f32(const float* g, const float* W2, const void* W2_planes, const float* Sm, const float* B, float* gB,
                         float* dSm, int64_t E, int S, int C, int I, int O, float alpha, int accumulate, void* stream);
/* gn_bil_project_bwd*_f32 accept dY == NULL (gB and dSm only).  The deferred Y gradient of up to 4 blocks that
 * share one basis tensor (S = 49, C = 32 or S = 7, C = 64; else hipErrorInvalidValue) is then produced in one pass:
 *   dY[t,s] = sum_b sum_c x_b[g(t),c] dSm_b[r(t),s,c]        (dSm_list / x_list: host arrays of nb device pointers) */
int gn_bil_dy_multi_f32(const float* const* dSm_list, const float* const* x_list, int nb, const int32_t* expand_idx,
                        const int32_t* seg_off, float* dY, int64_t E, int S, int C, void* stream);

/* ---- basis functions (P6-P8; closed forms of SURVEY.md Appendix A, evaluated in f64 in-kernel) */
/* out[e,n] = d^kd/dd^kd d^kf/df_n^kf [ u(d/c) sqrt(2/c) sin(f_n d/c)/d ]   (basis_layers.py:45-49);
 * (kd,kf) in {(0,0),(1,0),(2,0),(0,1),(1,1)} */
int gn_bessel_rbf_f32(const float* d, const float* freq, float* out, int64_t E, int R,
                      float cutoff, int p, int kd, int kf, void* stream);
/* out[e,l,n] = d^kd/dd^kd [ u(d/c) c^-1.5 N_ln j_l(z_ln d/c) ], kd in {0,1,2}
 * (basis_layers.py:121-128,241-250; z (S,R) f32 roots and nrm (S,R) f64 normalisers are host tables) */
int gn_sph_radial_f32(const float* d, const float* z, const double* nrm, float* out, int64_t E,
                      int S, int R, float cutoff, int p, int kd, void* stream);
/* out[t,l] = d^k/dtheta^k Y_l0(theta), l < S, k in {0,1,2}   (basis_layers.py:130-131 with zero_m_only) */
int gn_ylm0_f32(const float* theta, float* out, int64_t T, int S, int k, void* stream);
/* out[q,j] = d^kt/dtheta^kt d^kp/dphi^kp Y_j(theta,phi), j < S*S in the reference order
 * (m = 0,+1..+l,-l..-1 per l; basis_layers.py:269), kt+kp <= 2 */
int gn_ylm_f32(const float* theta, const float* phi, float* out, int64_t Q, int S, int kt, int kp,
               void* stream);

/* ---- geometry fused with the basis evaluation (P9 + P6/P7, first-order path) ---------------
 * gemnet.py:261-286 (interatomic vectors) -> basis_layers.py:45-49,121-128:
 *   D[e] = |R[id_a[e]] - R[id_c[e]]|, V[e,:] unit vector (V may be NULL), rbf (E,NR), rad (E,S,NR). */
int gn_edge_basis_fwd_f32(const float* R, const int32_t* id_c, const int32_t* id_a, const float* freq,
                          const float* z, const double* nrm, float* D, float* V, float* rbf, float* rad,
                          int64_t E, int NR, int S, float cutoff, int p, void* stream);
/* adjoint: W[e,:] = (g_D[e] + sum g_rbf * d rbf/dd + sum g_rad * d rad/dd) * V[e,:]  (any g_* may be NULL);
 * dE/dR = segsum(W, id_a) - segsum(W, id_c) */
int gn_edge_basis_bwd_f32(const float* g_D, const float* g_rbf, const float* g_rad, const float* R,
                          const int32_t* id_c, const int32_t* id_a, const float* freq, const float* z,
                          const double* nrm, float* W, int64_t E, int NR, int S, float cutoff, int p,
                          void* stream);
/* gemnet.py:288-311,420-451 -> basis_layers.py:130-131: Y[t,l] = Y_l0(atan2(max(|u x v|,1e-9), u.v)),
 * u = R[tc]-R[ta], v = R[tb]-R[ta]; theta (T,) optional output */
int gn_trip_basis_fwd_f32(const float* R, const int32_t* tc, const int32_t* ta, const int32_t* tb, float* Y,
                          float* theta, int64_t T, int S, void* stream);
/* adjoint: Gc[t,:] = dE/dR_c, Gb[t,:] = dE/dR_b per triplet (dE/dR_a = -(Gc+Gb)) given gY (T,S) */
int gn_trip_basis_bwd_f32(const float* gY, const float* R, const int32_t* tc, const int32_t* ta,
                          const int32_t* tb, float* Gc, float* Gb, int64_t T, int S, void* stream);

/* ---- twice-differentiable geometry of the force-training step (csrc/geometry2.hip) -----------------------------------
 * calculate_interatomic_vectors (gemnet.py:261-286) and calculate_neighbor_angles / calculate_angles3 (gemnet.py:288-311,
 * :420-451) as value / first adjoint / tangent kernels, so that `loss.backward()` through `autograd.grad(E, R,
 * create_graph=True)` (trainer.py:338-346) needs no pointwise launches for the geometry:
 *   D[e]      = |R[id_a[e]] - R[id_c[e]]|                       gn_dist_fwd_f32
 *   W[e,:]    = gD[e] dD/dR_a   (dD/dR_c = -W)                  gn_dist_bwd_f32     (reduced onto atoms by gn_segsum_multi_f32)
 *   Ddot[e]   = dD . (tR[a] - tR[c]);  H[e,:] = d/dR_a [gD dD/dR_a] (tR[a] - tR[c])      gn_dist_jvp_f32 (Ddot / H may be NULL)
 *   theta[t]  = atan2(max(|u x v|, 1e-9), u . v), u = R[c] - R[a], v = R[b] - R[a]       gn_angle_fwd_f32
 *   Gc, Gb    = g dtheta/dR_c, g dtheta/dR_b  (dtheta/dR_a = -(Gc + Gb))                  gn_angle_bwd_f32
 *   thdot[t]  = dtheta . (du, dv);  (Hc, Hb)[t] = d/d(R_c, R_b) [g dtheta] (du, dv), du = tR[c] - tR[a], dv = tR[b] - tR[a]
 *               — the directional derivative of the first adjoint, by dual numbers          gn_angle_jvp_f32 (thdot / Hc+Hb may be NULL;
 *               g may be NULL when only thdot is wanted) */
int gn_dist_fwd_f32(const float* R, const int32_t* id_c, const int32_t* id_a, float* D, int64_t E, void* stream);
int gn_dist_bwd_f32(const float* gD, const float* R, const int32_t* id_c, const int32_t* id_a, float* W, int64_t E, void* stream);
int gn_dist_jvp_f32(const float* R, const float* tR, const float* gD, const int32_t* id_c, const int32_t* id_a, float* Ddot,
                    float* H, int64_t E, void* stream);
int gn_angle_fwd_f32(const float* R, const int32_t* tc, const int32_t* ta, const int32_t* tb, float* theta, int64_t T,
                     void* stream);
int gn_angle_bwd_f32(const float* g, const float* R, const int32_t* tc, const int32_t* ta, const int32_t* tb, float* Gc,
                     float* Gb, int64_t T, void* stream);
int gn_angle_jvp_f32(const float* R, const float* tR, const float* g, const int32_t* tc, const int32_t* ta, const int32_t* tb,
                     float* thdot, float* Hc, float* Hb, int64_t T, void* stream);

/* Quadruplets c -> a - b <- d (gemnet.py:334-418: two neighbour angles, two vector rejections, the
 * dihedral) fused with the real Y_lm (basis_layers.py:269): Y[q,:] = Y_lm(Phi_cab, Theta_cabd) from
 * the four atom indices of each quadruplet. */
int gn_quad_basis_fwd_f32(const float* R, const int32_t* qc, const int32_t* qa, const int32_t* qb,
                          const int32_t* qd, float* Y, int64_t Q, int S, void* stream);
/* adjoint: Gc, Gb, Gd (Q,3) = dE/dR of atoms c, b, d per quadruplet; dE/dR_a = -(Gc+Gb+Gd) */
int gn_quad_basis_bwd_f32(const float* gY, const float* R, const int32_t* qc, const int32_t* qa,
                          const int32_t* qb, const int32_t* qd, float* Gc, float* Gb, float* Gd, int64_t Q,
                          int S, void* stream);
/* Same with explicit row pitches of the three outputs (Gb and Gd interleaved in one (Q,8) array let the sum over
 * the quadruplets of an intermediate triplet run as ONE float4 segmented sum). */
int gn_quad_basis_bwd_ld_f32(const float* gY, const float* R, const int32_t* qc, const int32_t* qa, const int32_t* qb,
                             const int32_t* qd, float* Gc, int ldc, float* Gb, int ldb, float* Gd, int ldd, int64_t Q,
                             int S, void* stream);

/* ---- pointwise -------------------------------------------------------------------------
 * out[i] = d^k/dx^k ssilu(x[i]), k in {0,1,2,3}   (base_layers.py:51-58) */
int gn_ssilu_f32(const float* x, float* out, int64_t n, int k, void* stream);
/* the pointwise product the composite (twice differentiable) path is built from:
 *   out[i] = c * (k >= 0 ? d^k/dz^k ssilu(z[i]) : 1) * (a ? a[i] : 1) * (b ? b[i] : 1) * (d ? d[i] : 1)
 * ScaledSiLU, its Hadamard / scale-factor epilogues (interaction_block.py:531-552,670-683) and every term of their
 * first and second derivatives are single launches of this form. */
int gn_pm_f32(const float* z, int k, const float* a, const float* b, const float* d, float c, float* out,
              int64_t n, void* stream);
/* backward glue of a fused Dense: with a = act ? ssilu'(z) : 1 and y0 = act ? ssilu(z) : z,
 *   dz[i] = g[i] * c * (mul ? mul[i] : 1) * a      gmul[i] = g[i] * c * y0   (if gmul != NULL) */
int gn_dact_mul_f32(const float* g, const float* z, int act, const float* mul, float c, float* dz,
                    float* gmul, int64_t n, void* stream);

/* ---- trainer-step fusion ---------------------------------------------------------------------
 * Everything Trainer.train_on_batch does between loss.backward() and the next forward (trainer.py:250-278 shared-
 * gradient rescale, :353-356 clip_grad_norm_, :115-160 AdamW / Adam (amsgrad, eps) step, ema_decay.py:68-93) over ONE
 * flat fp32 parameter buffer in two launches.  gscale[i]: gradient rescale (1/num_blocks for shared projections);
 * wd[i]: decoupled weight decay of the element's group (0 = plain Adam).  `partial`: caller-owned workspace of
 * gn_optim_blocks(n) doubles.  ema may be NULL.  norm_out (device float, optional) <- pre-clip global gradient norm.
 * step >= 1 is the Adam step count (bias correction computed on the host in fp32 like torch.optim).
 * `flag` (device int32, optional; ABI 13): with it a step whose global gradient norm is NOT FINITE leaves p, m, v, vmax and ema
 * untouched and sets flag[0] |= flag_bit (the reference would write NaN into every parameter, trainer.py:353-358; here the
 * caller polls the word, warns and — when the fp16-plane arithmetic overflowed — falls back and repeats).  NULL: the
 * reference's behaviour. */
int gn_optim_blocks(int64_t n);
int gn_adamw_ema_step_f32(float* p, const float* g, const float* gscale, const float* wd, float* m, float* v,
                          float* vmax, float* ema, int64_t n, double* partial, float max_norm, float lr, float beta1,
                          float beta2, float eps, int step, float ema_decay, float* norm_out, int32_t* flag, int flag_bit,
                          void* stream);

/* ---- range check of replayed graphs (ABI 13) -------------------------------------------------------------------------
 * flag[0] |= bit when x[0 .. n) holds an inf or a NaN.  The default Dense arithmetic keeps activations in two fp16 planes
 * (GN_CHAIN_F16X2: beyond 65 504 -> inf, which propagates to the energies and forces); the reference's fp32 has no such
 * cliff (base_layers.py:44-48).  An eager forward reads its outputs back; a captured hipGraph (MD loop, padded batches, the
 * training step) cannot: it ends with this launch on its energies and forces plus a copy of the word to pinned host memory,
 * and the host polls that copy without synchronising (runtime.RangeFlag).  The word is sticky until the host clears it. */
int gn_nonfinite_flag_f32(const float* x, int64_t n, int32_t* flag, int bit, void* stream);

/* ---- circular basis of GemNet-Q's intermediate triplets, projected (ABI 14) --------------------------------------------------
 * basis_layers.py:119-131 (rbf_env[id4_expand_intm_ab] * Y_l0) followed by mlp_cbf4 (gemnet.py): one pass, no (I, S R) array:
 *   out[i, n] = sum_{l, r} rad[ie[i], l, r] y[i, l] W[n, l R + r]         rad (Eint, S, R), ie (I) int32, y (I, S), W (N, S R)
 * adjoint (W frozen): g_rad (Eint, S, R) and g_y (I, S) from g (I, N); the rows of an interaction edge are contiguous (ie is
 * sorted: seg_off (Eint + 1) = its segment offsets), one wave per edge sums them in order.  S R <= 64, N <= 16. */
int gn_cbf_project_fwd_f32(const float* rad, const int32_t* ie, const float* y, const float* W, float* out, int64_t I, int S, int R,
                           int N, void* stream);
int gn_cbf_project_bwd_f32(const float* g, const float* rad, const int32_t* seg_off, const float* y, const float* W, float* g_rad,
                           float* g_y, int64_t E, int S, int R, int N, void* stream);

/* ---- the training loss with its cotangents (ABI 14) -----------------------------------------------------------------------
 * trainer.py:330-343: loss = (1 - rho) MAE(E) + rho mean_a |F_a - Ft_a|_2, here with the weights folded by the caller:
 *   loss = w_e sum_i |E_i - Et_i| + w_f sum_a m_a |F_a - Ft_a|_2,  w_f *= *w_f_dev when given (a device scalar: 1 / atoms of a
 *   padded step), m = mask (A) or all ones;  gE = dloss/dE, gF = dloss/dF (0 where the norm is 0, as ATen's norm backward).
 * E, Et, gE: nE values; F, Ft, gF: (A,3).  One workgroup, fixed order of addition. */
int gn_force_loss_f32(const float* E, const float* Et, int64_t nE, const float* F, const float* Ft, int64_t A, const float* mask,
                      float w_e, float w_f, const float* w_f_dev, float* loss, float* gE, float* gF, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* GEMNET_HIP_H */
