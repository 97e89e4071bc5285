// Dense contractions on the f32 MFMA (v_mfma_f32_32x32x2_f32): exact-f32 numerics at the
// f32 vector rate (157 TF peak on MI355X), fused prologue (activation derivative on the A
// operand) and fused epilogue (gather-add, ScaledSiLU, Hadamard, scale, two residual adds).
//
// Replaces every `Dense` (nn.Linear + ScaledSiLU) of the reference
// (gemnet/model/layers/base_layers.py:5-58), ResidualLayer (:61-89), the concat-Dense of
// embedding_block.py:60-75 and the final (E, C*I) x (C*I, O) contraction of the bilinear
// layer (efficient.py:185-188).
//
// Tiling: NW waves (wave64) per block, each wave owns TM x TN tiles of 32x32.  K-step 32 staged
// through LDS as As[BM][36] / Bs[BN][36] (k contiguous, +4 pad: rows 144 B apart -> ds_read_b128
// of 16 consecutive rows hits 16 distinct 16-B slots).  Each lane fetches one float4 (4
// consecutive k) per 32-row fragment; lanes 0-31 take k = kb..kb+3, lanes 32-63 take
// k = kb+4..kb+7, and MFMA step s consumes component s of both operands — the k permutation is
// the same for A and B so the sum over k is unchanged.
//
// Two kernels:
//   gemm_nt_pipe   "NT" operands (both k-contiguous, 16-B aligned rows): the next K-step's global
//                  loads are issued into registers BEFORE the MFMA loop of the current step and
//                  written to LDS after it, so HBM/L2 latency hides under the matrix pipe even at
//                  one wave per SIMD (the E = 18 k-row GEMMs of a 32-molecule batch fill the chip
//                  about once).  Small-M shapes (atom-side layers, M = 1024) get 32-row tiles so
//                  the launch spreads over 4x more CUs and per-block latency drops 2x.
//   gemm_generic   any transposition / alignment (weight-gradient GEMMs, K = 6 radial inputs,
//                  unaligned weight slices): synchronous staging.
#include "common.h"

typedef float v16f __attribute__((ext_vector_type(16)));

namespace {

constexpr int BK = 32;          // K-step of the generic kernel
constexpr int LDS_LD = BK + 4;

__device__ __forceinline__ float4 dact4(float4 v, float4 z) {
  v.x *= gn_dssilu(z.x); v.y *= gn_dssilu(z.y); v.z *= gn_dssilu(z.z); v.w *= gn_dssilu(z.w);
  return v;
}

// Stage a ROWS x 32 tile T[r][k] = op(G)[row0 + r][k0 + k] into LDS (zero-filled out of range).
//   trans == 0: op(G)[r][k] = G[r*ld + k]      trans == 1: op(G)[r][k] = G[k*ld + r]
// If Z != nullptr the element is multiplied by dssilu(Z[same index]).
template <int ROWS, int NT>
__device__ __forceinline__ void stage_tile(float (*Ts)[LDS_LD], const float* __restrict__ G,
                                           const float* __restrict__ Z, int trans, int vec,
                                           int row0, int k0, int nrows, int K, int ld, int tid) {
  if (!trans) {
    if (vec) {
      for (int f = tid; f < ROWS * (BK / 4); f += NT) {
        const int r = f >> 3;
        const int kv = (f & 7) << 2;
        const int gr = row0 + r;
        const int gk = k0 + kv;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (gr < nrows && gk < K) {
          const size_t off = (size_t)gr * ld + gk;
          if (gk + 3 < K) {
            v = *reinterpret_cast<const float4*>(G + off);
            if (Z) v = dact4(v, *reinterpret_cast<const float4*>(Z + off));
          } else {
            float t[4] = {0.f, 0.f, 0.f, 0.f};
            for (int j = 0; j < 4; ++j)
              if (gk + j < K) t[j] = G[off + j] * (Z ? gn_dssilu(Z[off + j]) : 1.0f);
            v = make_float4(t[0], t[1], t[2], t[3]);
          }
        }
        *reinterpret_cast<float4*>(&Ts[r][kv]) = v;
      }
    } else {
      for (int e = tid; e < ROWS * BK; e += NT) {
        const int r = e >> 5;
        const int k = e & 31;
        const int gr = row0 + r;
        const int gk = k0 + k;
        float v = 0.f;
        if (gr < nrows && gk < K) {
          const size_t off = (size_t)gr * ld + gk;
          v = G[off];
          if (Z) v *= gn_dssilu(Z[off]);
        }
        Ts[r][k] = v;
      }
    }
  } else {
    if (vec) {
      constexpr int RV = ROWS / 4;
      for (int f = tid; f < BK * RV; f += NT) {
        const int k = f / RV;
        const int rv = (f % RV) << 2;
        const int gk = k0 + k;
        const int gr = row0 + rv;
        float t[4] = {0.f, 0.f, 0.f, 0.f};
        if (gk < K && gr < nrows) {
          const size_t off = (size_t)gk * ld + gr;
          if (gr + 3 < nrows) {
            float4 v = *reinterpret_cast<const float4*>(G + off);
            if (Z) v = dact4(v, *reinterpret_cast<const float4*>(Z + off));
            t[0] = v.x; t[1] = v.y; t[2] = v.z; t[3] = v.w;
          } else {
            for (int j = 0; j < 4; ++j)
              if (gr + j < nrows) t[j] = G[off + j] * (Z ? gn_dssilu(Z[off + j]) : 1.0f);
          }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) Ts[rv + j][k] = t[j];
      }
    } else {
      for (int e = tid; e < ROWS * BK; e += NT) {
        const int k = e / ROWS;
        const int r = e % ROWS;
        const int gk = k0 + k;
        const int gr = row0 + r;
        float v = 0.f;
        if (gk < K && gr < nrows) {
          const size_t off = (size_t)gk * ld + gr;
          v = G[off];
          if (Z) v *= gn_dssilu(Z[off]);
        }
        Ts[r][k] = v;
      }
    }
  }
}

// 8 k-values of the staged tile: TM*TN*4 MFMAs
template <int TM, int TN, int LD>
__device__ __forceinline__ void mma_step(const float (*As)[LD], const float (*Bs)[LD],
                                         int arow, int brow, int kcol, v16f (&acc)[TM][TN]) {
  float4 a[TM], b[TN];
#pragma unroll
  for (int i = 0; i < TM; ++i) a[i] = *reinterpret_cast<const float4*>(&As[arow + i * 32][kcol]);
#pragma unroll
  for (int j = 0; j < TN; ++j) b[j] = *reinterpret_cast<const float4*>(&Bs[brow + j * 32][kcol]);
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].x, b[j].x, acc[i][j], 0, 0, 0);
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].y, b[j].y, acc[i][j], 0, 0, 0);
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].z, b[j].z, acc[i][j], 0, 0, 0);
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].w, b[j].w, acc[i][j], 0, 0, 0);
    }
}

// Epilogue.  C/D layout of the 32x32 MFMA: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
// Fused epilogue over the NV accumulator values a lane owns.  Stage-major: one uniform branch per stage and
// the NV values of a stage unrolled and independent, so their loads are all in flight together (the value-major
// form serialised ~8 branches and up to 5 dependent global loads per value).  FULL = tile entirely inside C.
template <int NV, bool FULL>
__device__ __forceinline__ void epilogue_vals(const gn_gemm_args& p, float (&v)[NV], const int (&row)[NV],
                                              const int (&col)[NV]) {
  const int M = p.M, N = p.N, ldc = p.ldc;
  const float* __restrict__ const gadd1 = p.gadd1;
  const float* __restrict__ const gadd2 = p.gadd2;
  float* __restrict__ const pre_out = p.pre_out;
  const float* __restrict__ const mul = p.mul;
  const float* __restrict__ const res = p.res;
  const float* __restrict__ const res2 = p.res2;
  float* __restrict__ const C = p.C;
  bool ok[NV];
#pragma unroll
  for (int n = 0; n < NV; ++n) ok[n] = FULL || (row[n] < M && col[n] < N);
#define GN_EV(body) _Pragma("unroll") for (int n = 0; n < NV; ++n) { if (FULL || ok[n]) { body } }
  if (gadd1) { const int32_t* __restrict__ const gi = p.gidx1; const int ldg = p.ldg;
               GN_EV(v[n] += gadd1[(size_t)gi[row[n]] * ldg + col[n]];) }
  if (gadd2) { const int32_t* __restrict__ const gi = p.gidx2; const int ldg = p.ldg;
               GN_EV(v[n] += gadd2[(size_t)gi[row[n]] * ldg + col[n]];) }
  if (pre_out) GN_EV(pre_out[(size_t)row[n] * ldc + col[n]] = v[n];)
  if (p.act) {
#pragma unroll
    for (int n = 0; n < NV; ++n) v[n] = gn_ssilu(v[n]);
  }
  if (mul) { const int ld = p.ldmul; GN_EV(v[n] *= mul[(size_t)row[n] * ld + col[n]];) }
  const float alpha = p.alpha;
  if (alpha != 1.0f) {
#pragma unroll
    for (int n = 0; n < NV; ++n) v[n] *= alpha;
  }
  if (res) {
    const int ld = p.ldres; const float beta = p.beta;
    const int32_t* __restrict__ const ridx = p.ridx;
    if (ridx) GN_EV(v[n] = (v[n] + res[(size_t)ridx[row[n]] * ld + col[n]]) * beta;)
    else GN_EV(v[n] = (v[n] + res[(size_t)row[n] * ld + col[n]]) * beta;)
  }
  if (res2) { const int ld = p.ldres2; const float beta2 = p.beta2;
              GN_EV(v[n] = (v[n] + res2[(size_t)row[n] * ld + col[n]]) * beta2;) }
  GN_EV(C[(size_t)row[n] * ldc + col[n]] = v[n];)
#undef GN_EV
}

template <int TM, int TN>
__device__ __forceinline__ void epilogue(const gn_gemm_args& p, v16f (&acc)[TM][TN], int rbase,
                                         int cbase, int lane) {
  const int l31 = lane & 31;
  const int rh = (lane >> 5) << 2;
  const bool full = rbase + TM * 32 <= p.M && cbase + TN * 32 <= p.N;
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      float v[16];
      int row[16], col[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        v[r] = acc[i][j][r];
        row[r] = rbase + i * 32 + (r & 3) + 8 * (r >> 2) + rh;
        col[r] = cbase + j * 32 + l31;
      }
      if (full) epilogue_vals<16, true>(p, v, row, col);
      else epilogue_vals<16, false>(p, v, row, col);
    }
}

template <int BM, int BN, int WR, int WC>
__global__ __launch_bounds__(WR * WC * 64) void gemm_generic(const gn_gemm_args p, const int vecA,
                                                             const int vecB) {
  constexpr int NT = WR * WC * 64;
  constexpr int TM = BM / (WR * 32);
  constexpr int TN = BN / (WC * 32);
  static_assert(TM >= 1 && TN >= 1, "tile too small");
  __shared__ __attribute__((aligned(16))) float As[BM][LDS_LD];
  __shared__ __attribute__((aligned(16))) float Bs[BN][LDS_LD];

  const int tid = threadIdx.x;
  const int wave = tid >> 6;
  const int lane = tid & 63;
  const int wr = wave / WC;
  const int wc = wave % WC;
  const int row0 = blockIdx.x * BM;
  const int col0 = blockIdx.y * BN;
  const int arow = wr * TM * 32 + (lane & 31);
  const int brow = wc * TN * 32 + (lane & 31);
  const int kh = (lane >> 5) << 2;  // 0 or 4

  v16f acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // split-K: blockIdx.z owns the K-range [kbeg, kend) (multiples of BK)
  int kbeg = 0, kend = p.K;
  if (p.splitk > 1) {
    const int steps = (p.K + BK - 1) / BK;
    const int per = (steps + p.splitk - 1) / p.splitk;
    kbeg = blockIdx.z * per * BK;
    kend = min(p.K, kbeg + per * BK);
  }
  for (int k0 = kbeg; k0 < kend; k0 += BK) {
    stage_tile<BM, NT>(As, p.A, p.a_dact_pre, p.trans_a, vecA, row0, k0, p.M, p.K, p.lda, tid);
    // Bs[n][k] = opB(B)[k][n]: trans_b == 0 means B is stored (N,K) = "row n, k contiguous"
    stage_tile<BN, NT>(Bs, p.B, nullptr, p.trans_b, vecB, col0, k0, p.N, p.K, p.ldb, tid);
    __syncthreads();
#pragma unroll
    for (int kb = 0; kb < BK; kb += 8) mma_step<TM, TN, LDS_LD>(As, Bs, arow, brow, kb + kh, acc);
    __syncthreads();
  }
  if (p.splitk > 1) {
    // raw partial sums -> workspace slice z (epilogue applied by splitk_reduce)
    float* __restrict__ ws = p.splitk_ws + (size_t)blockIdx.z * p.M * p.N;
    const int l31 = lane & 31, rh = (lane >> 5) << 2;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int col = col0 + (wc * TN + j) * 32 + l31;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = row0 + (wr * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + rh;
          if (row < p.M && col < p.N) ws[(size_t)row * p.N + col] = acc[i][j][r];
        }
      }
    return;
  }
  epilogue<TM, TN>(p, acc, row0 + wr * TM * 32, col0 + wc * TN * 32, lane);
}

__global__ void splitk_reduce(const float* __restrict__ ws, float* __restrict__ C, int M, int N, int ldc,
                              int splitk, float alpha) {
  const int64_t n = (int64_t)M * N;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    float acc = 0.f;
    for (int z = 0; z < splitk; ++z) acc += ws[(size_t)z * n + i];
    C[(i / N) * ldc + (i % N)] = acc * alpha;
  }
}

// NT operands, rows 16-B aligned, K % 4 == 0.  Register-prefetch pipeline over K-steps of KS.
template <int BM, int BN, int WR, int WC, int KS>
__global__ __launch_bounds__(WR * WC * 64) void gemm_nt_pipe(const gn_gemm_args p) {
  constexpr int NT = WR * WC * 64;
  constexpr int TM = BM / (WR * 32);
  constexpr int TN = BN / (WC * 32);
  constexpr int LD = KS + 4;
  constexpr int V = KS / 4;                    // float4 per tile row
  constexpr int NA = (BM * V + NT - 1) / NT;   // float4 loads per thread per K-step
  constexpr int NB = (BN * V + NT - 1) / NT;
  static_assert(TM >= 1 && TN >= 1, "tile too small");
  __shared__ __attribute__((aligned(16))) float As[BM][LD];
  __shared__ __attribute__((aligned(16))) float Bs[BN][LD];

  const int tid = threadIdx.x;
  const int wave = tid >> 6;
  const int lane = tid & 63;
  const int wr = wave / WC;
  const int wc = wave % WC;
  const int row0 = blockIdx.x * BM;
  const int col0 = blockIdx.y * BN;
  const int arow = wr * TM * 32 + (lane & 31);
  const int brow = wc * TN * 32 + (lane & 31);
  const int kh = (lane >> 5) << 2;
  const bool dact = p.a_dact_pre != nullptr;

  float4 ra[NA], rz[NA], rb[NB];
  auto load = [&](int k0) {
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int f = tid + i * NT;
      const int r = f / V, kv = (f % V) << 2;
      const int gr = row0 + r, gk = k0 + kv;
      ra[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      rz[i] = ra[i];
      if (f < BM * V && gr < p.M && gk < p.K) {
        const size_t off = (size_t)gr * p.lda + gk;
        ra[i] = *reinterpret_cast<const float4*>(p.A + off);
        if (dact) rz[i] = *reinterpret_cast<const float4*>(p.a_dact_pre + off);
      }
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int f = tid + i * NT;
      const int r = f / V, kv = (f % V) << 2;
      const int gr = col0 + r, gk = k0 + kv;
      rb[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (f < BN * V && gr < p.N && gk < p.K)
        rb[i] = *reinterpret_cast<const float4*>(p.B + (size_t)gr * p.ldb + gk);
    }
  };
  auto store = [&]() {
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int f = tid + i * NT;
      if (f < BM * V) *reinterpret_cast<float4*>(&As[f / V][(f % V) << 2]) = dact ? dact4(ra[i], rz[i]) : ra[i];
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int f = tid + i * NT;
      if (f < BN * V) *reinterpret_cast<float4*>(&Bs[f / V][(f % V) << 2]) = rb[i];
    }
  };

  v16f acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  load(0);
  for (int k0 = 0; k0 < p.K; k0 += KS) {
    store();
    __syncthreads();
    if (k0 + KS < p.K) load(k0 + KS);  // in flight while the matrix pipe works on this step
#pragma unroll
    for (int kb = 0; kb < KS; kb += 8) mma_step<TM, TN, LD>(As, Bs, arow, brow, kb + kh, acc);
    __syncthreads();
  }
  epilogue<TM, TN>(p, acc, row0 + wr * TM * 32, col0 + wc * TN * 32, lane);
}


typedef float v4f __attribute__((ext_vector_type(4)));

// Variant on v_mfma_f32_16x16x4_f32 with 8 waves per 32x128 block tile: wave (rg, cg) owns rows
// 16*rg..+16 and columns 32*cg..+32 (two 16x16 tiles, two independent accumulators for the 40-cycle
// dependent latency).  Same LDS image and register-prefetch pipeline as gemm_nt_pipe, but twice the
// waves per SIMD (4.4 instead of 2.2 at E = 18 k rows) so more of the load/barrier latency of one
// wave hides under the matrix work of another.  Epilogue: epilogue_vals().
// BKM = false: B is (N,K) (a Linear weight, k contiguous).  BKM = true: B is (K,N) (n contiguous; the
// input-gradient product dX = dY W of every Dense): the tile is staged k-major, Bs[k][n], so the global
// load and the LDS store stay float4 along n and a fragment is four conflict-free ds_read_b32
// (row stride 132 words: the four 16-lane groups, 4 rows apart, land 16 banks apart).
template <int KS, bool BKM>
__global__ __launch_bounds__(512) void gemm_nt_pipe16(const gn_gemm_args p) {
  constexpr int BM = 32, BN = 128, NT = 512;
  constexpr int LD = KS + 4;
  constexpr int LDB = BKM ? BN + 4 : KS + 4;
  constexpr int V = KS / 4;
  constexpr int NA = (BM * V + NT - 1) / NT;
  constexpr int NB = (BN * V + NT - 1) / NT;
  __shared__ __attribute__((aligned(16))) float As[BM][LD];
  __shared__ __attribute__((aligned(16))) float Bs[BKM ? KS : BN][LDB];
  const int tid = threadIdx.x;
  const int wave = tid >> 6;
  const int lane = tid & 63;
  const int rg = wave >> 2;   // 0..1
  const int cg = wave & 3;    // 0..3
  const int row0 = blockIdx.x * BM;
  const int col0 = blockIdx.y * BN;
  const int l15 = lane & 15;
  const int kq = (lane >> 4) << 2;   // 0,4,8,12: this lane group's 4 consecutive k of a 16-k chunk
  const bool dact = p.a_dact_pre != nullptr;

  float4 ra[NA], rz[NA], rb[NB];
  auto load = [&](int k0) {
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int f = tid + i * NT;
      const int r = f / V, kv = (f % V) << 2;
      const int gr = row0 + r, gk = k0 + kv;
      ra[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      rz[i] = ra[i];
      if (f < BM * V && gr < p.M && gk < p.K) {
        const size_t off = (size_t)gr * p.lda + gk;
        ra[i] = *reinterpret_cast<const float4*>(p.A + off);
        if (dact) rz[i] = *reinterpret_cast<const float4*>(p.a_dact_pre + off);
      }
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int f = tid + i * NT;
      rb[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (BKM) {
        const int k = f / (BN / 4), nv = (f % (BN / 4)) << 2;
        const int gk = k0 + k, gn = col0 + nv;
        if (f < BN * V && gk < p.K && gn < p.N)
          rb[i] = *reinterpret_cast<const float4*>(p.B + (size_t)gk * p.ldb + gn);
      } else {
        const int r = f / V, kv = (f % V) << 2;
        const int gr = col0 + r, gk = k0 + kv;
        if (f < BN * V && gr < p.N && gk < p.K)
          rb[i] = *reinterpret_cast<const float4*>(p.B + (size_t)gr * p.ldb + gk);
      }
    }
  };
  auto store = [&]() {
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int f = tid + i * NT;
      if (f < BM * V) *reinterpret_cast<float4*>(&As[f / V][(f % V) << 2]) = dact ? dact4(ra[i], rz[i]) : ra[i];
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int f = tid + i * NT;
      if (f < BN * V) {
        if (BKM) *reinterpret_cast<float4*>(&Bs[f / (BN / 4)][(f % (BN / 4)) << 2]) = rb[i];
        else *reinterpret_cast<float4*>(&Bs[f / V][(f % V) << 2]) = rb[i];
      }
    }
  };

  v4f acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
  const int arow = rg * 16 + l15;
  const int brow = cg * 32 + l15;
  load(0);
  for (int k0 = 0; k0 < p.K; k0 += KS) {
    store();
    __syncthreads();
    if (k0 + KS < p.K) load(k0 + KS);
#pragma unroll
    for (int kb = 0; kb < KS; kb += 16) {
      const float4 a = *reinterpret_cast<const float4*>(&As[arow][kb + kq]);
      float4 b0, b1;
      if (BKM) {
        b0 = make_float4(Bs[kb + kq][brow], Bs[kb + kq + 1][brow], Bs[kb + kq + 2][brow], Bs[kb + kq + 3][brow]);
        b1 = make_float4(Bs[kb + kq][brow + 16], Bs[kb + kq + 1][brow + 16], Bs[kb + kq + 2][brow + 16],
                         Bs[kb + kq + 3][brow + 16]);
      } else {
        b0 = *reinterpret_cast<const float4*>(&Bs[brow][kb + kq]);
        b1 = *reinterpret_cast<const float4*>(&Bs[brow + 16][kb + kq]);
      }
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b0.x, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b1.x, acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b0.y, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b1.y, acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b0.z, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b1.z, acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b0.w, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b1.w, acc1, 0, 0, 0);
    }
    __syncthreads();
  }
  // C/D layout of the 16x16 MFMA: col = lane&15, row = (lane>>4)*4 + reg
  const int rbase = row0 + rg * 16 + ((lane >> 4) << 2);
  const int cbase = col0 + cg * 32 + l15;
  float v[8];
  int row[8], col[8];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    v[r] = acc0[r]; row[r] = rbase + r; col[r] = cbase;
    v[4 + r] = acc1[r]; row[4 + r] = rbase + r; col[4 + r] = cbase + 16;
  }
  if (row0 + BM <= p.M && col0 + BN <= p.N) epilogue_vals<8, true>(p, v, row, col);
  else epilogue_vals<8, false>(p, v, row, col);
}

// C[b] = opA(A[b]) opB(B[b]) for tiny per-edge blocks (m*k, k*n <= 2048 floats).
__global__ __launch_bounds__(256) void bmm_f32_kernel(const float* __restrict__ A,
                                                      const float* __restrict__ B,
                                                      float* __restrict__ C, int m, int n, int k,
                                                      int trans_a, int trans_b) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;              // [m][k]
  float* Bs = smem + m * k;      // [k][n]
  const size_t b = blockIdx.x;
  const float* Ab = A + b * (size_t)m * k;
  const float* Bb = B + b * (size_t)k * n;
  float* Cb = C + b * (size_t)m * n;
  for (int e = threadIdx.x; e < m * k; e += blockDim.x) {
    // storage index e is coalesced; compute its logical (i,kk)
    int i, kk;
    if (trans_a) { kk = e / m; i = e % m; } else { i = e / k; kk = e % k; }
    As[i * k + kk] = Ab[e];
  }
  for (int e = threadIdx.x; e < k * n; e += blockDim.x) {
    int kk, j;
    if (trans_b) { j = e / k; kk = e % k; } else { kk = e / n; j = e % n; }
    Bs[kk * n + j] = Bb[e];
  }
  __syncthreads();
  for (int o = threadIdx.x; o < m * n; o += blockDim.x) {
    const int i = o / n;
    const int j = o % n;
    float acc = 0.f;
    for (int kk = 0; kk < k; ++kk) acc = fmaf(As[i * k + kk], Bs[kk * n + j], acc);
    Cb[o] = acc;
  }
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

template <int BM, int BN, int WR, int WC, int KS>
void launch(const gn_gemm_args& p, bool fast, int vecA, int vecB, hipStream_t st) {
  dim3 grid(gn_cdiv(p.M, BM), gn_cdiv(p.N, BN), p.splitk > 1 ? p.splitk : 1), block(WR * WC * 64);
  if (fast) hipLaunchKernelGGL((gemm_nt_pipe<BM, BN, WR, WC, KS>), grid, block, 0, st, p);
  else hipLaunchKernelGGL((gemm_generic<BM, BN, WR, WC>), grid, block, 0, st, p, vecA, vecB);
}

// ---- shapes the MFMA tiles do not fit -----------------------------------------------------------------------------
// K <= 64 that is tiny or not a multiple of 4 (the 6 radial basis functions -> 16 / 128, the 42-column circular basis,
// the K = 1 outer product of the energy head's input gradient): these are write-bound row operations, but the generic
// tile kernel stages both operands through LDS dword by dword (42 us for the (18 k, 6) x (6, 128) edge embedding
// against 9 MB of output).  Here the B tile (K x 128) sits in LDS, a thread owns one row and 4 columns 32 apart
// (row-contiguous 128-byte stores), the row of A is a broadcast load, and the usual fused epilogue applies.
template <bool NARROW>
__global__ __launch_bounds__(256) void gemm_smallk(const gn_gemm_args p) {
  // NARROW (N <= 32: the radial projections onto 16 columns — 9 of the 10 launches of a forward+force step): a thread owns
  // ONE column and four rows 8 apart, the B tile is (K, 32) — a quarter of the LDS reads and products of the wide form,
  // which computed 128 columns whatever N was (round 5).  Same order of products per output: bit-identical.
  constexpr int TN = NARROW ? 32 : 128;
  extern __shared__ float Bs[];   // [K][TN]
  const int K = p.K, N = p.N, M = p.M;
  const int tid = threadIdx.x;
  const int col0 = blockIdx.y * TN, row0 = blockIdx.x * 32;
  const float* __restrict__ const B = p.B;
  if (p.trans_b) {   // B is (K,N)
    for (int i = tid; i < K * TN; i += 256) {
      const int k = i / TN, n = i % TN;
      Bs[i] = col0 + n < N ? B[(size_t)k * p.ldb + col0 + n] : 0.f;
    }
  } else {           // B is (N,K): walk it in memory order
    for (int i = tid; i < K * TN; i += 256) {
      const int n = i / K, k = i - n * K;
      Bs[k * TN + n] = col0 + n < N ? B[(size_t)(col0 + n) * p.ldb + k] : 0.f;
    }
  }
  __syncthreads();
  const int l32 = tid & 31, rg = tid >> 5;
  if (NARROW) {
    const float* a[4];
    int row[4], col[4];
    float v[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int n = 0; n < 4; ++n) {
      row[n] = row0 + rg + 8 * n;
      col[n] = col0 + l32;
      a[n] = p.A + (size_t)(row[n] < M ? row[n] : M - 1) * p.lda;     // (rows behind M: loads clamped, stores masked)
    }
    for (int k = 0; k < K; ++k) {
      const float b = Bs[k * 32 + l32];
#pragma unroll
      for (int n = 0; n < 4; ++n) v[n] = fmaf(a[n][k], b, v[n]);
    }
    epilogue_vals<4, false>(p, v, row, col);
    return;
  }
  const bool full = col0 + 128 <= N;
  for (int rr = rg; rr < 32; rr += 8) {
    const int r = row0 + rr;
    if (r >= M) break;
    const float* __restrict__ a = p.A + (size_t)r * p.lda;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < K; ++k) {
      const float av = a[k];
      const float* b = Bs + k * 128 + l32;
#pragma unroll
      for (int n = 0; n < 4; ++n) v[n] = fmaf(av, b[32 * n], v[n]);
    }
    int row[4], col[4];
#pragma unroll
    for (int n = 0; n < 4; ++n) row[n] = r, col[n] = col0 + l32 + 32 * n;
    if (full) epilogue_vals<4, true>(p, v, row, col);
    else epilogue_vals<4, false>(p, v, row, col);
  }
}

// N = 1 (the energy / scalar heads): C[m] = (alpha * <A[m,:], b> + res[m]) * beta, one wave per row; the optional
// ungathered residual is the only epilogue stage (the running sum of the output blocks' energies).
__global__ __launch_bounds__(256) void gemm_n1(const float* __restrict__ A, int lda, const float* __restrict__ b,
                                               float* __restrict__ C, int ldc, int M, int K, float alpha, int vec,
                                               const float* __restrict__ res, int ldres, float beta) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= M) return;
  const float* __restrict__ a = A + (size_t)r * lda;
  float acc = 0.f;
  if (vec) {
    for (int k = 4 * lane; k < K; k += 256) {
      const float4 u = *reinterpret_cast<const float4*>(a + k), w = *reinterpret_cast<const float4*>(b + k);
      acc = fmaf(u.x, w.x, fmaf(u.y, w.y, fmaf(u.z, w.z, fmaf(u.w, w.w, acc))));
    }
  } else {
    for (int k = lane; k < K; k += 64) acc = fmaf(a[k], b[k], acc);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
  if (lane == 0) C[(size_t)r * ldc] = res ? (acc * alpha + res[(size_t)r * ldres]) * beta : acc * alpha;
}

}  // namespace

// cfg < 0: automatic tile selection; cfg >= 0: explicit variant (tuning / tests).
extern "C" int gn_gemm_f32_cfg(const gn_gemm_args* args, int cfg, void* stream) {
  const gn_gemm_args p = *args;
  if (p.M <= 0 || p.N <= 0) return 0;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int vecA = (p.lda % 4 == 0) && aligned16(p.A) && (!p.a_dact_pre || aligned16(p.a_dact_pre));
  const int vecB = (p.ldb % 4 == 0) && aligned16(p.B);
  const bool fast = !p.trans_a && !p.trans_b && vecA && vecB && (p.K % 4 == 0) && p.splitk <= 1;
  const bool plain = !p.a_dact_pre && !p.act && !p.pre_out && !p.mul && !p.res && !p.res2 && !p.gadd1 && !p.gadd2;
  const bool plain_res = !p.a_dact_pre && !p.act && !p.pre_out && !p.mul && !p.res2 && !p.gadd1 && !p.gadd2 && !p.ridx;
  if (cfg < 0 && !p.trans_a && p.splitk <= 1 && p.N == 1 && plain_res && (p.trans_b ? p.ldb == 1 : true)) {
    const int vec = vecA && aligned16(p.B) && (p.K % 4 == 0);
    hipLaunchKernelGGL(gemm_n1, dim3(gn_cdiv(p.M, 4)), dim3(256), 0, st, p.A, p.lda, p.B, p.C, p.ldc, p.M, p.K, p.alpha, vec,
                       p.res, p.ldres, p.beta);
    GN_LAUNCH_CHECK();
    return 0;
  }
  if (cfg < 0 && !p.trans_a && !p.a_dact_pre && p.splitk <= 1 && p.K >= 1 && p.K <= 64 && (p.K % 4 != 0 || p.K < 8)) {
#ifdef GN_SMALLK_WIDE_ONLY      // (A/B builds: the round-1..4 form for every N)
    if (false)
#else
    if (p.N <= 32)
#endif
      hipLaunchKernelGGL(gemm_smallk<true>, dim3(gn_cdiv(p.M, 32), 1), dim3(256), (size_t)p.K * 32 * sizeof(float), st, p);
    else
      hipLaunchKernelGGL(gemm_smallk<false>, dim3(gn_cdiv(p.M, 32), gn_cdiv(p.N, 128)), dim3(256),
                         (size_t)p.K * 128 * sizeof(float), st, p);
    GN_LAUNCH_CHECK();
    return 0;
  }
  // x @ B with B (K,N): k-major staging in the 8-wave kernel (any N; rows of B 16-byte aligned)
  if (cfg < 0 && !p.trans_a && p.trans_b && vecA && vecB && (p.K % 4 == 0) && (p.N % 4 == 0) && p.splitk <= 1 &&
      p.M >= 512) {
    dim3 grid(gn_cdiv(p.M, 32), gn_cdiv(p.N, 128));
    if (p.M > 4096) hipLaunchKernelGGL((gemm_nt_pipe16<32, true>), grid, dim3(512), 0, st, p);
    else hipLaunchKernelGGL((gemm_nt_pipe16<64, true>), grid, dim3(512), 0, st, p);
    GN_LAUNCH_CHECK();
    return 0;
  }
  if (p.splitk > 1) {
    if (!p.splitk_ws) return (int)hipErrorInvalidValue;
    if (p.N > 64) launch<32, 128, 1, 4, 32>(p, false, vecA, vecB, st);
    else if (p.N > 32) launch<32, 64, 1, 2, 32>(p, false, vecA, vecB, st);
    else launch<32, 32, 1, 1, 32>(p, false, vecA, vecB, st);
    GN_LAUNCH_CHECK();
    const int64_t n = (int64_t)p.M * p.N;
    hipLaunchKernelGGL(splitk_reduce, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, p.splitk_ws, p.C, p.M,
                       p.N, p.ldc, p.splitk, p.alpha);
    GN_LAUNCH_CHECK();
    return 0;
  }
  if (cfg < 0) {
    // Measured on MI355X (tools/gemm_bench.py, profiles/r1_gemm_tiles.txt): up to ~40 k rows the
    // launch is latency-bound, and 32-row tiles with one 32x32 tile per wave (2.2 waves/SIMD at
    // E = 18 k) beat 64x128 (12.7 vs 17.1 us at N = K = 128); large M wants 128x128 tiles.
    // The 8-wave 16x16x4 variant (cfg 14/15) doubles the waves per SIMD: 12.5 vs 12.8 us at E = 18 k,
    // 5.9 vs 7.3 us at M = 1024.
    const bool huge = p.M > 40000;
    if (p.N > 64) cfg = huge ? 13 : (p.M > 4096 ? 14 : 15);
    else if (p.N > 32) cfg = p.M > 4096 ? 11 : 3;
    else cfg = p.M > 4096 ? 4 : 5;
  }
  switch (cfg) {
    case 0: launch<64, 128, 2, 2, 32>(p, fast, vecA, vecB, st); break;
    case 1: launch<32, 128, 1, 4, 32>(p, fast, vecA, vecB, st); break;
    case 2: launch<64, 64, 2, 2, 32>(p, fast, vecA, vecB, st); break;
    case 3: launch<32, 64, 1, 2, 32>(p, fast, vecA, vecB, st); break;
    case 4: launch<128, 32, 4, 1, 32>(p, fast, vecA, vecB, st); break;
    case 5: launch<32, 32, 1, 1, 32>(p, fast, vecA, vecB, st); break;
    case 6: launch<64, 128, 2, 2, 64>(p, fast, vecA, vecB, st); break;
    case 7: launch<32, 128, 1, 4, 64>(p, fast, vecA, vecB, st); break;
    case 8: launch<64, 128, 2, 4, 32>(p, fast, vecA, vecB, st); break;
    case 9: launch<64, 128, 2, 4, 64>(p, fast, vecA, vecB, st); break;
    case 10: launch<32, 64, 1, 2, 64>(p, fast, vecA, vecB, st); break;
    case 11: launch<64, 64, 2, 2, 64>(p, fast, vecA, vecB, st); break;
    case 12: launch<128, 128, 4, 2, 32>(p, fast, vecA, vecB, st); break;
    case 13: launch<128, 128, 4, 4, 32>(p, fast, vecA, vecB, st); break;
    case 14:
    case 15:
      if (!fast) { launch<32, 128, 1, 4, 32>(p, fast, vecA, vecB, st); break; }
      if (cfg == 14) hipLaunchKernelGGL((gemm_nt_pipe16<32, false>), dim3(gn_cdiv(p.M, 32), gn_cdiv(p.N, 128)), dim3(512), 0, st, p);
      else hipLaunchKernelGGL((gemm_nt_pipe16<64, false>), dim3(gn_cdiv(p.M, 32), gn_cdiv(p.N, 128)), dim3(512), 0, st, p);
      break;
    default: return (int)hipErrorInvalidValue;
  }
  GN_LAUNCH_CHECK();
  return 0;
}

extern "C" int gn_gemm_f32(const gn_gemm_args* args, void* stream) { return gn_gemm_f32_cfg(args, -1, stream); }

extern "C" int gn_bmm_f32(const float* A, const float* B, float* C, int batch, int m, int n, int k,
                          int trans_a, int trans_b, void* stream) {
  if (batch <= 0 || m <= 0 || n <= 0) return 0;
  const size_t smem = (size_t)(m * k + k * n) * sizeof(float);
  if (smem > 64 * 1024) return (int)hipErrorInvalidValue;
  int threads = m * n;
  threads = threads > 256 ? 256 : ((threads + 63) / 64) * 64;
  hipLaunchKernelGGL(bmm_f32_kernel, dim3(batch), dim3(threads), smem,
                     static_cast<hipStream_t>(stream), A, B, C, m, n, k, trans_a, trans_b);
  GN_LAUNCH_CHECK();
  return 0;
}
