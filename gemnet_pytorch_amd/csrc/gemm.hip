// Dense contractions on the f32 MFMA (v_mfma_f32_32x32x2_f32): exact-f32 numerics at the
// f32 vector rate (157 TF peak on MI355X), fused prologue (activation derivative on the A
// operand) and fused epilogue (gather-add, ScaledSiLU, Hadamard, scale, residual).
//
// Replaces every `Dense` (nn.Linear + ScaledSiLU) of the reference
// (gemnet/model/layers/base_layers.py:5-58), ResidualLayer (:61-89), the concat-Dense of
// embedding_block.py:60-75 and the final (E, C*I) x (C*I, O) contraction of the bilinear
// layer (efficient.py:185-188).
//
// Tiling: 256 threads = 4 waves (wave64).  Block tile BM x BN, K-step 32 staged through LDS
// as As[BM][36] / Bs[BN][36] (k contiguous, +4 pad: rows 144 B apart -> ds_read_b128 of 16
// consecutive rows hits 16 distinct 16-B slots).  Each lane fetches one float4 (4 consecutive
// k) per 32-row fragment; lanes 0-31 take k = kb..kb+3, lanes 32-63 take k = kb+4..kb+7, and
// MFMA step s consumes component s of both operands — the k permutation is the same for A and
// B so the sum over k is unchanged.
#include "common.h"

typedef float v16f __attribute__((ext_vector_type(16)));

namespace {

constexpr int BK = 32;
constexpr int LDS_LD = BK + 4;

// Stage a ROWS x 32 tile T[r][k] = op(G)[row0 + r][k0 + k] into LDS (zero-filled out of range).
//   trans == 0: op(G)[r][k] = G[r*ld + k]      trans == 1: op(G)[r][k] = G[k*ld + r]
// If Z != nullptr the element is multiplied by dssilu(Z[same index]).
template <int ROWS>
__device__ __forceinline__ void stage_tile(float (*Ts)[LDS_LD], const float* __restrict__ G,
                                           const float* __restrict__ Z, int trans, int vec,
                                           int row0, int k0, int nrows, int K, int ld, int tid) {
  if (!trans) {
    if (vec) {
      for (int f = tid; f < ROWS * (BK / 4); f += 256) {
        const int r = f >> 3;
        const int kv = (f & 7) << 2;
        const int gr = row0 + r;
        const int gk = k0 + kv;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (gr < nrows && gk < K) {
          const size_t off = (size_t)gr * ld + gk;
          if (gk + 3 < K) {
            v = *reinterpret_cast<const float4*>(G + off);
            if (Z) {
              const float4 z = *reinterpret_cast<const float4*>(Z + off);
              v.x *= gn_dssilu(z.x); v.y *= gn_dssilu(z.y);
              v.z *= gn_dssilu(z.z); v.w *= gn_dssilu(z.w);
            }
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
      for (int e = tid; e < ROWS * BK; e += 256) {
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
      for (int f = tid; f < BK * RV; f += 256) {
        const int k = f / RV;
        const int rv = (f % RV) << 2;
        const int gk = k0 + k;
        const int gr = row0 + rv;
        float t[4] = {0.f, 0.f, 0.f, 0.f};
        if (gk < K && gr < nrows) {
          const size_t off = (size_t)gk * ld + gr;
          if (gr + 3 < nrows) {
            const float4 v = *reinterpret_cast<const float4*>(G + off);
            t[0] = v.x; t[1] = v.y; t[2] = v.z; t[3] = v.w;
            if (Z) {
              const float4 z = *reinterpret_cast<const float4*>(Z + off);
              t[0] *= gn_dssilu(z.x); t[1] *= gn_dssilu(z.y);
              t[2] *= gn_dssilu(z.z); t[3] *= gn_dssilu(z.w);
            }
          } else {
            for (int j = 0; j < 4; ++j)
              if (gr + j < nrows) t[j] = G[off + j] * (Z ? gn_dssilu(Z[off + j]) : 1.0f);
          }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) Ts[rv + j][k] = t[j];
      }
    } else {
      for (int e = tid; e < ROWS * BK; e += 256) {
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

template <int BM, int BN, int WR, int WC>
__global__ __launch_bounds__(256) void gemm_f32_kernel(const gn_gemm_args p, const int vecA,
                                                       const int vecB) {
  static_assert(WR * WC == 4, "4 waves per block");
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
  const int l31 = lane & 31;
  const int kh = (lane >> 5) << 2;  // 0 or 4

  v16f acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  for (int k0 = 0; k0 < p.K; k0 += BK) {
    stage_tile<BM>(As, p.A, p.a_dact_pre, p.trans_a, vecA, row0, k0, p.M, p.K, p.lda, tid);
    // Bs[n][k] = opB(B)[k][n]: trans_b == 0 means B is stored (N,K) = "row n, k contiguous"
    stage_tile<BN>(Bs, p.B, nullptr, p.trans_b, vecB, col0, k0, p.N, p.K, p.ldb, tid);
    __syncthreads();
#pragma unroll
    for (int kb = 0; kb < BK; kb += 8) {
      float4 a[TM], b[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i)
        a[i] = *reinterpret_cast<const float4*>(&As[(wr * TM + i) * 32 + l31][kb + kh]);
#pragma unroll
      for (int j = 0; j < TN; ++j)
        b[j] = *reinterpret_cast<const float4*>(&Bs[(wc * TN + j) * 32 + l31][kb + kh]);
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
    __syncthreads();
  }

  // epilogue.  C/D layout of the 32x32 MFMA: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
  const int rh = (lane >> 5) << 2;
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col = col0 + (wc * TN + j) * 32 + l31;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = row0 + (wr * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + rh;
        if (row < p.M && col < p.N) {
          float z = acc[i][j][r];
          if (p.gadd1) z += p.gadd1[(size_t)p.gidx1[row] * p.ldg + col];
          if (p.gadd2) z += p.gadd2[(size_t)p.gidx2[row] * p.ldg + col];
          const size_t co = (size_t)row * p.ldc + col;
          if (p.pre_out) p.pre_out[co] = z;
          float y = p.act ? gn_ssilu(z) : z;
          if (p.mul) y *= p.mul[(size_t)row * p.ldmul + col];
          y *= p.alpha;
          if (p.res) y = (y + p.res[(size_t)row * p.ldres + col]) * p.beta;
          p.C[co] = y;
        }
      }
    }
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

}  // namespace

extern "C" int gn_gemm_f32(const gn_gemm_args* args, void* stream) {
  const gn_gemm_args p = *args;
  if (p.M <= 0 || p.N <= 0) return 0;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int vecA = (p.lda % 4 == 0) && aligned16(p.A) && (!p.a_dact_pre || aligned16(p.a_dact_pre));
  const int vecB = (p.ldb % 4 == 0) && aligned16(p.B);
  if (p.N > 64) {
    dim3 grid(gn_cdiv(p.M, 64), gn_cdiv(p.N, 128));
    hipLaunchKernelGGL((gemm_f32_kernel<64, 128, 2, 2>), grid, dim3(256), 0, st, p, vecA, vecB);
  } else if (p.N > 32) {
    dim3 grid(gn_cdiv(p.M, 64), 1);
    hipLaunchKernelGGL((gemm_f32_kernel<64, 64, 2, 2>), grid, dim3(256), 0, st, p, vecA, vecB);
  } else {
    dim3 grid(gn_cdiv(p.M, 128), 1);
    hipLaunchKernelGGL((gemm_f32_kernel<128, 32, 4, 1>), grid, dim3(256), 0, st, p, vecA, vecB);
  }
  GN_LAUNCH_CHECK();
  return 0;
}

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
