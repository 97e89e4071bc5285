// C (M,N) = alpha * A^T B for two ROW-contracted operands A (K,M), B (K,N): the weight-gradient shape
// dW = dY^T X of every Dense (gemnet/model/layers/base_layers.py:5-48) and the W-adjoints of the
// double backward.  K is the number of edges / atoms (1 k .. 10^5), M, N <= a few hundred, so the
// product has a handful of output tiles and a very long contraction.
//
// Layout decisions (MI355X):
//  * both operands are m-/n-contiguous in HBM; v_mfma_f32_32x32x2_f32 wants lane -> (row = lane & 31,
//    k = lane >> 5), i.e. 32 consecutive m for one k.  Staging the tile k-major in LDS (Ts[k][m]) makes the
//    global load (float4 along m), the LDS store (float4) and the LDS read (ds_read_b32, 32 consecutive
//    words per half-wave) all conflict-free with no transposition anywhere.
//  * split-K over blockIdx.z so that ~3 blocks per CU are resident (the contraction is the only
//    parallelism there is); raw 64x64 partial tiles go to a workspace and a second, wide kernel
//    (16 z-groups x 64 columns per block) folds them deterministically — no float atomics.
//  * next K-step is prefetched into registers before the MFMA loop, LDS is double buffered: one barrier
//    per 16-row step.
#include "common.h"

typedef float v16f __attribute__((ext_vector_type(16)));

namespace {

constexpr int TBM = 64, TBN = 64, TKS = 16, TNT = 256;

template <bool VEC>
__device__ __forceinline__ void tn_load(const float* __restrict__ G, int ld, int k0, int kend, int c0, int ncols,
                                        int tid, float (&r)[4]) {
  if (VEC) {
    const int k = k0 + (tid >> 4);
    const int c = c0 + ((tid & 15) << 2);
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (k < kend && c < ncols) v = *reinterpret_cast<const float4*>(G + (size_t)k * ld + c);
    r[0] = v.x; r[1] = v.y; r[2] = v.z; r[3] = v.w;
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int e = tid + j * TNT;
      const int k = k0 + (e >> 6);
      const int c = c0 + (e & 63);
      r[j] = (k < kend && c < ncols) ? G[(size_t)k * ld + c] : 0.f;
    }
  }
}

template <bool VEC>
__device__ __forceinline__ void tn_store(float (*Ts)[TBM], int tid, const float (&r)[4]) {
  if (VEC) {
    *reinterpret_cast<float4*>(&Ts[tid >> 4][(tid & 15) << 2]) = make_float4(r[0], r[1], r[2], r[3]);
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int e = tid + j * TNT;
      Ts[e >> 6][e & 63] = r[j];
    }
  }
}

// One 64x64 tile of out = alpha * A^T B over rows [kbeg, kend) of the operands (one workgroup).
template <bool VA, bool VB>
__device__ __forceinline__ void tn_tile(const float* __restrict__ A, const float* __restrict__ B,
                                        float* __restrict__ o, int M, int N, int lda, int ldb, int ldo, int row0,
                                        int col0, int kbeg, int kend, float alpha, float (*As)[TKS][TBM],
                                        float (*Bs)[TKS][TBN]) {
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int wr = wave >> 1, wc = wave & 1;
  const int l31 = lane & 31, kh = lane >> 5;
  const int am = wr * 32 + l31, bn = wc * 32 + l31;

  v16f acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  float ra[4], rb[4];
  tn_load<VA>(A, lda, kbeg, kend, row0, M, tid, ra);
  tn_load<VB>(B, ldb, kbeg, kend, col0, N, tid, rb);
  tn_store<VA>(As[0], tid, ra);
  tn_store<VB>(Bs[0], tid, rb);
  __syncthreads();
  int buf = 0;
  for (int k0 = kbeg; k0 < kend; k0 += TKS) {
    const bool more = k0 + TKS < kend;
    if (more) {
      tn_load<VA>(A, lda, k0 + TKS, kend, row0, M, tid, ra);
      tn_load<VB>(B, ldb, k0 + TKS, kend, col0, N, tid, rb);
    }
#pragma unroll
    for (int kp = 0; kp < TKS; kp += 2)
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(As[buf][kp + kh][am], Bs[buf][kp + kh][bn], acc, 0, 0, 0);
    if (more) {
      tn_store<VA>(As[buf ^ 1], tid, ra);
      tn_store<VB>(Bs[buf ^ 1], tid, rb);
    }
    __syncthreads();
    buf ^= 1;
  }
  // acc[r]: row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), col = lane & 31 of the wave's 32x32 tile
  const int col = col0 + wc * 32 + l31;
  if (col < N) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = row0 + wr * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
      if (row < M) o[(size_t)row * ldo + col] = acc[r] * alpha;
    }
  }
}

template <bool VA, bool VB>
__global__ __launch_bounds__(TNT) void gemm_tn_kernel(const float* __restrict__ A, const float* __restrict__ B,
                                                      float* __restrict__ out, int M, int N, int K, int lda,
                                                      int ldb, int ldo, int kchunk, float alpha) {
  __shared__ __attribute__((aligned(16))) float As[2][TKS][TBM];
  __shared__ __attribute__((aligned(16))) float Bs[2][TKS][TBN];
  const int kbeg = blockIdx.z * kchunk;
  float* __restrict__ o = out + (gridDim.z > 1 ? (size_t)blockIdx.z * M * ldo : 0);
  tn_tile<VA, VB>(A, B, o, M, N, lda, ldb, ldo, blockIdx.x * TBM, blockIdx.y * TBN, kbeg, min(K, kbeg + kchunk), alpha,
                  As, Bs);
}

// Grouped form: every workgroup looks up its problem (binary search over the prefix of workgroup counts) — all
// weight-gradient products of one training step in ONE launch (training/wgrad_queue.py).  Partials go to
// ws + ws_off as [z][M][N].
__global__ __launch_bounds__(TNT) void gemm_tn_grouped_kernel(const gn_tn_problem* __restrict__ probs, int n_prob,
                                                              float* __restrict__ ws) {
  __shared__ __attribute__((aligned(16))) float As[2][TKS][TBM];
  __shared__ __attribute__((aligned(16))) float Bs[2][TKS][TBN];
  const int wg = blockIdx.x;
  int lo = 0, hi = n_prob - 1;
  while (lo < hi) {   // last problem with wg_begin <= wg
    const int mid = (lo + hi + 1) >> 1;
    if (probs[mid].wg_begin <= wg) lo = mid; else hi = mid - 1;
  }
  const gn_tn_problem p = probs[lo];
  int local = wg - p.wg_begin;
  const int tiles_n = (p.N + TBN - 1) / TBN, tiles_m = (p.M + TBM - 1) / TBM;
  const int z = local / (tiles_m * tiles_n);
  local -= z * tiles_m * tiles_n;
  const int tm = local / tiles_n, tn = local - tm * tiles_n;
  const int kbeg = z * p.kchunk;
  const int kend = min(p.K, kbeg + p.kchunk);
  float* __restrict__ o = ws + p.ws_off + (size_t)z * p.M * p.N;
  const bool va = (p.ldx % 4 == 0) && (p.M % 4 == 0) && ((reinterpret_cast<uintptr_t>(p.X) & 15) == 0);
  const bool vb = (p.ldy % 4 == 0) && (p.N % 4 == 0) && ((reinterpret_cast<uintptr_t>(p.Y) & 15) == 0);
  if (va && vb) tn_tile<true, true>(p.X, p.Y, o, p.M, p.N, p.ldx, p.ldy, p.N, tm * TBM, tn * TBN, kbeg, kend, p.alpha, As, Bs);
  else tn_tile<false, false>(p.X, p.Y, o, p.M, p.N, p.ldx, p.ldy, p.N, tm * TBM, tn * TBN, kbeg, kend, p.alpha, As, Bs);
}

// Grouped fold: target t owns elements [0, n_t) and a list of partial slices (offsets into ws); one workgroup
// folds 64 consecutive elements over all its slices in list order and ADDS the sum to the target (param.grad).
__global__ __launch_bounds__(1024) void tn_fold_grouped_kernel(const gn_tn_target* __restrict__ targets, int n_target,
                                                               const int64_t* __restrict__ slice_off,
                                                               const float* __restrict__ ws) {
  __shared__ float part[16][64];
  const int wg = blockIdx.x;
  int lo = 0, hi = n_target - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (targets[mid].wg_begin <= wg) lo = mid; else hi = mid - 1;
  }
  const gn_tn_target t = targets[lo];
  const int lane = threadIdx.x & 63, g = threadIdx.x >> 6;
  const int64_t i = (int64_t)(wg - t.wg_begin) * 64 + lane;
  float s = 0.f;
  if (i < t.n)
    for (int k = t.slice_begin + g; k < t.slice_end; k += 16) s += ws[slice_off[k] + i];
  part[g][lane] = s;
  __syncthreads();
  if (g == 0 && i < t.n) {
    float acc = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) acc += part[j][lane];
    // a column slice of a wider parameter (the three blocks of a concat-Dense weight): rows of `cols` elements, pitch `ld`
    const int64_t o = t.cols > 0 ? (i / t.cols) * (int64_t)t.ld + (i % t.cols) : i;
    t.out[o] += acc;
  }
}

// ws (Z, n) -> C: 1024 threads = 16 z-groups x 64 consecutive elements; each thread strides over z.
__global__ __launch_bounds__(1024) void tn_fold_kernel(const float* __restrict__ ws, float* __restrict__ C,
                                                       int64_t n, int N, int ldc, int Z, float alpha) {
  __shared__ float part[16][64];
  const int lane = threadIdx.x & 63, g = threadIdx.x >> 6;
  const int64_t i = (int64_t)blockIdx.x * 64 + lane;
  float s0 = 0.f, s1 = 0.f;
  if (i < n) {
    int z = g;
    for (; z + 16 < Z; z += 32) {
      s0 += ws[(size_t)z * n + i];
      s1 += ws[(size_t)(z + 16) * n + i];
    }
    if (z < Z) s0 += ws[(size_t)z * n + i];
  }
  part[g][lane] = s0 + s1;
  __syncthreads();
  if (g == 0 && i < n) {
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) s += part[j][lane];
    C[(i / N) * ldc + (i % N)] = s * alpha;
  }
}

inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

// See include/gemnet_hip.h.  `ws` must hold splitk * M * N floats when splitk > 1.
extern "C" int gn_gemm_tn_f32(const float* A, const float* B, float* C, int M, int N, int K, int lda, int ldb,
                              int ldc, float alpha, float* ws, int splitk, void* stream) {
  if (M <= 0 || N <= 0) return 0;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (K <= 0) {
    return (int)hipMemset2DAsync(C, (size_t)ldc * sizeof(float), 0, (size_t)N * sizeof(float), M, st);
  }
  if (splitk < 1) splitk = 1;
  int kchunk = gn_cdiv(gn_cdiv(K, splitk), TKS) * TKS;
  splitk = gn_cdiv(K, kchunk);  // no empty slices
  if (splitk > 1 && !ws) return (int)hipErrorInvalidValue;
  const bool va = (lda % 4 == 0) && (M % 4 == 0) && al16(A);
  const bool vb = (ldb % 4 == 0) && (N % 4 == 0) && al16(B);
  dim3 grid(gn_cdiv(M, TBM), gn_cdiv(N, TBN), splitk);
  float* out = splitk > 1 ? ws : C;
  const int ldo = splitk > 1 ? N : ldc;
  const float a1 = splitk > 1 ? 1.0f : alpha;
  if (va && vb) hipLaunchKernelGGL((gemm_tn_kernel<true, true>), grid, dim3(TNT), 0, st, A, B, out, M, N, K, lda, ldb, ldo, kchunk, a1);
  else if (va) hipLaunchKernelGGL((gemm_tn_kernel<true, false>), grid, dim3(TNT), 0, st, A, B, out, M, N, K, lda, ldb, ldo, kchunk, a1);
  else if (vb) hipLaunchKernelGGL((gemm_tn_kernel<false, true>), grid, dim3(TNT), 0, st, A, B, out, M, N, K, lda, ldb, ldo, kchunk, a1);
  else hipLaunchKernelGGL((gemm_tn_kernel<false, false>), grid, dim3(TNT), 0, st, A, B, out, M, N, K, lda, ldb, ldo, kchunk, a1);
  GN_LAUNCH_CHECK();
  if (splitk > 1) {
    const int64_t n = (int64_t)M * N;
    hipLaunchKernelGGL(tn_fold_kernel, dim3((unsigned)((n + 63) / 64)), dim3(1024), 0, st, ws, C, n, N, ldc, splitk,
                       alpha);
    GN_LAUNCH_CHECK();
  }
  return 0;
}

extern "C" int gn_gemm_tn_grouped_f32(const gn_tn_problem* probs, int n_prob, int total_wg, const gn_tn_target* targets,
                                      int n_target, int total_fold_wg, const int64_t* slice_off, float* ws,
                                      void* stream) {
  if (n_prob <= 0 || total_wg <= 0) return 0;
  hipStream_t st = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(gemm_tn_grouped_kernel, dim3(total_wg), dim3(TNT), 0, st, probs, n_prob, ws);
  GN_LAUNCH_CHECK();
  if (n_target > 0 && total_fold_wg > 0) {
    hipLaunchKernelGGL(tn_fold_grouped_kernel, dim3(total_fold_wg), dim3(1024), 0, st, targets, n_target, slice_off, ws);
    GN_LAUNCH_CHECK();
  }
  return 0;
}

// Number of K-slices the host should provision a workspace for (pure function of the shape).
extern "C" int gn_gemm_tn_splitk(int M, int N, int K) {
  const int tiles = gn_cdiv(M, TBM) * gn_cdiv(N, TBN);
  int s = gn_cdiv(768, tiles);                 // ~3 blocks per CU
  const int smax = gn_cdiv(K, 2 * TKS);        // at least two K-steps per slice
  if (s > smax) s = smax;
  if (s < 1) s = 1;
  if (s > 512) s = 512;
  return s;
}
