// The circular basis of the intermediate triplets of GemNet-Q projected in one pass (round 5).
//
// Reference: SphericalBasisLayer.forward with an index (basis_layers.py:119-131: rbf_env[id4_expand_intm_ab] * Y_l0(angle))
// followed by mlp_cbf4 = Dense(num_spherical * num_radial -> emb_size_cbf) (gemnet.py: cbf4 = self.mlp_cbf4(cbf4)):
//   out[i, n] = sum_{l, r} rad[e(i), l, r] * y[i, l] * W[n, l R + r]          i: intermediate triplet, e(i): its interaction edge
// The composite form gathered the (I, S R) rows, multiplied them by Y_l0 (two ATen launches over 94 MB at the 32 x 32 batch)
// and ran a K = 42 GEMM; its adjoint was a GEMM, two (I, S R) products, a reduction and a segmented sum (0.28 ms per
// forward+force step together).  Here: one kernel each way, nothing of size (I, S R) in memory.
//   adjoint:  t[i, k] = sum_n g[i, n] W[n, k];   g_y[i, l] = sum_r t[i, l R + r] rad[e(i), l, r];
//             g_rad[e, k] = sum_{i in seg(e)} t[i, k] y[i, l(k)]      (the rows of an interaction edge are contiguous: one wave
//             per edge walks them in order — deterministic, no atomics)
#include "common.h"

namespace {

constexpr int KMAX = 64, NMAX = 16;

// 16 lanes per row (one per output column), 16 rows per workgroup of 256
__global__ __launch_bounds__(256) void cbf_project_fwd_kernel(const float* __restrict__ rad, const int32_t* __restrict__ ie,
                                                              const float* __restrict__ y, const float* __restrict__ W,
                                                              float* __restrict__ out, int64_t I, int S, int R, int N) {
  __shared__ float Ws[KMAX * NMAX];      // [k][n]
  const int K = S * R;
  for (int i = threadIdx.x; i < K * NMAX; i += 256) {
    const int k = i / NMAX, n = i - k * NMAX;
    Ws[i] = n < N ? W[n * K + k] : 0.f;
  }
  __syncthreads();
  const int n = threadIdx.x & 15;
  const int64_t i = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4);
  const bool ok = i < I;
  const int64_t ii = ok ? i : I - 1;
  const float* __restrict__ re = rad + (int64_t)ie[ii] * K;
  const float* __restrict__ yi = y + ii * S;
  // lane n of the row's group holds the products p[k] for k = n, n + 16, n + 32, n + 48
  float p[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int k = n + 16 * j;
    p[j] = k < K ? re[k] * yi[k / R] : 0.f;
  }
  float acc = 0.f;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    if (16 * j >= K) break;
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const int k = 16 * j + s;
      const float pk = __shfl(p[j], s, 16);
      if (k < K) acc = fmaf(pk, Ws[k * NMAX + n], acc);
    }
  }
  if (ok && n < N) out[i * N + n] = acc;
}

// one wave per interaction edge; lane k < K owns column k of t
__global__ __launch_bounds__(256) void cbf_project_bwd_kernel(const float* __restrict__ g, const float* __restrict__ rad,
                                                              const int32_t* __restrict__ seg_off, const float* __restrict__ y,
                                                              const float* __restrict__ W, float* __restrict__ g_rad,
                                                              float* __restrict__ g_y, int64_t E, int S, int R, int N) {
  const int lane = threadIdx.x & 63;
  const int64_t e = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (e >= E) return;
  const int K = S * R;
  const bool mine = lane < K;
  const int k = mine ? lane : 0, l = k / R;
  float w[NMAX];
#pragma unroll
  for (int n = 0; n < NMAX; ++n) w[n] = (mine && n < N) ? W[n * K + k] : 0.f;
  const float rk = mine ? rad[e * K + k] : 0.f;
  float acc = 0.f;
  const int i0 = seg_off[e], i1 = seg_off[e + 1];
  for (int i = i0; i < i1; ++i) {
    const float* __restrict__ gi = g + (int64_t)i * N;
    float t = 0.f;
#pragma unroll
    for (int n = 0; n < NMAX; ++n)
      if (n < N) t = fmaf(gi[n], w[n], t);
    if (!mine) t = 0.f;
    acc = fmaf(t, y[(int64_t)i * S + l], acc);
    const float q = t * rk;
    // g_y[i, l] = sum_r q[l R + r]: lane l < S collects its R lanes in order
    float s = 0.f;
    for (int r = 0; r < R; ++r) s += __shfl(q, (lane < S ? lane : 0) * R + r, 64);
    if (lane < S) g_y[(int64_t)i * S + lane] = s;
  }
  if (mine) g_rad[e * K + k] = acc;
}

}  // namespace

extern "C" int gn_cbf_project_fwd_f32(const float* rad, const int32_t* ie, const float* y, const float* W, float* out, int64_t I,
                                      int S, int R, int N, void* stream) {
  if (I <= 0) return 0;
  if (S < 1 || R < 1 || S * R > KMAX || N < 1 || N > NMAX) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(cbf_project_fwd_kernel, dim3((unsigned)gn_cdiv(I, 16)), dim3(256), 0, static_cast<hipStream_t>(stream), rad, ie,
                     y, W, out, I, S, R, N);
  GN_LAUNCH_CHECK();
  return 0;
}

extern "C" int gn_cbf_project_bwd_f32(const float* g, const float* rad, const int32_t* seg_off, const float* y, const float* W,
                                      float* g_rad, float* g_y, int64_t E, int S, int R, int N, void* stream) {
  if (E <= 0) return 0;
  if (S < 1 || R < 1 || S * R > KMAX || N < 1 || N > NMAX) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(cbf_project_bwd_kernel, dim3((unsigned)gn_cdiv(E, 4)), dim3(256), 0, static_cast<hipStream_t>(stream), g, rad,
                     seg_off, y, W, g_rad, g_y, E, S, R, N);
  GN_LAUNCH_CHECK();
  return 0;
}
