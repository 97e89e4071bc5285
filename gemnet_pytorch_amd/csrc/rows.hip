// Row gather and deterministic CSR segmented sum (mutual adjoints).
//
// Replaces the ATen `index` gathers of the reference (interaction_block.py:543,548,562,678,693;
// embedding_block.py:70-71) and torch_scatter.scatter(reduce="add")
// (atom_update_block.py:67,172; gemnet.py:580) — the latter without atomics: rows are grouped
// by destination through a CSR (offsets + optional permutation) built once per batch.
#include "common.h"

namespace {

__global__ void gather_rows_v4(const float4* __restrict__ x, const int32_t* __restrict__ idx,
                               float4* __restrict__ y, int64_t T, int C4) {
  const int64_t n = T * C4;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t t = i / C4;
    const int c = (int)(i - t * C4);
    y[i] = x[(int64_t)idx[t] * C4 + c];
  }
}

__global__ void gather_rows_s(const float* __restrict__ x, const int32_t* __restrict__ idx,
                              float* __restrict__ y, int64_t T, int C) {
  const int64_t n = T * C;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t t = i / C;
    const int c = (int)(i - t * C);
    y[i] = x[(int64_t)idx[t] * C + c];
  }
}

__global__ void segsum_rows_v4(const float4* __restrict__ y, const int32_t* __restrict__ perm,
                               const int32_t* __restrict__ seg_off, float4* __restrict__ x,
                               int64_t N, int C4) {
  const int64_t n = N * C4;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / C4;
    const int c = (int)(i - r * C4);
    const int k0 = seg_off[r], k1 = seg_off[r + 1];
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int k = k0; k < k1; ++k) {
      const int64_t src = perm ? perm[k] : k;
      const float4 v = y[src * C4 + c];
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    x[i] = acc;
  }
}

__global__ void segsum_rows_s(const float* __restrict__ y, const int32_t* __restrict__ perm,
                              const int32_t* __restrict__ seg_off, float* __restrict__ x,
                              int64_t N, int C) {
  const int64_t n = N * C;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / C;
    const int c = (int)(i - r * C);
    const int k0 = seg_off[r], k1 = seg_off[r + 1];
    float acc = 0.f;
    for (int k = k0; k < k1; ++k) {
      const int64_t src = perm ? perm[k] : k;
      acc += y[src * C + c];
    }
    x[i] = acc;
  }
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
inline int grid_for(int64_t n) {
  int64_t b = (n + 255) / 256;
  return (int)(b < 1 ? 1 : (b > 8192 ? 8192 : b));
}

}  // namespace

extern "C" int gn_gather_rows_f32(const float* x, const int32_t* idx, float* y, int64_t T, int C,
                                  void* stream) {
  if (T <= 0 || C <= 0) return 0;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (C % 4 == 0 && aligned16(x) && aligned16(y)) {
    hipLaunchKernelGGL(gather_rows_v4, dim3(grid_for(T * (C / 4))), dim3(256), 0, st,
                       reinterpret_cast<const float4*>(x), idx, reinterpret_cast<float4*>(y), T, C / 4);
  } else {
    hipLaunchKernelGGL(gather_rows_s, dim3(grid_for(T * C)), dim3(256), 0, st, x, idx, y, T, C);
  }
  GN_LAUNCH_CHECK();
  return 0;
}

extern "C" int gn_segsum_rows_f32(const float* y, const int32_t* perm, const int32_t* seg_off,
                                  float* x, int64_t N, int C, void* stream) {
  if (N <= 0 || C <= 0) return 0;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (C % 4 == 0 && aligned16(x) && aligned16(y)) {
    hipLaunchKernelGGL(segsum_rows_v4, dim3(grid_for(N * (C / 4))), dim3(256), 0, st,
                       reinterpret_cast<const float4*>(y), perm, seg_off,
                       reinterpret_cast<float4*>(x), N, C / 4);
  } else {
    hipLaunchKernelGGL(segsum_rows_s, dim3(grid_for(N * C)), dim3(256), 0, st, y, perm, seg_off, x, N, C);
  }
  GN_LAUNCH_CHECK();
  return 0;
}
