// EXPERIMENT (not part of the product library): K1 + K2 + K3 of the bilinear layer in one launch for the spherical basis
// (S = 7, C = 64, I = 16, O = 64).  16 reduce edges per workgroup: 16 waves, each runs K1/K2 for one edge exactly like
// bil_reduce_project_mfma7_kernel but parks P[e] (16 x 64 = the 1024-long K3 row) in LDS instead of HBM; then the
// workgroup multiplies its 32 x 1024 tile with the (64 x 1024, k-contiguous) bilinear weight on the matrix cores.
// Built and timed by tools/k3fused_bench.py.
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float v4f __attribute__((ext_vector_type(4)));
constexpr int TE = 16, LDP = 1024 + 4;   // 16 edges per workgroup: 70 KB of LDS, two workgroups per CU

__global__ __launch_bounds__(1024) void bil_fused_fwd(const float* __restrict__ Y, const float* __restrict__ x,
                                                     const int32_t* __restrict__ expand_idx,
                                                     const int32_t* __restrict__ seg_off, const float* __restrict__ B,
                                                     const float* __restrict__ W2T, float* __restrict__ Sm,
                                                     float* __restrict__ out, int64_t E, float alpha) {
  constexpr int S = 7, C = 64, I = 16;
  extern __shared__ __attribute__((aligned(16))) float Pl[];   // [TE][LDP]
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int l15 = lane & 15, lg = lane >> 4;
  const int64_t e0 = (int64_t)blockIdx.x * TE;
  const bool srow = l15 < S;
  const int scl = min(l15, S - 1);
  for (int q = 0; q < 1; ++q) {
    const int row = wave;
    const int64_t e = e0 + row;
    float* __restrict__ prow = Pl + row * LDP;
    v4f pacc[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) pacc[nt] = (v4f){0.f, 0.f, 0.f, 0.f};
    if (e < E) {   // (wave-uniform)
      const int t0 = seg_off[e], t1 = seg_off[e + 1];
      const float* __restrict__ be = B + e * (int64_t)S * I;
      float bk[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float v = be[min(4 * lg + r, S - 1) * I + l15];
        bk[r] = (4 * lg + r) < S ? v : 0.f;
      }
      v4f acc[4];
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) acc[nt] = (v4f){0.f, 0.f, 0.f, 0.f};
      const int tlast = max(t1 - 1, t0);
      auto load = [&](int t, float& a, float (&b)[4]) {
        const int tq = t + lg;
        const bool ok = tq < t1;
        const int tc = min(tq, tlast);
        const float yv = Y[(int64_t)tc * S + scl];
        a = (ok && srow) ? yv : 0.f;
        const float* __restrict__ xr = x + (int64_t)expand_idx[tc] * C + l15;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
          const float v = xr[16 * nt];
          b[nt] = ok ? v : 0.f;
        }
      };
      float a0 = 0.f, b0[4] = {0.f, 0.f, 0.f, 0.f}, a1, b1[4];
      if (t0 < t1) load(t0, a0, b0);
      for (int t = t0; t < t1; t += 8) {
        load(t + 4, a1, b1);
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b0[nt], acc[nt], 0, 0, 0);
        load(t + 8, a0, b0);
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b1[nt], acc[nt], 0, 0, 0);
      }
      float* __restrict__ so = Sm + e * (int64_t)S * C;
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (4 * lg + r < S) so[(4 * lg + r) * C + 16 * nt + l15] = acc[nt][r];
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) pacc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(bk[r], acc[nt][r], pacc[nt], 0, 0, 0);
    }
    // P[e][i = 4 lg + r][c = 16 nt + l15] -> K3 row, k = i * 64 + c  (zeros for the rows past E)
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) prow[(4 * lg + r) * C + 16 * nt + l15] = pacc[nt][r];
  }
  __syncthreads();
  // K3: out[16 x 64] = Pl[16 x 1024] @ W2T^T; wave (kq, nt) owns one 16 x 16 tile over a quarter of K; the quarters meet
  // in LDS.  K-step (j, comp): lane group lg supplies k = 16 j + 4 lg + comp for both operands (float4 along k).
  const int kq = wave >> 2, nt = wave & 3;
  const float* __restrict__ wrow = W2T + (int64_t)(16 * nt + l15) * 1024 + 256 * kq + 4 * lg;
  const float* arow = Pl + l15 * LDP + 256 * kq + 4 * lg;
  float* red = Pl + TE * LDP;   // [3 kq][4 nt][64 lanes][4]
  v4f c0 = (v4f){0.f, 0.f, 0.f, 0.f}, c1 = (v4f){0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
  for (int j = 0; j < 16; j += 2) {
    const float4 w0 = *reinterpret_cast<const float4*>(wrow + 16 * j);
    const float4 w1 = *reinterpret_cast<const float4*>(wrow + 16 * j + 16);
    const float4 p0 = *reinterpret_cast<const float4*>(arow + 16 * j);
    const float4 p1 = *reinterpret_cast<const float4*>(arow + 16 * j + 16);
    c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(p0.x, w0.x, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(p1.x, w1.x, c1, 0, 0, 0);
    c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(p0.y, w0.y, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(p1.y, w1.y, c1, 0, 0, 0);
    c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(p0.z, w0.z, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(p1.z, w1.z, c1, 0, 0, 0);
    c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(p0.w, w0.w, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(p1.w, w1.w, c1, 0, 0, 0);
  }
  c0 += c1;
  if (kq > 0) *reinterpret_cast<v4f*>(red + (((kq - 1) * 4 + nt) * 64 + lane) * 4) = c0;
  __syncthreads();
  if (kq == 0) {
#pragma unroll
    for (int z = 0; z < 3; ++z) c0 += *reinterpret_cast<const v4f*>(red + ((z * 4 + nt) * 64 + lane) * 4);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int64_t e = e0 + 4 * lg + r;
      if (e < E) out[e * 64 + 16 * nt + l15] = alpha * c0[r];
    }
  }
}

extern "C" int k3fused_fwd(const float* Y, const float* x, const int32_t* expand_idx, const int32_t* seg_off,
                           const float* B, const float* W2T, float* Sm, float* out, int64_t E, float alpha, void* stream) {
  const size_t lds = ((size_t)TE * LDP + 3 * 4 * 64 * 4) * sizeof(float);
  static bool cfg = false;
  if (!cfg) {
    if (hipFuncSetAttribute((const void*)bil_fused_fwd, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return -2;
    cfg = true;
  }
  hipLaunchKernelGGL(bil_fused_fwd, dim3((unsigned)((E + TE - 1) / TE)), dim3(1024), lds, (hipStream_t)stream, Y, x, expand_idx,
                     seg_off, B, W2T, Sm, out, E, alpha);
  return (int)hipGetLastError();
}
