// EXPERIMENT (not part of the product library).  Checked with the last GPU seconds of round 1 by
// tools/k3fused_bwd_bench.py: gB and dSm bit-identical to gn_gemm_f32 + gn_bil_project_bwd_f32(dY = NULL), 65 us vs
// 74 us on the bench batch.  Integration (C ABI entry, ops._FusedBilinear.backward) is left for the next round.
// Adjoint of the bilinear tail in one launch for the spherical basis (S = 7, C = 64, I = 16, O = 64), Y gradient
// deferred (gn_bil_dy_multi_f32):
//   dP[e, k] = alpha * sum_o g[e,o] W2[k,o]      16-edge tile x 1024, K = 64: stays in LDS (never the 74 MB in HBM)
//   gB[e]    = Sm[e] dP[e]^T,   dSm[e] = B[e] dP[e]      one wave per edge, as bil_project_bwd_mfma7_kernel
// W2 = the bilinear weight as (I*C, O), o contiguous.
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float v4f __attribute__((ext_vector_type(4)));
constexpr int TE = 16, LDP = 1024 + 4;

__global__ __launch_bounds__(1024) void bil_fused_bwd(const float* __restrict__ g, const float* __restrict__ W2,
                                                      const float* __restrict__ Sm, const float* __restrict__ B,
                                                      float* __restrict__ gB, float* __restrict__ dSm, int64_t E,
                                                      float alpha) {
  constexpr int S = 7, C = 64, I = 16;
  extern __shared__ __attribute__((aligned(16))) float dPl[];   // [TE][LDP]
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int l15 = lane & 15, lg = lane >> 4;
  const int64_t e0 = (int64_t)blockIdx.x * TE;
  auto comp = [](const float4& v, int c) { return c == 0 ? v.x : (c == 1 ? v.y : (c == 2 ? v.z : v.w)); };
  // ---- phase 1: dP tile = g tile (16 x 64) @ W2^T (64 x 1024).  Wave w owns the 64 columns k = 64 w .. 64 w + 63
  // (= i = w, all c) as four 16 x 16 tiles; K-step (j, comp): lane group lg supplies o = 16 j + 4 lg + comp.
  {
    const int64_t er = min(e0 + l15, E - 1);   // rows past E: duplicates, never used
    float4 ga[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) ga[j] = *reinterpret_cast<const float4*>(g + er * 64 + 16 * j + 4 * lg);
    v4f acc[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) acc[nt] = (v4f){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      const float* __restrict__ wr = W2 + (int64_t)(64 * wave + 16 * nt + l15) * 64 + 4 * lg;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float4 wb = *reinterpret_cast<const float4*>(wr + 16 * j);
#pragma unroll
        for (int cp = 0; cp < 4; ++cp)
          acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(comp(ga[j], cp), comp(wb, cp), acc[nt], 0, 0, 0);
      }
    }
    // D: row = edge 4 lg + r, col = k = 64 w + 16 nt + l15
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) dPl[(4 * lg + r) * LDP + 64 * wave + 16 * nt + l15] = alpha * acc[nt][r];
  }
  __syncthreads();
  // ---- phase 2: one wave per edge, dP[e] (16 x 64, k = i * 64 + c) read from the LDS tile
  const int64_t e = e0 + wave;
  if (e >= E) return;
  const float* dPe = dPl + wave * LDP;
  const float* __restrict__ Sme = Sm + e * (int64_t)S * C;
  const float* __restrict__ Be = B + e * (int64_t)S * I;
  const int scl = min(l15, S - 1);
  float4 smf[4], dpf[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    smf[j] = *reinterpret_cast<const float4*>(Sme + scl * C + 16 * j + 4 * lg);
    dpf[j] = *reinterpret_cast<const float4*>(dPe + l15 * C + 16 * j + 4 * lg);
  }
  const float4 bf = *reinterpret_cast<const float4*>(Be + scl * I + 4 * lg);
  float dpk[4][4];   // dP[i = 4 lg + comp][c = 16 nt + l15]
#pragma unroll
  for (int cp = 0; cp < 4; ++cp)
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) dpk[cp][nt] = dPe[(4 * lg + cp) * C + 16 * nt + l15];
  v4f gb = (v4f){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int cp = 0; cp < 4; ++cp) gb = __builtin_amdgcn_mfma_f32_16x16x4f32(comp(smf[j], cp), comp(dpf[j], cp), gb, 0, 0, 0);
  float* __restrict__ gbo = gB + e * (int64_t)S * I;
#pragma unroll
  for (int r = 0; r < 4; ++r)
    if (4 * lg + r < S) gbo[(4 * lg + r) * I + l15] = gb[r];
  v4f d[4];
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) d[nt] = (v4f){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int cp = 0; cp < 4; ++cp)
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) d[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(comp(bf, cp), dpk[cp][nt], d[nt], 0, 0, 0);
  float* __restrict__ dso = dSm + e * (int64_t)S * C;
#pragma unroll
  for (int nt = 0; nt < 4; ++nt)
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (4 * lg + r < S) dso[(4 * lg + r) * C + 16 * nt + l15] = d[nt][r];
}

extern "C" int k3fused_bwd(const float* g, const float* W2, const float* Sm, const float* B, float* gB, float* dSm,
                           int64_t E, float alpha, void* stream) {
  const size_t lds = (size_t)TE * LDP * sizeof(float);
  static bool cfg = false;
  if (!cfg) {
    if (hipFuncSetAttribute((const void*)bil_fused_bwd, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return -2;
    cfg = true;
  }
  hipLaunchKernelGGL(bil_fused_bwd, dim3((unsigned)((E + TE - 1) / TE)), dim3(1024), lds, (hipStream_t)stream, g, W2, Sm, B, gB,
                     dSm, E, alpha);
  return (int)hipGetLastError();
}
