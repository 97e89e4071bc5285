// Shared device helpers for the gfx950 (CDNA4, wave64) GemNet kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>

#include "../../include/gemnet_hip.h"

#define GN_WAVE 64

#define GN_LAUNCH_CHECK()                         \
  do {                                            \
    hipError_t e__ = hipGetLastError();           \
    if (e__ != hipSuccess) return (int)e__;       \
  } while (0)

static inline int gn_cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// ScaledSiLU and its derivatives (reference: gemnet/model/layers/base_layers.py:51-58,
// y = silu(x)/0.6).  s = sigmoid(x):
//   f   = x s / 0.6
//   f'  = s (1 + x (1-s)) / 0.6
//   f'' = s (1-s) (2 + x (1-2s)) / 0.6
//   f'''= s (1-s) (3 (1-2s) + x (1 - 6s + 6s^2)) / 0.6
#define GN_INV_06 1.6666666666666667f

// v_exp_f32 + v_rcp_f32 (about 1 ulp each; |rel err| of s below ~3e-7 for |x| < 30) instead of the
// ~40-instruction IEEE expf + division: the activation epilogue of the LDS-resident layer chain was
// VALU-bound (4.5 k cycles per op for 20 values per lane, traced with -DGN_CHAIN_TRACE).
// (`__frcp_rn` still expands to the IEEE div_scale / div_fmas / div_fixup sequence; the builtin is the bare
// v_rcp_f32.)
__device__ __forceinline__ float gn_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }

__device__ __forceinline__ float gn_ssilu(float x) { return x * gn_sigmoid(x) * GN_INV_06; }

__device__ __forceinline__ float gn_dssilu(float x) {
  float s = gn_sigmoid(x);
  return s * (1.0f + x * (1.0f - s)) * GN_INV_06;
}

// f(x) and f'(x) from one sigmoid (x and y may alias)
__device__ __forceinline__ void gn_ssilu_pair(float x, float& y, float& d) {
  const float s = gn_sigmoid(x), xs = x * s;
  d = (s + xs - xs * s) * GN_INV_06;
  y = xs * GN_INV_06;
}

__device__ __forceinline__ float gn_d2ssilu(float x) {
  float s = gn_sigmoid(x);
  return s * (1.0f - s) * (2.0f + x * (1.0f - 2.0f * s)) * GN_INV_06;
}

__device__ __forceinline__ float gn_d3ssilu(float x) {
  float s = gn_sigmoid(x);
  return s * (1.0f - s) * (3.0f * (1.0f - 2.0f * s) + x * (1.0f - 6.0f * s + 6.0f * s * s)) * GN_INV_06;
}
