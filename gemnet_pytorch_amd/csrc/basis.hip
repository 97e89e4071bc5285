// Basis functions of GemNet in closed form, with the analytic derivatives force training needs.
//
// The reference builds these formulas symbolically with sympy at constructor time (17-59 s) and
// evaluates them as ~50 lambdified Python closures, one ATen launch per elementary op
// (gemnet/model/layers/basis_layers.py:45-49,119-131,239-270; basis_utils.py:47-80,174-253).
// Here one kernel per basis family evaluates value or d/d^2 derivative directly; arithmetic is
// done in f64 in-kernel (E*42 resp. T*7 values: negligible cost) so the f32 results are
// correctly rounded even where the reference's expanded sympy form cancels catastrophically
// (small z*d/c, SURVEY.md Appendix A "numerical-stability finding").
#include "common.h"
#include "basis_math.h"

namespace {

__global__ void bessel_rbf_kernel(const float* __restrict__ dist, const float* __restrict__ freq,
                                  float* __restrict__ out, int64_t E, int R, double cutoff, int p,
                                  int kd, int kf) {
  const int64_t n = E * R;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t e = i / R;
    const int r = (int)(i - e * R);
    out[i] = (float)bessel_rbf_eval((double)dist[e], (double)freq[r], cutoff, p, kd, kf);
  }
}

__global__ void sph_radial_kernel(const float* __restrict__ dist, const float* __restrict__ z,
                                  const double* __restrict__ nrm, float* __restrict__ out,
                                  int64_t E, int S, int R, double cutoff, int p, int kd) {
  const int SR = S * R;
  const int64_t n = E * SR;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t e = i / SR;
    const int lr = (int)(i - e * SR);
    out[i] = (float)sph_radial_eval((double)dist[e], (double)z[lr], nrm[lr], lr / R, cutoff, p, kd);
  }
}

__global__ void ylm0_kernel(const float* __restrict__ theta, float* __restrict__ out, int64_t T,
                            int S, int k) {
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < T;
       t += (int64_t)gridDim.x * blockDim.x)
    ylm0_row((double)theta[t], S, k, out + t * S);
}

constexpr int YL_MAX = 7;
__global__ void ylm_kernel(const float* __restrict__ theta, const float* __restrict__ phi,
                           float* __restrict__ out, int64_t Q, int S, int kt, int kp) {
  for (int64_t q = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; q < Q;
       q += (int64_t)gridDim.x * blockDim.x)
    ylm_row((double)theta[q], (double)phi[q], S, kt, kp, out + q * (int64_t)S * S);
}

__global__ void ssilu_kernel(const float* __restrict__ x, float* __restrict__ out, int64_t n, int k) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    const float v = x[i];
    out[i] = k == 0 ? gn_ssilu(v) : (k == 1 ? gn_dssilu(v) : (k == 2 ? gn_d2ssilu(v) : gn_d3ssilu(v)));
  }
}

__device__ __forceinline__ void dact_mul_one(float g, float z, int act, float m, float c, bool want_gmul, float& dz,
                                             float& gm) {
  const float gv = g * c;
  const float a = act ? gn_dssilu(z) : 1.0f;
  dz = gv * m * a;
  if (want_gmul) gm = gv * (act ? gn_ssilu(z) : z);
}

// dz = c g f'(z) mul,  gmul = c g f(z)   (z, mul, gmul optional); float4 body + scalar tail when everything is aligned
__global__ void dact_mul_kernel(const float* __restrict__ g, const float* __restrict__ z, int act,
                                const float* __restrict__ mul, float c, float* __restrict__ dz,
                                float* __restrict__ gmul, int64_t n, int vec) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int64_t i0 = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  const int64_t n4 = vec ? (n >> 2) : 0;
  const float4 one = make_float4(1.f, 1.f, 1.f, 1.f), zero = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int64_t i = i0; i < n4; i += stride) {
    const float4 gv = reinterpret_cast<const float4*>(g)[i];
    const float4 zv = z ? reinterpret_cast<const float4*>(z)[i] : zero;
    const float4 mv = mul ? reinterpret_cast<const float4*>(mul)[i] : one;
    float4 d, m = zero;
    dact_mul_one(gv.x, zv.x, act, mv.x, c, gmul, d.x, m.x);
    dact_mul_one(gv.y, zv.y, act, mv.y, c, gmul, d.y, m.y);
    dact_mul_one(gv.z, zv.z, act, mv.z, c, gmul, d.z, m.z);
    dact_mul_one(gv.w, zv.w, act, mv.w, c, gmul, d.w, m.w);
    reinterpret_cast<float4*>(dz)[i] = d;
    if (gmul) reinterpret_cast<float4*>(gmul)[i] = m;
  }
  for (int64_t i = (n4 << 2) + i0; i < n; i += stride) {
    float d, m = 0.f;
    dact_mul_one(g[i], z ? z[i] : 0.f, act, mul ? mul[i] : 1.0f, c, gmul, d, m);
    dz[i] = d;
    if (gmul) gmul[i] = m;
  }
}

// out = c * f^(k)(z) * a * b * d  (k = -1: no activation factor; a, b, d optional), float4 body + scalar tail
__device__ __forceinline__ float pm_act(float v, int k) {
  return k == 0 ? gn_ssilu(v) : (k == 1 ? gn_dssilu(v) : (k == 2 ? gn_d2ssilu(v) : gn_d3ssilu(v)));
}
__global__ void pm_kernel(const float* __restrict__ z, int k, const float* __restrict__ a,
                          const float* __restrict__ b, const float* __restrict__ d, float c,
                          float* __restrict__ out, int64_t n, int vec) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int64_t i0 = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (vec) {
    const int64_t n4 = n >> 2;
    for (int64_t i = i0; i < n4; i += stride) {
      float4 r = make_float4(c, c, c, c);
      if (k >= 0) {
        const float4 v = reinterpret_cast<const float4*>(z)[i];
        r.x *= pm_act(v.x, k); r.y *= pm_act(v.y, k); r.z *= pm_act(v.z, k); r.w *= pm_act(v.w, k);
      }
      if (a) { const float4 v = reinterpret_cast<const float4*>(a)[i]; r.x *= v.x; r.y *= v.y; r.z *= v.z; r.w *= v.w; }
      if (b) { const float4 v = reinterpret_cast<const float4*>(b)[i]; r.x *= v.x; r.y *= v.y; r.z *= v.z; r.w *= v.w; }
      if (d) { const float4 v = reinterpret_cast<const float4*>(d)[i]; r.x *= v.x; r.y *= v.y; r.z *= v.z; r.w *= v.w; }
      reinterpret_cast<float4*>(out)[i] = r;
    }
    for (int64_t i = (n4 << 2) + i0; i < n; i += stride) {
      float r = c;
      if (k >= 0) r *= pm_act(z[i], k);
      if (a) r *= a[i];
      if (b) r *= b[i];
      if (d) r *= d[i];
      out[i] = r;
    }
  } else {
    for (int64_t i = i0; i < n; i += stride) {
      float r = c;
      if (k >= 0) r *= pm_act(z[i], k);
      if (a) r *= a[i];
      if (b) r *= b[i];
      if (d) r *= d[i];
      out[i] = r;
    }
  }
}

inline int grid_for(int64_t n) {
  int64_t b = (n + 255) / 256;
  return (int)(b < 1 ? 1 : (b > 8192 ? 8192 : b));
}

}  // namespace

extern "C" int gn_bessel_rbf_f32(const float* d, const float* freq, float* out, int64_t E, int R,
                                 float cutoff, int p, int kd, int kf, void* stream) {
  if (E <= 0) return 0;
  if (kd < 0 || kf < 0 || kf > 1 || kd + kf > 2 || p < 2) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(bessel_rbf_kernel, dim3(grid_for(E * R)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), d, freq, out, E, R, (double)cutoff, p, kd, kf);
  GN_LAUNCH_CHECK();
  return 0;
}

extern "C" int gn_sph_radial_f32(const float* d, const float* z, const double* nrm, float* out,
                                 int64_t E, int S, int R, float cutoff, int p, int kd, void* stream) {
  if (E <= 0) return 0;
  if (kd < 0 || kd > 2 || p < 2 || S > YL_MAX + 1) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(sph_radial_kernel, dim3(grid_for(E * S * R)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), d, z, nrm, out, E, S, R, (double)cutoff, p, kd);
  GN_LAUNCH_CHECK();
  return 0;
}

extern "C" int gn_ylm0_f32(const float* theta, float* out, int64_t T, int S, int k, void* stream) {
  if (T <= 0) return 0;
  if (k < 0 || k > 2) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(ylm0_kernel, dim3(grid_for(T)), dim3(256), 0, static_cast<hipStream_t>(stream),
                     theta, out, T, S, k);
  GN_LAUNCH_CHECK();
  return 0;
}

extern "C" int gn_ylm_f32(const float* theta, const float* phi, float* out, int64_t Q, int S, int kt,
                          int kp, void* stream) {
  if (Q <= 0) return 0;
  if (kt < 0 || kp < 0 || kt + kp > 2) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(ylm_kernel, dim3(grid_for(Q)), dim3(256), 0, static_cast<hipStream_t>(stream),
                     theta, phi, out, Q, S, kt, kp);
  GN_LAUNCH_CHECK();
  return 0;
}

extern "C" int gn_ssilu_f32(const float* x, float* out, int64_t n, int k, void* stream) {
  if (n <= 0) return 0;
  if (k < 0 || k > 3) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(ssilu_kernel, dim3(grid_for(n)), dim3(256), 0, static_cast<hipStream_t>(stream),
                     x, out, n, k);
  GN_LAUNCH_CHECK();
  return 0;
}

extern "C" int gn_dact_mul_f32(const float* g, const float* z, int act, const float* mul, float c,
                               float* dz, float* gmul, int64_t n, void* stream) {
  if (n <= 0) return 0;
  if ((act || gmul) && !z) return (int)hipErrorInvalidValue;
  auto al = [](const void* p) { return !p || (reinterpret_cast<uintptr_t>(p) & 15u) == 0; };
  const int vec = al(g) && al(z) && al(mul) && al(dz) && al(gmul);
  hipLaunchKernelGGL(dact_mul_kernel, dim3(grid_for(vec ? (n + 3) / 4 : n)), dim3(256), 0, static_cast<hipStream_t>(stream),
                     g, z, act, mul, c, dz, gmul, n, vec);
  GN_LAUNCH_CHECK();
  return 0;
}

extern "C" int gn_pm_f32(const float* z, int k, const float* a, const float* b, const float* d, float c, float* out,
                         int64_t n, void* stream) {
  if (n <= 0) return 0;
  if (k < -1 || k > 3 || (k >= 0 && !z)) return (int)hipErrorInvalidValue;
  auto al = [](const void* p) { return !p || (reinterpret_cast<uintptr_t>(p) & 15u) == 0; };
  const int vec = al(z) && al(a) && al(b) && al(d) && al(out);
  hipLaunchKernelGGL(pm_kernel, dim3(grid_for(vec ? (n + 3) / 4 : n)), dim3(256), 0, static_cast<hipStream_t>(stream),
                     z, k, a, b, d, c, out, n, vec);
  GN_LAUNCH_CHECK();
  return 0;
}

extern "C" int gn_abi_version(void) { return 15; }
extern "C" const char* gn_error_string(int code) { return hipGetErrorString((hipError_t)code); }
