// Twice-differentiable geometry of the force-TRAINING step (include/gemnet_hip.h, gn_dist_* / gn_angle_*).
//
// Reference: calculate_interatomic_vectors gemnet/model/gemnet.py:261-286 and calculate_neighbor_angles / calculate_angles3
// :288-311, :420-451 under `loss.backward()` through `autograd.grad(E, R, create_graph=True)` (trainer.py:338-346): in the
// reference (and on the composite closure here) the gathers, differences, cross products, norms, clamp and atan2 of the
// forward pass, of its first adjoint and of the second-order pass are ~200 pointwise launches over (E,3)/(T,3)
// temporaries per training step.  Here each of
//     distance   D[e]     = |R[a(e)] - R[c(e)]|
//     angle      theta[t] = atan2(max(|u x v|, 1e-9), u.v),  u = R[c] - R[a],  v = R[b] - R[a]
// is one kernel for the value, one for the first adjoint (per-edge / per-triplet position gradients that the
// deterministic CSR sums of rows.hip reduce onto atoms) and one for the tangent pass of the double backward:
//     *_jvp:   d out = J dR            (the gradient w.r.t. the incoming adjoint g)
//              H     = d/dR [J^T g] dR  (per-edge / per-triplet second-order position terms; optional)
// The second-order terms come from FORWARD-MODE DUAL NUMBERS through the very code of the first adjoint (`Dual`: value +
// directional derivative; the adjoint routine is a template over the scalar type), not from a hand-derived Hessian.
#include "common.h"

namespace {

struct Dual {
  float v, d;
};
__device__ __forceinline__ Dual mk(float v, float d = 0.f) { return {v, d}; }
__device__ __forceinline__ Dual operator+(Dual a, Dual b) { return {a.v + b.v, a.d + b.d}; }
__device__ __forceinline__ Dual operator-(Dual a, Dual b) { return {a.v - b.v, a.d - b.d}; }
__device__ __forceinline__ Dual operator-(Dual a) { return {-a.v, -a.d}; }
__device__ __forceinline__ Dual operator*(Dual a, Dual b) { return {a.v * b.v, a.d * b.v + a.v * b.d}; }
__device__ __forceinline__ Dual operator/(Dual a, Dual b) {
  const float q = a.v / b.v;
  return {q, (a.d - q * b.d) / b.v};
}
__device__ __forceinline__ Dual dsqrt(Dual a) {
  const float s = sqrtf(a.v);
  return {s, 0.5f * a.d / s};
}
__device__ __forceinline__ float dsqrt(float a) { return sqrtf(a); }
__device__ __forceinline__ float val(float a) { return a; }
__device__ __forceinline__ float val(Dual a) { return a.v; }
__device__ __forceinline__ void lift(float& o, float v) { o = v; }
__device__ __forceinline__ void lift(Dual& o, float v) { o = {v, 0.f}; }

template <class T>
struct V3 {
  T x, y, z;
};
template <class T>
__device__ __forceinline__ T dot(const V3<T>& a, const V3<T>& b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
template <class T>
__device__ __forceinline__ V3<T> cross(const V3<T>& a, const V3<T>& b) {
  return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}

// First adjoint of theta(u, v) = atan2(max(|u x v|, 1e-9), u.v):  gu = g dtheta/du, gv = g dtheta/dv  (the clamp has a
// zero gradient through y, gemnet.py:309: torch.max(y, 1e-9)).  Same formulas as trip_basis_bwd_kernel (geometry.hip).
template <class T>
__device__ __forceinline__ void angle_adjoint(const V3<T>& u, const V3<T>& v, const T g, V3<T>& gu, V3<T>& gv) {
  const T x = dot(u, v);
  const V3<T> w = cross(u, v);
  const T yn = dsqrt(dot(w, w));
  const bool clamped = val(yn) < 1e-9f;
  T y, zero;
  lift(zero, 0.f);
  lift(y, 1e-9f);
  if (!clamped) y = yn;
  const T r2 = x * x + y * y;
  const T dx = -(y / r2) * g;
  T dy = zero;
  V3<T> n = {zero, zero, zero};
  if (!clamped) {
    dy = (x / r2) * g;
    n = {w.x / y, w.y / y, w.z / y};
  }
  const V3<T> vn = cross(v, n), nu = cross(n, u);      // d|u x v|/du = v x n,  d|u x v|/dv = n x u
  gu = {dx * v.x + dy * vn.x, dx * v.y + dy * vn.y, dx * v.z + dy * vn.z};
  gv = {dx * u.x + dy * nu.x, dx * u.y + dy * nu.y, dx * u.z + dy * nu.z};
}

__global__ void dist_fwd_kernel(const float* __restrict__ R, const int32_t* __restrict__ id_c, const int32_t* __restrict__ id_a,
                                float* __restrict__ D, int64_t E) {
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < E; e += (int64_t)gridDim.x * blockDim.x) {
    const float* Ra = R + 3 * (int64_t)id_a[e];
    const float* Rc = R + 3 * (int64_t)id_c[e];
    const float vx = Ra[0] - Rc[0], vy = Ra[1] - Rc[1], vz = Ra[2] - Rc[2];
    D[e] = sqrtf(vx * vx + vy * vy + vz * vz);      // the reference's f32 arithmetic: sqrt(sum(V^2))
  }
}

// W[e] = gD[e] * (R[a] - R[c]) / D[e]:  dE/dR = segsum(W, id_a) - segsum(W, id_c)
__global__ void dist_bwd_kernel(const float* __restrict__ gD, const float* __restrict__ R, const int32_t* __restrict__ id_c,
                                const int32_t* __restrict__ id_a, float* __restrict__ W, int64_t E) {
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < E; e += (int64_t)gridDim.x * blockDim.x) {
    const float* Ra = R + 3 * (int64_t)id_a[e];
    const float* Rc = R + 3 * (int64_t)id_c[e];
    const float vx = Ra[0] - Rc[0], vy = Ra[1] - Rc[1], vz = Ra[2] - Rc[2];
    const float sc = gD[e] / sqrtf(vx * vx + vy * vy + vz * vz);
    W[3 * e] = sc * vx; W[3 * e + 1] = sc * vy; W[3 * e + 2] = sc * vz;
  }
}

// Ddot[e] = vhat . dv,  H[e] = gD[e] (dv - vhat (vhat . dv)) / D[e]   with dv = tR[a] - tR[c]   (either output may be null)
__global__ void dist_jvp_kernel(const float* __restrict__ R, const float* __restrict__ tR, const float* __restrict__ gD,
                                const int32_t* __restrict__ id_c, const int32_t* __restrict__ id_a, float* __restrict__ Ddot,
                                float* __restrict__ H, int64_t E) {
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < E; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t a = id_a[e], c = id_c[e];
    const float vx = R[3 * a] - R[3 * c], vy = R[3 * a + 1] - R[3 * c + 1], vz = R[3 * a + 2] - R[3 * c + 2];
    const float tx = tR[3 * a] - tR[3 * c], ty = tR[3 * a + 1] - tR[3 * c + 1], tz = tR[3 * a + 2] - tR[3 * c + 2];
    const float d = sqrtf(vx * vx + vy * vy + vz * vz), id = 1.0f / d;
    const float hx = vx * id, hy = vy * id, hz = vz * id;
    const float pr = hx * tx + hy * ty + hz * tz;
    if (Ddot) Ddot[e] = pr;
    if (H) {
      const float sc = gD[e] * id;
      H[3 * e] = sc * (tx - hx * pr); H[3 * e + 1] = sc * (ty - hy * pr); H[3 * e + 2] = sc * (tz - hz * pr);
    }
  }
}

__device__ __forceinline__ void load_uv(const float* __restrict__ R, int64_t c, int64_t a, int64_t b, V3<float>& u, V3<float>& v) {
  u = {R[3 * c] - R[3 * a], R[3 * c + 1] - R[3 * a + 1], R[3 * c + 2] - R[3 * a + 2]};
  v = {R[3 * b] - R[3 * a], R[3 * b + 1] - R[3 * a + 1], R[3 * b + 2] - R[3 * a + 2]};
}

__global__ void angle_fwd_kernel(const float* __restrict__ R, const int32_t* __restrict__ tc, const int32_t* __restrict__ ta,
                                 const int32_t* __restrict__ tb, float* __restrict__ theta, int64_t T) {
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < T; t += (int64_t)gridDim.x * blockDim.x) {
    V3<float> u, v;
    load_uv(R, tc[t], ta[t], tb[t], u, v);
    const V3<float> w = cross(u, v);
    const float yn = sqrtf(dot(w, w));
    theta[t] = atan2f(yn < 1e-9f ? 1e-9f : yn, dot(u, v));
  }
}

// Gc[t] = g dtheta/dR_c, Gb[t] = g dtheta/dR_b   (dtheta/dR_a = -(Gc + Gb))
__global__ void angle_bwd_kernel(const float* __restrict__ g, const float* __restrict__ R, const int32_t* __restrict__ tc,
                                 const int32_t* __restrict__ ta, const int32_t* __restrict__ tb, float* __restrict__ Gc,
                                 float* __restrict__ Gb, int64_t T) {
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < T; t += (int64_t)gridDim.x * blockDim.x) {
    V3<float> u, v, gu, gv;
    load_uv(R, tc[t], ta[t], tb[t], u, v);
    angle_adjoint<float>(u, v, g[t], gu, gv);
    Gc[3 * t] = gu.x; Gc[3 * t + 1] = gu.y; Gc[3 * t + 2] = gu.z;
    Gb[3 * t] = gv.x; Gb[3 * t + 1] = gv.y; Gb[3 * t + 2] = gv.z;
  }
}

// thdot[t] = dtheta/du . du + dtheta/dv . dv;   (Hc, Hb)[t] = d/d(u, v) [g (dtheta/du, dtheta/dv)] (du, dv)
// with du = tR[c] - tR[a], dv = tR[b] - tR[a]: the directional derivative of the first adjoint (dual numbers)
__global__ void angle_jvp_kernel(const float* __restrict__ R, const float* __restrict__ tR, const float* __restrict__ g,
                                 const int32_t* __restrict__ tc, const int32_t* __restrict__ ta, const int32_t* __restrict__ tb,
                                 float* __restrict__ thdot, float* __restrict__ Hc, float* __restrict__ Hb, int64_t T) {
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < T; t += (int64_t)gridDim.x * blockDim.x) {
    V3<float> u, v, du, dv;
    load_uv(R, tc[t], ta[t], tb[t], u, v);
    load_uv(tR, tc[t], ta[t], tb[t], du, dv);
    const V3<Dual> U = {mk(u.x, du.x), mk(u.y, du.y), mk(u.z, du.z)};
    const V3<Dual> Vv = {mk(v.x, dv.x), mk(v.y, dv.y), mk(v.z, dv.z)};
    V3<Dual> gu, gv;
    const float gt = g ? g[t] : 1.0f;
    angle_adjoint<Dual>(U, Vv, mk(gt), gu, gv);
    if (thdot) {
      // value parts are g dtheta/d(u, v): the tangent of theta itself is their contraction with (du, dv) at g = 1
      V3<float> g1u, g1v;
      angle_adjoint<float>(u, v, 1.0f, g1u, g1v);
      thdot[t] = dot(g1u, du) + dot(g1v, dv);
    }
    if (Hc) {
      Hc[3 * t] = gu.x.d; Hc[3 * t + 1] = gu.y.d; Hc[3 * t + 2] = gu.z.d;
      Hb[3 * t] = gv.x.d; Hb[3 * t + 1] = gv.y.d; Hb[3 * t + 2] = gv.z.d;
    }
  }
}

inline int grid_for2(int64_t n) {
  const int64_t b = (n + 255) / 256;
  return (int)(b < 1 ? 1 : (b > 65535 ? 65535 : b));
}

}  // namespace

extern "C" int gn_dist_fwd_f32(const float* R, const int32_t* id_c, const int32_t* id_a, float* D, int64_t E, void* stream) {
  if (E <= 0) return 0;
  hipLaunchKernelGGL(dist_fwd_kernel, dim3(grid_for2(E)), dim3(256), 0, static_cast<hipStream_t>(stream), R, id_c, id_a, D, E);
  GN_LAUNCH_CHECK();
  return 0;
}

extern "C" int gn_dist_bwd_f32(const float* gD, const float* R, const int32_t* id_c, const int32_t* id_a, float* W, int64_t E,
                               void* stream) {
  if (E <= 0) return 0;
  hipLaunchKernelGGL(dist_bwd_kernel, dim3(grid_for2(E)), dim3(256), 0, static_cast<hipStream_t>(stream), gD, R, id_c, id_a, W,
                     E);
  GN_LAUNCH_CHECK();
  return 0;
}

extern "C" int gn_dist_jvp_f32(const float* R, const float* tR, const float* gD, const int32_t* id_c, const int32_t* id_a,
                               float* Ddot, float* H, int64_t E, void* stream) {
  if (E <= 0) return 0;
  if (H && !gD) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(dist_jvp_kernel, dim3(grid_for2(E)), dim3(256), 0, static_cast<hipStream_t>(stream), R, tR, gD, id_c, id_a,
                     Ddot, H, E);
  GN_LAUNCH_CHECK();
  return 0;
}

extern "C" int gn_angle_fwd_f32(const float* R, const int32_t* tc, const int32_t* ta, const int32_t* tb, float* theta, int64_t T,
                                void* stream) {
  if (T <= 0) return 0;
  hipLaunchKernelGGL(angle_fwd_kernel, dim3(grid_for2(T)), dim3(256), 0, static_cast<hipStream_t>(stream), R, tc, ta, tb, theta,
                     T);
  GN_LAUNCH_CHECK();
  return 0;
}

extern "C" int gn_angle_bwd_f32(const float* g, const float* R, const int32_t* tc, const int32_t* ta, const int32_t* tb,
                                float* Gc, float* Gb, int64_t T, void* stream) {
  if (T <= 0) return 0;
  hipLaunchKernelGGL(angle_bwd_kernel, dim3(grid_for2(T)), dim3(256), 0, static_cast<hipStream_t>(stream), g, R, tc, ta, tb, Gc,
                     Gb, T);
  GN_LAUNCH_CHECK();
  return 0;
}

extern "C" int gn_angle_jvp_f32(const float* R, const float* tR, const float* g, const int32_t* tc, const int32_t* ta,
                                const int32_t* tb, float* thdot, float* Hc, float* Hb, int64_t T, void* stream) {
  if (T <= 0) return 0;
  if ((Hc == nullptr) != (Hb == nullptr) || (Hc && !g)) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(angle_jvp_kernel, dim3(grid_for2(T)), dim3(256), 0, static_cast<hipStream_t>(stream), R, tR, g, tc, ta, tb,
                     thdot, Hc, Hb, T);
  GN_LAUNCH_CHECK();
  return 0;
}
