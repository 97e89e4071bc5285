// Geometry fused with the basis evaluation (first-order path).
//
// Reference: calculate_interatomic_vectors gemnet/model/gemnet.py:261-286, calculate_neighbor_angles
// :288-311, calculate_angles3 :420-451 feeding BesselBasisLayer / SphericalBasisLayer
// (layers/basis_layers.py:45-49,119-131) — ~60 pointwise ATen launches over (E,3)/(T,3) temporaries
// in the reference.  Here: one kernel per edge (distance -> Bessel rbf + spherical-Bessel radial
// basis), one per triplet (angle -> Y_l0), and their adjoints, which recompute the geometry from R
// instead of reading saved temporaries and emit per-edge / per-triplet position gradients that the
// deterministic CSR segmented sum (rows.hip) reduces onto atoms.
#include "common.h"
#include "basis_math.h"

namespace {

// D[e] = |R[a]-R[c]|; rbf[e,n]; rad[e,l,n].  16 lanes per edge (round 5; was one work item per basis value: 49 threads
// per edge each fetching the edge's two atoms again): lane `sub` evaluates the functions sub, sub + 16, sub + 32 of the
// NR + S NR (= 48) and lane 0 writes the distance — same values as before.
__global__ void edge_basis_fwd_kernel(const float* __restrict__ R, const int32_t* __restrict__ id_c,
                                      const int32_t* __restrict__ id_a, const float* __restrict__ freq,
                                      const float* __restrict__ z, const double* __restrict__ nrm,
                                      float* __restrict__ D, float* __restrict__ V, float* __restrict__ rbf,
                                      float* __restrict__ rad, int64_t E, int NR, int S, double cutoff, int p) {
  const int sub = threadIdx.x & 15;
  const int nfun = NR + S * NR;
  for (int64_t e = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 4; e < E; e += ((int64_t)gridDim.x * blockDim.x) >> 4) {
    const float* Ra = R + 3 * (int64_t)id_a[e];
    const float* Rc = R + 3 * (int64_t)id_c[e];
    // same f32 arithmetic as the reference: V = Rt - Rs; D = sqrt(sum(V^2))
    const float vx = Ra[0] - Rc[0], vy = Ra[1] - Rc[1], vz = Ra[2] - Rc[2];
    const float d = sqrtf(vx * vx + vy * vy + vz * vz);
    if (sub == 0) {
      D[e] = d;
      if (V) { V[3 * e] = vx / d; V[3 * e + 1] = vy / d; V[3 * e + 2] = vz / d; }
    }
    for (int j = sub; j < nfun; j += 16) {
      if (j < NR) {
        rbf[e * NR + j] = (float)bessel_rbf_eval((double)d, (double)freq[j], cutoff, p, 0, 0);
      } else {
        const int lr = j - NR;
        rad[e * S * NR + lr] = (float)sph_radial_eval((double)d, (double)z[lr], nrm[lr], lr / NR, cutoff, p, 0);
      }
    }
  }
}

// W[e,:] = gD[e] * V[e,:] with gD = g_D + sum_n g_rbf f'_n(d) + sum_{l,n} g_rad R'_ln(d).
// 16 lanes per edge share the NR + S*NR (= 48) f64 derivative evaluations and combine them with
// xor-shuffles inside their 16-lane group (fixed order: deterministic).
__global__ void edge_basis_bwd_kernel(const float* __restrict__ g_D, const float* __restrict__ g_rbf,
                                      const float* __restrict__ g_rad, const float* __restrict__ R,
                                      const int32_t* __restrict__ id_c, const int32_t* __restrict__ id_a,
                                      const float* __restrict__ freq, const float* __restrict__ z,
                                      const double* __restrict__ nrm, float* __restrict__ Wout, int64_t E,
                                      int NR, int S, double cutoff, int p) {
  const int sub = threadIdx.x & 15;
  const int64_t e = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 4;
  const bool ok = e < E;
  float vx = 0.f, vy = 0.f, vz = 0.f, d = 1.f;
  double g = 0.0;
  if (ok) {
    const float* Ra = R + 3 * (int64_t)id_a[e];
    const float* Rc = R + 3 * (int64_t)id_c[e];
    vx = Ra[0] - Rc[0]; vy = Ra[1] - Rc[1]; vz = Ra[2] - Rc[2];
    d = sqrtf(vx * vx + vy * vy + vz * vz);
    const int nfun = NR + S * NR;
    for (int j = sub; j < nfun; j += 16) {
      if (j < NR) {
        if (g_rbf) g += (double)g_rbf[e * NR + j] * bessel_rbf_eval((double)d, (double)freq[j], cutoff, p, 1, 0);
      } else if (g_rad) {
        const int lr = j - NR;
        g += (double)g_rad[e * S * NR + lr] * sph_radial_eval((double)d, (double)z[lr], nrm[lr], lr / NR, cutoff, p, 1);
      }
    }
  }
#pragma unroll
  for (int m = 8; m >= 1; m >>= 1) g += __shfl_xor(g, m, 16);
  if (ok && sub == 0) {
    if (g_D) g += (double)g_D[e];
    const float sc = (float)(g / (double)d);
    Wout[3 * e] = sc * vx; Wout[3 * e + 1] = sc * vy; Wout[3 * e + 2] = sc * vz;
  }
}

struct Ang { float ux, uy, uz, vx, vy, vz, wx, wy, wz, x, y; bool clamped; };

__device__ __forceinline__ Ang angle_of(const float* __restrict__ R, int c, int a, int b) {
  Ang g;
  const float* Ra = R + 3 * (int64_t)a;
  const float* Rc = R + 3 * (int64_t)c;
  const float* Rb = R + 3 * (int64_t)b;
  g.ux = Rc[0] - Ra[0]; g.uy = Rc[1] - Ra[1]; g.uz = Rc[2] - Ra[2];
  g.vx = Rb[0] - Ra[0]; g.vy = Rb[1] - Ra[1]; g.vz = Rb[2] - Ra[2];
  g.x = g.ux * g.vx + g.uy * g.vy + g.uz * g.vz;
  g.wx = g.uy * g.vz - g.uz * g.vy;
  g.wy = g.uz * g.vx - g.ux * g.vz;
  g.wz = g.ux * g.vy - g.uy * g.vx;
  const float yn = sqrtf(g.wx * g.wx + g.wy * g.wy + g.wz * g.wz);
  g.clamped = yn < 1e-9f;          // torch.max(y, 1e-9): gradient through y vanishes (gemnet.py:309)
  g.y = g.clamped ? 1e-9f : yn;
  return g;
}

// Y[t,l] = Y_l0(angle(c<-a->b)); optionally theta[t]
__global__ void trip_basis_fwd_kernel(const float* __restrict__ R, const int32_t* __restrict__ tc,
                                      const int32_t* __restrict__ ta, const int32_t* __restrict__ tb,
                                      float* __restrict__ Y, float* __restrict__ theta, int64_t T, int S) {
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < T;
       t += (int64_t)gridDim.x * blockDim.x) {
    const Ang g = angle_of(R, tc[t], ta[t], tb[t]);
    const float th = atan2f(g.y, g.x);
    if (theta) theta[t] = th;
    ylm0_row((double)th, S, 0, Y + t * S);
  }
}

// Gc[t,:] = dE/dR_c, Gb[t,:] = dE/dR_b of triplet t given gY[t,:]  (dE/dR_a = -(Gc+Gb))
__global__ void trip_basis_bwd_kernel(const float* __restrict__ gY, const float* __restrict__ R,
                                      const int32_t* __restrict__ tc, const int32_t* __restrict__ ta,
                                      const int32_t* __restrict__ tb, float* __restrict__ Gc,
                                      float* __restrict__ Gb, int64_t T, int S) {
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < T;
       t += (int64_t)gridDim.x * blockDim.x) {
    const Ang g = angle_of(R, tc[t], ta[t], tb[t]);
    const float th = atan2f(g.y, g.x);
    float dY[8];
    ylm0_row((double)th, S, 1, dY);
    float gth = 0.f;
    for (int l = 0; l < S; ++l) gth += gY[t * S + l] * dY[l];
    const float r2 = g.x * g.x + g.y * g.y;
    const float dx = -g.y / r2 * gth;                    // dtheta/dx * g
    const float dy = g.clamped ? 0.f : g.x / r2 * gth;   // dtheta/dy * g
    // y = |w|, w = u x v, n = w/|w|: dy/du = v x n, dy/dv = n x u
    const float iy = g.clamped ? 0.f : 1.0f / g.y;
    const float nx = g.wx * iy, ny = g.wy * iy, nz = g.wz * iy;
    Gc[3 * t] = dx * g.vx + dy * (g.vy * nz - g.vz * ny);
    Gc[3 * t + 1] = dx * g.vy + dy * (g.vz * nx - g.vx * nz);
    Gc[3 * t + 2] = dx * g.vz + dy * (g.vx * ny - g.vy * nx);
    Gb[3 * t] = dx * g.ux + dy * (ny * g.uz - nz * g.uy);
    Gb[3 * t + 1] = dx * g.uy + dy * (nz * g.ux - nx * g.uz);
    Gb[3 * t + 2] = dx * g.uz + dy * (nx * g.uy - ny * g.ux);
  }
}

// ---- quadruplets c -> a - b <- d (gemnet.py:334-418) fused with the real Y_lm (basis_layers.py:269) --
struct V3 { float x, y, z; };
__device__ __forceinline__ V3 v3(const float* p) { return {p[0], p[1], p[2]}; }
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ V3 operator*(float s, V3 a) { return {s * a.x, s * a.y, s * a.z}; }
__device__ __forceinline__ float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }

// theta = atan2(max(|u x v|, 1e-9), u.v) and, for a given dL/dtheta, dL/du and dL/dv
__device__ __forceinline__ float angle_uv(V3 u, V3 v) {
  const V3 w = cross(u, v);
  const float yn = sqrtf(dot(w, w));
  return atan2f(yn < 1e-9f ? 1e-9f : yn, dot(u, v));
}
// (sin, cos) of that angle without atan2 / sincos: with y = max(|u x v|, 1e-9), x = u.v: sin = y / hypot(x, y),
// cos = x / hypot(x, y)  (exactly the sine and cosine of atan2(y, x))
__device__ __forceinline__ void angle_uv_sc(V3 u, V3 v, double& sn, double& cs) {
  const V3 w = cross(u, v);
  const float yn = sqrtf(dot(w, w));
  const double y = yn < 1e-9f ? 1e-9 : (double)yn;
  const double x = (double)dot(u, v);
  const double ir = 1.0 / sqrt(x * x + y * y);
  sn = y * ir;
  cs = x * ir;
}
__device__ __forceinline__ void angle_uv_bwd(V3 u, V3 v, float gth, V3& gu, V3& gv) {
  const V3 w = cross(u, v);
  const float x = dot(u, v);
  const float yn = sqrtf(dot(w, w));
  const bool clamped = yn < 1e-9f;
  const float y = clamped ? 1e-9f : yn;
  const float r2 = x * x + y * y;
  const float dx = -y / r2 * gth;
  const float dy = clamped ? 0.f : x / r2 * gth;
  const V3 n = (clamped ? 0.f : 1.0f / y) * w;
  gu = dx * v + dy * cross(v, n);
  gv = dx * u + dy * cross(n, u);
}
// r = x - (x.n / n.n) n  (vector_rejection, gemnet.py:313-332) and its adjoint
__device__ __forceinline__ V3 reject(V3 x, V3 n) { return x - (dot(x, n) / dot(n, n)) * n; }
__device__ __forceinline__ void reject_bwd(V3 x, V3 n, V3 gr, V3& gx, V3& gn) {
  const float s = dot(x, n), q = dot(n, n), a = s / q, gn_dot = dot(gr, n);
  gx = gr - (gn_dot / q) * n;
  gn = (-a) * gr - gn_dot * ((1.0f / q) * x - (2.0f * s / (q * q)) * n);
}

// Y[q, :] = Y_lm(Phi_cab, Theta_cabd).  One thread per quadruplet; a thread's S^2 = 49 outputs are 196 B apart from
// its neighbour's, so rows go through LDS (row stride 49 words: conflict-free) and leave as one contiguous,
// coalesced block per workgroup (the direct 4-byte stores ran at a quarter of the write bandwidth).
__global__ __launch_bounds__(256) void quad_basis_fwd_kernel(const float* __restrict__ R, const int32_t* __restrict__ qc,
                                                             const int32_t* __restrict__ qa,
                                                             const int32_t* __restrict__ qb,
                                                             const int32_t* __restrict__ qd, float* __restrict__ Y,
                                                             int64_t Q, int S) {
  extern __shared__ float rows[];   // [256][S*S]
  const int SS = S * S;
  for (int64_t q0 = (int64_t)blockIdx.x * 256; q0 < Q; q0 += (int64_t)gridDim.x * 256) {
    const int64_t q = q0 + threadIdx.x;
    if (q < Q) {
      const V3 Ra = v3(R + 3 * (int64_t)qa[q]), Rb = v3(R + 3 * (int64_t)qb[q]);
      const V3 uac = v3(R + 3 * (int64_t)qc[q]) - Ra, uab = Rb - Ra, ubd = v3(R + 3 * (int64_t)qd[q]) - Rb;
      const V3 uba = (-1.0f) * uab;
      double sn, cs, s1, c1;   // polar angle Phi_cab, azimuth Theta_cabd
      angle_uv_sc(uab, uac, sn, cs);
      angle_uv_sc(reject(uac, uab), reject(ubd, uba), s1, c1);
      ylm_row_sc(sn, cs, s1, c1, S, rows + threadIdx.x * SS);
    }
    __syncthreads();
    const int64_t n = (Q - q0 < 256 ? Q - q0 : 256) * SS;
    float* __restrict__ dst = Y + q0 * SS;
    for (int64_t i = threadIdx.x; i < n; i += 256) dst[i] = rows[i];
    __syncthreads();
  }
}

// Gc, Gb, Gd (Q,3): dE/dR of atoms c, b, d per quadruplet (dE/dR_a = -(Gc+Gb+Gd)) given gY (Q,S^2); the gY rows of
// a workgroup are fetched coalesced into LDS first
__global__ __launch_bounds__(256) void quad_basis_bwd_kernel(const float* __restrict__ gY, const float* __restrict__ R,
                                                             const int32_t* __restrict__ qc,
                                                             const int32_t* __restrict__ qa,
                                                             const int32_t* __restrict__ qb,
                                                             const int32_t* __restrict__ qd, float* __restrict__ Gc,
                                                             float* __restrict__ Gb, float* __restrict__ Gd, int64_t Q,
                                                             int S, int ldc, int ldb, int ldd) {
  extern __shared__ float rows[];   // [256][S*S]
  const int SS = S * S;
  for (int64_t q0 = (int64_t)blockIdx.x * 256; q0 < Q; q0 += (int64_t)gridDim.x * 256) {
    const int64_t n = (Q - q0 < 256 ? Q - q0 : 256) * SS;
    const float* __restrict__ src = gY + q0 * SS;
    for (int64_t i = threadIdx.x; i < n; i += 256) rows[i] = src[i];
    __syncthreads();
    const int64_t q = q0 + threadIdx.x;
    if (q < Q) {
      const V3 Ra = v3(R + 3 * (int64_t)qa[q]), Rb = v3(R + 3 * (int64_t)qb[q]);
      const V3 uac = v3(R + 3 * (int64_t)qc[q]) - Ra, uab = Rb - Ra, ubd = v3(R + 3 * (int64_t)qd[q]) - Rb;
      const V3 uba = (-1.0f) * uab;
      const V3 p1 = reject(uac, uab), p2 = reject(ubd, uba);
      double sn, cs, s1, c1;
      angle_uv_sc(uab, uac, sn, cs);
      angle_uv_sc(p1, p2, s1, c1);
      double g_first, g_second;   // d/d(polar angle = Phi_cab), d/d(azimuth = Theta_cabd)
      ylm_dot_grad_sc(sn, cs, s1, c1, S, rows + threadIdx.x * SS, g_first, g_second);
      const float g_phi = (float)g_first, g_th = (float)g_second;
      V3 g_ab, g_ac, gp1, gp2, t1, t2, g_bd, g_ba;
      angle_uv_bwd(uab, uac, g_phi, g_ab, g_ac);
      angle_uv_bwd(p1, p2, g_th, gp1, gp2);
      reject_bwd(uac, uab, gp1, t1, t2);   // d p1 / d(uac, uab)
      g_ac = g_ac + t1; g_ab = g_ab + t2;
      reject_bwd(ubd, uba, gp2, g_bd, g_ba);
      g_ab = g_ab - g_ba;                   // uba = -uab
      // uac = Rc - Ra, uab = Rb - Ra, ubd = Rd - Rb
      const V3 gc = g_ac, gd = g_bd, gb = g_ab - g_bd;
      Gc[ldc * q] = gc.x; Gc[ldc * q + 1] = gc.y; Gc[ldc * q + 2] = gc.z;
      Gb[ldb * q] = gb.x; Gb[ldb * q + 1] = gb.y; Gb[ldb * q + 2] = gb.z;
      Gd[ldd * q] = gd.x; Gd[ldd * q + 1] = gd.y; Gd[ldd * q + 2] = gd.z;
    }
    __syncthreads();
  }
}

// Angle form of the tensor basis: instead of the (Q, 49) harmonics (196 B per quadruplet, re-read by every interaction
// block: 1.8 GB per pass at 9 M quadruplets) only (sin, cos) of the polar angle Phi_cab and of the azimuth Theta_cabd
// leave this kernel — 16 B per quadruplet; the bilinear kernels rebuild Y_lm from them in registers / LDS
// (csrc/bilinear.hip, *_ang kernels; basis_layers.py:239-295 evaluates sympy expressions of the same two angles).
__global__ __launch_bounds__(256) void quad_angles_fwd_kernel(const float* __restrict__ R, const int32_t* __restrict__ qc,
                                                              const int32_t* __restrict__ qa, const int32_t* __restrict__ qb,
                                                              const int32_t* __restrict__ qd, float4* __restrict__ ang,
                                                              int64_t Q) {
  for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; q < Q; q += (int64_t)gridDim.x * 256) {
    const V3 Ra = v3(R + 3 * (int64_t)qa[q]), Rb = v3(R + 3 * (int64_t)qb[q]);
    const V3 uac = v3(R + 3 * (int64_t)qc[q]) - Ra, uab = Rb - Ra, ubd = v3(R + 3 * (int64_t)qd[q]) - Rb;
    const V3 uba = (-1.0f) * uab;
    double sn, cs, s1, c1;
    angle_uv_sc(uab, uac, sn, cs);
    angle_uv_sc(reject(uac, uab), reject(ubd, uba), s1, c1);
    ang[q] = make_float4((float)sn, (float)cs, (float)s1, (float)c1);
  }
}

// Adjoint of the angle form: g_ang[q] = (dE/dPhi_cab, dE/dTheta_cabd, -, -) -> per-quadruplet force contributions.
__global__ __launch_bounds__(256) void quad_angles_bwd_kernel(const float4* __restrict__ g_ang, const float* __restrict__ R,
                                                              const int32_t* __restrict__ qc, const int32_t* __restrict__ qa,
                                                              const int32_t* __restrict__ qb, const int32_t* __restrict__ qd,
                                                              float* __restrict__ Gc, float* __restrict__ Gb,
                                                              float* __restrict__ Gd, int64_t Q, int ldc, int ldb, int ldd) {
  for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; q < Q; q += (int64_t)gridDim.x * 256) {
    const V3 Ra = v3(R + 3 * (int64_t)qa[q]), Rb = v3(R + 3 * (int64_t)qb[q]);
    const V3 uac = v3(R + 3 * (int64_t)qc[q]) - Ra, uab = Rb - Ra, ubd = v3(R + 3 * (int64_t)qd[q]) - Rb;
    const V3 uba = (-1.0f) * uab;
    const V3 p1 = reject(uac, uab), p2 = reject(ubd, uba);
    const float4 g = g_ang[q];
    const float g_phi = g.x, g_th = g.y;
    V3 g_ab, g_ac, gp1, gp2, t1, t2, g_bd, g_ba;
    angle_uv_bwd(uab, uac, g_phi, g_ab, g_ac);
    angle_uv_bwd(p1, p2, g_th, gp1, gp2);
    reject_bwd(uac, uab, gp1, t1, t2);
    g_ac = g_ac + t1; g_ab = g_ab + t2;
    reject_bwd(ubd, uba, gp2, g_bd, g_ba);
    g_ab = g_ab - g_ba;
    const V3 gc = g_ac, gd = g_bd, gb = g_ab - g_bd;
    Gc[ldc * q] = gc.x; Gc[ldc * q + 1] = gc.y; Gc[ldc * q + 2] = gc.z;
    if (ldb == 8 && ldd == 8 && Gd == Gb + 4) {
      // packed rows [Gb xyz, 0, Gd xyz, 0] (the two-level force reduction of ops._quad_adjoint): two 16-byte stores, the pad
      // lanes written here (round 5: the caller used to zero-fill the 288 MB buffer first)
      float4* __restrict__ row = reinterpret_cast<float4*>(Gb + 8 * q);
      row[0] = make_float4(gb.x, gb.y, gb.z, 0.f);
      row[1] = make_float4(gd.x, gd.y, gd.z, 0.f);
    } else {
      Gb[ldb * q] = gb.x; Gb[ldb * q + 1] = gb.y; Gb[ldb * q + 2] = gb.z;
      Gd[ldd * q] = gd.x; Gd[ldd * q + 1] = gd.y; Gd[ldd * q + 2] = gd.z;
    }
  }
}

// Tangent pass of the angle form (the double backward of quad_angles_bwd_kernel under trainer.py:346): for a position tangent
// tR (A,3) — the cotangent dL/dF of the force — tang[q] = (dPhi_cab . tR, dTheta_cabd . tR, 0, 0): the directional derivatives
// of the two angles, i.e. the gradient of  sum_q <G(q), tR>  w.r.t. the incoming g_ang.  Formed from the SAME closed-form
// adjoints as the first-order kernel (unit cotangents), so first adjoint and tangent agree to rounding.
__global__ __launch_bounds__(256) void quad_angles_jvp_kernel(const float* __restrict__ R, const float* __restrict__ tR,
                                                              const int32_t* __restrict__ qc, const int32_t* __restrict__ qa,
                                                              const int32_t* __restrict__ qb, const int32_t* __restrict__ qd,
                                                              float4* __restrict__ tang, int64_t Q) {
  for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; q < Q; q += (int64_t)gridDim.x * 256) {
    const int64_t ia = qa[q], ib = qb[q], ic = qc[q], id = qd[q];
    const V3 Ra = v3(R + 3 * ia), Rb = v3(R + 3 * ib);
    const V3 uac = v3(R + 3 * ic) - Ra, uab = Rb - Ra, ubd = v3(R + 3 * id) - Rb;
    const V3 uba = (-1.0f) * uab;
    const V3 ta = v3(tR + 3 * ia), tb = v3(tR + 3 * ib);
    const V3 dac = v3(tR + 3 * ic) - ta, dab = tb - ta, dbd = v3(tR + 3 * id) - tb;
    const V3 p1 = reject(uac, uab), p2 = reject(ubd, uba);
    V3 g_ab, g_ac, gp1, gp2, t1, t2, g_bd, g_ba;
    angle_uv_bwd(uab, uac, 1.0f, g_ab, g_ac);
    const float dphi = dot(g_ab, dab) + dot(g_ac, dac);
    angle_uv_bwd(p1, p2, 1.0f, gp1, gp2);
    reject_bwd(uac, uab, gp1, t1, t2);      // d p1 / d(uac, uab)
    reject_bwd(ubd, uba, gp2, g_bd, g_ba);  // d p2 / d(ubd, uba);  d uba = -d uab
    const float dth = dot(t1, dac) + dot(t2, dab) + dot(g_bd, dbd) - dot(g_ba, dab);
    tang[q] = make_float4(dphi, dth, 0.f, 0.f);
  }
}

inline int grid_for(int64_t n) {
  int64_t b = (n + 255) / 256;
  return (int)(b < 1 ? 1 : (b > 8192 ? 8192 : b));
}

}  // namespace

extern "C" int gn_edge_basis_fwd_f32(const float* R, const int32_t* id_c, const int32_t* id_a,
                                     const float* freq, const float* z, const double* nrm, float* D, float* V,
                                     float* rbf, float* rad, int64_t E, int NR, int S, float cutoff, int p,
                                     void* stream) {
  if (E <= 0) return 0;
  if (p < 2 || S > 8) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(edge_basis_fwd_kernel, dim3(grid_for(E * 16)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), R, id_c, id_a, freq, z, nrm, D, V, rbf, rad, E, NR, S,
                     (double)cutoff, p);
  GN_LAUNCH_CHECK();
  return 0;
}

extern "C" int gn_edge_basis_bwd_f32(const float* g_D, const float* g_rbf, const float* g_rad, const float* R,
                                     const int32_t* id_c, const int32_t* id_a, const float* freq,
                                     const float* z, const double* nrm, float* W, int64_t E, int NR, int S,
                                     float cutoff, int p, void* stream) {
  if (E <= 0) return 0;
  if (p < 2 || S > 8) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(edge_basis_bwd_kernel, dim3((unsigned)((E * 16 + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                     g_D, g_rbf, g_rad, R, id_c, id_a, freq, z, nrm, W, E, NR, S, (double)cutoff, p);
  GN_LAUNCH_CHECK();
  return 0;
}

extern "C" int gn_trip_basis_fwd_f32(const float* R, const int32_t* tc, const int32_t* ta, const int32_t* tb,
                                     float* Y, float* theta, int64_t T, int S, void* stream) {
  if (T <= 0) return 0;
  if (S > 8) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(trip_basis_fwd_kernel, dim3(grid_for(T)), dim3(256), 0, static_cast<hipStream_t>(stream),
                     R, tc, ta, tb, Y, theta, T, S);
  GN_LAUNCH_CHECK();
  return 0;
}

extern "C" int gn_trip_basis_bwd_f32(const float* gY, const float* R, const int32_t* tc, const int32_t* ta,
                                     const int32_t* tb, float* Gc, float* Gb, int64_t T, int S, void* stream) {
  if (T <= 0) return 0;
  if (S > 8) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(trip_basis_bwd_kernel, dim3(grid_for(T)), dim3(256), 0, static_cast<hipStream_t>(stream),
                     gY, R, tc, ta, tb, Gc, Gb, T, S);
  GN_LAUNCH_CHECK();
  return 0;
}

extern "C" int gn_quad_basis_fwd_f32(const float* R, const int32_t* qc, const int32_t* qa, const int32_t* qb,
                                     const int32_t* qd, float* Y, int64_t Q, int S, void* stream) {
  if (Q <= 0) return 0;
  if (S > 7) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(quad_basis_fwd_kernel, dim3(grid_for(Q)), dim3(256), (size_t)256 * S * S * sizeof(float),
                     static_cast<hipStream_t>(stream), R, qc, qa, qb, qd, Y, Q, S);
  GN_LAUNCH_CHECK();
  return 0;
}

extern "C" int gn_quad_basis_bwd_f32(const float* gY, const float* R, const int32_t* qc, const int32_t* qa,
                                     const int32_t* qb, const int32_t* qd, float* Gc, float* Gb, float* Gd,
                                     int64_t Q, int S, void* stream) {
  return gn_quad_basis_bwd_ld_f32(gY, R, qc, qa, qb, qd, Gc, 3, Gb, 3, Gd, 3, Q, S, stream);
}

extern "C" int gn_quad_basis_bwd_ld_f32(const float* gY, const float* R, const int32_t* qc, const int32_t* qa,
                                        const int32_t* qb, const int32_t* qd, float* Gc, int ldc, float* Gb, int ldb,
                                        float* Gd, int ldd, int64_t Q, int S, void* stream) {
  if (Q <= 0) return 0;
  if (S > 7 || ldc < 3 || ldb < 3 || ldd < 3) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(quad_basis_bwd_kernel, dim3(grid_for(Q)), dim3(256), (size_t)256 * S * S * sizeof(float),
                     static_cast<hipStream_t>(stream), gY, R, qc, qa, qb, qd, Gc, Gb, Gd, Q, S, ldc, ldb, ldd);
  GN_LAUNCH_CHECK();
  return 0;
}

extern "C" int gn_quad_angles_fwd_f32(const float* R, const int32_t* qc, const int32_t* qa, const int32_t* qb,
                                      const int32_t* qd, float* ang, int64_t Q, void* stream) {
  if (Q <= 0) return 0;
  hipLaunchKernelGGL(quad_angles_fwd_kernel, dim3(grid_for(Q)), dim3(256), 0, static_cast<hipStream_t>(stream), R, qc, qa,
                     qb, qd, reinterpret_cast<float4*>(ang), Q);
  GN_LAUNCH_CHECK();
  return 0;
}

extern "C" int gn_quad_angles_bwd_ld_f32(const float* g_ang, const float* R, const int32_t* qc, const int32_t* qa,
                                         const int32_t* qb, const int32_t* qd, float* Gc, int ldc, float* Gb, int ldb,
                                         float* Gd, int ldd, int64_t Q, void* stream) {
  if (Q <= 0) return 0;
  if (ldc < 3 || ldb < 3 || ldd < 3) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(quad_angles_bwd_kernel, dim3(grid_for(Q)), dim3(256), 0, static_cast<hipStream_t>(stream),
                     reinterpret_cast<const float4*>(g_ang), R, qc, qa, qb, qd, Gc, Gb, Gd, Q, ldc, ldb, ldd);
  GN_LAUNCH_CHECK();
  return 0;
}

extern "C" int gn_quad_angles_jvp_f32(const float* R, const float* tR, const int32_t* qc, const int32_t* qa, const int32_t* qb,
                                      const int32_t* qd, float* tang, int64_t Q, void* stream) {
  if (Q <= 0) return 0;
  if ((reinterpret_cast<uintptr_t>(tang) & 15u) != 0) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(quad_angles_jvp_kernel, dim3(grid_for(Q)), dim3(256), 0, static_cast<hipStream_t>(stream), R, tR, qc, qa,
                     qb, qd, reinterpret_cast<float4*>(tang), Q);
  GN_LAUNCH_CHECK();
  return 0;
}
