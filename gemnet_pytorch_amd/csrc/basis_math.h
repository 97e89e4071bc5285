// Pure math of the GemNet basis functions, shared by the HIP kernels (basis.hip) and by the
// host-side unit-test shim (tests/host_math_shim.cpp, compiled with g++): every function here is
// GN_HD so the formulas can be checked against the reference goldens without a GPU.
#pragma once
#include <math.h>
#include <stdint.h>
#if defined(__HIPCC__)
#define GN_HD __host__ __device__ __forceinline__
#else
#define GN_HD inline
#endif

struct Env3 { double u, u1, u2; };  // envelope value, d/dd, d2/dd2

// u(x) = 1 + a x^p + b x^(p+1) + c x^(p+2) for x < 1 else 0, x = d / cutoff  (envelope.py:14-29)
GN_HD Env3 envelope(double x, int p, double inv_c) {
  Env3 r = {0.0, 0.0, 0.0};
  if (x < 1.0) {
    const double a = -(p + 1) * (p + 2) / 2.0, b = (double)p * (p + 2), c = -p * (p + 1) / 2.0;
    // x^(p-2) by repeated multiplication (p >= 2 in all GemNet configs, p = 5): the f64 pow() was ~200 instructions of every
    // one of the 49 work items per edge of the basis kernels
    double xp2 = 1.0;
    for (int i = 0; i < p - 2; ++i) xp2 *= x;
    const double xp1 = xp2 * x, xp = xp1 * x, xq = xp * x, xr = xq * x;
    r.u = 1.0 + a * xp + b * xq + c * xr;
    r.u1 = (a * p * xp1 + b * (p + 1) * xp + c * (p + 2) * xq) * inv_c;
    r.u2 = (a * p * (p - 1) * xp2 + b * (p + 1) * p * xp1 + c * (p + 2) * (p + 1) * xp) * inv_c * inv_c;
  }
  return r;
}

// spherical Bessel j_l(y) for l = 0..L (L <= 7), stable: series for y < l, upward recurrence else
// 1/x for a small positive integer-valued x (< 2^24: exact in f32): f32 reciprocal + two Newton steps in f64 (relative
// error ~1e-28 before rounding) instead of the ~40-instruction f64 division — the power series below used to spend 30
// of them per evaluation, and small arguments (y < l) are the common case for the short edges of l >= 3.
GN_HD double rcp_small_int(double x) {
  double r = (double)(1.0f / (float)x);
  r = r * (2.0 - x * r);
  return r * (2.0 - x * r);
}

GN_HD double sph_jl_series(int l, double y) {
  double dfact = 1.0;
  for (int i = 1; i <= 2 * l + 1; i += 2) dfact *= i;
  const double q = -0.5 * y * y;
  double term = 1.0, acc = 1.0;
  for (int k = 1; k < 30; ++k) {
    term *= q * rcp_small_int(k * (2.0 * l + 2.0 * k + 1.0));
    acc += term;
    if (fabs(term) < 1e-18 * fabs(acc)) break;   // (y < l <= 7: a dozen terms; the fixed 30 were most of the small-y cost)
  }
  double yl = 1.0;
  for (int i = 0; i < l; ++i) yl *= y;
  return yl * rcp_small_int(dfact) * acc;
}

GN_HD double sph_jl(int l, double y, double sn, double cs) {
  if (l > 0 && y < (double)l) return sph_jl_series(l, y);
  const double iy = 1.0 / y;
  double jm = sn * iy;  // j0
  if (l == 0) return jm;
  double j = sn * iy * iy - cs * iy;  // j1
  for (int k = 1; k < l; ++k) {
    const double jn = (2 * k + 1) * iy * j - jm;
    jm = j;
    j = jn;
  }
  return j;
}

// (j_{l-1}(y), j_l(y)) in ONE pass (l >= 1): the derivative j_l' = j_{l-1} - (l + 1) / y j_l needs both — two separate
// recurrences (or two series) before
GN_HD void sph_jl_pair(int l, double y, double sn, double cs, double& jlm1, double& jl) {
  if (y < (double)l) {                 // series region
    jl = sph_jl_series(l, y);
    jlm1 = (l - 1 > 0 && y < (double)(l - 1)) ? sph_jl_series(l - 1, y) : sph_jl(l - 1, y, sn, cs);
    return;
  }
  const double iy = 1.0 / y;
  double jm = sn * iy;                 // j0
  double j = sn * iy * iy - cs * iy;   // j1
  for (int k = 1; k < l; ++k) {
    const double jn = (2 * k + 1) * iy * j - jm;
    jm = j;
    j = jn;
  }
  jlm1 = jm;
  jl = j;
}

GN_HD double ylm_prefactor(int l, int m) {
  // sqrt((2l+1)/(4 pi) (l-m)!/(l+m)!)   (basis_utils.py:83-104), m >= 0
  double r = (2.0 * l + 1.0) / (4.0 * 3.14159265358979323846);
  for (int k = l - m + 1; k <= l + m; ++k) r /= k;
  return sqrt(r);
}

// second-order jet in theta
struct Jet { double v, d1, d2; };
GN_HD Jet jmul(const Jet& a, const Jet& b) {
  return {a.v * b.v, a.d1 * b.v + a.v * b.d1, a.d2 * b.v + 2.0 * a.d1 * b.d1 + a.v * b.d2};
}
GN_HD Jet jscale(const Jet& a, double s) { return {a.v * s, a.d1 * s, a.d2 * s}; }
GN_HD Jet jsub(const Jet& a, const Jet& b) { return {a.v - b.v, a.d1 - b.d1, a.d2 - b.d2}; }


// d^kd/dd^kd d^kf/df^kf [ u(d/c) sqrt(2/c) sin(f d/c) / d ]    (basis_layers.py:45-49)
GN_HD double bessel_rbf_eval(double d, double f, double cutoff, int p, int kd, int kf) {
  const double inv_c = 1.0 / cutoff;
  const double A = sqrt(2.0 * inv_c);
  const double w = f * inv_c;
  const Env3 u = envelope(d * inv_c, p, inv_c);
  double sn, cs;
  sincos(w * d, &sn, &cs);
  const double id = 1.0 / d;
  double v;
  if (kf == 0) {
    const double h = sn * id;
    const double h1 = w * cs * id - sn * id * id;
    if (kd == 0) v = u.u * h;
    else if (kd == 1) v = u.u1 * h + u.u * h1;
    else {
      const double h2 = -w * w * sn * id - 2.0 * w * cs * id * id + 2.0 * sn * id * id * id;
      v = u.u2 * h + 2.0 * u.u1 * h1 + u.u * h2;
    }
  } else {  // d/df: dh/df = cos(wd)/c, dh'/df = -w sin(wd)/c
    if (kd == 0) v = u.u * cs * inv_c;
    else v = (u.u1 * cs - u.u * w * sn) * inv_c;
  }
  return A * v;
}

// d^kd/dd^kd [ u(d/c) c^-1.5 N j_l(z d/c) ]    (basis_layers.py:121-128; basis_utils.py:47-80)
GN_HD double sph_radial_eval(double d, double z, double nrm, int l, double cutoff, int p, int kd) {
  const double inv_c = 1.0 / cutoff;
  const double a = z * inv_c;  // y = a d
  const double y = a * d;
  const Env3 u = envelope(d * inv_c, p, inv_c);
  double sn, cs;
  sincos(y, &sn, &cs);
  double v;
  if (kd == 0) {
    v = u.u * sph_jl(l, y, sn, cs);
  } else {
    // j_l' = j_{l-1} - (l+1)/y j_l  (l >= 1);  j_0' = -j_1
    double jl, jlm1 = 0.0;
    if (l == 0) jl = sph_jl(0, y, sn, cs); else sph_jl_pair(l, y, sn, cs, jlm1, jl);
    const double j1 = (l == 0) ? -sph_jl(1, y, sn, cs) : jlm1 - (l + 1) / y * jl;
    const double J1 = a * j1;
    if (kd == 1) {
      v = u.u1 * jl + u.u * J1;
    } else {
      // y^2 j'' + 2 y j' + (y^2 - l(l+1)) j = 0
      const double j2 = -2.0 / y * j1 - (1.0 - l * (l + 1) / (y * y)) * jl;
      v = u.u2 * jl + 2.0 * u.u1 * J1 + u.u * a * a * j2;
    }
  }
  return v * nrm * inv_c * sqrt(inv_c);
}

// out[l] = d^k/dtheta^k [ N_l0 P_l(cos theta) ], l < S    (zero_m_only branch, basis_utils.py:221-222)
GN_HD void ylm0_row(double theta, int S, int k, float* out) {
  double sn, c;
  sincos(theta, &sn, &c);
  // P_l = ((2l-1) c P_{l-1} - (l-1) P_{l-2})/l,  P'_l = P'_{l-2} + (2l-1) P_{l-1},
  // P''_l = P''_{l-2} + (2l-1) P'_{l-1}
  double P0 = 1.0, P1 = c, D0 = 0.0, D1 = 1.0, H0 = 0.0, H1 = 0.0;
  for (int l = 0; l < S; ++l) {
    double P, D, H;
    if (l == 0) { P = P0; D = D0; H = H0; }
    else if (l == 1) { P = P1; D = D1; H = H1; }
    else {
      P = ((2 * l - 1) * c * P1 - (l - 1) * P0) / l;
      D = D0 + (2 * l - 1) * P1;
      H = H0 + (2 * l - 1) * D1;
      P0 = P1; P1 = P; D0 = D1; D1 = D; H0 = H1; H1 = H;
    }
    double v;
    if (k == 0) v = P;
    else if (k == 1) v = -sn * D;
    else v = sn * sn * H - c * D;
    out[l] = (float)(ylm_prefactor(l, 0) * v);
  }
}

// ylm_prefactor(l, m) for l, m < 7 (the loop + sqrt per (l, m) per quadruplet was a third of the tensor-basis kernels)
GN_HD double ylm_prefactor_tab(int l, int m) {
  constexpr double T[7][7] = {
    {0.28209479177387814, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0},
    {0.4886025119029199, 0.3454941494713355, 0.0, 0.0, 0.0, 0.0, 0.0},
    {0.6307831305050401, 0.2575161346821264, 0.1287580673410632, 0.0, 0.0, 0.0, 0.0},
    {0.7463526651802308, 0.21545345607610045, 0.06813236509555216, 0.02781492157551894, 0.0, 0.0, 0.0},
    {0.8462843753216345, 0.18923493915151202, 0.044603102903819275, 0.011920680675222404, 0.004214597070904597, 0.0, 0.0},
    {0.9356025796273888, 0.17081687924064806, 0.03228135587163618, 0.006589404174225528, 0.0015531374585246046, 0.000491145188826305, 0.0},
    {1.0171072362820548, 0.156943053829006, 0.024814875652103455, 0.00413581260868391, 0.0007550926197968212, 0.0001609862874555169, 4.6472738199140574e-05}};
  return (l < 7 && m < 7) ? T[l][m] : ylm_prefactor(l, m);
}

// Jet arithmetic that only carries the derivative orders <= JO (the unused components fold away)
template <int JO> GN_HD Jet jmul_o(const Jet& a, const Jet& b) {
  Jet r = {a.v * b.v, 0.0, 0.0};
  if (JO >= 1) r.d1 = a.d1 * b.v + a.v * b.d1;
  if (JO >= 2) r.d2 = a.d2 * b.v + 2.0 * a.d1 * b.d1 + a.v * b.d2;
  return r;
}

// Core recurrence over (m, l): visit(l, m, base, ql, cos(m phi), sin(m phi), prefactor) with the theta-jet of the
// phase-free associated Legendre Q_l^m carried to order JO.  cos/sin(m phi) by the angle-addition recurrence from
// one sincos (error ~ m eps) instead of one f64 sincos per m.
template <int JO, typename F>
GN_HD void ylm_core_sc(double sn, double cs, double s1, double c1, int S, F visit) {
  const Jet js = {sn, cs, -sn};
  const Jet jc = {cs, -sn, -cs};
  double cm = 1.0, sm = 0.0;   // cos(m phi), sin(m phi)
  Jet qmm = {1.0, 0.0, 0.0};
  for (int m = 0; m < S; ++m) {
    if (m > 0) {
      qmm = jscale(jmul_o<JO>(js, qmm), (double)(2 * m - 1));
      const double cn = cm * c1 - sm * s1;
      sm = sm * c1 + cm * s1;
      cm = cn;
    }
    Jet qa = qmm;              // q[l-1][m]
    Jet qb = {0.0, 0.0, 0.0};  // q[l-2][m]
    for (int l = m; l < S; ++l) {
      Jet ql;
      if (l == m) ql = qmm;
      else if (l == m + 1) ql = jscale(jmul_o<JO>(jc, qa), (double)(2 * m + 1));
      else ql = jscale(jsub(jscale(jmul_o<JO>(jc, qa), (double)(2 * l - 1)), jscale(qb, (double)(l + m - 1))),
                       1.0 / (l - m));
      if (l > m) { qb = qa; qa = ql; }
      visit(l, m, l * l, ql, cm, sm, (m == 0 ? 1.0 : 1.4142135623730951) * ylm_prefactor_tab(l, m));
    }
  }
}

template <int JO, typename F>
GN_HD void ylm_core(double theta, double ph, int S, F visit) {
  double sn, cs, s1, c1;
  sincos(theta, &sn, &cs);
  sincos(ph, &s1, &c1);
  ylm_core_sc<JO>(sn, cs, s1, c1, S, visit);
}

// Visit every real Y_j(theta, phi), j < S*S, with its kt-th theta / kp-th phi derivative:
// emit(slot, value).  Per degree l the slots hold m = 0, +1..+l, -l..-1 (negative list indices at
// basis_utils.py:237).  Y_l,+m = sqrt2 N_lm Q_l^m cos(m phi), Y_l,-m = sqrt2 N_lm Q_l^m sin(m phi),
// Q_l^m = phase-free associated Legendre in (cos theta, sin theta)  (basis_utils.py:137-159,226-243).
template <int KT, typename F>
GN_HD void ylm_visit_t(double theta, double ph, int S, int kp, F emit) {
  ylm_core<KT>(theta, ph, S, [&](int l, int m, int base, const Jet& ql, double cm, double sm, double pf) {
    const double tv = pf * (KT == 0 ? ql.v : (KT == 1 ? ql.d1 : ql.d2));
    if (m == 0) {
      emit(base, kp == 0 ? tv : 0.0);
    } else {
      double fc, fs;  // phi factors of +m (cos) and -m (sin), kp-th derivative
      if (kp == 0) { fc = cm; fs = sm; }
      else if (kp == 1) { fc = -m * sm; fs = m * cm; }
      else { fc = -(double)m * m * cm; fs = -(double)m * m * sm; }
      emit(base + m, tv * fc);              // +m
      emit(base + 2 * l + 1 - m, tv * fs);  // -m
    }
  });
}

template <typename F>
GN_HD void ylm_visit(double theta, double ph, int S, int kt, int kp, F emit) {
  if (kt == 0) ylm_visit_t<0>(theta, ph, S, kp, emit);
  else if (kt == 1) ylm_visit_t<1>(theta, ph, S, kp, emit);
  else ylm_visit_t<2>(theta, ph, S, kp, emit);
}

// Both first derivatives contracted with g in one pass (the geometry adjoint needs exactly these two):
//   g_theta = sum_j g[j] dY_j/dtheta,   g_phi = sum_j g[j] dY_j/dphi
// (sin, cos) of both angles given directly: angles that come from atan2(y, x) never need the atan2 nor the sincos)
GN_HD void ylm_dot_grad_sc(double sn, double cs, double s1, double c1, int S, const float* g, double& g_theta,
                           double& g_phi) {
  double at = 0.0, ap = 0.0;
  ylm_core_sc<1>(sn, cs, s1, c1, S, [&](int l, int m, int base, const Jet& ql, double cm, double sm, double pf) {
    if (m == 0) {
      at += (double)g[base] * pf * ql.d1;
    } else {
      const double gp = (double)g[base + m], gm = (double)g[base + 2 * l + 1 - m];
      at += pf * ql.d1 * (gp * cm + gm * sm);
      ap += pf * ql.v * (double)m * (gm * cm - gp * sm);
    }
  });
  g_theta = at;
  g_phi = ap;
}

GN_HD void ylm_dot_grad(double theta, double ph, int S, const float* g, double& g_theta, double& g_phi) {
  double sn, cs, s1, c1;
  sincos(theta, &sn, &cs);
  sincos(ph, &s1, &c1);
  ylm_dot_grad_sc(sn, cs, s1, c1, S, g, g_theta, g_phi);
}

// values only, from (sin, cos) of both angles
GN_HD void ylm_row_sc(double sn, double cs, double s1, double c1, int S, float* o) {
  ylm_core_sc<0>(sn, cs, s1, c1, S, [o](int l, int m, int base, const Jet& ql, double cm, double sm, double pf) {
    const double tv = pf * ql.v;
    if (m == 0) {
      o[base] = (float)tv;
    } else {
      o[base + m] = (float)(tv * cm);
      o[base + 2 * l + 1 - m] = (float)(tv * sm);
    }
  });
}

// out[j] = d^kt/dtheta^kt d^kp/dphi^kp Y_j(theta, phi), j < S*S
GN_HD void ylm_row(double theta, double ph, int S, int kt, int kp, float* o) {
  ylm_visit(theta, ph, S, kt, kp, [o](int slot, double v) { o[slot] = (float)v; });
}

// sum_j g[j] * d^kt/dtheta^kt d^kp/dphi^kp Y_j(theta, phi)
GN_HD double ylm_dot(double theta, double ph, int S, int kt, int kp, const float* g) {
  double acc = 0.0;
  ylm_visit(theta, ph, S, kt, kp, [&acc, g](int slot, double v) { acc += (double)g[slot] * v; });
  return acc;
}

// ---- S = 7 rows with every loop bound a compile-time constant, in the scalar type T -----------------------------------
// The generic visitors above run f64 jets through runtime loops (an f64 division and a table lookup per (l, m)): about
// 10 k cycles per 64 quadruplets on gfx950.  The bilinear kernels that rebuild the tensor basis on the fly
// (csrc/bilinear_ang.hip) evaluate a row per quadruplet per pass, so they use these fully unrolled forms in f32:
// ~200 FMAs per row; deviation from the f64 row <= 2e-6 of the largest component (tests/test_host_math.py).
// emit(slot, value of type T) for the 49 harmonics in the reference order
template <typename T, typename F>
GN_HD void ylm7_visit_T(T sn, T cs, T s1, T c1, F emit) {
  constexpr int L = 7;
  T cm[L], sm[L];
  cm[0] = T(1); sm[0] = T(0);
#pragma unroll
  for (int m = 1; m < L; ++m) {
    cm[m] = cm[m - 1] * c1 - sm[m - 1] * s1;
    sm[m] = sm[m - 1] * c1 + cm[m - 1] * s1;
  }
  T qmm = T(1);
#pragma unroll
  for (int m = 0; m < L; ++m) {
    if (m > 0) qmm = qmm * sn * T(2 * m - 1);
    T qa = qmm, qb = T(0);
#pragma unroll
    for (int l = m; l < L; ++l) {
      T ql;
      if (l == m) ql = qmm;
      else if (l == m + 1) ql = cs * qa * T(2 * m + 1);
      else ql = (T(2 * l - 1) * cs * qa - T(l + m - 1) * qb) * T(1.0 / (l - m));
      if (l > m) { qb = qa; qa = ql; }
      const T tv = ql * T((m == 0 ? 1.0 : 1.4142135623730951) * ylm_prefactor_tab(l, m));
      if (m == 0) {
        emit(l * l, tv);
      } else {
        emit(l * l + m, tv * cm[m]);
        emit(l * l + 2 * l + 1 - m, tv * sm[m]);
      }
    }
  }
}

template <typename T>
GN_HD void ylm7_row_T(T sn, T cs, T s1, T c1, float* o) {
  ylm7_visit_T<T>(sn, cs, s1, c1, [o](int slot, T v) { o[slot] = (float)v; });
}

// First-order dual number in f32: value + one directional derivative.  ylm7_visit_T<DualF> with
//   sn = (sin th, cos th * dth), cs = (cos th, -sin th * dth), s1 = (sin ph, cos ph * dph), c1 = (cos ph, -sin ph * dph)
// yields every Y_j together with its TANGENT  dY_j = dY_j/dth * dth + dY_j/dph * dph  — the rows the second-order sweeps of
// force training contract with (trainer.py:346 through basis_layers.py:239-295: the tangent of the tensor basis along the
// position tangent u = dL/dF), at about twice the cost of the plain row and without a hand-derived derivative table.
struct DualF {
  float v, d;
  GN_HD DualF() : v(0.f), d(0.f) {}
  GN_HD DualF(float v_, float d_) : v(v_), d(d_) {}
  GN_HD explicit DualF(double c) : v((float)c), d(0.f) {}
  GN_HD explicit DualF(int c) : v((float)c), d(0.f) {}
};
GN_HD DualF operator+(DualF a, DualF b) { return DualF(a.v + b.v, a.d + b.d); }
GN_HD DualF operator-(DualF a, DualF b) { return DualF(a.v - b.v, a.d - b.d); }
GN_HD DualF operator*(DualF a, DualF b) { return DualF(a.v * b.v, a.d * b.v + a.v * b.d); }

// o_val[j] = Y_j, o_tan[j] = dY_j (either may be null) for the tangent (dth, dph) of the two angles
GN_HD void ylm7_row_tangent(float sn, float cs, float s1, float c1, float dth, float dph, float* o_val, float* o_tan) {
  ylm7_visit_T<DualF>(DualF(sn, cs * dth), DualF(cs, -sn * dth), DualF(s1, c1 * dph), DualF(c1, -s1 * dph),
                      [o_val, o_tan](int slot, DualF y) {
                        if (o_val) o_val[slot] = y.v;
                        if (o_tan) o_tan[slot] = y.d;
                      });
}

// g_theta = sum_j g[j] dY_j/dtheta, g_phi = sum_j g[j] dY_j/dphi for the 49 harmonics of S = 7 (ylm_dot_grad_sc in T)
template <typename T>
GN_HD void ylm7_dot_grad_T(T sn, T cs, T s1, T c1, const float* g, T& g_theta, T& g_phi) {
  constexpr int L = 7;
  T cm[L], sm[L];
  cm[0] = T(1); sm[0] = T(0);
#pragma unroll
  for (int m = 1; m < L; ++m) {
    cm[m] = cm[m - 1] * c1 - sm[m - 1] * s1;
    sm[m] = sm[m - 1] * c1 + cm[m - 1] * s1;
  }
  T at = T(0), ap = T(0);
  T qmm = T(1), dmm = T(0);          // Q_m^m and its theta derivative
#pragma unroll
  for (int m = 0; m < L; ++m) {
    if (m > 0) {
      const T nd = T(2 * m - 1) * (cs * qmm + sn * dmm);
      qmm = T(2 * m - 1) * sn * qmm;
      dmm = nd;
    }
    T qa = qmm, da = dmm, qb = T(0), db = T(0);
#pragma unroll
    for (int l = m; l < L; ++l) {
      T ql, dl;
      if (l == m) { ql = qmm; dl = dmm; }
      else if (l == m + 1) { ql = T(2 * m + 1) * cs * qa; dl = T(2 * m + 1) * (cs * da - sn * qa); }
      else {
        const T a = T((2.0 * l - 1.0) / (l - m)), b = T((double)(l + m - 1) / (l - m));
        ql = a * cs * qa - b * qb;
        dl = a * (cs * da - sn * qa) - b * db;
      }
      if (l > m) { qb = qa; db = da; qa = ql; da = dl; }
      const T pf = T((m == 0 ? 1.0 : 1.4142135623730951) * ylm_prefactor_tab(l, m));
      if (m == 0) {
        at += (T)g[l * l] * pf * dl;
      } else {
        const T gp = (T)g[l * l + m], gm = (T)g[l * l + 2 * l + 1 - m];
        at += pf * dl * (gp * cm[m] + gm * sm[m]);
        ap += pf * ql * T(m) * (gm * cm[m] - gp * sm[m]);
      }
    }
  }
  g_theta = at;
  g_phi = ap;
}

