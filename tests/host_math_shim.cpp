// TEST-ONLY host shim: exposes the GN_HD math of gemnet_pytorch_amd/csrc/basis_math.h and the
// ScaledSiLU derivative formulas to ctypes so they can be checked against the reference goldens
// on a machine without a GPU.  Compiled with g++ by tests/test_host_math.py; not product code.
#include <math.h>
#include <stdint.h>
#include "../gemnet_pytorch_amd/csrc/basis_math.h"

extern "C" {
void shim_bessel_rbf(const double* d, const double* f, double* out, int n, int R, double cutoff, int p, int kd, int kf) {
  for (int i = 0; i < n; ++i) for (int r = 0; r < R; ++r) out[i * R + r] = bessel_rbf_eval(d[i], f[r], cutoff, p, kd, kf);
}
void shim_sph_radial(const double* d, const float* z, const double* nrm, double* out, int n, int S, int R, double cutoff, int p, int kd) {
  for (int i = 0; i < n; ++i) for (int lr = 0; lr < S * R; ++lr)
    out[i * S * R + lr] = sph_radial_eval(d[i], (double)z[lr], nrm[lr], lr / R, cutoff, p, kd);
}
void shim_ylm0(const double* th, float* out, int n, int S, int k) { for (int i = 0; i < n; ++i) ylm0_row(th[i], S, k, out + i * S); }
void shim_ylm(const double* th, const double* ph, float* out, int n, int S, int kt, int kp) {
  for (int i = 0; i < n; ++i) ylm_row(th[i], ph[i], S, kt, kp, out + i * S * S);
}
void shim_ylm7_row_f32(const double* th, const double* ph, float* out, int n) {
  for (int i = 0; i < n; ++i)
    ylm7_row_T<float>((float)sin(th[i]), (float)cos(th[i]), (float)sin(ph[i]), (float)cos(ph[i]), out + i * 49);
}
void shim_ylm7_row_tangent(const double* th, const double* ph, const float* dth, const float* dph, float* val, float* tan, int n) {
  for (int i = 0; i < n; ++i)
    ylm7_row_tangent((float)sin(th[i]), (float)cos(th[i]), (float)sin(ph[i]), (float)cos(ph[i]), dth[i], dph[i], val + i * 49,
                     tan + i * 49);
}
void shim_ylm_row_f64(const double* th, const double* ph, float* out, int n) {
  for (int i = 0; i < n; ++i) ylm_row_sc(sin(th[i]), cos(th[i]), sin(ph[i]), cos(ph[i]), 7, out + i * 49);
}
void shim_ylm7_dot_grad(const double* th, const double* ph, const float* g, double* out, int n, int use_f32) {
  for (int i = 0; i < n; ++i) {
    if (use_f32) {
      float a, b;
      ylm7_dot_grad_T<float>((float)sin(th[i]), (float)cos(th[i]), (float)sin(ph[i]), (float)cos(ph[i]), g + i * 49, a, b);
      out[2 * i] = a; out[2 * i + 1] = b;
    } else {
      double a, b;
      ylm_dot_grad_sc(sin(th[i]), cos(th[i]), sin(ph[i]), cos(ph[i]), 7, g + i * 49, a, b);
      out[2 * i] = a; out[2 * i + 1] = b;
    }
  }
}
}
