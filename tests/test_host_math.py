"""The closed-form basis math shared with the HIP kernels (csrc/basis_math.h), compiled for the
host with g++ and checked against the reference goldens and against autograd derivatives of the
oracle.  CPU-only: validates the formulas (value, 1st, 2nd derivatives) before any GPU run."""
import ctypes
import os
import subprocess

import numpy as np
import pytest
import torch

from oracle import basis_oracle as B

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def shim(tmp_path_factory):
    out = tmp_path_factory.mktemp("shim") / "libshim.so"
    subprocess.check_call(["g++", "-O2", "-shared", "-fPIC",
                           os.path.join(ROOT, "tests", "host_math_shim.cpp"), "-o", str(out)])
    return ctypes.CDLL(str(out))


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _deriv(fn, x, order):
    """order-th derivative of each output column of fn(x) w.r.t. the 1-D tensor x (autograd)."""
    x = x.clone().requires_grad_(True)
    y = fn(x)
    cols = []
    for k in range(y.shape[1]):
        g = y[:, k]
        for _ in range(order):
            g = torch.autograd.grad(g.sum(), x, create_graph=True, allow_unused=True)[0]
            if g is None:
                g = torch.zeros_like(x)
        cols.append(g.detach())
    return torch.stack(cols, 1).numpy()


@pytest.mark.parametrize("kd", [0, 1, 2])
def test_bessel_rbf_d(shim, golden_basis, kd):
    d = np.ascontiguousarray(golden_basis["d"])
    f = np.ascontiguousarray(golden_basis["freq"])
    out = np.zeros((len(d), 6))
    shim.shim_bessel_rbf(_p(d), _p(f), _p(out), len(d), 6, ctypes.c_double(5.0), 5, kd, 0)
    ref = _deriv(lambda x: B.bessel_rbf(x, torch.tensor(f), 5.0, 5), torch.tensor(d), kd)
    np.testing.assert_allclose(out, ref, rtol=1e-9, atol=1e-11)
    if kd == 0:
        np.testing.assert_allclose(out, golden_basis["bessel_rbf"], rtol=1e-12, atol=1e-13)
    if kd == 1:
        np.testing.assert_allclose(out, golden_basis["bessel_rbf_dd"], rtol=1e-10, atol=1e-12)


@pytest.mark.parametrize("kd", [0, 1])
def test_bessel_rbf_df(shim, golden_basis, kd):
    d = np.ascontiguousarray(golden_basis["d"])
    f = np.ascontiguousarray(golden_basis["freq"])
    out = np.zeros((len(d), 6))
    shim.shim_bessel_rbf(_p(d), _p(f), _p(out), len(d), 6, ctypes.c_double(5.0), 5, kd, 1)
    ft = torch.tensor(f, requires_grad=True)
    dt = torch.tensor(d, requires_grad=True)
    y = B.bessel_rbf(dt, ft, 5.0, 5)
    ref = np.zeros_like(out)
    for e in range(len(d)):
        for n in range(6):
            v = y[e, n]
            if kd == 1:
                v = torch.autograd.grad(v, dt, create_graph=True)[0][e]
            ref[e, n] = torch.autograd.grad(v, ft, retain_graph=True)[0][n]
    np.testing.assert_allclose(out, ref, rtol=1e-9, atol=1e-11)


@pytest.mark.parametrize("cutoff,dkey,gkey", [(5.0, "d", "c5"), (10.0, "d10", "c10")])
@pytest.mark.parametrize("kd", [0, 1, 2])
def test_sph_radial(shim, golden_basis, cutoff, dkey, gkey, kd):
    d = np.ascontiguousarray(golden_basis[dkey])
    z = B.jn_zeros(7, 6)
    nrm = B.sph_bessel_normalizer(7, 6)
    out = np.zeros((len(d), 42))
    shim.shim_sph_radial(_p(d), _p(z), _p(nrm), _p(out), len(d), 7, 6, ctypes.c_double(cutoff), 5, kd)
    ref = _deriv(lambda x: B.sph_bessel_radial(x, 7, 6, cutoff, 5).reshape(-1, 42), torch.tensor(d), kd)
    np.testing.assert_allclose(out, ref, rtol=1e-8, atol=1e-9)
    if kd == 0:
        np.testing.assert_allclose(out.reshape(-1, 7, 6), golden_basis[f"radial_{gkey}"], rtol=1e-7, atol=2e-9)
    if kd == 1:
        np.testing.assert_allclose(out.reshape(-1, 7, 6), golden_basis[f"radial_{gkey}_dd"], rtol=1e-6, atol=1e-7)


@pytest.mark.parametrize("k", [0, 1, 2])
def test_ylm0(shim, golden_basis, k):
    th = np.ascontiguousarray(golden_basis["theta"])
    out = np.zeros((len(th), 7), dtype=np.float32)
    shim.shim_ylm0(_p(th), _p(out), len(th), 7, k)
    ref = _deriv(lambda x: B.real_sph_harm_l0(7, x), torch.tensor(th), k)
    np.testing.assert_allclose(out, ref, rtol=2e-6, atol=2e-6)
    if k == 0:
        np.testing.assert_allclose(out, golden_basis["y_l0"], rtol=2e-6, atol=1e-6)
    if k == 1:
        np.testing.assert_allclose(out, golden_basis["y_l0_dtheta"], rtol=2e-6, atol=2e-6)


@pytest.mark.parametrize("kt,kp", [(0, 0), (1, 0), (0, 1), (2, 0), (1, 1), (0, 2)])
def test_ylm(shim, golden_basis, kt, kp):
    th = np.ascontiguousarray(golden_basis["theta"])
    ph = np.ascontiguousarray(golden_basis["phi"])
    out = np.zeros((len(th), 49), dtype=np.float32)
    shim.shim_ylm(_p(th), _p(ph), _p(out), len(th), 7, kt, kp)
    tt = torch.tensor(th, requires_grad=True)
    pp = torch.tensor(ph, requires_grad=True)
    y = B.real_sph_harm_full(7, tt, pp)
    ref = np.zeros((len(th), 49))
    for j in range(49):
        g = y[:, j]
        for _ in range(kt):
            g = torch.autograd.grad(g.sum(), tt, create_graph=True, allow_unused=True)[0]
            g = torch.zeros_like(tt) if g is None else g
        for _ in range(kp):
            if not g.requires_grad:
                g = torch.zeros_like(tt)
                break
            g = torch.autograd.grad(g.sum(), pp, create_graph=True, allow_unused=True)[0]
            g = torch.zeros_like(tt) if g is None else g
        ref[:, j] = g.detach().numpy()
    scale = np.abs(ref).max()
    np.testing.assert_allclose(out, ref, rtol=5e-6, atol=5e-7 * max(1.0, scale))
    if (kt, kp) == (0, 0):
        np.testing.assert_allclose(out, golden_basis["y_lm"], rtol=2e-6, atol=1e-6)
    if (kt, kp) == (1, 0):
        np.testing.assert_allclose(out, golden_basis["y_lm_dtheta"], rtol=5e-6, atol=5e-6)
    if (kt, kp) == (0, 1):
        np.testing.assert_allclose(out, golden_basis["y_lm_dphi"], rtol=5e-6, atol=5e-6)


def test_fixed_size_f32_rows_match_the_f64_visitors(shim):
    """ylm7_row_T<float> / ylm7_dot_grad_T<float> (what csrc/bilinear_ang.hip evaluates per quadruplet and pass) against
    the f64 visitors used by the geometry kernels: <= 2e-6 of the largest harmonic, 2e-5 relative on the two gradients."""
    rs = np.random.RandomState(0)
    n = 4000
    th = np.concatenate([rs.uniform(0, np.pi, n - 6), [1e-4, np.pi - 1e-4, 0.5 * np.pi, 1e-7, 1.0, 3.0]])
    ph = np.concatenate([rs.uniform(0, np.pi, n - 6), [0.3, 2.0, 1e-5, 3.1, np.pi - 1e-6, 0.0]])
    a, b = np.zeros((n, 49), np.float32), np.zeros((n, 49), np.float32)
    shim.shim_ylm7_row_f32(_p(th), _p(ph), _p(a), n)
    shim.shim_ylm_row_f64(_p(th), _p(ph), _p(b), n)
    ref = B.real_sph_harm_full(7, torch.tensor(th), torch.tensor(ph)).numpy()
    assert np.abs(b - ref).max() <= 2e-6                       # the f64 visitor is the oracle's formula
    assert np.abs(a.astype(np.float64) - ref).max() <= 2e-6 * np.abs(ref).max()
    g = rs.standard_normal((n, 49)).astype(np.float32)
    o32, o64 = np.zeros((n, 2)), np.zeros((n, 2))
    shim.shim_ylm7_dot_grad(_p(th), _p(ph), _p(g), _p(o32), n, 1)
    shim.shim_ylm7_dot_grad(_p(th), _p(ph), _p(g), _p(o64), n, 0)
    scale = np.abs(o64).max()
    assert np.abs(o32 - o64).max() <= 2e-5 * scale, (np.abs(o32 - o64).max(), scale)


def test_tangent_rows_by_dual_numbers_match_the_derivative_rows(shim):
    """ylm7_row_tangent (csrc/basis_math.h: DualF through the unrolled f32 recurrences) — the rows dY = Y_theta dtheta +
    Y_phi dphi that the second-order sweeps of GemNet-Q force training rebuild per quadruplet (csrc/bilinear_ang.hip, *_tan
    kernels) — against the f64 derivative visitors (ylm_row with kt / kp = 1), for tangents of any magnitude."""
    rs = np.random.RandomState(1)
    n = 3000
    th = np.concatenate([rs.uniform(0.01, np.pi - 0.01, n - 4), [1e-3, np.pi - 1e-3, 0.5 * np.pi, 1.0]])
    ph = np.concatenate([rs.uniform(0, np.pi, n - 4), [0.3, 2.0, 1e-5, np.pi - 1e-6]])
    scale = 10.0 ** rs.uniform(-6, 3, n)
    dth = (rs.standard_normal(n) * scale).astype(np.float32)
    dph = (rs.standard_normal(n) * scale).astype(np.float32)
    val, tan = np.zeros((n, 49), np.float32), np.zeros((n, 49), np.float32)
    shim.shim_ylm7_row_tangent(_p(th), _p(ph), _p(dth), _p(dph), _p(val), _p(tan), n)
    y, yt, yp = (np.zeros((n, 49), np.float32) for _ in range(3))
    shim.shim_ylm(_p(th), _p(ph), _p(y), n, 7, 0, 0)
    shim.shim_ylm(_p(th), _p(ph), _p(yt), n, 7, 1, 0)
    shim.shim_ylm(_p(th), _p(ph), _p(yp), n, 7, 0, 1)
    assert np.abs(val - y).max() <= 2e-6 * np.abs(y).max()
    ref = yt.astype(np.float64) * dth[:, None] + yp.astype(np.float64) * dph[:, None]
    row = np.abs(ref).max(axis=1, keepdims=True) + 1e-30
    assert (np.abs(tan - ref) / row).max() <= 2e-5, (np.abs(tan - ref) / row).max()
