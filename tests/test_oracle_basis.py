"""Oracle basis closed forms vs goldens produced by the reference's sympy formulas (fp64)."""
import numpy as np
import torch

from oracle import basis_oracle as B


def test_jn_zeros_bit_exact(golden_basis):
    z = B.jn_zeros(7, 6)
    assert z.dtype == np.float32
    assert np.array_equal(z, golden_basis["jn_zeros"])


def test_normalizer_and_prefactor(golden_basis):
    np.testing.assert_allclose(B.sph_bessel_normalizer(7, 6), golden_basis["normalizer"], rtol=1e-14)
    pf = np.array([[B.sph_harm_prefactor(l, m) if m <= l else 0.0 for m in range(7)] for l in range(7)])
    np.testing.assert_allclose(pf, golden_basis["prefactor"], rtol=1e-14)


def test_sph_harm_order():
    o = B.sph_harm_order(3)
    assert o == [(0, 0), (1, 0), (1, 1), (1, -1), (2, 0), (2, 1), (2, 2), (2, -2), (2, -1)]


def test_bessel_rbf(golden_basis):
    d = torch.tensor(golden_basis["d"], requires_grad=True)
    f = torch.tensor(golden_basis["freq"])
    y = B.bessel_rbf(d, f, 5.0, 5)
    np.testing.assert_allclose(y.detach().numpy(), golden_basis["bessel_rbf"], rtol=1e-12, atol=1e-14)
    g = torch.stack([torch.autograd.grad(y[:, k].sum(), d, retain_graph=True)[0] for k in range(6)], 1)
    np.testing.assert_allclose(g.numpy(), golden_basis["bessel_rbf_dd"], rtol=1e-10, atol=1e-12)


def _check_radial(golden_basis, key, dkey, cutoff):
    d = torch.tensor(golden_basis[dkey], requires_grad=True)
    rad = B.sph_bessel_radial(d, 7, 6, cutoff, 5)
    ref = golden_basis[f"radial_{key}"]
    # the reference's expanded sympy form loses digits at small z*d/c;
    # tolerance is absolute on an O(1..10) quantity
    np.testing.assert_allclose(rad.detach().numpy(), ref, rtol=1e-7, atol=2e-9)
    g = torch.stack([torch.autograd.grad(rad[:, l, k].sum(), d, retain_graph=True)[0]
                     for l in range(7) for k in range(6)], 1).reshape(-1, 7, 6)
    np.testing.assert_allclose(g.numpy(), golden_basis[f"radial_{key}_dd"], rtol=1e-6, atol=1e-7)


def test_radial_c5(golden_basis):
    _check_radial(golden_basis, "c5", "d", 5.0)


def test_radial_c10(golden_basis):
    _check_radial(golden_basis, "c10", "d10", 10.0)


def test_tensor_radial_repeat(golden_basis):
    d = torch.tensor(golden_basis["d"])
    rad = B.sph_bessel_radial(d, 7, 6, 5.0, 5)
    rep = torch.repeat_interleave(rad, torch.arange(7) * 2 + 1, dim=1)
    np.testing.assert_allclose(rep.numpy(), golden_basis["radial_tensor_c5"], rtol=1e-7, atol=2e-9)


def test_y_l0(golden_basis):
    th = torch.tensor(golden_basis["theta"], requires_grad=True)
    y = B.real_sph_harm_l0(7, th)
    np.testing.assert_allclose(y.detach().numpy(), golden_basis["y_l0"], rtol=1e-12, atol=1e-13)
    g = torch.stack([torch.autograd.grad(y[:, l].sum(), th, retain_graph=True)[0] for l in range(7)], 1)
    np.testing.assert_allclose(g.numpy(), golden_basis["y_l0_dtheta"], rtol=1e-10, atol=1e-12)


def test_cbf_product(golden_basis):
    d = torch.tensor(golden_basis["d"])
    th = torch.tensor(golden_basis["theta"])
    out = (B.sph_bessel_radial(d, 7, 6, 5.0, 5) * B.real_sph_harm_l0(7, th)[:, :, None]).reshape(-1, 42)
    np.testing.assert_allclose(out.numpy(), golden_basis["cbf_c5"], rtol=1e-7, atol=2e-9)


def test_y_lm(golden_basis):
    th = torch.tensor(golden_basis["theta"], requires_grad=True)
    ph = torch.tensor(golden_basis["phi"], requires_grad=True)
    y = B.real_sph_harm_full(7, th, ph)
    np.testing.assert_allclose(y.detach().numpy(), golden_basis["y_lm"], rtol=1e-11, atol=1e-12)
    gt = torch.stack([torch.autograd.grad(y[:, k].sum(), th, retain_graph=True)[0] for k in range(49)], 1)
    gp = torch.stack([torch.autograd.grad(y[:, k].sum(), ph, retain_graph=True)[0] for k in range(49)], 1)
    np.testing.assert_allclose(gt.numpy(), golden_basis["y_lm_dtheta"], rtol=1e-9, atol=1e-10)
    np.testing.assert_allclose(gp.numpy(), golden_basis["y_lm_dphi"], rtol=1e-9, atol=1e-10)
