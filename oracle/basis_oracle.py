"""TEST INFRASTRUCTURE (oracle) — closed-form restatement of GemNet's basis functions.

Plain PyTorch, dtype-generic (run it in float64 for goldens), differentiable to any
order by autograd.  No sympy: the reference derives these formulas symbolically at
constructor time (17-59 s); here they are evaluated by recurrences.

Reference followed (file:line under /root/reference):
  * Jn_zeros                      gemnet/model/layers/basis_utils.py:14-29
  * bessel_basis (normaliser)     gemnet/model/layers/basis_utils.py:47-80
  * sph_harm_prefactor            gemnet/model/layers/basis_utils.py:83-104
  * associated_legendre_polynomials  basis_utils.py:107-171
  * real_sph_harm                 basis_utils.py:174-253
  * Envelope                      gemnet/model/layers/envelope.py:4-29
  * BesselBasisLayer.forward      gemnet/model/layers/basis_layers.py:45-49
  * SphericalBasisLayer.forward   basis_layers.py:119-162
  * TensorBasisLayer.forward      basis_layers.py:239-295
Pinned by tests/golden/basis_*.npz (generated from the reference, fp64).
"""
import math
from functools import lru_cache

import numpy as np
import torch
from scipy import special as sp
from scipy.optimize import brentq


@lru_cache(maxsize=None)
def jn_zeros(n: int, k: int) -> np.ndarray:
    """First k positive zeros of j_l, l < n, stored as float32 (basis_utils.py:14-29).

    The reference stores every intermediate bracket in a float32 array, so the bracket
    ends of order l are the float32-rounded zeros of order l-1 (interlacing property).
    """
    zerosj = np.zeros((n, k), dtype="float32")
    zerosj[0] = np.arange(1, k + 1) * np.pi
    points = np.arange(1, k + n) * np.pi
    racines = np.zeros(k + n - 1, dtype="float32")
    for i in range(1, n):
        for j in range(k + n - 1 - i):
            racines[j] = brentq(lambda r: sp.spherical_jn(i, r), points[j], points[j + 1])
        points = racines
        zerosj[i][:k] = racines[:k]
    return zerosj


@lru_cache(maxsize=None)
def sph_bessel_normalizer(n: int, k: int) -> np.ndarray:
    """N_ln = (0.5 * j_{l+1}(z_ln)^2)^(-1/2), float64 from float32 roots (basis_utils.py:57-66)."""
    z = jn_zeros(n, k)
    out = np.zeros((n, k), dtype=np.float64)
    for l in range(n):
        for i in range(k):
            out[l, i] = 1.0 / np.sqrt(0.5 * sp.spherical_jn(l + 1, z[l, i]) ** 2)
    return out


def sph_harm_prefactor(l: int, m: int) -> float:
    """sqrt((2l+1)/(4 pi) (l-|m|)!/(l+|m|)!)  (basis_utils.py:83-104)."""
    return math.sqrt(
        (2 * l + 1) / (4 * math.pi) * math.factorial(l - abs(m)) / math.factorial(l + abs(m))
    )


def envelope(x: torch.Tensor, p: int) -> torch.Tensor:
    """u(x) = 1 + a x^p + b x^(p+1) + c x^(p+2) for x<1 else 0 (envelope.py:14-29)."""
    a = -(p + 1) * (p + 2) / 2
    b = p * (p + 2)
    c = -p * (p + 1) / 2
    val = 1 + a * x**p + b * x ** (p + 1) + c * x ** (p + 2)
    return torch.where(x < 1, val, torch.zeros_like(x))


def bessel_rbf(d: torch.Tensor, freq: torch.Tensor, cutoff: float, p: int) -> torch.Tensor:
    """(E,) -> (E, num_radial)   (basis_layers.py:45-49)."""
    inv_cutoff = 1 / cutoff
    norm_const = (2 * inv_cutoff) ** 0.5
    d = d[:, None]
    ds = d * inv_cutoff
    return envelope(ds, p) * norm_const * torch.sin(freq * ds) / d


def _sph_jl_series(l: int, x: torch.Tensor, nterms: int = 30) -> torch.Tensor:
    """j_l(x) = x^l/(2l+1)!! * sum_k (-x^2/2)^k / (k! (2l+3)(2l+5)...(2l+2k+1))."""
    dfact = 1.0
    for i in range(1, 2 * l + 2, 2):
        dfact *= i
    q = -0.5 * x * x
    term = torch.ones_like(x)
    acc = torch.ones_like(x)
    for k in range(1, nterms):
        term = term * q / (k * (2 * l + 2 * k + 1))
        acc = acc + term
    return x**l / dfact * acc


def sph_jl_all(L: int, x: torch.Tensor):
    """[j_0(x), ..., j_{L-1}(x)], stable: upward recurrence for x >= l, series below."""
    j0 = torch.sin(x) / x
    out = [j0]
    if L == 1:
        return out
    j1 = torch.sin(x) / (x * x) - torch.cos(x) / x
    up = [j0, j1]
    for l in range(1, L - 1):
        up.append((2 * l + 1) / x * up[l] - up[l - 1])
    for l in range(1, L):
        ser = _sph_jl_series(l, x)
        out.append(torch.where(x < float(l), ser, up[l]))
    return out


def sph_bessel_radial(d: torch.Tensor, num_spherical: int, num_radial: int, cutoff: float, p: int):
    """rbf_env (E, num_spherical, num_radial) = u(d/c) c^-1.5 N_ln j_l(z_ln d/c).

    basis_layers.py:121-128 (same in TensorBasisLayer :241-250) with the lambdified
    sympy formulas of basis_utils.py:47-80 evaluated in closed form.
    """
    z = torch.as_tensor(jn_zeros(num_spherical, num_radial).astype(np.float64)).to(d.dtype)
    nrm = torch.as_tensor(sph_bessel_normalizer(num_spherical, num_radial)).to(d.dtype)
    inv_cutoff = 1 / cutoff
    ds = d * inv_cutoff
    u = envelope(ds, p)
    cols = []
    for l in range(num_spherical):
        # argument differs per (l, n): evaluate only order l at z_ln * ds
        arg = z[l][None, :] * ds[:, None]  # (E, num_radial)
        jl = sph_jl_all(l + 1, arg)[l]
        cols.append(nrm[l][None, :] * jl)
    rbf = torch.stack(cols, dim=1)  # (E, S, R)
    return rbf * (inv_cutoff**1.5) * u[:, None, None]


def legendre_q(L: int, c: torch.Tensor, s: torch.Tensor):
    """Q[l][m] = (-1)^m P_l^m(c)  (phase-free associated Legendre), 0<=m<=l<L, s = sin(theta).

    Recurrences of basis_utils.py:137-159; the reference's P_l^l carries the
    Condon-Shortley factor (1-2l) = -(2l-1) which real_sph_harm cancels with (-1)^m
    (basis_utils.py:226-243), hence the phase-free form here.
    """
    Q = [[None] * (l + 1) for l in range(L)]
    Q[0][0] = torch.ones_like(c)
    for l in range(1, L):
        Q[l][l] = (2 * l - 1) * s * Q[l - 1][l - 1]
    for m in range(0, L - 1):
        Q[m + 1][m] = (2 * m + 1) * c * Q[m][m]
    for l in range(2, L):
        for m in range(l - 1):
            Q[l][m] = ((2 * l - 1) * c * Q[l - 1][m] - (l + m - 1) * Q[l - 2][m]) / (l - m)
    return Q


def real_sph_harm_l0(L: int, theta: torch.Tensor) -> torch.Tensor:
    """(N,) -> (N, L): Y_l0(theta), zero_m_only=True branch (basis_utils.py:221-222)."""
    c = torch.cos(theta)
    s = torch.sin(theta)
    Q = legendre_q(L, c, s)
    return torch.stack([sph_harm_prefactor(l, 0) * Q[l][0] for l in range(L)], dim=1)


def sph_harm_order(L: int):
    """List of (l, m) in the reference's storage order: per l: 0, +1..+l, -l..-1.

    Negative python list indices at basis_utils.py:237 put m=-1 in the LAST slot.
    """
    order = []
    for l in range(L):
        order.append((l, 0))
        for m in range(1, l + 1):
            order.append((l, m))
        for m in range(l, 0, -1):
            order.append((l, -m))
    return order


def real_sph_harm_full(L: int, theta: torch.Tensor, phi: torch.Tensor) -> torch.Tensor:
    """(N,),(N,) -> (N, L^2) real Y_lm(theta, phi) in the reference order (basis_utils.py:224-243)."""
    c = torch.cos(theta)
    s = torch.sin(theta)
    Q = legendre_q(L, c, s)
    cols = []
    for (l, m) in sph_harm_order(L):
        if m == 0:
            cols.append(sph_harm_prefactor(l, 0) * Q[l][0])
        elif m > 0:
            cols.append(math.sqrt(2.0) * sph_harm_prefactor(l, m) * Q[l][m] * torch.cos(m * phi))
        else:
            cols.append(math.sqrt(2.0) * sph_harm_prefactor(l, m) * Q[l][-m] * torch.sin(-m * phi))
    return torch.stack(cols, dim=1)
