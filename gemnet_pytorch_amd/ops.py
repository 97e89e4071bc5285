"""torch.autograd.Function wrappers around the HIP kernel launchers (gemnet_pytorch_amd/kernels.py).

Design rule: the op set is CLOSED under differentiation.  Every backward
is written in terms of `.apply` of ops from this file (plus native torch pointwise arithmetic),
so `torch.autograd.grad(E, R, create_graph=True)` followed by `loss.backward()` — the
second-order pass GemNet's force training needs (gemnet.py:605-611, trainer.py:346) — works
without op-specific double-backward code:

    gather_rows        <->  segsum_rows            (mutual adjoints, linear)
    mm(A,B,ta,tb)       ->  mm                      (GEMM on the f32 MFMA)
    bmm(A,B,ta,tb)      ->  bmm
    bil_reduce / bil_reduce_t / bil_dot             (the three faces of one trilinear form)
    ssilu^(k)           ->  ssilu^(k+1)
    bessel_rbf^(kd,kf), sph_radial^(kd), ylm0^(k), ylm^(kt,kp)  ->  next derivative order

`param_grads(False)` lets GemNet.forward tell the ops that only dE/dR is wanted while it
computes forces, so no weight-gradient GEMMs are launched for them.
"""
import contextlib

import torch

from . import kernels as K

_PARAM_GRADS = True


@contextlib.contextmanager
def param_grads(enabled: bool):
    """While disabled, backward passes skip gradients of weight-like operands."""
    global _PARAM_GRADS
    old = _PARAM_GRADS
    _PARAM_GRADS = enabled
    try:
        yield
    finally:
        _PARAM_GRADS = old


# ------------------------------------------------------------------------ gather <-> segsum
class _Gather(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, ri):
        ctx.ri = ri
        return K.gather(x, ri.idx32)

    @staticmethod
    def backward(ctx, g):
        ri = ctx.ri
        if ri.inverse is not None:
            return _Gather.apply(g, ri.inverse), None
        return _SegSum.apply(g, ri), None


class _SegSum(torch.autograd.Function):
    @staticmethod
    def forward(ctx, y, ri):
        ctx.ri = ri
        perm, seg = ri.csr
        return K.segsum(y, perm, seg, ri.n_rows)

    @staticmethod
    def backward(ctx, g):
        return _Gather.apply(g, ctx.ri), None


def gather_rows(x, ri):
    """y[t] = x[ri.idx[t]]   (interaction_block.py:543,548,678,693; embedding_block.py:70-71)."""
    return _Gather.apply(x, ri)


def segsum_rows(y, ri):
    """x[n] = sum_{t: ri.idx[t]==n} y[t]   (torch_scatter.scatter(reduce='add'): atom_update_block.py:67)."""
    return _SegSum.apply(y, ri)


# ---------------------------------------------------------------------------------- GEMMs
class _MM(torch.autograd.Function):
    """mm(A, B, ta, tb) = (A^T if ta else A) @ (B if tb else B^T)  — tb=False: B is a Linear weight."""

    @staticmethod
    def forward(ctx, A, B, ta, tb):
        ctx.save_for_backward(A, B)
        ctx.ta, ctx.tb = ta, tb
        return K.gemm(A, B, ta, tb)

    @staticmethod
    def backward(ctx, g):
        A, B = ctx.saved_tensors
        ta, tb = ctx.ta, ctx.tb
        gA = gB = None
        # C = a @ b with a = (A^T if ta else A), b = (B if tb else B^T); da = g b^T, db = a^T g.
        if ctx.needs_input_grad[0]:
            if not ta:
                gA = _MM.apply(g, B, False, not tb)   # g @ b^T
            else:
                gA = _MM.apply(B, g, not tb, False)   # (g b^T)^T = b @ g^T
        if ctx.needs_input_grad[1] and _PARAM_GRADS:
            if tb:
                gB = _MM.apply(A, g, not ta, True)    # a^T @ g
            else:
                gB = _MM.apply(g, A, True, not ta)    # (a^T g)^T = g^T @ a
        return gA, gB, None, None


def mm(A, B, ta=False, tb=False):
    return _MM.apply(A, B, ta, tb)


def linear(x, W):
    """x @ W^T with W a torch Linear weight (out, in)   (base_layers.py:46)."""
    return _MM.apply(x, W, False, False)


class _BMM(torch.autograd.Function):
    """bmm(A, B, ta, tb) = (A^T if ta else A) @ (B^T if tb else B), batched over dim 0."""

    @staticmethod
    def forward(ctx, A, B, ta, tb):
        ctx.save_for_backward(A, B)
        ctx.ta, ctx.tb = ta, tb
        return K.bmm(A, B, ta, tb)

    @staticmethod
    def backward(ctx, g):
        A, B = ctx.saved_tensors
        ta, tb = ctx.ta, ctx.tb
        gA = gB = None
        # C = a @ b, a = (A^T if ta else A), b = (B^T if tb else B); da = g b^T, db = a^T g
        if ctx.needs_input_grad[0]:
            gA = _BMM.apply(B, g, tb, True) if ta else _BMM.apply(g, B, False, not tb)
        if ctx.needs_input_grad[1]:
            gB = _BMM.apply(g, A, True, ta) if tb else _BMM.apply(A, g, not ta, False)
        return gA, gB, None, None


def bmm(A, B, ta=False, tb=False):
    return _BMM.apply(A, B, ta, tb)


# ----------------------------------------------------------------------------- activation
class _SSiLU(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, k):
        ctx.save_for_backward(x)
        ctx.k = k
        return K.ssilu(x, k)

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        if ctx.k >= 3:
            raise NotImplementedError("ssilu derivative order > 3")
        return g * _SSiLU.apply(x, ctx.k + 1), None


def ssilu(x, k=0):
    """k-th derivative of ScaledSiLU (base_layers.py:51-58)."""
    return _SSiLU.apply(x, k)


# ------------------------------------------------------------------- bilinear aggregation
class _BilReduce(torch.autograd.Function):
    """Sm[e,s,c] = sum_{t in seg(e)} Y[t,s] x[g(t),c]."""

    @staticmethod
    def forward(ctx, Y, x, sp):
        ctx.save_for_backward(Y, x)
        ctx.sp = sp
        return K.bil_reduce(Y, x, sp)

    @staticmethod
    def backward(ctx, g):
        Y, x = ctx.saved_tensors
        gY = _BilDot.apply(g, x, ctx.sp) if ctx.needs_input_grad[0] else None
        gx = _BilReduceT.apply(Y, g, ctx.sp) if ctx.needs_input_grad[1] else None
        return gY, gx, None


class _BilReduceT(torch.autograd.Function):
    """dx[j,c] = sum_{t: g(t)=j} sum_s Y[t,s] D[r(t),s,c]."""

    @staticmethod
    def forward(ctx, Y, D, sp):
        ctx.save_for_backward(Y, D)
        ctx.sp = sp
        return K.bil_reduce_t(Y, D, sp)

    @staticmethod
    def backward(ctx, g):
        Y, D = ctx.saved_tensors
        gY = _BilDot.apply(D, g, ctx.sp) if ctx.needs_input_grad[0] else None
        gD = _BilReduce.apply(Y, g, ctx.sp) if ctx.needs_input_grad[1] else None
        return gY, gD, None


class _BilDot(torch.autograd.Function):
    """dY[t,s] = sum_c D[r(t),s,c] x[g(t),c]."""

    @staticmethod
    def forward(ctx, D, x, sp):
        ctx.save_for_backward(D, x)
        ctx.sp = sp
        return K.bil_dot(D, x, sp)

    @staticmethod
    def backward(ctx, g):
        D, x = ctx.saved_tensors
        gD = _BilReduce.apply(g, x, ctx.sp) if ctx.needs_input_grad[0] else None
        gx = _BilReduceT.apply(g, D, ctx.sp) if ctx.needs_input_grad[1] else None
        return gD, gx, None


def bil_reduce(Y, x, sp):
    """K1 of SURVEY.md Appendix D (the scatter-to-padded + first bmm of efficient.py:173-177)."""
    return _BilReduce.apply(Y, x, sp)


# ----------------------------------------------------------------------------------- basis
class _BesselRBF(torch.autograd.Function):
    @staticmethod
    def forward(ctx, d, freq, cutoff, p, kd, kf):
        ctx.save_for_backward(d, freq)
        ctx.cfg = (cutoff, p, kd, kf)
        return K.bessel_rbf(d, freq, cutoff, p, kd, kf)

    @staticmethod
    def backward(ctx, g):
        d, freq = ctx.saved_tensors
        cutoff, p, kd, kf = ctx.cfg
        gd = gf = None
        if ctx.needs_input_grad[0]:
            if kd + kf + 1 > 2:
                raise NotImplementedError("bessel_rbf: third-order derivative requested")
            gd = (g * _BesselRBF.apply(d, freq, cutoff, p, kd + 1, kf)).sum(dim=1)
        if ctx.needs_input_grad[1] and _PARAM_GRADS:
            if kf >= 1 or kd + 1 > 2:
                raise NotImplementedError("bessel_rbf: second derivative w.r.t. frequencies requested")
            gf = (g * _BesselRBF.apply(d, freq, cutoff, p, kd, kf + 1)).sum(dim=0)
        return gd, gf, None, None, None, None


def bessel_rbf(d, freq, cutoff, p):
    """(E,) -> (E, num_radial)   (BesselBasisLayer.forward, basis_layers.py:45-49)."""
    return _BesselRBF.apply(d, freq, float(cutoff), int(p), 0, 0)


class _SphRadial(torch.autograd.Function):
    @staticmethod
    def forward(ctx, d, z, nrm, cutoff, p, kd):
        ctx.save_for_backward(d, z, nrm)
        ctx.cfg = (cutoff, p, kd)
        return K.sph_radial(d, z, nrm, cutoff, p, kd)

    @staticmethod
    def backward(ctx, g):
        d, z, nrm = ctx.saved_tensors
        cutoff, p, kd = ctx.cfg
        if kd >= 2:
            raise NotImplementedError("sph_radial: third-order derivative requested")
        gd = (g * _SphRadial.apply(d, z, nrm, cutoff, p, kd + 1)).sum(dim=(1, 2))
        return gd, None, None, None, None, None


def sph_radial(d, z, nrm, cutoff, p):
    """(E,) -> (E, S, R): u(d/c) c^-1.5 N_ln j_l(z_ln d/c)   (basis_layers.py:121-128)."""
    return _SphRadial.apply(d, z, nrm, float(cutoff), int(p), 0)


class _Ylm0(torch.autograd.Function):
    @staticmethod
    def forward(ctx, theta, S, k):
        ctx.save_for_backward(theta)
        ctx.cfg = (S, k)
        return K.ylm0(theta, S, k)

    @staticmethod
    def backward(ctx, g):
        (theta,) = ctx.saved_tensors
        S, k = ctx.cfg
        if k >= 2:
            raise NotImplementedError("ylm0: third-order derivative requested")
        return (g * _Ylm0.apply(theta, S, k + 1)).sum(dim=1), None, None


def ylm0(theta, S):
    """(T,) -> (T, S): Y_l0(theta)   (SphericalBasisLayer angular part, basis_layers.py:130-131)."""
    return _Ylm0.apply(theta, int(S), 0)


class _Ylm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, theta, phi, S, kt, kp):
        ctx.save_for_backward(theta, phi)
        ctx.cfg = (S, kt, kp)
        return K.ylm(theta, phi, S, kt, kp)

    @staticmethod
    def backward(ctx, g):
        theta, phi = ctx.saved_tensors
        S, kt, kp = ctx.cfg
        if kt + kp >= 2:
            raise NotImplementedError("ylm: third-order derivative requested")
        gt = (g * _Ylm.apply(theta, phi, S, kt + 1, kp)).sum(dim=1) if ctx.needs_input_grad[0] else None
        gp = (g * _Ylm.apply(theta, phi, S, kt, kp + 1)).sum(dim=1) if ctx.needs_input_grad[1] else None
        return gt, gp, None, None, None


def ylm(theta, phi, S):
    """(Q,),(Q,) -> (Q, S^2) real Y_lm in the reference order   (TensorBasisLayer, basis_layers.py:269)."""
    return _Ylm.apply(theta, phi, int(S), 0, 0)
