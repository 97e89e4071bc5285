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

import os
import threading

import torch

from . import _lib
from . import kernels as K

# The switches below (`param_grads`, `fused_first_order`, `weight_cache`, `train2`, `wgrad_queue`, ...) are host-side state of
# ONE pass that the caller's thread sets and the autograd engine's thread reads — process-wide by necessity (the engine's
# worker inherits nothing thread-local).  Two Python threads driving two models at once would interleave them, so every
# section that sets them — GemNet.forward (with the force pass inside it) and the Trainer's / TrainStep's loss.backward() —
# runs under this one re-entrant lock: passes of different threads serialise on the host (the GPU work they enqueue still
# overlaps on their streams).  What is NOT under the lock and needs none: the arithmetic mode (thread-local + recorded per
# autograd node, kernels.use_mode / _in_mode below) and the C ABI itself (no library state, include/gemnet_hip.h).
_EXCLUSIVE = threading.RLock()


def exclusive():
    """The lock of a pass that sets the process-wide host switches of this module (re-entrant)."""
    return _EXCLUSIVE


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


# Weight-gradient queue (training/wgrad_queue.py): in the final, non-differentiable backward of a training step the
# dW = X^T Y products are leaves of the graph; with a queue set they are deferred into one grouped launch.
_WGRAD_QUEUE = None


@contextlib.contextmanager
def wgrad_queue(queue):
    global _WGRAD_QUEUE
    prev, _WGRAD_QUEUE = _WGRAD_QUEUE, queue
    try:
        yield
    finally:
        _WGRAD_QUEUE = prev


def _queueable(P):
    """May the weight gradient of P be deferred into the grouped launch?  Leaf parameters and their column-block views
    (training/wgrad_queue.py::grad_target) in the final, non-differentiable backward."""
    if _WGRAD_QUEUE is None or torch.is_grad_enabled():
        return False
    from .training.wgrad_queue import grad_target
    return grad_target(P) is not None


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


class _CbfProject(torch.autograd.Function):
    """out = Dense_W((rad[ri] * y[:, :, None]).reshape(I, S R)) in one pass (csrc/cbf.hip); W frozen, first-order backward."""

    @staticmethod
    def forward(ctx, rad, y, W, ri):
        out = K.cbf_project_fwd(rad, ri.idx32, y, W)
        ctx.save_for_backward(rad, y, W)
        ctx.ri = ri
        ctx.acc = _acc_join(rad)
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        rad, y, W = ctx.saved_tensors
        perm, seg = ctx.ri.csr
        assert perm is None, "cbf_project: the interaction-edge index of the intermediate triplets is sorted"
        g_rad, g_y = K.cbf_project_bwd(g.contiguous(), rad, seg, y, W)
        acc = ctx.acc
        if acc is not None:      # (rad has one consumer today; kept for symmetry with the other radial tables)
            prev, last = acc.enter()
            if prev is not None:
                g_rad = prev.add_(g_rad)
            acc.leave(g_rad, last)
            if not last:
                g_rad = None
        return g_rad, g_y, None, None


USE_CBF_PROJECT = os.environ.get("GEMNET_CBF_PROJECT", "1") == "1"


def cbf_project(rad, ri, y, W):
    """Circular basis of GemNet-Q's intermediate triplets through mlp_cbf4 (basis_layers.py:119-131 + a bias-free Dense), fused
    when the weight is frozen (forward+force inference) and the index is sorted by interaction edge; None otherwise (the caller
    keeps the composite form)."""
    if not (USE_CBF_PROJECT and constant_weights() and ri.is_sorted and K.cbf_project_supported(rad, y, W)):
        return None
    return _CbfProject.apply(rad, y, contiguous_weight(W), ri)


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
            if not ta and _queueable(B):
                if tb:
                    _WGRAD_QUEUE.add(B, A, g)         # B.grad += A^T @ g
                else:
                    _WGRAD_QUEUE.add(B, g, A)         # B.grad += g^T @ A
            elif tb:
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
class _PM(torch.autograd.Function):
    """y = c * ssilu^(k)(z) * a * b   (k = -1: no activation factor; a, b optional) as ONE launch (gn_pm_f32).

    Closed under differentiation: d/dz is the same form with k+1 and one more factor, d/da is the same form with a
    replaced by the incoming gradient.  The kernel takes three factors, which is exactly what the double backward
    of `ssilu(z) * mul` needs (f''(z) * mul * g * gg); a third derivative is never taken on this path."""

    @staticmethod
    def forward(ctx, k, c, z, *factors):
        ctx.k, ctx.c = k, c
        ctx.n_f = len(factors)
        ctx.save_for_backward(z, *factors)
        f = list(factors) + [None] * (3 - len(factors))
        return K.pm(z, k, f[0], f[1], f[2], c)

    @staticmethod
    def backward(ctx, g):
        z, *factors = ctx.saved_tensors
        k, c = ctx.k, ctx.c
        need = ctx.needs_input_grad  # (k, c, z, *factors)
        gz = None
        if k >= 0 and need[2]:
            if len(factors) >= 3:
                raise NotImplementedError("pm: more than three tensor factors (third derivative of the activation)")
            if k >= 3:
                raise NotImplementedError("ssilu derivative order > 3")
            gz = _PM.apply(k + 1, c, z, *factors, g)
        gf = []
        for i in range(len(factors)):
            if need[3 + i]:
                others = [f for j, f in enumerate(factors) if j != i]
                gf.append(_PM.apply(k, c, z, *others, g))
            else:
                gf.append(None)
        return (None, None, gz, *gf)


def pm(z, k=0, *factors, c=1.0):
    """c * ssilu^(k)(z) * prod(factors); k = -1 drops the activation factor (then z may be None)."""
    factors = [f for f in factors if f is not None]
    if k < 0:
        if not factors:
            raise ValueError("pm without activation needs at least one factor")
        if len(factors) == 1 and c == 1.0:
            return factors[0]
    return _PM.apply(int(k), float(c), z if k >= 0 else None, *factors)


def ssilu(x, k=0):
    """k-th derivative of ScaledSiLU (base_layers.py:51-58)."""
    return _PM.apply(int(k), 1.0, x)


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


# =========================================================================================
# Fused first-order path
# =========================================================================================
# When no second-order graph is needed (inference forces, direct-force training) every Dense is ONE
# GEMM launch forward (gather-add / ScaledSiLU / Hadamard / scale / residual(s) in the epilogue) and
# ONE backward (the activation derivative applied to the A operand while it is staged into LDS),
# instead of the ~6 launches of the composite closure above.  `fused_first_order(False)` — set by
# GemNet.forward when it builds the force with create_graph=True — routes the same calls through
# the differentiable composite ops instead.
_FUSED = True


@contextlib.contextmanager
def fused_first_order(enabled: bool):
    global _FUSED
    old = _FUSED
    _FUSED = enabled
    try:
        yield
    finally:
        _FUSED = old


def is_fused():
    return _FUSED


def chain_mode(mode):
    """Select the arithmetic of the Dense stacks (kernels.CHAIN_MODES) for the launches THIS THREAD issues inside the block;
    None keeps the current one (kernels.use_mode: thread-local, not a process global).  Packed weights are cached per weight
    and plane format (kernels.SPLIT_FORMAT: the bf16-plane modes share one form, "h3" has its own)."""
    return K.use_mode(mode)


def _in_mode(backward):
    """Decorator of the `backward` of a Function whose forward recorded `ctx.mode = K.current_mode()`: the backward runs on the
    autograd engine's thread — inside the forward's `chain_mode` block (inference forces) or long after it closed and another
    model with another `matmul_precision` ran (direct-force training: loss.backward()) — and issues its launches in the
    arithmetic of ITS forward, never in whatever a process-global switch happens to hold."""
    import functools

    @functools.wraps(backward)
    def wrapped(ctx, *grads):
        with K.use_mode(ctx.mode):
            return backward(ctx, *grads)
    return wrapped


# Derived-weight cache (transposes / contiguous copies of FROZEN weights).  Entries are keyed by the address
# of the source, so the dict must not outlive the tensors it was filled from: it is owned by a model
# (GemNet._wcache, dropped on _apply / load_state_dict / deepcopy) and only active inside
# `weight_cache(...)`; outside of one nothing is cached (a process-global dict returned another model's
# transposes once the allocator reused a freed weight's address).
_WT_CACHE = None


@contextlib.contextmanager
def weight_cache(cache):
    global _WT_CACHE
    prev, _WT_CACHE = _WT_CACHE, cache
    try:
        yield
    finally:
        _WT_CACHE = prev


def _cached(key, version, make):
    if _WT_CACHE is None:
        return make()
    hit = _WT_CACHE.get(key)
    if hit is not None and hit[0] == version:
        return hit[1]
    if len(_WT_CACHE) > 4096:
        _WT_CACHE.clear()
    val = make()
    _WT_CACHE[key] = (version, val)
    return val


def _frozen(W):
    """May derived forms of W be cached?  Yes when it does not require grad, and also while the weights are treated as
    constants (force-by-autograd inference on a model whose nn.Parameters still require grad — the default after
    `.eval()`): the cache is keyed on `W._version`, and GemNet drops it on every train()/eval() switch, on
    load_state_dict and after a fused optimizer step (whose kernel updates the flat buffer without touching versions)."""
    return (not W.requires_grad) or constant_weights()


def cached_form(tag, W, make):
    """A derived form of the frozen weight W, cached on its address and version inside `weight_cache`."""
    return _cached((tag, W.data_ptr(), tuple(W.shape), tuple(W.stride())), W._version, make)


def transposed(W):
    """W^T contiguous, so backward GEMMs also run the k-contiguous ("NT") pipelined kernel.  Cached
    for frozen weights (inference); recomputed per call for trainable ones."""
    if not _frozen(W):
        return W.detach().t().contiguous()
    return _cached(("t", W.data_ptr(), tuple(W.shape), tuple(W.stride())), W._version,
                   lambda: W.detach().t().contiguous())


class _FusedDense(torch.autograd.Function):
    """y = epilogue(x @ W^T); see gn_gemm_f32.  First-order backward only."""

    @staticmethod
    def forward(ctx, x, W, mul, res, res2, g1, g2, cfg):
        act, alpha, beta, beta2, res_rows, i1, i2 = cfg
        ctx.acc = _acc_join(x)
        need_z = act or mul is not None
        if _frozen(W) and (W.stride(0) % 4 or W.data_ptr() % 16) and W.shape[1] % 4 == 0:
            W = contiguous_weight(W)   # column slice of a wider frozen matrix (edge embedding): copied once, not per call
        out = K.gemm(x, W, act=act, pre_out=need_z, mul=mul, alpha=alpha,
                     res=res, ridx=None if res_rows is None else res_rows.idx32, beta=beta,
                     res2=res2, beta2=beta2,
                     gadd1=g1, gidx1=None if i1 is None else i1.idx32,
                     gadd2=g2, gidx2=None if i2 is None else i2.idx32)
        y, z = out if need_z else (out, None)
        ctx.cfg = cfg
        ctx.has = (mul is not None, res is not None, res2 is not None, g1 is not None, g2 is not None)
        keep_x = W.requires_grad and _PARAM_GRADS
        ctx.save_for_backward(x if keep_x else None, W, z, mul)
        return y

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        x, W, z, mul = ctx.saved_tensors
        act, alpha, beta, beta2, res_rows, i1, i2 = ctx.cfg
        has_mul, has_res, has_res2, has_g1, has_g2 = ctx.has
        need = ctx.needs_input_grad
        g = g.contiguous()
        gx = gW = gmul = gres = gres2 = gg1 = gg2 = None
        c = 1.0
        if has_res2:
            if need[4]:
                gres2 = g * beta2
            c *= beta2
        if has_res:
            c *= beta
            if need[3]:
                t = g * c if c != 1.0 else g
                if res_rows is None:
                    gres = t
                elif res_rows.inverse is not None:
                    gres = K.gather(t, res_rows.inverse.idx32)
                else:
                    gres = K.segsum(t, *res_rows.csr, res_rows.n_rows)
        c *= alpha
        want_w = need[1] and _PARAM_GRADS and x is not None
        want_gmul = has_mul and need[2]
        explicit = has_mul or want_w or (has_g1 and need[5]) or (has_g2 and need[6])
        acc = ctx.acc
        out = prev = None
        last = True
        if acc is not None:
            if need[0]:
                out, prev, last = acc.target((g.shape[0], W.shape[1]), g)
            else:
                gx = acc.skip()
        if explicit:
            dz, gmul = K.dact_mul(g, z, act, mul, c, want_gmul=want_gmul)
            if need[0]:
                gx = K.gemm(dz, transposed(W), res2=prev, out=out)
            if has_g1 and need[5]:
                gg1 = K.segsum(dz, *i1.csr, i1.n_rows)
            if has_g2 and need[6]:
                gg2 = K.segsum(dz, *i2.csr, i2.n_rows)
            if want_w:
                gW = K.gemm(dz, x, True, True)          # dz^T @ x  (N, K)
        elif need[0]:
            gx = K.gemm(g, transposed(W), a_dact_pre=z if act else None, alpha=c, res2=prev, out=out)
        if not last:
            gx = None
        return gx, gW, gmul, gres, gres2, gg1, gg2, None


def dense(x, W, act=False, *, mul=None, alpha=1.0, res=None, res_rows=None, beta=1.0, res2=None, beta2=1.0,
          g1=None, i1=None, g2=None, i2=None):
    """y0 = act(x W^T + g1[i1] + g2[i2]) (* mul) * alpha;  y1 = (y0 + res[res_rows]) * beta;
    y = (y1 + res2) * beta2   — every optional stage skipped when its tensor is None."""
    if _FUSED:
        return _FusedDense.apply(x, W, mul, res, res2, g1, g2, (bool(act), float(alpha), float(beta), float(beta2),
                                                                res_rows, i1, i2))
    if (_TRAIN2 and USE_STACKS and mul is None and alpha == 1.0 and res_rows is None and W.dim() == 2
            and W.shape[0] % 16 == 0 and W.shape[0] <= 128 and W.shape[1] % 4 == 0 and W.shape[1] <= 128
            and x.dim() == 2 and x.dtype == W.dtype and (x.is_cuda or K.current_mode() != "f32")):
        # a single Dense as a one-GEMM stack: twice differentiable, one launch per sweep (ops_train.py)
        from . import ops_train
        return ops_train.stack(x, first=dict(W=W, act=act, res=res, beta=beta, res2=res2, beta2=beta2,
                                             g1=g1, i1=i1, g2=g2, i2=i2))
    z = linear(x, W)
    if g1 is not None:
        z = z + gather_rows(g1, i1)
    if g2 is not None:
        z = z + gather_rows(g2, i2)
    # ((f(z) * mul * alpha + res) * beta + res2) * beta2 with the scalars folded: one fused pointwise launch for the
    # activation / Hadamard / scale and one `add` per residual
    c = alpha * (beta if res is not None else 1.0) * (beta2 if res2 is not None else 1.0)
    if act or mul is not None or c != 1.0:
        y = pm(z, 0 if act else -1, *((z,) if not act else ()), mul, c=c)
    else:
        y = z
    if res is not None:
        r = res if res_rows is None else gather_rows(res, res_rows)
        y = torch.add(y, r, alpha=beta * (beta2 if res2 is not None else 1.0))
    if res2 is not None:
        y = torch.add(y, res2, alpha=beta2)
    return y


USE_AGGREGATE = os.environ.get("GEMNET_AGGREGATE", "1") == "1"


class _RbfAggregate(torch.autograd.Function):
    """out[a] = scale * sum_{e -> a} m[e] * (W rbf[e]): Dense(rbf) + Hadamard + scatter-add of AtomUpdateBlock /
    OutputBlock (atom_update_block.py:60-68) as ONE pass forward and ONE pass backward; W constant (force pass)."""

    @staticmethod
    def forward(ctx, m, rbf, W, ri, scale):
        perm, seg = ri.csr
        ctx.acc_m, ctx.acc_rbf = _acc_join(m, cross=True), _acc_join(rbf)
        ctx.save_for_backward(m, rbf, W)
        ctx.ri, ctx.scale = ri, scale
        return K.rbf_aggregate_fwd(m, rbf, W, perm, seg, ri.n_rows, scale)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        m, rbf, W = ctx.saved_tensors
        need = ctx.needs_input_grad
        acc_m, acc_r = ctx.acc_m, ctx.acc_rbf
        prev_m, last_m = acc_m.enter() if acc_m is not None else (None, True)
        prev_r, last_r = acc_r.enter() if acc_r is not None else (None, True)
        g_m, g_rbf = K.rbf_aggregate_bwd(g.contiguous(), m, rbf, W, ctx.ri.idx32, ctx.scale,
                                         want_m=need[0] or acc_m is not None, want_rbf=need[1] or acc_r is not None,
                                         acc_m=prev_m, acc_rbf=prev_r)
        if acc_m is not None:
            acc_m.leave(g_m, last_m)
        if acc_r is not None:
            acc_r.leave(g_rbf, last_r)
        return g_m if last_m else None, g_rbf if last_r else None, None, None, None


def rbf_aggregate(m, rbf, W, ri, scale):
    """Fused path of `AtomUpdateBlock._aggregate` for constant weights; None when the shapes are not the fused ones."""
    if _TRAIN2 and USE_AGGREGATE and K.rbf_aggregate_supported(m, rbf, W):
        from . import ops_train
        return ops_train.rbf_aggregate(m, rbf, W, ri, scale)
    if not (USE_AGGREGATE and constant_weights() and K.rbf_aggregate_supported(m, rbf, W)):
        return None
    return _RbfAggregate.apply(m, rbf, contiguous_weight(W), ri, float(scale))


class GradSink:
    """Running sum of the gradient of one tensor that several fused ops consume (the angular basis is shared by all
    interaction blocks).  Each consumer adds its contribution into `buf` inside its own kernel and returns None to
    autograd, except the consumer whose backward runs last, which returns the buffer — so autograd's own
    accumulation (three (T,S)-sized adds per step: 1.8 GB each for the quadruplet basis) never happens.

    Re-entrant: `consumers` is fixed by the forward; every backward pass over the graph (one per target when
    GemNet.forward differentiates several energies with retain_graph=True) counts its own `left` down from
    `consumers` and clears the running state when it reaches zero."""

    def __init__(self):
        self.consumers = 0  # fused bilinear layers that consumed the tensor in the forward
        self.left = 0       # consumers whose backward has not run yet in the CURRENT backward pass
        self.buf = None
        self.pending = []   # (dSm_b, x_b) of the consumers whose Y gradient is deferred to one combined pass
        self.stream = None  # accumulate_gradient: HIP stream of the participating consumers
        self.cross = False  # consumers on several streams take part (events order their writes, see enter / leave)
        self.buf_stream = self.buf_event = None
        self.task = None    # id of the backward pass (autograd graph task) the running state belongs to
        self.home = None    # HIP stream the shared tensor was produced on (= the stream its producer's backward runs on)

    def arrive(self):
        """Called once per consumer backward; returns True for the last consumer of this pass."""
        task = torch._C._current_graph_task_id()
        if self.left == 0 or task != self.task:
            # first consumer of a new pass — also after a pass that died half way (an exception in some backward):
            # its leftovers must not be mistaken for this pass's running sum
            self.task = task
            self.left = self.consumers
            self.buf = None
            self.pending = []
            # the sum only reaches autograd through the LAST consumer: a pass that prunes one of them (a gradient
            # of something that does not depend on every consumer) must fail loudly, not return a partial sum
            torch.autograd.Variable._execution_engine.queue_callback(self._end_of_pass)
        self.left -= 1
        return self.left == 0

    def _end_of_pass(self):
        if self.left != 0:
            missing, self.left, self.buf, self.pending = self.left, 0, None, []
            raise RuntimeError(
                f"shared-gradient accumulation: {missing} of {self.consumers} fused consumers did not run in this "
                "backward pass, the gradient would be incomplete; set GEMNET_GRAD_ACC=0 for such graphs")

    def target(self, shape, like):
        """-> (out, prev, last): the tensor this consumer's kernel writes, the running sum it must add in the same
        kernel (None for the first consumer of the pass; `out is prev` otherwise: in place, element by element) and
        whether this consumer hands the sum to autograd."""
        prev, last = self.enter()
        out = prev if prev is not None else torch.empty(shape, device=like.device, dtype=like.dtype)
        self.leave(out, last)
        return out, prev, last

    def enter(self):
        """-> (running sum so far or None, is this the last consumer of the pass); pair with `leave`.
        A consumer on ANOTHER stream than the one that wrote the running sum last (an output block on the side stream
        followed by the interaction block on the main stream) first orders itself behind that write and tells the
        allocator that the buffer is in use on its stream too."""
        last = self.arrive()
        buf = self.buf
        if buf is not None and self.cross and buf.is_cuda:
            cur = torch.cuda.current_stream(buf.device)
            if self.buf_stream is not None and self.buf_stream != cur.cuda_stream:
                cur.wait_event(self.buf_event)
                buf.record_stream(cur)
        return buf, last

    def leave(self, out, last):
        self.buf = None if last else out
        if self.cross and not last and out is not None and out.is_cuda:
            cur = torch.cuda.current_stream(out.device)
            self.buf_stream = cur.cuda_stream
            self.buf_event = torch.cuda.Event()
            self.buf_event.record(cur)

    def skip(self):
        """A consumer that has no contribution in this pass (undefined incoming gradient)."""
        last = self.arrive()
        out, self.buf = self.buf, (None if last else self.buf)
        return out if last else None


def share_gradient(t):
    """Mark tensor `t` (a non-leaf that requires grad) as shared between fused bilinear layers of one forward."""
    if t.requires_grad and _FUSED:
        t._gn_sink = GradSink()
        if t.is_cuda:
            t._gn_sink.home = torch.cuda.current_stream(t.device)
    return t


# The combined Y gradient of all consumers of a shared basis (bil_dy_multi: 75 us at the headline shape) is launched by the
# LAST consumer's backward, in front of the rest of that block's adjoint, although only the basis' own backward needs it.
# When the basis was produced on another stream than the consumers run on (the forked head of GemNet.forward: autograd
# replays the producer's backward on that stream), the launch goes to THAT stream: ordered in front of its consumer by
# stream order, beside the block's remaining adjoint kernels instead of in front of them.
USE_LATE_DY = os.environ.get("GEMNET_LATE_DY", "0") == "1"   # measured: profiles/r5_late_dy_ab.txt (no gain: off)


def _on_home_stream(sink, inputs, fn):
    home = sink.home
    t0 = inputs[0]
    if not (USE_LATE_DY and home is not None and t0.is_cuda):
        return fn()
    cur = torch.cuda.current_stream(t0.device)
    if home.cuda_stream == cur.cuda_stream:
        return fn()
    home.wait_stream(cur)
    with torch.cuda.stream(home):
        out = fn()
    for t in inputs:
        t.record_stream(home)
    return out


USE_GRAD_ACC = os.environ.get("GEMNET_GRAD_ACC", "1") == "1"


def accumulate_gradient(t, stream=None):
    """Mark the activation `t` as consumed by several fused ops: each of them adds its gradient contribution to the
    running sum inside its own backward kernel (residual input of the last GEMM / an accumulate flag) and the last one
    returns the sum, so the autograd engine never launches its own elementwise adds (32 per forward+force step of the
    4-block model, 4-6 us each: tools/exp/grad_fanin.py, tools/exp/aten_ops.py).  Consumers on another HIP stream than
    the first one (the output blocks) do not take part: their gradient reaches `t` through autograd as before."""
    if (_FUSED or _TRAIN2) and USE_GRAD_ACC and t.requires_grad:
        # a NEW running sum per call: a long-lived tensor (a leaf fed to a block again and again) must not count the
        # consumers of an earlier forward; the ops of that forward keep their reference to the sink they joined
        t._gn_acc = GradSink()
        # the stream of the participating CONSUMERS (default: the current one); the producer may live elsewhere
        if stream is not None:
            t._gn_acc.stream = stream.cuda_stream
        else:
            t._gn_acc.stream = torch.cuda.current_stream(t.device).cuda_stream if t.is_cuda else 0
    return t


# Off by default: it removes five (E, 128) adds per step but the main-stream consumers then wait for the output block's
# backward BEFORE they start instead of meeting it in one add at the end — same-box A/B on MI355X (profiles/r3_ab.txt):
# 2.789 / 2.792 ms off vs 2.792 / 2.791 ms on.  No gain, so the simpler ordering stays.
USE_CROSS_ACC = os.environ.get("GEMNET_CROSS_ACC", "0") == "1"


def _acc_join(t, cross=False):
    """Forward side of `accumulate_gradient`: register the calling fused op as a consumer of `t`.  A consumer on another
    stream than the sink's only takes part when it asks for it (`cross`: the output block's aggregation of the block
    input m — its backward runs first, on the side stream, and the main-stream consumers then add into its result behind
    an event instead of through a separate elementwise add of the autograd engine)."""
    acc = getattr(t, "_gn_acc", None)
    if acc is None:
        return None
    st = torch.cuda.current_stream(t.device).cuda_stream if t.is_cuda else 0
    if acc.stream != st:
        if not (cross and USE_CROSS_ACC and constant_weights()):
            return None
        acc.cross = True
    acc.consumers += 1
    return acc


class _FusedBilinear(torch.autograd.Function):
    """out = alpha * K3(K2(rbf_W1, K1(sph, x)))  (SURVEY.md Appendix D), first-order backward."""

    @staticmethod
    def forward(ctx, rbf_W1, sph, x, W, sp, alpha):
        ctx.sink = getattr(sph, "_gn_sink", None)
        if ctx.sink is not None:
            ctx.sink.consumers += 1
        ctx.acc_B = _acc_join(rbf_W1)
        C, I, O = W.shape
        keep_p = W.requires_grad and _PARAM_GRADS
        ctx.ang = K.is_angle_form(sph, rbf_W1.shape[1])
        if ctx.ang and (C, I) != (32, 32):
            raise ValueError("the angle-form tensor basis needs emb_size_quad = emb_size_sbf = 32")
        if not keep_p and not ctx.ang and K.bil_fused_fwd_supported(sph.shape[1], C, I, O):
            # (K3 on the fp16 pipe keeps P unscaled in fp16 planes: only under the fp16-plane Dense arithmetic and its
            # overflow guard; the adjoint scales its cotangent rows per edge and has no range limit)
            Sm, out = K.bil_fused_fwd(sph, x, rbf_W1, bilinear_weight(W, True), sp, alpha,   # K1 + K2 + K3, P stays in LDS
                                      W2T_planes=bilinear_weight_planes(W) if K.current_mode() == "h3" else None)
            P = None
        else:
            Sm, P = K.bil_reduce_project(sph, x, rbf_W1, sp)    # K1 + K2 in one launch: (E,S,C), (E,I,C)
            out = K.gemm(P.reshape(-1, I * C), bilinear_weight(W, True), alpha=alpha)
        ctx.save_for_backward(rbf_W1, sph, x, W, Sm, P if keep_p else None)
        ctx.sp, ctx.alpha = sp, alpha
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        rbf_W1, sph, x, W, Sm, P = ctx.saved_tensors
        C, I, O = W.shape
        sp, alpha = ctx.sp, ctx.alpha
        need = ctx.needs_input_grad
        g = g.contiguous()
        sink = ctx.sink
        # K3^T + the gB / dSm part in ONE launch when the Y gradient is deferred anyway (dP stays in LDS)
        fused_tail = (need[1] and not ctx.ang and sink is not None and sink.consumers <= 4
                      and tuple(Sm.shape[1:]) == (7, 64) and K.bil_fused_bwd_supported(Sm.shape[1], C, I, O)
                      and not (need[3] and _PARAM_GRADS and P is not None))
        dP = None if fused_tail else \
            K.gemm(g, bilinear_weight(W, False), alpha=alpha).reshape(-1, I, C)   # g @ W2^T: W2 is (N=I*C, K=O)
        accB = ctx.acc_B   # running gradient of the radial part of the basis (shared by the blocks)
        prevB, lastB = accB.enter() if accB is not None else (None, True)
        if need[1] and ctx.ang and (sink is None or sink.consumers > 4):
            # angle form without a shared sink: this block's angle gradient alone
            gB, dSm, _ = K.bil_project_bwd(dP, Sm, rbf_W1, x, sp, want_dY=False, gB_accum=prevB)
            gsph = K.bil_dy_multi([dSm], [x], sp, ang=sph)
            if sink is not None:
                sink.arrive()
        elif sink is not None and need[1] and tuple(Sm.shape[1:]) in ((49, 32), (7, 64)) and sink.consumers <= 4:
            # gB and dSm now, the Y gradient of all consumers of this basis in ONE pass when the last one arrives
            if fused_tail:
                gB, dSm = K.bil_fused_bwd(g, bilinear_weight(W, False), Sm, rbf_W1, alpha, gB_accum=prevB,
                                          W2_planes=bilinear_weight_planes(W, False))
            else:
                gB, dSm, _ = K.bil_project_bwd(dP, Sm, rbf_W1, x, sp, want_dY=False, gB_accum=prevB)
            last = sink.arrive()
            sink.pending.append((dSm, x))
            gsph = None
            if last:
                ds, xs = [d for d, _ in sink.pending], [xx for _, xx in sink.pending]
                gsph = _on_home_stream(sink, ds + xs + [sph], lambda: K.bil_dy_multi(ds, xs, sp, ang=sph if ctx.ang else None))
                sink.pending = []
        elif sink is not None and need[1]:
            # the Y gradient is summed across the consumers of `sph` inside the kernel (see GradSink)
            last = sink.arrive()
            gB, dSm, sink.buf = K.bil_project_bwd(dP, Sm, rbf_W1, x, sp, dY_accum=sink.buf, gB_accum=prevB)
            gsph = sink.buf if last else None
            if last:
                sink.buf = None
        else:
            gB, dSm, gsph = K.bil_project_bwd(dP, Sm, rbf_W1, x, sp, gB_accum=prevB)      # 2 bmm + bil_dot in one launch
        if accB is not None:
            accB.leave(gB, lastB)
            if not lastB:
                gB = None
        gx = gW = None
        if not need[0]:
            gB = None
        if not need[1]:
            gsph = None
        if need[2]:
            gx = K.bil_reduce_t(sph, dSm, sp)
        if need[3] and _PARAM_GRADS and P is not None:
            gW2 = K.gemm(P.reshape(-1, I * C), g, True, True, alpha=alpha)   # P^T @ g  (I*C, O)
            gW = gW2.reshape(I, C, O).permute(1, 0, 2)
        return gB, gsph, gx, gW, None, None


def bilinear_weight(W, transposed_form):
    """The (C, I, O) weight of the bilinear layer as W2 (I*C, O) with rows (i, c) — or W2^T — contiguous; both are
    permuted COPIES, cached on the owning parameter's version when it is frozen."""
    C, I, O = W.shape

    def make():
        W2 = W.detach().permute(1, 0, 2).reshape(I * C, O)
        return W2.t().contiguous() if transposed_form else W2
    if not _frozen(W):
        return make()
    return _cached(("bilT" if transposed_form else "bil", W.data_ptr(), tuple(W.shape)), W._version, make)


def bilinear_weight_planes(W, transposed_form=True):
    """W2^T (or W2) of a FROZEN bilinear weight as two fp16 planes in MFMA fragment order (K3 of the fused forward / its
    transpose in the fused adjoint on the fp16 matrix pipe, kernels.bil_fused_fwd / bil_fused_bwd); None for a trainable
    weight (its planes would have to be repacked every step) and off the device."""
    if not (_frozen(W) and W.is_cuda and K.USE_K3_F16):
        return None
    return _cached(("bilTp" if transposed_form else "bilp", W.data_ptr(), tuple(W.shape)), W._version,
                   lambda: K.pack_weight_split(bilinear_weight(W, transposed_form), fmt=1))


def bilinear(rbf_W1, sph, x, W, sp, alpha=1.0):
    """efficient.py:159-189: out[e,o] = alpha * sum_{t in seg(e)} sum_{s,i,c} sph[t,s] rbf_W1[e,s,i] x[g(t),c] W[c,i,o]."""
    if _FUSED:
        return _FusedBilinear.apply(rbf_W1, sph, x, W, sp, float(alpha))
    C, I, O = W.shape
    if _TRAIN2 and USE_TRAIN2_BILINEAR and K.is_angle_form(sph, rbf_W1.shape[1]):
        # the tensor basis in angle form (GemNet.forward chose it: quad_train2_enabled): fused twins on the *_ang kernels
        from . import ops_train
        return ops_train.bilinear_ang(rbf_W1, sph, x, W, sp, alpha)
    if _TRAIN2 and USE_TRAIN2_BILINEAR and K.bil_train_supported(sph.shape[1], C, I):
        from . import ops_train
        return ops_train.bilinear(rbf_W1, sph, x, W, sp, alpha)
    Sm = bil_reduce(sph, x, sp)
    P = bmm(rbf_W1, Sm, True, False)
    W2 = W.permute(1, 0, 2).reshape(I * C, O)
    out = mm(P.reshape(-1, I * C), W2, False, True)
    return out * alpha if alpha != 1.0 else out


# ---------------------------------------------------------------- fused geometry + basis
class _EdgeBasis(torch.autograd.Function):
    """(R) -> D, V, rbf, rad in one launch; adjoint recomputes the geometry (first-order only)."""

    @staticmethod
    def forward(ctx, R, freq, ri_c, ri_a, z, nrm, cutoff, p, want_V, want_rbf):
        D, V, rbf, rad = K.edge_basis_fwd(R, ri_c.idx32, ri_a.idx32, freq, z, nrm, cutoff, p, want_V, want_rbf)
        if freq is None:
            freq = R.new_zeros(0)
        ctx.save_for_backward(R, freq, z, nrm, D)
        ctx.cfg = (ri_c, ri_a, cutoff, p, want_V, want_rbf)
        ctx.mark_non_differentiable(*[t for t in (V,) if t is not None])
        outs = (D, V if want_V else R.new_zeros(0), rbf if want_rbf else R.new_zeros(0), rad)
        return outs

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gD, gV, g_rbf, g_rad):
        R, freq, z, nrm, D = ctx.saved_tensors
        ri_c, ri_a, cutoff, p, want_V, want_rbf = ctx.cfg
        gR = gf = None
        if not want_rbf:
            g_rbf = None
        if ctx.needs_input_grad[0]:
            W = K.edge_basis_bwd(gD, g_rbf, g_rad, R, ri_c.idx32, ri_a.idx32, freq if want_rbf else None,
                                 z, nrm, cutoff, p)
            gR = K.segsum_multi([(W, *ri_a.csr, 1.0), (W, *ri_c.csr, -1.0)], ri_a.n_rows)
        if want_rbf and ctx.needs_input_grad[1] and _PARAM_GRADS and g_rbf is not None:
            gf = (g_rbf * K.bessel_rbf(D, freq, cutoff, p, 0, 1)).sum(dim=0)
        return gR, gf, None, None, None, None, None, None, None, None


def edge_basis(R, freq, ri_c, ri_a, z, nrm, cutoff, p, want_V=False, want_rbf=True):
    D, V, rbf, rad = _EdgeBasis.apply(R, freq, ri_c, ri_a, z, nrm, float(cutoff), int(p), want_V, want_rbf)
    return D, (V if want_V else None), (rbf if want_rbf else None), rad


class _TripBasis(torch.autograd.Function):
    """(R) -> Y_l0 of every triplet angle in one launch (first-order adjoint)."""

    @staticmethod
    def forward(ctx, R, ri_c, ri_a, ri_b, S):
        Y, _ = K.trip_basis_fwd(R, ri_c.idx32, ri_a.idx32, ri_b.idx32, S)
        ctx.save_for_backward(R)
        ctx.cfg = (ri_c, ri_a, ri_b)
        return Y

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gY):
        (R,) = ctx.saved_tensors
        ri_c, ri_a, ri_b = ctx.cfg
        if not ctx.needs_input_grad[0]:
            return None, None, None, None, None
        Gc, Gb = K.trip_basis_bwd(gY, R, ri_c.idx32, ri_a.idx32, ri_b.idx32)
        gR = K.segsum_multi([(Gc, *ri_c.csr, 1.0), (Gb, *ri_b.csr, 1.0), (Gc, *ri_a.csr, -1.0), (Gb, *ri_a.csr, -1.0)],
                            ri_c.n_rows)
        return gR, None, None, None, None


def trip_basis(R, ri_c, ri_a, ri_b, S):
    return _TripBasis.apply(R, ri_c, ri_a, ri_b, int(S))


# ------------------------------------------------------------------ LDS-resident layer stacks
def constant_weights():
    """True while weights are treated as constants (force-by-autograd inference: the graph of E is
    consumed inside GemNet.forward, so parameter gradients can never be requested)."""
    return _FUSED and not _PARAM_GRADS


# LDS-resident layer stacks (gn_chain_f32) are correct and tested but, measured on MI355X at the
# 32-molecule batch (profiles/r1_chain_bench.txt), one chain op costs 8 us (M = 1024) / 15-17 us
# (E = 18 k) against 5.9 / 12.5 us for the stand-alone 8-wave GEMM: per-op latency inside the chain
# (4 barrier-separated K-steps on one accumulator per wave, 153 VGPRs -> 2 waves/SIMD) is not yet
# lower than a launch.  Off by default until the persistent weight-stationary variant lands.
USE_STACKS = os.environ.get("GEMNET_STACKS", "1") == "1"


def stacks_enabled():
    return USE_STACKS and (constant_weights() or _TRAIN2)


# Fused force TRAINING (ops_train.py): the Dense stacks as twice-differentiable single-launch Functions, everything else
# on the composite ops above.  Set by GemNet.forward while it builds the force with create_graph=True.
_TRAIN2 = False
USE_TRAIN2 = os.environ.get("GEMNET_TRAIN2", "1") == "1"
USE_TRAIN2_BILINEAR = os.environ.get("GEMNET_TRAIN2_BILINEAR", "1") == "1"
_STEP_PACKED = None     # per-forward cache of split-bf16 weight planes (the weights change every step)


class PackRegistry:
    """Split planes (in the format of the requesting sweep, kernels.split_format) of every (weight, orientation) a model's training step asks for, refreshed by ONE grouped launch
    at the start of each step (gn_pack_weight_split_grouped) instead of one launch per weight and sweep (~220 per step).
    The first step packs on demand and registers what it saw; from then on `begin_step` repacks all registered entries in
    place (their buffers — and the device job table — are persistent: captured hipGraphs replay the same launch).
    Keyed by the weight's address / shape / strides: `clear()` when the parameters move (GemNet._apply, load_state_dict)."""

    def __init__(self):
        self.entries = {}      # key -> (W, trans, packed uint8 tensor)
        self.table = None      # device job table of the entries registered when it was built
        self.n_table = 0
        self.total_units = 0

    def clear(self):
        self.entries, self.table, self.n_table, self.total_units = {}, None, 0, 0

    @staticmethod
    def key(W, trans):
        # the packed format follows the chain mode of the requesting sweep ("h3": fp16 planes for S1 / S2, bf16 planes for
        # the loss-scaled S3 / S4 — kernels.linear_mode): one entry per format
        return (W.data_ptr(), tuple(W.shape), tuple(W.stride()), bool(trans), K.split_format())

    def get(self, W, trans):
        k = self.key(W, trans)
        hit = self.entries.get(k)
        if hit is None:
            packed = K.pack_weight_split(W.detach(), trans=trans)
            self.entries[k] = (W.detach(), bool(trans), packed)
            return packed
        return hit[2]

    def begin_step(self):
        """Repack every registered entry (one launch).  Entries registered after the table was built are packed on
        demand in `get` until the table is rebuilt — outside of stream capture only (it needs a host -> device copy)."""
        if not self.entries:
            return
        capturing = torch.cuda.is_current_stream_capturing()
        if self.n_table != len(self.entries):
            if capturing:
                raise RuntimeError("PackRegistry: run one eager training step of this model before capturing a hipGraph")
            self.table, self.total_units = K.pack_job_table([(W, t, p) for W, t, p in self.entries.values()])
            self.n_table = len(self.entries)
        if _lib.TRACE is not None:      # operands of the grouped launch live in the device job table (hbcheck.py)
            ents = list(self.entries.values())[:self.n_table]
            _lib.note(reads=[W for W, _, _ in ents], writes=[p for _, _, p in ents])
        K.pack_weight_split_grouped(self.table, self.n_table, self.total_units)


@contextlib.contextmanager
def train2(enabled: bool, registry=None):
    """`registry`: the model's PackRegistry (GPU only); None packs per weight and step."""
    global _TRAIN2, _STEP_PACKED
    old, old_cache = _TRAIN2, _STEP_PACKED
    _TRAIN2 = bool(enabled)
    if enabled:
        if registry is not None:
            registry.begin_step()
        _STEP_PACKED = registry if registry is not None else {}
    try:
        yield
    finally:
        _TRAIN2, _STEP_PACKED = old, old_cache


def train2_enabled():
    return _TRAIN2


# Second-order position terms d/dR [J^T g] dR of the geometry ops (ops_train._Dist2B / _Angle2B): needed for the gradient
# of the loss w.r.t. the POSITIONS through the force (loss.backward() on a leaf R), never for parameter gradients — the
# data-parallel training step (training/ddp.py) asks for the parameters only and switches them off.
_POSITION_2ND = True


@contextlib.contextmanager
def position_second_order_grads(enabled: bool):
    global _POSITION_2ND
    old, _POSITION_2ND = _POSITION_2ND, bool(enabled)
    try:
        yield
    finally:
        _POSITION_2ND = old


def position_second_order():
    return _POSITION_2ND


# Does the CALLER of this pass differentiate w.r.t. positions a second time?  GemNet.forward knows: it differentiates the energy
# w.r.t. a fresh leaf of its own unless the caller's R already takes part in an autograd graph (model/gemnet.py) — only then can
# anybody ask for d(loss)/dR through the force.  The fused twins provide the second-order POSITION terms of distances and
# triplet angles (dual numbers, csrc/geometry2.hip) but not those of the quadruplet geometry / tensor basis (the Hessian of the
# two angles and the second derivatives of the 49 harmonics): with a position graph requested the quadruplet path stays on the
# composite closure, which is closed under differentiation to any order.
_POSITION_GRAPH = False
USE_TRAIN2_QUAD = os.environ.get("GEMNET_TRAIN2_QUAD", "1") == "1"


@contextlib.contextmanager
def position_graph(enabled: bool):
    global _POSITION_GRAPH
    old = _POSITION_GRAPH
    _POSITION_GRAPH = bool(enabled)
    try:
        yield
    finally:
        _POSITION_GRAPH = old


def quad_train2_enabled(C=32, I=32, S=49):
    """Force training of the quadruplet interaction on the fused angle-form twins (ops_train._QuadAngles2 / _BilinearAng2)?"""
    return (_TRAIN2 and USE_TRAIN2_QUAD and USE_TRAIN2_BILINEAR and USE_QUAD_ANGLES and not _POSITION_GRAPH
            and K.bil_ang_train_supported(S, C, I))


def step_cache():
    """The per-step store of packed weights (None outside `train2`): a stack keeps a reference for its later sweeps."""
    return _STEP_PACKED


def step_packed(W, trans, cache=None):
    """Split fragment form (plane format of the current chain mode) of a TRAINABLE weight (or of its transpose), packed once per training step and shared
    by the four sweeps of that step (ops_train.py); None on the f32 chain kernel / the host emulation.
    `cache`: the store a stack captured in its forward — the S3 / S4 sweeps run inside loss.backward(), after the
    `train2` context of the forward has closed."""
    if K.current_mode() == "f32" or not W.is_cuda:
        return None
    if cache is None:
        cache = _STEP_PACKED
    if cache is None:
        return K.pack_weight_split(W.detach(), trans=trans)
    if isinstance(cache, PackRegistry):
        return cache.get(W, trans)
    key = PackRegistry.key(W, trans)
    hit = cache.get(key)
    if hit is None:
        hit = cache[key] = K.pack_weight_split(W.detach(), trans=trans)
    return hit


def contiguous_weight(W):
    """Row-contiguous copy of a (possibly sliced) frozen weight, cached on its version."""
    if W.is_contiguous():
        return W.detach()
    return _cached(("c", W.data_ptr(), tuple(W.shape), tuple(W.stride())), W._version,
                   lambda: W.detach().contiguous())


def packed_weight(W, trans):
    """Split fragment form (gn_pack_weight_split_fmt, plane format of the current chain mode) of the weight W (trans: of W^T) for the split-operand chain
    kernel.  Cached with the other derived forms when W is frozen; packed per call when W is trainable (a cache keyed
    by the address of a per-call temporary would hand a later weight the earlier one's planes); None when the chain
    runs on the f32 MFMA or on the host emulation."""
    if K.current_mode() == "f32" or not W.is_cuda:
        return None
    if not _frozen(W):
        return K.pack_weight_split(W.detach(), trans=trans)
    return _cached(("pkt" if trans else "pk", K.split_format(), W.data_ptr(), tuple(W.shape), tuple(W.stride())), W._version,
                   lambda: K.pack_weight_split(W.detach(), trans=trans))


def _gemm(prog, W, trans=False, **kw):
    """Append `x @ W^T` (trans: `x @ W`) to a chain program: the fp32 (N,K) form for the f32-MFMA kernel / the host
    emulation and the packed split form for the bf16 kernel."""
    prog.gemm(transposed(W) if trans else contiguous_weight(W), packed=packed_weight(W, trans), **kw)


class _Stack(torch.autograd.Function):
    """[Dense] + ResidualLayer* as ONE launch (gn_chain_f32), activations resident in LDS.

    y0 = ((act(x W0^T + g1[i1] + g2[i2]) + res) * beta + res2) * beta2            (optional leading Dense)
    y_k = ((y_{k-1} + act(act(y_{k-1} W1^T) W2^T)) * s + skip_k) * skip_beta_k     (ResidualLayers)
    Weights are constants here (see constant_weights()); the adjoint is another chain program."""

    @staticmethod
    def forward(ctx, spec, x, res, res2, g1, g2, *skips):
        first, layers, s = spec["first"], spec["layers"], spec["s"]
        M = x.shape[0]
        dev, dt = x.device, x.dtype
        ctx.acc = _acc_join(x)
        prog = K.ChainProgram(M)
        x = x.contiguous()
        prog.load(0, x)
        cur, oth = 0, 1
        # the backward below is first order and only ever needs ssilu'(z): the split-operand kernel stores that factor
        # (from the sigmoid it evaluates anyway) in place of z, and the adjoint program multiplies instead of evaluating
        # exp + rcp per element again (its epilogues were VALU-bound)
        deriv = K.current_mode() != "f32"
        zs = []
        width = x.shape[1]
        n_out = (layers[-1]["W2"].shape[0] if layers else first["W"].shape[0])
        y = torch.empty((M, n_out), device=dev, dtype=dt)
        if first is not None:
            W0 = contiguous_weight(first["W"])
            z0 = torch.empty((M, W0.shape[0]), device=dev, dtype=dt) if first["act"] else None
            rr = first.get("res_rows")
            _gemm(prog, first["W"], a_slot=0, y_slot=1, act=first["act"], pre_deriv=deriv and bool(first["act"]),
                      gadd1=g1, gidx1=None if g1 is None else first["i1"].idx32,
                      gadd2=g2, gidx2=None if g2 is None else first["i2"].idx32,
                      pre_out=z0, res=res, res_rows=None if rr is None else rr.idx32, beta=first["beta"],
                      res2=res2, beta2=first["beta2"], out=None if layers else y)
            zs.append(z0)
            cur, oth = 1, 0
            width = W0.shape[0]
        for k, L in enumerate(layers):
            W1, W2 = contiguous_weight(L["W1"]), contiguous_weight(L["W2"])
            z1 = torch.empty((M, width), device=dev, dtype=dt)
            z2 = torch.empty((M, width), device=dev, dtype=dt)
            last = k + 1 == len(layers)
            _gemm(prog, L["W1"], a_slot=cur, y_slot=oth, act=True, pre_out=z1, pre_deriv=deriv)
            _gemm(prog, L["W2"], a_slot=oth, y_slot=cur, act=True, pre_out=z2, pre_deriv=deriv, res=cur, beta=s,
                      res2=skips[k], beta2=L["skip_beta"], out=y if last else None)
            zs += [z1, z2]
        # tail projections y @ Wt^T of the final rows while they are still in LDS (the concat-Dense atom terms,
        # embedding_block.py:70-72, ride on the atom stack instead of being two more launches)
        tails = []
        for Wt in spec.get("tails", ()):
            Wt_c = contiguous_weight(Wt)
            t = torch.empty((M, Wt_c.shape[0]), device=dev, dtype=dt)
            _gemm(prog, Wt, a_slot=cur, y_slot=-1, out=t)
            tails.append(t)
        if deriv and x.is_cuda and not K.chain_split_supported(prog):   # falls back to the f32 chain kernel: plain z
            deriv = False
            for o in prog.ops:
                o["pre_deriv"] = False
        ctx.deriv = deriv
        ctx.mode = K.current_mode()
        K.chain(prog)
        ctx.set_materialize_grads(False)
        ctx.spec = spec
        ctx.has = (res is not None, res2 is not None, g1 is not None, g2 is not None,
                   tuple(sk is not None for sk in skips))
        ctx.save_for_backward(*[z for z in zs if z is not None])
        ctx.z_mask = [z is not None for z in zs]
        ctx.in_width = x.shape[1]
        ctx.out_shape = (M, n_out)
        # a skip connection fed by the stack's own input (m -> ... + m): its gradient joins dL/dx inside the backward
        # program (parked in LDS slot 2) instead of as a second (M,128) tensor that autograd then adds to gx
        ctx.skip_is_x = spec.get("skip_is_x", -1) if first is not None else -1
        return (y, *tails) if tails else y

    @staticmethod
    @torch.autograd.function.once_differentiable
    @_in_mode
    def backward(ctx, g, *g_tails):
        spec = ctx.spec
        first, layers, s = spec["first"], spec["layers"], spec["s"]
        has_res, has_res2, has_g1, has_g2, has_skips = ctx.has
        need = ctx.needs_input_grad  # (spec, x, res, res2, g1, g2, *skips)
        saved = list(ctx.saved_tensors)
        zs = [saved.pop(0) if m else None for m in ctx.z_mask]
        live = [t for t in (g, *g_tails) if t is not None]
        acc = ctx.acc
        if not live:
            return (None, acc.skip() if acc is not None else None) + (None,) * (4 + len(layers))
        M, width = ctx.out_shape
        dev, dt = live[0].device, live[0].dtype
        if g is None:
            g = torch.zeros((M, width), device=dev, dtype=dt)
        g = g.contiguous()
        prog = K.ChainProgram(M)
        prog.load(0, g)
        cur, oth = 0, 1
        zmode = 1 if ctx.deriv else 0        # the saved tensors hold ssilu'(z) (forward) resp. z
        for Wt, gt in zip(spec.get("tails", ()), g_tails):
            if gt is not None:   # dL/dy += gt @ Wt
                prog.load(oth, gt.contiguous())
                _gemm(prog, Wt, trans=True, a_slot=oth, y_slot=cur, res=cur, beta=1.0)
        g_skips = [None] * len(layers)
        park = ctx.skip_is_x if ctx.in_width == width else -1
        zi = len(zs)
        for k in range(len(layers) - 1, -1, -1):
            L = layers[k]
            z2, z1 = zs[zi - 1], zs[zi - 2]
            zi -= 2
            c = s
            if has_skips[k]:
                if need[6 + k] and k == park and need[1]:
                    prog.scale(2, cur, L["skip_beta"], width=width)   # joins dL/dx in the last GEMM below
                    c = s * L["skip_beta"]
                elif need[6 + k]:
                    g_skips[k] = torch.empty((M, width), device=dev, dtype=dt)
                    prog.scale(cur, cur, L["skip_beta"], out=g_skips[k], width=width)
                else:
                    c = s * L["skip_beta"]
            prog.scale(cur, cur, c, width=width)                  # G = dL/d(x + f(x))
            prog.scale(oth, cur, 1.0, Z=z2, mode=zmode)            # dz2
            _gemm(prog, L["W2"], trans=True, a_slot=oth, y_slot=oth)   # dh1 = dz2 @ W2
            prog.scale(oth, oth, 1.0, Z=z1, mode=zmode)            # dz1
            _gemm(prog, L["W1"], trans=True, a_slot=oth, y_slot=cur, res=cur, beta=1.0)  # dx = dz1 @ W1 + G
        gx = g_res = g_res2 = gg1 = gg2 = None
        if first is not None:
            z0 = zs[0]
            c = 1.0
            if has_res2:
                if need[3]:
                    g_res2 = torch.empty((M, width), device=dev, dtype=dt)
                    prog.scale(cur, cur, first["beta2"], out=g_res2, width=width)
                else:
                    c *= first["beta2"]
            if has_res:
                c *= first["beta"]
                # tied residuals (up_project_pair): res2 + res[rows] with ONE gradient, handed over as dL/d res2
                if need[2] and not first.get("tied"):
                    g_res = torch.empty((M, width), device=dev, dtype=dt)
                    prog.scale(cur, cur, c, out=g_res, width=width)
                    c = 1.0
            want_dz = (has_g1 and need[4]) or (has_g2 and need[5])
            dz0 = torch.empty((M, width), device=dev, dtype=dt) if want_dz else None
            src = cur
            if z0 is not None or c != 1.0 or want_dz:
                # into the other slot: the producing GEMM then emits it as its second output (K.fuse_program)
                prog.scale(oth, cur, c, Z=z0, out=dz0, width=width, mode=zmode)
                src = oth
            last = True
            if need[1]:
                prev = None
                if acc is not None:   # running gradient of x: added in this GEMM's epilogue, in place
                    gx, prev, last = acc.target((M, ctx.in_width), g)
                else:
                    gx = torch.empty((M, ctx.in_width), device=dev, dtype=dt)
                parked = park >= 0 and has_skips[park] and need[6 + park]
                _gemm(prog, first["W"], trans=True, a_slot=src, y_slot=-1, out=gx, res=2 if parked else None, beta=1.0,
                      res2=prev, beta2=1.0)
            elif acc is not None:
                gx = acc.skip()
            K.chain(K.fuse_program(prog))
            if not last:
                gx = None
            if has_g1 and need[4]:
                gg1 = K.segsum(dz0, *first["i1"].csr, first["i1"].n_rows)
            if has_g2 and need[5]:
                gg2 = K.segsum(dz0, *first["i2"].csr, first["i2"].n_rows)
        else:
            gx = torch.empty((M, width), device=dev, dtype=dt)
            prog.store(cur, gx)
            K.chain(K.fuse_program(prog))
            if acc is not None:
                prev, last = acc.enter()
                if prev is not None:
                    gx = prev.add_(gx)
                acc.leave(gx, last)
                if not last:
                    gx = None
        return (None, gx, g_res, g_res2, gg1, gg2) + tuple(g_skips)


def stack(x, first=None, layers=(), s=0.7071067811865475, tails=()):
    """first: dict(W, act, res=None, beta=1, res2=None, beta2=1, g1=None, i1=None, g2=None, i2=None) or None;
    layers: sequence of dict(W1, W2, skip=None, skip_beta=1); tails: weights Wt -> extra outputs y @ Wt^T.
    Returns y, or (y, *tail outputs).  Constant weights (force pass), or the twice-differentiable training form
    (ops_train.stack) inside `train2`."""
    if _TRAIN2:
        from . import ops_train
        return ops_train.stack(x, first=first, layers=layers, s=s, tails=tails)
    assert constant_weights(), "ops.stack is the constant-weight inference path"
    spec = dict(first=None, layers=[dict(W1=L["W1"], W2=L["W2"], skip_beta=float(L.get("skip_beta", 1.0)))
                                    for L in layers], s=float(s), tails=tuple(tails))
    res = res2 = g1 = g2 = None
    if first is not None:
        spec["first"] = dict(W=first["W"], act=bool(first.get("act", False)), beta=float(first.get("beta", 1.0)),
                             beta2=float(first.get("beta2", 1.0)), i1=first.get("i1"), i2=first.get("i2"),
                             res_rows=first.get("res_rows"), tied=bool(first.get("tied", False)))
        if spec["first"]["res_rows"] is not None:
            # the row-gathered residual exists for the pair form only: its gradient is the one of res2 (see _UpPair)
            assert spec["first"]["tied"] and spec["first"]["beta"] == 1.0 and first.get("res2") is not None
        res, res2, g1, g2 = first.get("res"), first.get("res2"), first.get("g1"), first.get("g2")
    skips = [L.get("skip") for L in layers]
    spec["skip_is_x"] = next((k for k, sk in enumerate(skips) if sk is x), -1)
    return _Stack.apply(spec, x, res, res2, g1, g2, *skips)


class _DenseHadamardDown(torch.autograd.Function):
    """x -> Dense_a(x) (.) (rbf W_r^T) * alpha -> Dense_d(.) as ONE launch forward and ONE backward (gn_chain_f32):
    the head of TripletInteraction / QuadrupletInteraction (interaction_block.py:667-675, :531-541: dense_ba/db,
    the radial Hadamard with its scale factor, down_projection).  Constant weights (see constant_weights())."""

    @staticmethod
    def forward(ctx, x, rbf, Wa, Wr, Wd, cfg):
        act_a, act_d, alpha = cfg
        M = x.shape[0]
        dev, dt = x.device, x.dtype
        ctx.acc_x, ctx.acc_rbf = _acc_join(x), _acc_join(rbf)
        x, rbf = x.contiguous(), rbf.contiguous()
        Wa_c, Wr_c, Wd_c = contiguous_weight(Wa), contiguous_weight(Wr), contiguous_weight(Wd)
        z1 = torch.empty((M, Wa_c.shape[0]), device=dev, dtype=dt)
        r = torch.empty((M, Wr_c.shape[0]), device=dev, dtype=dt)
        z3 = torch.empty((M, Wd_c.shape[0]), device=dev, dtype=dt)
        y = torch.empty((M, Wd_c.shape[0]), device=dev, dtype=dt)
        prog = K.ChainProgram(M)
        prog.load(0, x)
        _gemm(prog, Wa, a_slot=0, y_slot=1, act=act_a, pre_out=z1)            # x_a = act(x Wa^T)
        prog.load(0, rbf)
        _gemm(prog, Wr, a_slot=0, y_slot=0, pre_out=r, mul=1, alpha=alpha)     # (rbf Wr^T) * x_a * alpha
        _gemm(prog, Wd, a_slot=0, y_slot=1, act=act_d, pre_out=z3, out=y)
        K.chain(prog)
        ctx.cfg = cfg
        ctx.mode = K.current_mode()
        ctx.save_for_backward(z1, r, z3, Wa, Wr, Wd)
        return y

    @staticmethod
    @torch.autograd.function.once_differentiable
    @_in_mode
    def backward(ctx, g):
        z1, r, z3, Wa, Wr, Wd = ctx.saved_tensors
        act_a, act_d, alpha = ctx.cfg
        need = ctx.needs_input_grad
        M, dev, dt = g.shape[0], g.device, g.dtype
        g = g.contiguous()
        nd, nh = Wd.shape[0], Wa.shape[0]
        # running gradients of x / rbf (ops.accumulate_gradient): joined in the epilogue of the GEMM that produces ours
        px = pr = None
        last_x = last_r = True
        if ctx.acc_x is not None and need[0]:
            gx, px, last_x = ctx.acc_x.target((M, Wa.shape[1]), g)
        else:
            gx = torch.empty((M, Wa.shape[1]), device=dev, dtype=dt) if need[0] else None
        if ctx.acc_rbf is not None and need[1]:
            grbf, pr, last_r = ctx.acc_rbf.target((M, Wr.shape[1]), g)
        else:
            grbf = torch.empty((M, Wr.shape[1]), device=dev, dtype=dt) if need[1] else None
        prog = K.ChainProgram(M)
        prog.load(0, g)
        if act_d:
            prog.scale(0, 0, 1.0, Z=z3, width=nd)                               # dz3
        if need[0] and need[1]:
            # dh = dz3 @ Wd feeds two products: d x_a = dh * r * alpha (the GEMM's own output, slot 1) and
            # d r = dh * x_a * alpha with x_a = act(z1) recomputed (second output, taken before the `mul` stage, slot 0)
            _gemm(prog, Wd, trans=True, a_slot=0, y_slot=1, mul=r, mul_mode=1, alpha=alpha,
                  y2=0, y2_src=1, alpha2=alpha, Z2=z1, mode2=2 if act_a else 1)
            _gemm(prog, Wr, trans=True, a_slot=0, y_slot=-1, out=grbf, res=pr, beta=1.0)   # d rbf = d r @ Wr
            if act_a:
                prog.scale(1, 1, 1.0, Z=z1, width=nh, mode=0)                    # dz1
            _gemm(prog, Wa, trans=True, a_slot=1, y_slot=-1, out=gx, res=px, beta=1.0)
        else:
            _gemm(prog, Wd, trans=True, a_slot=0, y_slot=1)                      # d(hadamard) in slot 1
            if need[1]:
                # d r = dh * x_a * alpha, x_a = act(z1) recomputed;  d rbf = d r @ Wr
                prog.scale(0, 1, alpha, Z=z1, width=nh, mode=2 if act_a else 1)
                _gemm(prog, Wr, trans=True, a_slot=0, y_slot=-1, out=grbf, res=pr, beta=1.0)
            if need[0]:
                prog.scale(1, 1, alpha, Z=r, width=nh, mode=1)                   # d x_a = dh * r * alpha
                if act_a:
                    prog.scale(1, 1, 1.0, Z=z1, width=nh, mode=0)                # dz1
                _gemm(prog, Wa, trans=True, a_slot=1, y_slot=-1, out=gx, res=px, beta=1.0)
        K.chain(K.fuse_program(prog))
        return gx if last_x else None, grbf if last_r else None, None, None, None, None


class _UpPair(torch.autograd.Function):
    """(y_ac, y_ca) = (act(x W_ac^T), act(x W_ca^T)) * alpha, both up projections of the interaction tail
    (interaction_block.py:696-705) in ONE launch.  The reference then forms x3 = (y_ca + y_ac[id_swap]) / sqrt2; here the
    pair is consumed by the next stack as two residuals (`tied`: the swapped one gathered in its epilogue), and that
    stack returns the ONE gradient G of their sum as dL/dy_ca.  The adjoint is one launch as well:
        dL/dx = (G[swap^-1] (.) act'(z_ac)) W_ac + (G (.) act'(z_ca)) W_ca
    with the inverse-permutation gather folded into the tile load.  Constant weights (force pass)."""

    @staticmethod
    def forward(ctx, x, W_ac, W_ca, swap, act, alpha):
        M = x.shape[0]
        dev, dt = x.device, x.dtype
        ctx.acc = _acc_join(x)
        x = x.contiguous()
        N = W_ac.shape[0]
        z_ac = torch.empty((M, N), device=dev, dtype=dt) if act else None
        z_ca = torch.empty((M, N), device=dev, dtype=dt) if act else None
        y_ac = torch.empty((M, N), device=dev, dtype=dt)
        y_ca = torch.empty((M, N), device=dev, dtype=dt)
        prog = K.ChainProgram(M)
        prog.load(0, x)
        _gemm(prog, W_ac, a_slot=0, y_slot=-1, act=act, alpha=alpha, pre_out=z_ac, out=y_ac)
        _gemm(prog, W_ca, a_slot=0, y_slot=-1, act=act, alpha=alpha, pre_out=z_ca, out=y_ca)
        K.chain(prog)
        ctx.set_materialize_grads(False)
        ctx.mode = K.current_mode()
        ctx.save_for_backward(z_ac, z_ca, W_ac, W_ca)
        ctx.swap, ctx.act, ctx.alpha, ctx.width = swap, act, alpha, x.shape[1]
        return y_ac, y_ca

    @staticmethod
    @torch.autograd.function.once_differentiable
    @_in_mode
    def backward(ctx, g_ac, g_ca):
        if g_ac is not None:
            raise RuntimeError("up_project_pair: y_ac may only be consumed together with y_ca as a tied residual pair")
        acc = ctx.acc
        if g_ca is None or not ctx.needs_input_grad[0]:
            return (acc.skip() if acc is not None else None), None, None, None, None, None
        z_ac, z_ca, W_ac, W_ca = ctx.saved_tensors
        G = g_ca.contiguous()
        M = G.shape[0]
        if acc is not None:
            gx, prev, last = acc.target((M, ctx.width), G)
        else:
            gx, prev, last = torch.empty((M, ctx.width), device=G.device, dtype=G.dtype), None, True
        inv = ctx.swap.inverse if ctx.swap.inverse is not None else None
        assert inv is not None, "id_swap is a permutation"
        prog = K.ChainProgram(M)
        mode = 0 if ctx.act else 1
        # slot 1 <- raw rows, slot 0 <- rows * alpha * act'(z): the second tensor of the load
        prog.load(1, G, rows=inv.idx32, y2=0, alpha2=ctx.alpha, Z2=z_ac, mode2=mode)
        _gemm(prog, W_ac, trans=True, a_slot=0, y_slot=2)                        # parked in registers
        prog.load(1, G, y2=0, alpha2=ctx.alpha, Z2=z_ca, mode2=mode)
        _gemm(prog, W_ca, trans=True, a_slot=0, y_slot=-1, res=2, beta=1.0, res2=prev, beta2=1.0, out=gx)
        K.chain(prog)
        return (gx if last else None), None, None, None, None, None


class SwappedPair:
    """y_ca + y_ac[swap] kept as its two terms (ops.up_project_pair -> ops.stack(first=dict(..., tied=True)))."""
    __slots__ = ("y_ca", "y_ac", "swap")

    def __init__(self, y_ca, y_ac, swap):
        self.y_ca, self.y_ac, self.swap = y_ca, y_ac, swap


def up_project_pair(x, W_ac, W_ca, swap, act, alpha):
    assert constant_weights(), "the fused up-projection pair is the constant-weight inference path"
    return _UpPair.apply(x, W_ac, W_ca, swap, bool(act), float(alpha))


def dense_hadamard_down(x, rbf, Wa, Wr, Wd, act_a, act_d, alpha):
    if _TRAIN2:
        from . import ops_train
        return ops_train.dense_hadamard_down(x, rbf, Wa, Wr, Wd, act_a, act_d, alpha)
    assert constant_weights(), "the fused interaction head is the constant-weight inference path"
    return _DenseHadamardDown.apply(x, rbf, Wa, Wr, Wd, (bool(act_a), bool(act_d), float(alpha)))


class _QuadBasis(torch.autograd.Function):
    """(R) -> real Y_lm(Phi_cab, Theta_cabd) of every quadruplet in one launch (first-order adjoint)."""

    @staticmethod
    def forward(ctx, R, ri_c, ri_a, ri_b, ri_d, S, plan=None, angle_form=False):
        if angle_form:   # (Q,4) = (sin, cos) of the two angles: the bilinear kernels rebuild Y_lm (csrc/bilinear_ang.hip)
            Y = K.quad_angles_fwd(R, ri_c.idx32, ri_a.idx32, ri_b.idx32, ri_d.idx32)
        else:
            Y = K.quad_basis_fwd(R, ri_c.idx32, ri_a.idx32, ri_b.idx32, ri_d.idx32, S)
        ctx.save_for_backward(R)
        ctx.cfg = (ri_c, ri_a, ri_b, ri_d, S, plan)
        ctx.angle_form = angle_form
        return Y

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gY):
        (R,) = ctx.saved_tensors
        ri_c, ri_a, ri_b, ri_d, S, plan = ctx.cfg
        if not ctx.needs_input_grad[0]:
            return (None,) * 8
        return (_quad_adjoint(gY, R, ri_c, ri_a, ri_b, ri_d, plan, S, ctx.angle_form),) + (None,) * 7


def _quad_adjoint(gY, R, ri_c, ri_a, ri_b, ri_d, plan, S, angle_form):
    """dE/dR (A,3) from the gradient w.r.t. the quadruplet basis — gY (Q,S^2), or in angle form g_ang (Q,4) = (dE/dPhi_cab,
    dE/dTheta_cabd, 0, 0): the per-quadruplet position terms of atoms c, b, d (a: minus their sum) reduced onto the atoms."""
    idx = (ri_c.idx32, ri_a.idx32, ri_b.idx32, ri_d.idx32)
    if plan is None:
        if angle_form:
            Gc, Gb, Gd = K.quad_angles_bwd(gY, R, *idx)
        else:
            Gc, Gb, Gd = K.quad_basis_bwd(gY, R, *idx, S)
        return (K.segsum(Gc, *ri_c.csr, ri_c.n_rows) + K.segsum(Gb, *ri_b.csr, ri_b.n_rows)
                + K.segsum(Gd, *ri_d.csr, ri_d.n_rows) - K.segsum(Gc + Gb + Gd, *ri_a.csr, ri_a.n_rows))
    # two-level sums: the quadruplets of one reduce edge (c -> a) are contiguous and share c and a, so their
    # contributions are first summed per edge with coalesced reads (9 M x 12 B gathered through a
    # permutation by atom ran at 180 GB/s), then the 18 k edge rows go to the atoms
    # b and d are shared by the quadruplets of one intermediate triplet (a, b, d): [Gb | Gd] rows of 32 B
    # are summed per intermediate triplet in one float4 pass, then the 0.6 M rows go to the atoms
    if angle_form:
        Gc, Gbd = K.quad_angles_bwd(gY, R, *idx, packed=True)
    else:
        Gc, Gbd = K.quad_basis_bwd_packed(gY, R, *idx, S)
    seg, E = plan.quad.seg_off, plan.n_edges
    Ec = K.segsum(Gc, None, seg, E)
    Ebd = K.segsum(Gbd, None, seg, E)
    Ea = Ec + Ebd[:, 0:3] + Ebd[:, 4:7]
    Ibd = K.segsum(Gbd, *plan.quad.expand.csr, plan.quad.n_expand)
    rb, rd = plan.quad_geom["b_of_exp"], plan.quad_geom["d_of_exp"]
    return (K.segsum(Ec, *plan.id_c.csr, plan.id_c.n_rows) - K.segsum(Ea, *plan.id_a.csr, plan.id_a.n_rows)
            + K.segsum(Ibd[:, 0:3].contiguous(), *rb.csr, rb.n_rows)
            + K.segsum(Ibd[:, 4:7].contiguous(), *rd.csr, rd.n_rows))


def quad_angles_adjoint(g_ang, R, ri_c, ri_a, ri_b, ri_d, plan=None):
    """First adjoint of the angle form (ops_train._QuadAngles2)."""
    return _quad_adjoint(g_ang, R, ri_c, ri_a, ri_b, ri_d, plan, 7, True)


# The tensor basis of GemNet-Q in angle form (16 B per quadruplet instead of the 196-B harmonics row, rebuilt inside the
# bilinear kernels).  Only for the published shapes (num_spherical 7, emb_size_quad = emb_size_sbf = 32).
USE_QUAD_ANGLES = os.environ.get("GEMNET_QUAD_ANGLES", "1") == "1"


def quad_basis(R, ri_c, ri_a, ri_b, ri_d, S, plan=None, angle_form=False):
    """plan: the GraphPlan the four atom indices came from (enables the two-level force reduction).
    angle_form: return (Q,4) (sin, cos) pairs of (Phi_cab, Theta_cabd) for the *_ang bilinear kernels."""
    return _QuadBasis.apply(R, ri_c, ri_a, ri_b, ri_d, int(S), plan, bool(angle_form and USE_QUAD_ANGLES and S == 7))
