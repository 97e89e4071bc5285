"""Twice-differentiable LDS-resident layer stacks: the fused force-TRAINING path.

The reference's training step is `loss.backward()` through `autograd.grad(E, R, create_graph=True)`
(gemnet/model/gemnet.py:603-611, gemnet/training/trainer.py:338-346).  With F = -dE/dR and u = dL/dF the parameter
gradient of the force term is the gradient of the directional derivative of E along u, so the whole step is FOUR
sweeps over the layer graph, each of them one chain program (csrc/chain2.hip) per stack:

    S1  forward                     x -> z_k, a_k                                  (values)
    S2  adjoint  (builds F)         mu_y -> mu_h_k, mu_z_k = mu_h_k f'(z_k)        (create_graph=True pass)
    S3  tangent  (along u)          dx -> dz_k, da_k = f'(z_k) dz_k                (double backward of S2)
    S4  second adjoint              ybar -> zbar_k = hbar_k f'(z_k) + mu_h_k f''(z_k) dz_k        (backward of S1)
        weight gradients            dW_k = zbar_k^T a_{k-1} + mu_z_k^T da_{k-1}    (queued: one grouped launch per step)

Inside PyTorch's autograd each stack is a pair of Functions: `X` (forward = S1; its backward is S4 — or, while the force
graph is being built, a call of `XB`) and `XB` (forward = S2, backward = S3).  X and XB share a record (`_Rec`) through
which S4 finds what S2 and S3 left behind (mu_h, dz); a zero-size token output of X that XB takes as an input makes the
engine run XB's backward (S3) before X's (S4).  The source term mu_h f''(z) dz is added inside the S4 program
(`ChainProgram.source`, gn_chain_op.src_*), so no per-layer pointwise launch exists on this path.

Everything that is not a Dense stack (bases, bilinear layer, aggregation, geometry) stays on the composite ops of
`ops.py`, which are closed under differentiation; both kinds of nodes live in one autograd graph.
"""
import torch

from . import kernels as K
from . import ops


class _Rec:
    """What the four sweeps of one stack instance share (tensors are fp32 (M, N) unless noted)."""

    def __init__(self):
        self.s1 = None    # dict: z[site], a_in[gemm]
        self.s2 = None    # dict: mu_h[site], mu_z[gemm], g, g_tails
        self.s3 = None    # dict: zdot[site], adot[gemm]
        self.cache = ops.step_cache()   # packed weights of this training step (shared by all four sweeps)
        # arithmetic of the chain launches: S1 / S2 in the mode of the forward; the loss-scaled sweeps (S3, S4 and an
        # energy-only final adjoint) in kernels.linear_mode of it — the forward's `ops.chain_mode` context has closed by then
        self.mode = K.current_mode()


def _releases_record(backward):
    """Decorator of the `backward` of the first-level Functions (S2 under create_graph, S4 otherwise): when it ran as the FINAL
    sweep (no graph being recorded) the second-order records are dropped.  They hold cotangents whose autograd history
    leads — through C++ edges the cyclic garbage collector cannot follow — back to this very node (record -> g -> the
    downstream layer's second-level node -> its token -> the downstream first-level node -> this node's output -> this
    node -> record): without this, every eager training step's records, activations included, stayed alive for the life
    of the process (found with gc disabled: 41 -> 127 ms per eager step, tools/exp/train_leak_cpu.py)."""
    import functools

    @functools.wraps(backward)
    def wrapped(ctx, *grads):
        try:
            return backward(ctx, *grads)
        finally:
            if not torch.is_grad_enabled():
                rec = getattr(ctx, "rec", None)
                if rec is not None:
                    rec.s2 = rec.s3 = None
    return wrapped


def _sweep_mode(rec, linear):
    return ops.chain_mode(K.linear_mode(rec.mode) if linear else rec.mode)


def _linear_sweep(fn):
    """Decorator of a `backward(ctx, ...)` that runs S3 of `ctx.rec`."""
    def wrapped(ctx, *a):
        with _sweep_mode(ctx.rec, True):
            return fn(ctx, *a)
    wrapped.__name__, wrapped.__doc__ = fn.__name__, fn.__doc__
    return wrapped


def _new(M, N, like):
    return torch.empty((M, N), device=like.device, dtype=like.dtype)


def _wgrad(W, zb, a):
    """W.grad += zb^T a, queued into the grouped launch when W is a queueable leaf; returns the tensor gradient
    otherwise (weight slices, CPU emulation)."""
    if zb is None or a is None:
        return None
    if ops._queueable(W):
        ops._WGRAD_QUEUE.add(W, zb, a)
        return None
    return K.gemm(zb, a, True, True)


def _wgrad_bilinear(W, Pm, g, alpha):
    """Gradient of the bilinear weight W (C, I, O) from  dW2 = alpha * P2d^T g  with P (E, I, C), g (E, O):
    dW[c, i, o] = alpha sum_e P[e, i, c] g[e, o].  Queued as I products (P[:, i, :]^T g -> the (C, O) block of i, rows
    with pitch I * O) into the grouped launch when W is a leaf with a contiguous .grad; a tensor otherwise."""
    C, I, O = W.shape
    q = ops._WGRAD_QUEUE
    if (q is not None and not torch.is_grad_enabled() and W.is_leaf and W.is_cuda and W.grad is not None
            and W.grad.is_contiguous()):
        base = W.grad.data_ptr()
        for i in range(I):      # (alpha rides on the queued product: no scaled copy of g)
            q.add_region((base + 4 * i * O, C, O, I * O), Pm[:, i, :], g, keep=W, alpha=alpha)
        return None
    gW2 = K.gemm(Pm.reshape(-1, I * C), g, True, True, alpha=alpha)
    return gW2.reshape(I, C, O).permute(1, 0, 2)


def _add(a, b):
    if a is None:
        return b
    if b is None:
        return a
    return a + b


def _gemm(prog, W, trans=False, cache=None, **kw):
    """Append x @ W^T (trans: x @ W) with W a TRAINABLE weight: packed once per step (ops.step_packed)."""
    packed = ops.step_packed(W, trans, cache)
    if packed is not None:
        # the split-operand kernel reads the packed planes; the fp32 matrix only carries the shape
        prog.gemm(W.detach().t() if trans else W.detach(), packed=packed, **kw)
    else:
        prog.gemm(W.detach().t().contiguous() if trans else ops.contiguous_weight(W), packed=None, **kw)


# ============================================================================================ Dense + ResidualLayer*
class _Stack2(torch.autograd.Function):
    """ops._Stack with trainable weights, twice differentiable.  Inputs: x, res, res2, g1, g2, *skips, *weights
    (weights in the order W0?, (W1, W2) per layer, tails)."""

    @staticmethod
    def forward(ctx, spec, x, res, res2, g1, g2, *rest):
        first, layers, s = spec["first"], spec["layers"], spec["s"]
        nL, nT = len(layers), len(spec["tails"])
        skips, Ws = rest[:nL], rest[nL:]
        M = x.shape[0]
        ctx.acc = ops._acc_join(x)
        x = x.contiguous()
        rec = _Rec()
        z, a_in = {}, {}
        prog = K.ChainProgram(M)
        prog.load(0, x)
        cur, oth = 0, 1
        wi = 0
        y_prev = x
        if first is not None:
            W0 = Ws[wi]; wi += 1
            N0 = W0.shape[0]
            z0 = _new(M, N0, x) if first["act"] else None
            y0 = _new(M, N0, x)
            _gemm(prog, cache=rec.cache, W=W0, a_slot=0, y_slot=1, act=first["act"],
                  gadd1=g1, gidx1=None if g1 is None else first["i1"].idx32,
                  gadd2=g2, gidx2=None if g2 is None else first["i2"].idx32,
                  pre_out=z0, res=res, beta=first["beta"], res2=res2, beta2=first["beta2"], out=y0)
            z["0"] = z0
            a_in["0"] = x
            cur, oth = 1, 0
            y_prev = y0
        for k, L in enumerate(layers):
            W1, W2 = Ws[wi], Ws[wi + 1]; wi += 2
            width = W1.shape[0]
            z1, h1, z2, yk = (_new(M, width, x) for _ in range(4))
            _gemm(prog, cache=rec.cache, W=W1, a_slot=cur, y_slot=oth, act=True, pre_out=z1, out=h1)
            _gemm(prog, cache=rec.cache, W=W2, a_slot=oth, y_slot=cur, act=True, pre_out=z2, res=cur, beta=s,
                  res2=skips[k], beta2=L["skip_beta"], out=yk)
            z[(k, 1)], z[(k, 2)] = z1, z2
            a_in[(k, 1)], a_in[(k, 2)] = y_prev, h1
            y_prev = yk
        tails = []
        for j in range(nT):
            Wt = Ws[wi]; wi += 1
            t = _new(M, Wt.shape[0], x)
            _gemm(prog, cache=rec.cache, W=Wt, a_slot=cur, y_slot=-1, out=t)
            # (an alias, not the returned object: that one gets this node as its grad_fn, and record -> output -> node ->
            #  record would keep the step's tensors until the cyclic collector's next pass — ~1 GiB per eager step at B = 32)
            a_in[("t", j)] = y_prev.detach()
            tails.append(t)
        K.chain(prog)
        rec.s1 = dict(z=z, a_in=a_in)
        tok = x.new_empty(0)
        ctx.rec, ctx.spec, ctx.n = rec, spec, (nL, nT)
        ctx.has = (res is not None, res2 is not None, g1 is not None, g2 is not None, tuple(sk is not None for sk in skips))
        ctx.in_width, ctx.out_shape = x.shape[1], tuple(y_prev.shape)
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(tok, *Ws)
        return (y_prev, *tails, tok)

    @staticmethod
    @_releases_record
    def backward(ctx, g, *rest):
        nL, nT = ctx.n
        g_tails = rest[:nT]
        tok, *Ws = ctx.saved_tensors
        need = ctx.needs_input_grad      # (spec, x, res, res2, g1, g2, *skips, *Ws)
        n_in = 6 + nL + len(Ws)
        acc = ctx.acc
        if g is None and all(t is None for t in g_tails):
            return (None, acc.skip() if acc is not None else None) + (None,) * (n_in - 2)
        want = dict(x=need[1], res=need[2], res2=need[3], g1=need[4], g2=need[5], skips=tuple(need[6:6 + nL]))
        prev, last = (acc.enter() if want["x"] else (acc.skip(), True)) if acc is not None else (None, True)
        if torch.is_grad_enabled():
            # the force graph is being built: S2 as a differentiable node
            outs = _Stack2B.apply(ctx.rec, ctx.spec, ctx.has, want, (ctx.in_width, ctx.out_shape), g, *g_tails, tok,
                                  prev if want["x"] else None, *Ws)
            gx = outs[0]
            if acc is not None and want["x"]:
                acc.leave(gx, last)
                gx = gx if last else None
            elif acc is not None:
                gx = prev           # nothing of ours: hand the running sum on (acc.skip)
            return (None, gx, *outs[1:], *([None] * len(Ws)))
        # final pass: S4 (first-order adjoint when no tangent sweep ran through this stack)
        s3 = ctx.rec.s3
        ctx.rec.s3 = None
        out, zbar = _stack_adjoint(ctx.rec, ctx.spec, ctx.has, want, (ctx.in_width, ctx.out_shape), g, g_tails, Ws,
                                   second=s3, store="zbar", prev=prev if want["x"] else None, inplace=True)
        gx = out[0]
        if acc is not None and want["x"]:
            acc.leave(gx, last)
            gx = gx if last else None
        elif acc is not None:
            gx = prev
        gWs = [None] * len(Ws)
        if ops._PARAM_GRADS:
            a_in = ctx.rec.s1["a_in"]
            for i, key in enumerate(_gemm_keys(ctx.spec)):
                if need[6 + nL + i]:
                    gWs[i] = _wgrad(Ws[i], zbar.get(key), a_in.get(key))
        return (None, gx, *out[1:], *gWs)


def _gemm_keys(spec):
    """Keys of the GEMMs of a stack in weight order."""
    keys = []
    if spec["first"] is not None:
        keys.append("0")
    for k in range(len(spec["layers"])):
        keys += [(k, 1), (k, 2)]
    keys += [("t", j) for j in range(len(spec["tails"]))]
    return keys


def _stack_adjoint(rec, spec, has, want, shapes, g, g_tails, Ws, second, store, prev=None, inplace=False):
    with _sweep_mode(rec, store != "mu"):
        return _stack_adjoint_(rec, spec, has, want, shapes, g, g_tails, Ws, second, store, prev, inplace)


def _stack_adjoint_(rec, spec, has, want, shapes, g, g_tails, Ws, second, store, prev=None, inplace=False):
    """The reverse sweep of a stack as one chain program.  S2 (second is None, store = "mu": mu_h per activation and
    mu_z per GEMM are written out for the later sweeps) and S4 (store = "zbar"; `second` = the record of the tangent
    sweep, whose dz enter as source terms mu_h f''(z) dz).  Returns ((gx, g_res, g_res2, gg1, gg2, *g_skips), stored).
    `prev`: running gradient of x (ops.accumulate_gradient) that the last GEMM adds in its epilogue — into a new tensor,
    or (`inplace`, the non-differentiable S4) into `prev` itself."""
    first, layers, s = spec["first"], spec["layers"], spec["s"]
    has_res, has_res2, has_g1, has_g2, has_skips = has
    in_width, (M, width) = shapes
    z = rec.s1["z"]
    live = [t for t in (g, *g_tails) if t is not None]
    like = live[0]
    if g is None:
        g = torch.zeros((M, width), device=like.device, dtype=like.dtype)
    g = g.contiguous()
    keys = _gemm_keys(spec)
    Wof = dict(zip(keys, Ws))
    mu_h, stored = {}, {}

    def src(site):
        if second is None or rec.s2 is None or second["zdot"].get(site) is None or rec.s2["mu_h"].get(site) is None:
            return None
        return K.ChainProgram.source(rec.s2["mu_h"][site], second["zdot"][site], d2=True)

    prog = K.ChainProgram(M)
    prog.load(0, g)
    cur, oth = 0, 1
    for j, gt in enumerate(g_tails):
        if gt is not None:
            gt = gt.contiguous()
            stored[("t", j)] = gt.detach()             # dL/d(tail j) is the pre-activation adjoint of that GEMM (values only)
            prog.load(oth, gt)
            _gemm(prog, cache=rec.cache, W=Wof[("t", j)], trans=True, a_slot=oth, y_slot=cur, res=cur, beta=1.0)
    g_skips = [None] * len(layers)
    park = spec.get("skip_is_x", -1)
    if not (park >= 0 and in_width == width and has_skips[park] and want["skips"][park] and want["x"]):
        park = -1
    for k in range(len(layers) - 1, -1, -1):
        L = layers[k]
        c = s
        if has_skips[k]:
            if k == park:
                prog.scale(2, cur, L["skip_beta"], width=width)    # joins dL/dx in the last GEMM (res = slot 2)
                c = s * L["skip_beta"]
            elif want["skips"][k]:
                g_skips[k] = _new(M, width, like)
                prog.scale(cur, cur, L["skip_beta"], out=g_skips[k], width=width)
                c = s
            else:
                c = s * L["skip_beta"]
        mh2 = _new(M, width, like) if store == "mu" else None
        prog.scale(cur, cur, c, out=mh2, width=width)                       # Gs = adjoint of ssilu(z2)
        stored[(k, 2)] = _new(M, width, like)
        prog.scale(oth, cur, 1.0, Z=z[(k, 2)], add=src((k, 2)), out=stored[(k, 2)], width=width)   # adjoint of z2
        mh1 = _new(M, width, like) if store == "mu" else None
        _gemm(prog, cache=rec.cache, W=Wof[(k, 2)], trans=True, a_slot=oth, y_slot=oth, pre_out=mh1)                    # adjoint of h1
        stored[(k, 1)] = _new(M, width, like)
        prog.scale(oth, oth, 1.0, Z=z[(k, 1)], add=src((k, 1)), out=stored[(k, 1)], width=width)   # adjoint of z1
        _gemm(prog, cache=rec.cache, W=Wof[(k, 1)], trans=True, a_slot=oth, y_slot=cur, res=cur, beta=1.0)              # adjoint of y_{k-1}
        mu_h[(k, 2)], mu_h[(k, 1)] = mh2, mh1
    gx = g_res = g_res2 = gg1 = gg2 = None
    if first is not None:
        z0 = z["0"]
        c = 1.0
        if has_res2:
            if want["res2"]:
                g_res2 = _new(M, width, like)
                prog.scale(cur, cur, first["beta2"], out=g_res2, width=width)
            else:
                c *= first["beta2"]
        if has_res:
            c *= first["beta"]
            if want["res"]:
                g_res = _new(M, width, like)
                prog.scale(cur, cur, c, out=g_res, width=width)
                c = 1.0
        dz0 = _new(M, width, like)
        stored["0"] = dz0
        if z0 is not None:
            mh0 = _new(M, width, like) if store == "mu" else None
            if mh0 is not None or c != 1.0:
                prog.scale(cur, cur, c, out=mh0, width=width)
            mu_h["0"] = mh0
            prog.scale(oth, cur, 1.0, Z=z0, add=src("0"), out=dz0, width=width)
        else:
            prog.scale(oth, cur, c, out=dz0, width=width)
        if want["x"]:
            gx = prev if (inplace and prev is not None) else _new(M, in_width, like)
            _gemm(prog, cache=rec.cache, W=Wof["0"], trans=True, a_slot=oth, y_slot=-1, out=gx,
                  res=2 if park >= 0 else None, beta=1.0, res2=prev, beta2=1.0)
        K.chain(K.fuse_program(prog))
        if has_g1 and want["g1"]:
            gg1 = K.segsum(dz0, *first["i1"].csr, first["i1"].n_rows)
        if has_g2 and want["g2"]:
            gg2 = K.segsum(dz0, *first["i2"].csr, first["i2"].n_rows)
    else:
        gx = _new(M, width, like)
        prog.store(cur, gx)
        K.chain(K.fuse_program(prog))
        if prev is not None:
            gx = prev.add_(gx) if inplace else prev + gx
    if store == "mu":
        stored = dict(mu_h=mu_h, mu_z=stored)
    return (gx, g_res, g_res2, gg1, gg2, *g_skips), stored


class _Stack2B(torch.autograd.Function):
    """S2 of a stack as a differentiable node: forward = the adjoint program (+ mu stores), backward = S3."""

    @staticmethod
    def forward(ctx, rec, spec, has, want, shapes, g, *rest):
        nT = len(spec["tails"])
        g_tails, prev, Ws = rest[:nT], rest[nT + 1], rest[nT + 2:]
        out, st = _stack_adjoint(rec, spec, has, want, shapes, g, g_tails, Ws, second=None, store="mu", prev=prev)
        # the record keeps ALIASES of stored adjoints that are also returned (a returned tensor gets this node as its grad_fn:
        # record -> tensor -> node -> record would never be freed when the outer backward does not run)
        returned = {id(t) for t in out if t is not None}
        st = {name: {k: (v.detach() if (v is not None and id(v) in returned) else v) for k, v in d.items()}
              for name, d in st.items()}
        rec.s2 = st
        rec.s3 = None
        ctx.rec, ctx.spec, ctx.has, ctx.shapes, ctx.nT = rec, spec, has, shapes, nT
        ctx.g_none = (g is None, tuple(t is None for t in g_tails))
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(*Ws)
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    @_linear_sweep
    def backward(ctx, t_x, t_res, t_res2, t_g1, t_g2, *t_skips):
        rec, spec, nT = ctx.rec, ctx.spec, ctx.nT
        Ws = ctx.saved_tensors
        first, layers, s = spec["first"], spec["layers"], spec["s"]
        has_res, has_res2, has_g1, has_g2, has_skips = ctx.has
        in_width, (M, width) = ctx.shapes
        need = ctx.needs_input_grad    # (rec, spec, has, want, shapes, g, *g_tails, tok, prev, *Ws)
        n_lead = 5
        ts = [t for t in (t_x, t_res, t_res2, t_g1, t_g2, *t_skips) if t is not None]
        if not ts:
            return (None,) * (n_lead + 1 + nT + 2 + len(Ws))
        t_prev = t_x if need[n_lead + 1 + nT + 1] else None       # the running sum passes the cotangent through
        like = ts[0]
        z = rec.s1["z"]
        keys = _gemm_keys(spec)
        Wof = dict(zip(keys, Ws))
        zdot, adot = {}, {}
        prog = K.ChainProgram(M)
        cur, oth = 0, 1
        if first is not None:
            if t_x is None:
                t_x = torch.zeros((M, in_width), device=like.device, dtype=like.dtype)
            t_x = t_x.contiguous()
            prog.load(0, t_x)
            adot["0"] = t_x
            z0 = z["0"]
            zd0 = _new(M, width, like) if z0 is not None else None
            y0 = _new(M, width, like)
            # y0 = ((f(z0) + res) beta + res2) beta2  ->  dy0 = ((f'(z0) dz0 + dres) beta + dres2) beta2; a residual whose
            # tangent is absent still leaves its factor on the stages before it
            alpha, beta, beta2 = 1.0, first["beta"], first["beta2"]
            tr = t_res.contiguous() if (has_res and t_res is not None) else None
            tr2 = t_res2.contiguous() if (has_res2 and t_res2 is not None) else None
            if has_res and tr is None:
                alpha *= beta
            if has_res2 and tr2 is None:
                if tr is not None:
                    beta *= beta2
                else:
                    alpha *= beta2
            _gemm(prog, cache=rec.cache, W=Wof["0"], a_slot=0, y_slot=1,
                  gadd1=None if t_g1 is None else t_g1.contiguous(), gidx1=None if t_g1 is None else first["i1"].idx32,
                  gadd2=None if t_g2 is None else t_g2.contiguous(), gidx2=None if t_g2 is None else first["i2"].idx32,
                  pre_out=zd0, mul=z0, mul_mode=2 if z0 is not None else 1, alpha=alpha,
                  res=tr, beta=beta if tr is not None else 1.0, res2=tr2, beta2=beta2 if tr2 is not None else 1.0, out=y0)
            zdot["0"] = zd0
            cur, oth = 1, 0
            y_prev = y0
        else:
            if t_x is None:
                t_x = torch.zeros((M, width), device=like.device, dtype=like.dtype)
            t_x = t_x.contiguous()
            prog.load(0, t_x)
            y_prev = t_x
        for k, L in enumerate(layers):
            zd1, hd1, zd2, yk = (_new(M, width, like) for _ in range(4))
            tsk = t_skips[k] if has_skips[k] else None
            if has_skips[k] and tsk is None and spec.get("skip_is_x", -1) == k:
                tsk = t_x        # the skip IS the stack's input (its gradient rode on dL/dx in S2): same tangent
            _gemm(prog, cache=rec.cache, W=Wof[(k, 1)], a_slot=cur, y_slot=oth, pre_out=zd1, mul=z[(k, 1)], mul_mode=2, out=hd1)
            _gemm(prog, cache=rec.cache, W=Wof[(k, 2)], a_slot=oth, y_slot=cur, pre_out=zd2, mul=z[(k, 2)], mul_mode=2, res=cur, beta=s,
                  res2=None if tsk is None else tsk.contiguous(), beta2=L["skip_beta"] if tsk is not None else 1.0, out=yk)
            if has_skips[k] and tsk is None and L["skip_beta"] != 1.0:
                _scale_last_gemm(prog, L["skip_beta"])
            zdot[(k, 1)], zdot[(k, 2)] = zd1, zd2
            adot[(k, 1)], adot[(k, 2)] = y_prev, hd1
            y_prev = yk
        t_tails = []
        for j in range(nT):
            Wt = Wof[("t", j)]
            if need[n_lead + 1 + j]:
                tt = _new(M, Wt.shape[0], like)
                _gemm(prog, cache=rec.cache, W=Wt, a_slot=cur, y_slot=-1, out=tt)
            else:
                tt = None
            adot[("t", j)] = y_prev
            t_tails.append(tt)
        K.chain(prog)
        rec.s3 = dict(zdot=zdot, adot=adot)
        gWs = [None] * len(Ws)
        if ops._PARAM_GRADS:
            mu_z = rec.s2["mu_z"]
            for i, key in enumerate(keys):
                if need[n_lead + 1 + nT + 2 + i]:
                    gWs[i] = _wgrad(Ws[i], mu_z.get(key), adot.get(key))
        g_y = y_prev if need[n_lead] else None
        return (None,) * n_lead + (g_y, *t_tails, None, t_prev, *gWs)


def _scale_last_gemm(prog, c):
    """Fold a constant factor into the LAST stage of the most recent GEMM op of `prog`."""
    o = prog.ops[-1]
    if o["res2"] is not None:
        o["beta2"] *= c
    elif o["res"] is not None:
        o["beta"] *= c
    else:
        o["alpha"] *= c


def stack(x, first=None, layers=(), s=0.7071067811865475, tails=()):
    """ops.stack for trainable weights (same arguments), twice differentiable."""
    spec = dict(first=None, layers=[dict(skip_beta=float(L.get("skip_beta", 1.0))) for L in layers], s=float(s),
                tails=tuple(range(len(tails))))
    res = res2 = g1 = g2 = None
    Ws = []
    if first is not None:
        assert first.get("res_rows") is None and not first.get("tied"), \
            "the tied / row-gathered residual pair (ops.up_project_pair) is a constant-weight form"
        spec["first"] = dict(act=bool(first.get("act", False)), beta=float(first.get("beta", 1.0)),
                             beta2=float(first.get("beta2", 1.0)), i1=first.get("i1"), i2=first.get("i2"))
        res, res2, g1, g2 = first.get("res"), first.get("res2"), first.get("g1"), first.get("g2")
        Ws.append(first["W"])
    for L in layers:
        Ws += [L["W1"], L["W2"]]
    Ws += list(tails)
    skips = [L.get("skip") for L in layers]
    # a skip connection fed by the stack's own input (m -> ... + m): its gradient joins dL/dx inside the adjoint programs
    # (parked in the register slot) instead of travelling as a second (M, 128) tensor that autograd then adds
    spec["skip_is_x"] = next((k for k, sk in enumerate(skips) if sk is x), -1) if first is not None else -1
    out = _Stack2.apply(spec, x, res, res2, g1, g2, *skips, *Ws)
    out = out[:-1]              # drop the ordering token
    return out if tails else out[0]


# ======================================================================== Dense -> radial Hadamard -> down projection
class _Head2(torch.autograd.Function):
    """ops._DenseHadamardDown (interaction_block.py:667-675, :531-541) with trainable weights, twice differentiable:
    z1 = x Wa^T, xa = f(z1);  r = rbf Wr^T;  h = r (.) xa alpha;  z3 = h Wd^T, y = f(z3)."""

    @staticmethod
    def forward(ctx, cfg, x, rbf, Wa, Wr, Wd):
        act_a, act_d, alpha = cfg
        M = x.shape[0]
        ctx.acc_x, ctx.acc_rbf = ops._acc_join(x), ops._acc_join(rbf)
        x, rbf = x.contiguous(), rbf.contiguous()
        rec = _Rec()
        nh, nd = Wa.shape[0], Wd.shape[0]
        z1 = _new(M, nh, x) if act_a else None
        xa, r, h = _new(M, nh, x), _new(M, nh, x), _new(M, nh, x)
        z3 = _new(M, nd, x) if act_d else None
        y = _new(M, nd, x)
        prog = K.ChainProgram(M)
        prog.load(0, x)
        _gemm(prog, Wa, cache=rec.cache, a_slot=0, y_slot=1, act=act_a, pre_out=z1, out=xa)
        prog.load(0, rbf)
        _gemm(prog, Wr, cache=rec.cache, a_slot=0, y_slot=0, pre_out=r, mul=1, alpha=alpha, out=h)
        _gemm(prog, Wd, cache=rec.cache, a_slot=0, y_slot=1, act=act_d, pre_out=z3, out=y)
        K.chain(prog)
        rec.s1 = dict(x=x, rbf=rbf, z1=z1, xa=xa, r=r, h=h, z3=z3)
        tok = x.new_empty(0)
        ctx.rec, ctx.cfg = rec, cfg
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(tok, Wa, Wr, Wd)
        return y, tok

    @staticmethod
    @_releases_record
    def backward(ctx, g, g_tok):
        tok, Wa, Wr, Wd = ctx.saved_tensors
        need = ctx.needs_input_grad      # (cfg, x, rbf, Wa, Wr, Wd)
        ax, ar = ctx.acc_x, ctx.acc_rbf
        if g is None:
            return (None, ax.skip() if ax is not None else None, ar.skip() if ar is not None else None, None, None, None)
        px, last_x = ax.enter() if ax is not None else (None, True)
        pr, last_r = ar.enter() if ar is not None else (None, True)
        if torch.is_grad_enabled():
            gx, grbf = _Head2B.apply(ctx.rec, ctx.cfg, g, tok, px, pr, Wa, Wr, Wd)
        else:
            s3 = ctx.rec.s3
            ctx.rec.s3 = None
            (gx, grbf), st = _head_adjoint(ctx.rec, ctx.cfg, g, (Wa, Wr, Wd), second=s3, store="zbar", prev=(px, pr), inplace=True)
        if ax is not None:
            ax.leave(gx, last_x)
        if ar is not None:
            ar.leave(grbf, last_r)
        gx, grbf = (gx if last_x else None), (grbf if last_r else None)
        if torch.is_grad_enabled():
            return None, gx, grbf, None, None, None
        s1 = ctx.rec.s1
        gWa = gWr = gWd = None
        if ops._PARAM_GRADS:
            if need[3]:
                gWa = _wgrad(Wa, st["z1"], s1["x"])
            if need[4]:
                gWr = _wgrad(Wr, st["r"], s1["rbf"])
            if need[5]:
                gWd = _wgrad(Wd, st["z3"], s1["h"])
        return None, gx, grbf, gWa, gWr, gWd


def _head_adjoint(rec, cfg, g, Ws, second, store, prev=(None, None), inplace=False):
    with _sweep_mode(rec, store != "mu"):
        return _head_adjoint_(rec, cfg, g, Ws, second, store, prev, inplace)


def _head_adjoint_(rec, cfg, g, Ws, second, store, prev=(None, None), inplace=False):
    """Reverse sweep of the head: S2 (store = "mu") or S4 (store = "zbar", `second` = the tangent record).
    -> ((gx, grbf), dict of the adjoints at z3, dh, r, xa, z1).  `prev`: running gradients of (x, rbf) added in the
    epilogues of the two GEMMs that produce ours (into new tensors, or — `inplace` — into the running sums)."""
    px, pr = prev
    act_a, act_d, alpha = cfg
    Wa, Wr, Wd = Ws
    s1 = rec.s1
    M = g.shape[0]
    g = g.contiguous()
    nh, nd = Wa.shape[0], Wd.shape[0]
    S = K.ChainProgram.source
    sec = second if (second is not None and rec.s2 is not None) else None
    s2 = rec.s2
    st = {}
    prog = K.ChainProgram(M)
    prog.load(0, g)
    if act_d:
        st["z3"] = _new(M, nd, g)
        prog.scale(0, 0, 1.0, Z=s1["z3"], width=nd, out=st["z3"],
                   add=S(s2["g"], sec["zd3"], d2=True) if sec is not None else None)
    else:
        st["z3"] = g
    st["dh"] = _new(M, nh, g) if store == "mu" else None
    _gemm(prog, Wd, trans=True, cache=rec.cache, a_slot=0, y_slot=1, out=st["dh"])           # adjoint of h
    st["r"] = _new(M, nh, g)
    prog.scale(0, 1, alpha, Z=s1["xa"], mode=1, width=nh, out=st["r"],
               add=S(s2["dh"], sec["xad"], alpha=alpha) if sec is not None else None)          # adjoint of r
    grbf = pr if (inplace and pr is not None) else _new(M, Wr.shape[1], g)
    _gemm(prog, Wr, trans=True, cache=rec.cache, a_slot=0, y_slot=-1, out=grbf, res=pr, beta=1.0)
    st["xa"] = _new(M, nh, g) if (store == "mu" or not act_a) else None
    prog.scale(1, 1, alpha, Z=s1["r"], mode=1, width=nh, out=st["xa"],
               add=S(s2["dh"], sec["rd"], alpha=alpha) if sec is not None else None)           # adjoint of xa
    if act_a:
        st["z1"] = _new(M, nh, g)
        prog.scale(1, 1, 1.0, Z=s1["z1"], mode=0, width=nh, out=st["z1"],
                   add=S(s2["xa"], sec["zd1"], d2=True) if sec is not None else None)          # adjoint of z1
    else:
        st["z1"] = st["xa"]
    gx = px if (inplace and px is not None) else _new(M, Wa.shape[1], g)
    _gemm(prog, Wa, trans=True, cache=rec.cache, a_slot=1, y_slot=-1, out=gx, res=px, beta=1.0)
    K.chain(K.fuse_program(prog))
    return (gx, grbf), st


class _Head2B(torch.autograd.Function):
    @staticmethod
    def forward(ctx, rec, cfg, g, tok, px, pr, Wa, Wr, Wd):
        out, st = _head_adjoint(rec, cfg, g, (Wa, Wr, Wd), second=None, store="mu", prev=(px, pr))
        st["g"] = g.detach().contiguous()     # (values only: with its history the record would own a path back to this node)
        rec.s2, rec.s3 = st, None
        ctx.rec, ctx.cfg = rec, cfg
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(Wa, Wr, Wd)
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    @_linear_sweep
    def backward(ctx, t_x, t_rbf):
        rec, (act_a, act_d, alpha) = ctx.rec, ctx.cfg
        Wa, Wr, Wd = ctx.saved_tensors
        need = ctx.needs_input_grad      # (rec, cfg, g, tok, px, pr, Wa, Wr, Wd)
        if t_x is None and t_rbf is None:
            return (None,) * 9
        t_px = t_x if need[4] else None          # the running sums pass their cotangents through
        t_pr = t_rbf if need[5] else None
        s1, s2 = rec.s1, rec.s2
        like = t_x if t_x is not None else t_rbf
        M = like.shape[0]
        nh, nd = Wa.shape[0], Wd.shape[0]
        t_x = torch.zeros_like(s1["x"]) if t_x is None else t_x.contiguous()
        t_rbf = torch.zeros_like(s1["rbf"]) if t_rbf is None else t_rbf.contiguous()
        zd1 = _new(M, nh, like) if act_a else None
        xad, rd, hd = _new(M, nh, like), _new(M, nh, like), _new(M, nh, like)
        zd3 = _new(M, nd, like) if act_d else None
        yd = _new(M, nd, like)
        prog = K.ChainProgram(M)
        prog.load(0, t_x)
        # d xa = f'(z1) dz1 -> memory; d xa (.) r -> slot 1 (the second output of the same op)
        _gemm(prog, Wa, cache=rec.cache, a_slot=0, y_slot=-1, pre_out=zd1, mul=s1["z1"], mul_mode=2 if act_a else 1, out=xad,
              y2=1, y2_src=0, alpha2=1.0, Z2=s1["r"], mode2=1)
        prog.load(0, t_rbf)
        # dh = (dr (.) xa + d xa (.) r) alpha
        _gemm(prog, Wr, cache=rec.cache, a_slot=0, y_slot=0, pre_out=rd, mul=s1["xa"], mul_mode=1, res=1, beta=alpha, out=hd)
        _gemm(prog, Wd, cache=rec.cache, a_slot=0, y_slot=-1, pre_out=zd3, mul=s1["z3"], mul_mode=2 if act_d else 1, out=yd)
        K.chain(prog)
        rec.s3 = dict(zd1=zd1, xad=xad, rd=rd, hd=hd, zd3=zd3)
        gWa = gWr = gWd = None
        if ops._PARAM_GRADS:
            if need[6]:
                gWa = _wgrad(Wa, s2["z1"], t_x)
            if need[7]:
                gWr = _wgrad(Wr, s2["r"], t_rbf)
            if need[8]:
                gWd = _wgrad(Wd, s2["z3"], hd)
        return None, None, (yd if need[2] else None), None, t_px, t_pr, gWa, gWr, gWd


def dense_hadamard_down(x, rbf, Wa, Wr, Wd, act_a, act_d, alpha):
    y, _ = _Head2.apply((bool(act_a), bool(act_d), float(alpha)), x, rbf, Wa, Wr, Wd)
    return y


# ================================================================================ radial-weighted edge -> atom aggregation
class _Aggregate2(torch.autograd.Function):
    """out[a] = scale * sum_{e -> a} m[e] (.) (W rbf[e]) (atom_update_block.py:60-68) with a trainable W, twice
    differentiable.  The map is trilinear in (m, rbf, W), so every second-order term is the first-order kernel
    (csrc/aggregate.hip) called with one operand replaced by its tangent:
        S3   d g    = fwd(dm, rbf) + fwd(m, drbf)
        S4   (mbar, rbfbar) = bwd(obar; m, rbf) + bwd(g; dm, drbf)       (two launches into one pair of buffers — the
             running gradients of m and rbf when they have several fused consumers, ops.accumulate_gradient)
        dW  += q(obar, m)^T rbf + q(g, dm)^T rbf + q(g, m)^T drbf,      q(u, v)[e] = scale * u[id_a[e]] (.) v[e]."""

    @staticmethod
    def forward(ctx, m, rbf, W, ri, scale):
        perm, seg = ri.csr
        ctx.acc_m, ctx.acc_rbf = ops._acc_join(m), ops._acc_join(rbf)
        m, rbf = m.contiguous(), rbf.contiguous()
        Wc = W.detach().contiguous()
        out = K.rbf_aggregate_fwd(m, rbf, Wc, perm, seg, ri.n_rows, scale)
        rec = _Rec()
        rec.s1 = dict(m=m, rbf=rbf)
        tok = m.new_empty(0)
        ctx.rec, ctx.ri, ctx.scale = rec, ri, scale
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(tok, W)
        return out, tok

    @staticmethod
    @_releases_record
    def backward(ctx, g, g_tok):
        tok, W = ctx.saved_tensors
        need = ctx.needs_input_grad      # (m, rbf, W, ri, scale)
        am, ar = ctx.acc_m, ctx.acc_rbf
        if g is None:
            return (am.skip() if am is not None else None, ar.skip() if ar is not None else None, None, None, None)
        rec, ri, scale = ctx.rec, ctx.ri, ctx.scale
        pm, last_m = am.enter() if am is not None else (None, True)
        pr, last_r = ar.enter() if ar is not None else (None, True)
        want_m, want_r = need[0] or am is not None, need[1] or ar is not None
        gW = None
        if torch.is_grad_enabled():
            g_m, g_rbf = _Aggregate2B.apply(rec, ri, scale, (want_m, want_r), g, tok, W, pm, pr)
        else:
            s3 = rec.s3
            rec.s3 = None
            s1 = rec.s1
            g = g.contiguous()
            Wc = W.detach().contiguous()
            g_m, g_rbf = K.rbf_aggregate_bwd(g, s1["m"], s1["rbf"], Wc, ri.idx32, scale, want_m=want_m, want_rbf=want_r,
                                             acc_m=pm, acc_rbf=pr)
            if s3 is not None and rec.s2 is not None:
                # the cross terms: mbar += scale g[a] (.) (W drbf), rbfbar += scale W^T (g[a] (.) dm)
                t_m, t_rbf = s3["t_m"], s3["t_rbf"]
                zm = t_m if t_m is not None else s1["m"]          # (placeholder operand of an output that is not produced)
                zr = t_rbf if t_rbf is not None else s1["rbf"]
                do_m, do_r = want_m and t_rbf is not None, want_r and t_m is not None
                if do_m or do_r:
                    g_m2, g_r2 = K.rbf_aggregate_bwd(rec.s2["g"], zm, zr, Wc, ri.idx32, scale, want_m=False, want_rbf=False,
                                                     acc_m=g_m if do_m else None, acc_rbf=g_rbf if do_r else None)
            if need[2] and ops._PARAM_GRADS:
                gW = _wgrad(W, K.gather_mul(g, ri.idx32, s1["m"], scale), s1["rbf"])
        if am is not None:
            am.leave(g_m, last_m)
        if ar is not None:
            ar.leave(g_rbf, last_r)
        return (g_m if last_m else None), (g_rbf if last_r else None), gW, None, None


class _Aggregate2B(torch.autograd.Function):
    @staticmethod
    def forward(ctx, rec, ri, scale, want, g, tok, W, pm, pr):
        g = g.contiguous()
        s1 = rec.s1
        # running gradients (ops.accumulate_gradient): summed into copies — this pass is differentiated again, the sums
        # of the earlier consumers stay what their nodes returned
        acc_m = pm.clone() if pm is not None else None
        acc_r = pr.clone() if pr is not None else None
        g_m, g_rbf = K.rbf_aggregate_bwd(g, s1["m"], s1["rbf"], W.detach().contiguous(), ri.idx32, scale,
                                         want_m=want[0], want_rbf=want[1], acc_m=acc_m, acc_rbf=acc_r)
        rec.s2, rec.s3 = dict(g=g.detach()), None     # (values only, see _releases_record)
        ctx.rec, ctx.ri, ctx.scale = rec, ri, scale
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(W)
        return g_m, g_rbf

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, t_m, t_rbf):
        (W,) = ctx.saved_tensors
        need = ctx.needs_input_grad      # (rec, ri, scale, want, g, tok, W, pm, pr)
        if t_m is None and t_rbf is None:
            return (None,) * 9
        rec, ri, scale = ctx.rec, ctx.ri, ctx.scale
        s1, g = rec.s1, rec.s2["g"]
        perm, seg = ri.csr
        Wc = W.detach().contiguous()
        t_m = None if t_m is None else t_m.contiguous()
        t_rbf = None if t_rbf is None else t_rbf.contiguous()
        gd = None
        if need[4]:
            if t_m is not None:
                gd = K.rbf_aggregate_fwd(t_m, s1["rbf"], Wc, perm, seg, ri.n_rows, scale)
            if t_rbf is not None:
                gd = _add(gd, K.rbf_aggregate_fwd(s1["m"], t_rbf, Wc, perm, seg, ri.n_rows, scale))
        rec.s3 = dict(t_m=t_m, t_rbf=t_rbf)
        gW = None
        if need[6] and ops._PARAM_GRADS:
            if t_m is not None:
                gW = _wgrad(W, K.gather_mul(g, ri.idx32, t_m, scale), s1["rbf"])
            if t_rbf is not None:
                gW = _add(gW, _wgrad(W, K.gather_mul(g, ri.idx32, s1["m"], scale), t_rbf))
        return None, None, None, None, gd, None, gW, (t_m if need[7] else None), (t_rbf if need[8] else None)


def rbf_aggregate(m, rbf, W, ri, scale):
    out, _ = _Aggregate2.apply(m, rbf, W, ri, float(scale))
    return out


# ============================================================================================ efficient bilinear layer
class _Bilinear2(torch.autograd.Function):
    """out[e] = alpha * vec(P[e]) W2,  P[e] = B[e]^T Sm[e],  Sm[e] = sum_{t in seg(e)} Y[t] (x) x[g(t)]   (efficient.py:159-189,
    SURVEY.md Appendix D: K1, K2, K3) with a trainable W, twice differentiable, on the fused matrix-core kernels of the
    force path (csrc/bilinear.hip).  The map is multilinear in (B, Y, x, W): with tangents (dB, dY, dx) and the first
    adjoints mu_P = alpha g W2^T, mu_Sm = B mu_P of S2,
        S3   dSm = K1(dY, x) + K1(Y, dx);   dP = dB^T Sm + B^T dSm;   d out = alpha dP W2
        S4   Pbar = alpha obar W2^T;  Bbar = Sm Pbar^T + dSm mu_P^T;  Smbar = B Pbar + dB mu_P;
             xbar = K1^T(Y, Smbar) + K1^T(dY, mu_Sm);   Ybar = dot(Smbar, x) + dot(mu_Sm, dx)
        dW2  = alpha (P^T obar + dP^T g)
    — the cross terms are the SAME kernels called with one operand replaced by its tangent."""

    @staticmethod
    def forward(ctx, B, Y, x, W, sp, alpha):
        C, I, O = W.shape
        B, Y, x = B.contiguous(), Y.contiguous(), x.contiguous()
        W2 = W.detach().permute(1, 0, 2).reshape(I * C, O).contiguous()       # rows (i, c): one copy per step
        W2T = W2.t().contiguous()   # (O, I*C): the K = 1024 products run the k-contiguous GEMM (37 vs 68 us at E = 18 k)
        Sm, P = K.bil_reduce_project(Y, x, B, sp)
        out = K.gemm(P.reshape(-1, I * C), W2T, alpha=alpha)
        rec = _Rec()
        rec.s1 = dict(B=B, Y=Y, x=x, W2=W2, W2T=W2T, Sm=Sm, P=P)
        tok = x.new_empty(0)
        ctx.rec, ctx.sp, ctx.alpha, ctx.dims = rec, sp, alpha, (C, I, O)
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(tok, W)
        return out, tok

    @staticmethod
    @_releases_record
    def backward(ctx, g, g_tok):
        tok, W = ctx.saved_tensors
        need = ctx.needs_input_grad      # (B, Y, x, W, sp, alpha)
        if g is None:
            return (None,) * 6
        rec, sp, alpha = ctx.rec, ctx.sp, ctx.alpha
        C, I, O = ctx.dims
        if torch.is_grad_enabled():
            gB, gY, gx = _Bilinear2B.apply(rec, sp, alpha, ctx.dims, tuple(need[:3]), g, tok, W)
            return gB, gY, gx, None, None, None
        s1, s2, s3 = rec.s1, rec.s2, rec.s3
        rec.s3 = None
        g = g.contiguous()
        Pb = K.gemm(g, s1["W2"], alpha=alpha).reshape(-1, I, C)
        gB, Smb, _ = K.bil_project_bwd(Pb, s1["Sm"], s1["B"], s1["x"], sp, want_dY=False)
        terms_d, terms_x = [Smb], [s1["x"]]
        gx2 = None
        if s3 is not None and s2 is not None:
            Smd, tB = s3["Smd"], s3["tB"]
            if Smd is not None or tB is not None:
                zS = Smd if Smd is not None else torch.zeros_like(s1["Sm"])
                zB = tB if tB is not None else torch.zeros_like(s1["B"])
                gB, Smb, _ = K.bil_project_bwd(s2["mu_P"], zS, zB, s1["x"], sp, want_dY=False, gB_accum=gB, dSm_accum=Smb)
            if s3["tx"] is not None:
                terms_d.append(s2["mu_Sm"])
                terms_x.append(s3["tx"])
            if s3["tY"] is not None and need[2]:
                gx2 = K.bil_reduce_t(s3["tY"], s2["mu_Sm"], sp)
        gx = _add(K.bil_reduce_t(s1["Y"], Smb, sp), gx2) if need[2] else None
        gY = K.bil_dy_multi(terms_d, terms_x, sp) if need[1] else None
        gW = None
        if need[3] and ops._PARAM_GRADS:
            gW = _wgrad_bilinear(W, s1["P"], g, alpha)
        return (gB if need[0] else None), gY, gx, gW, None, None


class _Bilinear2B(torch.autograd.Function):
    @staticmethod
    def forward(ctx, rec, sp, alpha, dims, want, g, tok, W):
        C, I, O = dims
        s1 = rec.s1
        g = g.contiguous()
        mu_P = K.gemm(g, s1["W2"], alpha=alpha).reshape(-1, I, C)
        gB, mu_Sm, gY = K.bil_project_bwd(mu_P, s1["Sm"], s1["B"], s1["x"], sp, want_dY=want[1])
        gx = K.bil_reduce_t(s1["Y"], mu_Sm, sp) if want[2] else None
        rec.s2, rec.s3 = dict(g=g.detach(), mu_P=mu_P, mu_Sm=mu_Sm), None     # (values only, see _releases_record)
        ctx.rec, ctx.sp, ctx.alpha, ctx.dims = rec, sp, alpha, dims
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(W)
        return (gB if want[0] else None), gY, gx

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, tB, tY, tx):
        (W,) = ctx.saved_tensors
        need = ctx.needs_input_grad      # (rec, sp, alpha, dims, want, g, tok, W)
        if tB is None and tY is None and tx is None:
            return (None,) * 8
        rec, sp, alpha = ctx.rec, ctx.sp, ctx.alpha
        C, I, O = ctx.dims
        s1, s2 = rec.s1, rec.s2
        tB = None if tB is None else tB.contiguous()
        tY = None if tY is None else tY.contiguous()
        tx = None if tx is None else tx.contiguous()
        # dSm = K1(dY, x) + K1(Y, dx) and dP = B^T dSm + dB^T Sm: the second K1 call starts from the first one's sum and
        # takes the dB^T Sm term along (gn_bil_reduce_project2_f32)
        Smd = Pd = None
        if tY is not None and tx is not None:
            Smd, _ = K.bil_reduce_project(tY, s1["x"], s1["B"], sp, want_P=False)
            Smd, Pd = K.bil_reduce_project(s1["Y"], tx, s1["B"], sp, Sm_init=Smd, B2=tB, Sm2=s1["Sm"] if tB is not None else None)
        elif tY is not None or tx is not None:
            Ya, xa = (tY, s1["x"]) if tY is not None else (s1["Y"], tx)
            Smd, Pd = K.bil_reduce_project(Ya, xa, s1["B"], sp, B2=tB, Sm2=s1["Sm"] if tB is not None else None)
        else:
            Pd = K.bmm(tB, s1["Sm"], True, False)
        gd = K.gemm(Pd.reshape(-1, I * C), s1["W2T"], alpha=alpha) if need[5] else None
        rec.s3 = dict(Smd=Smd, tB=tB, tY=tY, tx=tx)
        gW = None
        if need[7] and ops._PARAM_GRADS:
            gW = _wgrad_bilinear(W, Pd, s2["g"], alpha)
        return None, None, None, None, None, gd, None, gW


def bilinear(rbf_W1, sph, x, W, sp, alpha=1.0):
    out, _ = _Bilinear2.apply(rbf_W1, sph, x, W, sp, float(alpha))
    return out


# ============================================================== quadruplet bilinear layer, tensor basis in ANGLE form
class _BilinearAng2(torch.autograd.Function):
    """The quadruplet bilinear layer (interaction_block.py:517-566, efficient.py:159-189; S = 49, C = I = 32) with the tensor
    basis given as `ang` (Q,4) = (sin, cos) of (Phi_cab, Theta_cabd) instead of the (Q,49) harmonics (basis_layers.py:239-295),
    trainable W, twice differentiable on the fused angle-form kernels (csrc/bilinear_ang.hip).  Same algebra as _Bilinear2
    with Y a FUNCTION of the two angles: the gradient slot of `ang` carries g_ang (Q,4) = (dE/dPhi, dE/dTheta, 0, 0) (the
    convention of ops._QuadBasis), its tangent is t_ang (Q,4) = (dPhi, dTheta, 0, 0) along u = dL/dF, and
        dY[q] = Y_theta[q] dPhi[q] + Y_phi[q] dTheta[q]
    is rebuilt in-kernel by dual numbers next to Y[q] (no (Q,49) array in any sweep):
        S1   Sm, P = K1K2(ang, x, B);  out = alpha P W2
        S2   mu_P = alpha g W2^T;  gB = Sm mu_P^T;  mu_Sm = B mu_P;  gx = K1^T(Y, mu_Sm);  g_ang = <mu_Sm . x, dY/d angle>
        S3   Smd = K1(dY, x) + K1(Y, tx);  Pd = B^T Smd + tB^T Sm          (ONE launch: gn_bil_reduce_project_ang_tan_f32)
             d g = alpha Pd W2
        S4   Pb = alpha ob W2^T;  gB = Sm Pb^T + Smd mu_P^T;  Smb = B Pb + tB mu_P;
             gx = K1^T(Y, Smb) + K1^T(dY, mu_Sm)                           (ONE launch + the CSR sum: gn_bil_expand_ang_tan_f32)
        dW2  = alpha (P^T ob + Pd^T g)
    The angles depend on the positions only, so S4 owes them a gradient only when the CALLER differentiates w.r.t. positions a
    second time (ops.position_graph): that case needs the second derivatives of the 49 harmonics and stays on the composite
    closure (ops.bilinear routes it there); here the angle slot of S4 is left empty."""

    @staticmethod
    def forward(ctx, B, ang, x, W, sp, alpha):
        C, I, O = W.shape
        B, ang, x = B.contiguous(), ang.contiguous(), x.contiguous()
        W2 = W.detach().permute(1, 0, 2).reshape(I * C, O).contiguous()
        W2T = W2.t().contiguous()
        Sm, P = K.bil_reduce_project(ang, x, B, sp)
        out = K.gemm(P.reshape(-1, I * C), W2T, alpha=alpha)
        rec = _Rec()
        rec.s1 = dict(B=B, ang=ang, x=x, W2=W2, W2T=W2T, Sm=Sm, P=P)
        tok = x.new_empty(0)
        ctx.rec, ctx.sp, ctx.alpha, ctx.dims = rec, sp, alpha, (C, I, O)
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(tok, W)
        return out, tok

    @staticmethod
    @_releases_record
    def backward(ctx, g, g_tok):
        tok, W = ctx.saved_tensors
        need = ctx.needs_input_grad      # (B, ang, x, W, sp, alpha)
        if g is None:
            return (None,) * 6
        rec, sp, alpha = ctx.rec, ctx.sp, ctx.alpha
        C, I, O = ctx.dims
        if torch.is_grad_enabled():
            gB, g_ang, gx = _BilinearAng2B.apply(rec, sp, alpha, ctx.dims, tuple(need[:3]), g, tok, W)
            return gB, g_ang, gx, None, None, None
        s1, s2, s3 = rec.s1, rec.s2, rec.s3
        rec.s3 = None
        g = g.contiguous()
        with ops.chain_mode(rec.mode):
            Pb = K.gemm(g, s1["W2"], alpha=alpha).reshape(-1, I, C)
            gB, Smb, _ = K.bil_project_bwd(Pb, s1["Sm"], s1["B"], s1["x"], sp, want_dY=False)
            t_ang = None
            if s3 is not None and s2 is not None:
                Smd, tB, t_ang = s3["Smd"], s3["tB"], s3["t_ang"]
                if Smd is not None or tB is not None:
                    zS = Smd if Smd is not None else torch.zeros_like(s1["Sm"])
                    zB = tB if tB is not None else torch.zeros_like(s1["B"])
                    gB, Smb2, _ = K.bil_project_bwd(s2["mu_P"], zS, zB, s1["x"], sp, want_dY=False, gB_accum=gB)
                    if tB is not None:
                        Smb = Smb.add_(Smb2)
            gx = None
            if need[2]:
                if t_ang is not None and s2 is not None:
                    gx = K.bil_reduce_t_tan(s1["ang"], t_ang, Smb, s2["mu_Sm"], sp)
                else:
                    gx = K.bil_reduce_t(s1["ang"], Smb, sp)
        gW = None
        if need[3] and ops._PARAM_GRADS:
            gW = _wgrad_bilinear(W, s1["P"], g, alpha)
        return (gB if need[0] else None), None, gx, gW, None, None


class _BilinearAng2B(torch.autograd.Function):
    @staticmethod
    def forward(ctx, rec, sp, alpha, dims, want, g, tok, W):
        C, I, O = dims
        s1 = rec.s1
        g = g.contiguous()
        with ops.chain_mode(rec.mode):
            mu_P = K.gemm(g, s1["W2"], alpha=alpha).reshape(-1, I, C)
            gB, mu_Sm, _ = K.bil_project_bwd(mu_P, s1["Sm"], s1["B"], s1["x"], sp, want_dY=False)
            g_ang = K.bil_dy_multi([mu_Sm], [s1["x"]], sp, ang=s1["ang"]) if want[1] else None
            gx = K.bil_reduce_t(s1["ang"], mu_Sm, sp) if want[2] else None
        rec.s2, rec.s3 = dict(g=g.detach(), mu_P=mu_P, mu_Sm=mu_Sm), None     # (values only, see _releases_record)
        ctx.rec, ctx.sp, ctx.alpha, ctx.dims = rec, sp, alpha, dims
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(W)
        return (gB if want[0] else None), g_ang, gx

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, tB, t_ang, tx):
        (W,) = ctx.saved_tensors
        need = ctx.needs_input_grad      # (rec, sp, alpha, dims, want, g, tok, W)
        if tB is None and t_ang is None and tx is None:
            return (None,) * 8
        rec, sp, alpha = ctx.rec, ctx.sp, ctx.alpha
        C, I, O = ctx.dims
        s1, s2 = rec.s1, rec.s2
        tB = None if tB is None else tB.contiguous()
        t_ang = None if t_ang is None else t_ang.contiguous()
        tx = None if tx is None else tx.contiguous()
        if t_ang is None and tx is None:
            Smd, Pd = None, K.bmm(tB, s1["Sm"], True, False)
        else:
            Smd, Pd = K.bil_reduce_project_tan(s1["ang"], t_ang, s1["x"], tx, s1["B"], tB, s1["Sm"], sp)
        gd = K.gemm(Pd.reshape(-1, I * C), s1["W2T"], alpha=alpha) if need[5] else None
        rec.s3 = dict(Smd=Smd, tB=tB, t_ang=t_ang, tx=tx)
        gW = None
        if need[7] and ops._PARAM_GRADS:
            gW = _wgrad_bilinear(W, Pd, s2["g"], alpha)
        return None, None, None, None, None, gd, None, gW


def bilinear_ang(rbf_W1, ang, x, W, sp, alpha=1.0):
    out, _ = _BilinearAng2.apply(rbf_W1, ang, x, W, sp, float(alpha))
    return out


# =================================================================================== distances and triplet angles from R
class _Dist2(torch.autograd.Function):
    """D[e] = |R[a(e)] - R[c(e)]| (gemnet.py:261-286), twice differentiable on three kernels (csrc/geometry2.hip) — the
    reference's gather / sub / pow / sum / sqrt and their two generations of autograd nodes are ~40 pointwise launches."""

    @staticmethod
    def forward(ctx, R, ri_c, ri_a):
        ctx.save_for_backward(R)
        ctx.ri = (ri_c, ri_a)
        return K.dist_fwd(R, ri_c.idx32, ri_a.idx32)

    @staticmethod
    def backward(ctx, gD):
        (R,) = ctx.saved_tensors
        ri_c, ri_a = ctx.ri
        if gD is None or not ctx.needs_input_grad[0]:
            return None, None, None
        if torch.is_grad_enabled():
            return _Dist2B.apply(gD, R, ri_c, ri_a), None, None
        W = K.dist_bwd(gD.contiguous(), R, ri_c.idx32, ri_a.idx32)
        return K.segsum_multi([(W, *ri_a.csr, 1.0), (W, *ri_c.csr, -1.0)], ri_a.n_rows), None, None


class _Dist2B(torch.autograd.Function):
    @staticmethod
    def forward(ctx, gD, R, ri_c, ri_a):
        gD = gD.contiguous()
        ctx.save_for_backward(gD, R)
        ctx.ri = (ri_c, ri_a)
        W = K.dist_bwd(gD, R, ri_c.idx32, ri_a.idx32)
        return K.segsum_multi([(W, *ri_a.csr, 1.0), (W, *ri_c.csr, -1.0)], ri_a.n_rows)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, tR):
        gD, R = ctx.saved_tensors
        ri_c, ri_a = ctx.ri
        if tR is None:
            return None, None, None, None
        want_H = ctx.needs_input_grad[1] and ops.position_second_order()
        Dd, H = K.dist_jvp(R, tR.contiguous(), gD, ri_c.idx32, ri_a.idx32, want_D=ctx.needs_input_grad[0], want_H=want_H)
        gR = K.segsum_multi([(H, *ri_a.csr, 1.0), (H, *ri_c.csr, -1.0)], ri_a.n_rows) if want_H else None
        return Dd, gR, None, None


def distances(R, ri_c, ri_a):
    return _Dist2.apply(R, ri_c, ri_a)


class _Angle2(torch.autograd.Function):
    """theta[t] = atan2(max(|u x v|, 1e-9), u . v), u = R[c] - R[a], v = R[b] - R[a] (gemnet.py:288-311, :420-451), twice
    differentiable: value, first adjoint and the tangent pass (dual numbers through the adjoint) are one kernel each."""

    @staticmethod
    def forward(ctx, R, ri_c, ri_a, ri_b):
        ctx.save_for_backward(R)
        ctx.ri = (ri_c, ri_a, ri_b)
        return K.angle_fwd(R, ri_c.idx32, ri_a.idx32, ri_b.idx32)

    @staticmethod
    def backward(ctx, g):
        (R,) = ctx.saved_tensors
        ri_c, ri_a, ri_b = ctx.ri
        if g is None or not ctx.needs_input_grad[0]:
            return None, None, None, None
        if torch.is_grad_enabled():
            return _Angle2B.apply(g, R, ri_c, ri_a, ri_b), None, None, None
        Gc, Gb = K.angle_bwd(g.contiguous(), R, ri_c.idx32, ri_a.idx32, ri_b.idx32)
        return _atoms3(Gc, Gb, ri_c, ri_a, ri_b), None, None, None


def _atoms3(Gc, Gb, ri_c, ri_a, ri_b):
    """Per-triplet position terms (c: Gc, b: Gb, a: -(Gc + Gb)) -> atoms, one multi-term segmented sum."""
    return K.segsum_multi([(Gc, *ri_c.csr, 1.0), (Gb, *ri_b.csr, 1.0), (Gc, *ri_a.csr, -1.0), (Gb, *ri_a.csr, -1.0)],
                          ri_c.n_rows)


class _Angle2B(torch.autograd.Function):
    @staticmethod
    def forward(ctx, g, R, ri_c, ri_a, ri_b):
        g = g.contiguous()
        ctx.save_for_backward(g, R)
        ctx.ri = (ri_c, ri_a, ri_b)
        Gc, Gb = K.angle_bwd(g, R, ri_c.idx32, ri_a.idx32, ri_b.idx32)
        return _atoms3(Gc, Gb, ri_c, ri_a, ri_b)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, tR):
        g, R = ctx.saved_tensors
        ri_c, ri_a, ri_b = ctx.ri
        if tR is None:
            return None, None, None, None, None
        want_H = ctx.needs_input_grad[1] and ops.position_second_order()
        thd, Hc, Hb = K.angle_jvp(R, tR.contiguous(), g, ri_c.idx32, ri_a.idx32, ri_b.idx32,
                                  want_theta=ctx.needs_input_grad[0], want_H=want_H)
        gR = _atoms3(Hc, Hb, ri_c, ri_a, ri_b) if want_H else None
        return thd, gR, None, None, None


def triplet_angles(R, ri_c, ri_a, ri_b):
    return _Angle2.apply(R, ri_c, ri_a, ri_b)


# ==================================================================== the two quadruplet angles from R, in angle form
class _QuadAngles2(torch.autograd.Function):
    """ang (Q,4) = (sin, cos) of Phi_cab and Theta_cabd from the four atoms of every quadruplet (gemnet.py:334-418:
    calculate_angles — two neighbour angles, two vector rejections, the dihedral), twice differentiable on three kernels
    (csrc/geometry.hip: value, first adjoint, tangent) — the composite closure spends ~25 ATen launches over (Q,3) arrays
    (9 M rows at B = 32) per pass on the same arithmetic.  Gradient convention of the angle form: the slot of `ang` carries
    g_ang (Q,4) = (dE/dPhi, dE/dTheta, 0, 0).  Second-order POSITION terms (the Hessian of the two angles) are not provided:
    the fused quadruplet path is only taken when the caller does not differentiate w.r.t. positions again
    (ops.position_graph)."""

    @staticmethod
    def forward(ctx, R, ri_c, ri_a, ri_b, ri_d, plan):
        ctx.save_for_backward(R)
        ctx.cfg = (ri_c, ri_a, ri_b, ri_d, plan)
        return K.quad_angles_fwd(R, ri_c.idx32, ri_a.idx32, ri_b.idx32, ri_d.idx32)

    @staticmethod
    def backward(ctx, g_ang):
        (R,) = ctx.saved_tensors
        if g_ang is None or not ctx.needs_input_grad[0]:
            return (None,) * 6
        if torch.is_grad_enabled():
            return (_QuadAngles2B.apply(g_ang, R, *ctx.cfg),) + (None,) * 5
        return (ops.quad_angles_adjoint(g_ang.contiguous(), R, *ctx.cfg),) + (None,) * 5


class _QuadAngles2B(torch.autograd.Function):
    @staticmethod
    def forward(ctx, g_ang, R, ri_c, ri_a, ri_b, ri_d, plan):
        ctx.save_for_backward(R)
        ctx.cfg = (ri_c, ri_a, ri_b, ri_d)
        return ops.quad_angles_adjoint(g_ang.contiguous(), R, ri_c, ri_a, ri_b, ri_d, plan)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, tR):
        (R,) = ctx.saved_tensors
        if tR is None or not ctx.needs_input_grad[0]:
            return (None,) * 7
        ri_c, ri_a, ri_b, ri_d = ctx.cfg
        t_ang = K.quad_angles_jvp(R, tR.contiguous(), ri_c.idx32, ri_a.idx32, ri_b.idx32, ri_d.idx32)
        return (t_ang,) + (None,) * 6


def quad_angles(R, ri_c, ri_a, ri_b, ri_d, plan=None):
    return _QuadAngles2.apply(R, ri_c, ri_a, ri_b, ri_d, plan)
