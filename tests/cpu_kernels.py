"""TEST-ONLY CPU emulation of the launchers in gemnet_pytorch_amd/kernels.py.

Lets the CPU test-suite exercise the host logic that sits ABOVE the C ABI — the autograd closure
of gemnet_pytorch_amd/ops.py (first- and second-order), the index plans and the GemNet module
wiring — on a machine without a GPU, by monkeypatching the launchers with restatements of the
kernel semantics documented in include/gemnet_hip.h.  It is not importable from the product
package and is never used when a GPU is present (the -m gpu tests call the real library).
"""
import contextlib

import torch

from oracle import basis_oracle as B

import gemnet_pytorch_amd.kernels as K


def _act(z, k):
    s = torch.sigmoid(z)
    if k == 0:
        return z * s / 0.6
    if k == 1:
        return s * (1 + z * (1 - s)) / 0.6
    if k == 2:
        return s * (1 - s) * (2 + z * (1 - 2 * s)) / 0.6
    return s * (1 - s) * (3 * (1 - 2 * s) + z * (1 - 6 * s + 6 * s * s)) / 0.6


def gemm(A, B_, trans_a=False, trans_b=False, *, a_dact_pre=None, act=False, pre_out=False, mul=None,
         alpha=1.0, res=None, beta=1.0, gadd1=None, gidx1=None, gadd2=None, gidx2=None,
         ridx=None, res2=None, beta2=1.0, cfg=-1, out=None):
    a = A
    if a_dact_pre is not None:
        a = a * _act(a_dact_pre, 1)
    a = a.t() if trans_a else a
    b = B_ if trans_b else B_.t()
    z = a @ b
    if gadd1 is not None:
        z = z + gadd1[gidx1.long()]
    if gadd2 is not None:
        z = z + gadd2[gidx2.long()]
    y = _act(z, 0) if act else z
    if mul is not None:
        y = y * mul
    y = y * alpha
    if res is not None:
        y = (y + (res if ridx is None else res[ridx.long()])) * beta
    if res2 is not None:
        y = (y + res2) * beta2
    if out is not None:
        y = out.copy_(y)
    return (y, z) if pre_out else y


def dact_mul(g, z, act, mul, c, want_gmul=False):
    a = _act(z, 1) if act else 1.0
    dz = g * c * (mul if mul is not None else 1.0) * a
    gmul = g * c * (_act(z, 0) if act else z) if want_gmul else None
    return dz, gmul


def gather(x, idx32):
    return x[idx32.long()]


def segsum(y, perm, seg_off, n_rows):
    src = y if perm is None else y[perm.long()]
    counts = (seg_off[1:] - seg_off[:-1]).long()
    dest = torch.repeat_interleave(torch.arange(n_rows), counts)
    out = torch.zeros((n_rows,) + tuple(y.shape[1:]), dtype=y.dtype)
    return out.index_add(0, dest, src[: dest.shape[0]])


def segsum_multi(terms, n_rows):
    return sum(sg * segsum(y, perm, seg, n_rows) for (y, perm, seg, sg) in terms)


def bmm(A, B_, ta, tb):
    a = A.transpose(1, 2) if ta else A
    b = B_.transpose(1, 2) if tb else B_
    return torch.bmm(a, b)


def ssilu(x, k):
    return _act(x, k)


def pm(z, k, a=None, b=None, d=None, c=1.0):
    out = _act(z, k) if k >= 0 else torch.ones_like(a)
    for t in (a, b, d):
        if t is not None:
            out = out * t
    return out * c


def bil_reduce(Y, x, sp):
    xt = x[sp.expand.idx32.long()]
    out = torch.zeros((sp.n_reduce, Y.shape[1], x.shape[1]), dtype=x.dtype)
    return out.index_add(0, sp.reduce.idx32.long(), Y[:, :, None] * xt[:, None, :])


def bil_reduce_t(Y, D, sp):
    if is_angle_form(Y, D.shape[1]):
        Y = _ang_to_Y(Y)
    contrib = torch.einsum("ts,tsc->tc", Y, D[sp.reduce.idx32.long()])
    out = torch.zeros((sp.n_expand, D.shape[2]), dtype=D.dtype)
    return out.index_add(0, sp.expand.idx32.long(), contrib)


def bil_dot(D, x, sp):
    return torch.einsum("tsc,tc->ts", D[sp.reduce.idx32.long()], x[sp.expand.idx32.long()])


def _nth(fn, x, order):
    """order-th derivative of every output column w.r.t. the 1-D x (autograd on the oracle formula)."""
    if order == 0:
        with torch.no_grad():
            return fn(x.detach())
    with torch.enable_grad():
        xx = x.detach().clone().requires_grad_(True)
        y = fn(xx)
        flat = y.reshape(y.shape[0], -1)
        cols = []
        for j in range(flat.shape[1]):
            g = flat[:, j]
            for _ in range(order):
                g = torch.autograd.grad(g.sum(), xx, create_graph=True, allow_unused=True)[0]
                if g is None or not g.requires_grad:
                    g = torch.zeros_like(xx) if g is None else g
                    break
            cols.append(g.detach())
        return torch.stack(cols, 1).reshape(y.shape)


def bessel_rbf(d, freq, cutoff, p, kd, kf):
    """Elementwise mixed derivative on the (E, R) grid via autograd on the oracle formula."""
    with torch.enable_grad():
        D = d.detach()[:, None].expand(-1, freq.shape[0]).clone().requires_grad_(True)
        F = freq.detach()[None, :].expand(d.shape[0], -1).clone().requires_grad_(True)
        ds = D / cutoff
        y = B.envelope(ds, p) * (2 / cutoff) ** 0.5 * torch.sin(F * ds) / D
        for _ in range(kd):
            y = torch.autograd.grad(y.sum(), D, create_graph=True)[0]
        for _ in range(kf):
            y = torch.autograd.grad(y.sum(), F, create_graph=True)[0]
    return y.detach()


def sph_radial(d, z, nrm, cutoff, p, kd):
    S, R = z.shape
    return _nth(lambda x: B.sph_bessel_radial(x, S, R, cutoff, p), d, kd)


def ylm0(theta, S, k):
    return _nth(lambda x: B.real_sph_harm_l0(S, x), theta, k)


def ylm(theta, phi, S, kt, kp):
    with torch.enable_grad():
        tt = theta.detach().clone().requires_grad_(True)
        pp = phi.detach().clone().requires_grad_(True)
        y = B.real_sph_harm_full(S, tt, pp)
        cols = []
        for j in range(S * S):
            g = y[:, j]
            for var, n in ((tt, kt), (pp, kp)):
                for _ in range(n):
                    if not g.requires_grad:
                        g = torch.zeros_like(tt)
                        break
                    g = torch.autograd.grad(g.sum(), var, create_graph=True, allow_unused=True)[0]
                    if g is None:
                        g = torch.zeros_like(tt)
            cols.append(g.detach())
    return torch.stack(cols, 1)


def _edge_geom(R, idx_c, idx_a):
    Vv = R[idx_a.long()] - R[idx_c.long()]
    D = torch.sqrt((Vv ** 2).sum(1))
    return D, Vv / D[:, None]


def edge_basis_fwd(R, idx_c, idx_a, freq, z, nrm, cutoff, p, want_V=False, want_rbf=True):
    D, V = _edge_geom(R, idx_c, idx_a)
    S, NR = z.shape
    rbf = B.bessel_rbf(D, freq, cutoff, p) if want_rbf else None
    rad = B.sph_bessel_radial(D, S, NR, cutoff, p)
    return D, (V if want_V else None), rbf, rad


def edge_basis_bwd(gD, g_rbf, g_rad, R, idx_c, idx_a, freq, z, nrm, cutoff, p):
    with torch.enable_grad():
        S, NR = z.shape
        Rr = R.detach().clone().requires_grad_(True)
        Vv = Rr[idx_a.long()] - Rr[idx_c.long()]
        D = torch.sqrt((Vv ** 2).sum(1))
        tot = 0.0
        if gD is not None:
            tot = tot + (gD * D).sum()
        if g_rbf is not None:
            tot = tot + (g_rbf * B.bessel_rbf(D, freq.detach(), cutoff, p)).sum()
        if g_rad is not None:
            tot = tot + (g_rad * B.sph_bessel_radial(D, S, NR, cutoff, p)).sum()
        (gDtot,) = torch.autograd.grad(tot, D)
    return (gDtot[:, None] * (Vv / D[:, None])).detach()


def _angle(R, tc, ta, tb):
    u = R[tc.long()] - R[ta.long()]
    v = R[tb.long()] - R[ta.long()]
    x = (u * v).sum(1)
    y = torch.linalg.cross(u, v, dim=-1).norm(dim=-1).clamp(min=1e-9)
    return torch.atan2(y, x)


def trip_basis_fwd(R, tc, ta, tb, S, want_theta=False):
    th = _angle(R, tc, ta, tb)
    return B.real_sph_harm_l0(S, th), (th if want_theta else None)


def trip_basis_bwd(gY, R, tc, ta, tb):
    with torch.enable_grad():
        Rc = R[tc.long()].detach().clone().requires_grad_(True)
        Ra = R[ta.long()].detach()
        Rb = R[tb.long()].detach().clone().requires_grad_(True)
        u, v = Rc - Ra, Rb - Ra
        x = (u * v).sum(1)
        y = torch.linalg.cross(u, v, dim=-1).norm(dim=-1).clamp(min=1e-9)
        Y = B.real_sph_harm_l0(gY.shape[1], torch.atan2(y, x))
        Gc, Gb = torch.autograd.grad((gY * Y).sum(), (Rc, Rb))
    return Gc, Gb


def chain(prog, mode=None):
    """Interpret a ChainProgram on whole matrices (slots = (M, 128) tensors).  Exact arithmetic; the launch rules of the
    mode still apply (a program the device kernel would refuse must not pass on the emulation)."""
    from gemnet_pytorch_amd import kernels as _K
    if (mode or _K.current_mode()) == "h3" and _K.CHAIN_LAYOUT != "row" and _K.h3_hazards(prog):
        raise RuntimeError(f"chain: ops {_K.h3_hazards(prog)} are not representable in mode 'h3'")
    M = prog.M
    dt = None
    for o in prog.ops:
        for k in ("src", "W"):
            if o.get(k) is not None:
                dt = o[k].dtype
    slots = [torch.zeros(M, 128, dtype=dt) for _ in range(3)]

    def sel(x, N):
        if x is None:
            return None
        return slots[x][:, :N] if isinstance(x, int) else x

    def source(src, Z):
        """alpha * (ssilu''(Z) | 1) * P * Q of a second-order source term (gn_chain_op.src_*)."""
        if src is None:
            return 0.0
        t = src["P"] * src["alpha"]
        if src["Q"] is not None:
            t = t * src["Q"]
        if src["mode"] == 1:
            t = t * _act(Z, 2)
        return t

    for o in prog.ops:
        if o["kind"] == "load":
            src = o["src"] if o["rows"] is None else o["src"][o["rows"].long()]
            src = src * o.get("alpha", 1.0)
            slots[o["slot"]] = torch.zeros(M, 128, dtype=dt)
            slots[o["slot"]][:, :src.shape[1]] = src
            if o.get("y2", -1) >= 0:
                y2 = src * o.get("alpha2", 1.0)
                if o.get("Z2") is not None:
                    m2 = o.get("mode2", 0)
                    y2 = y2 * (_act(o["Z2"], 1) if m2 == 0 else (o["Z2"] if m2 == 1 else _act(o["Z2"], 0)))
                y2 = y2 + source(o.get("add2"), o.get("Z2"))
                slots[o["y2"]] = torch.zeros(M, 128, dtype=dt)
                slots[o["y2"]][:, :src.shape[1]] = y2
        elif o["kind"] == "scale":
            Z, out = o["Z"], o["out"]
            w = o["width"] or (Z.shape[1] if Z is not None else out.shape[1])
            v = slots[o["a_slot"]][:, :w] * o["alpha"]
            if Z is not None:
                mode = o.get("mode", 0)
                v = v * (_act(Z, 1) if mode == 0 else (Z if mode == 1 else _act(Z, 0)))
            v = v + source(o.get("add"), Z)
            new = slots[o["slot"]].clone()
            new[:, :w] = v
            slots[o["slot"]] = new
            if out is not None:
                out.copy_(v)
        elif o["kind"] == "store":
            o["out"].copy_(slots[o["slot"]][:, :o["out"].shape[1]])
        else:
            W = o["W"]
            N, Kd = W.shape
            z = slots[o["a_slot"]][:, :Kd] @ W.t()
            if o["gadd1"] is not None:
                z = z + o["gadd1"][o["gidx1"].long()]
            if o["gadd2"] is not None:
                z = z + o["gadd2"][o["gidx2"].long()]
            if o["pre_out"] is not None:
                o["pre_out"].copy_(_act(z, 1) if o.get("pre_deriv") else z)
            y = _act(z, 0) if o["act"] else z
            a_val = y
            mul = sel(o["mul"], N)
            if mul is not None:
                mm = o.get("mul_mode", 1)
                y = y * (mul if mm <= 1 else _act(mul, 1 if mm == 2 else 0))
            y = y * o["alpha"]
            y = y + source(o.get("add"), mul)
            res = sel(o["res"], N)
            if res is not None:
                if o["res_rows"] is not None:
                    res = res[o["res_rows"].long()]
                y = (y + res) * o["beta"]
            res2 = sel(o["res2"], N)
            if res2 is not None:
                y = (y + res2) * o["beta2"]
            if o["out"] is not None:
                o["out"].copy_(y)
            y2 = None
            if o.get("y2", -1) >= 0 or o.get("out2") is not None:
                y2 = (a_val if o.get("y2_src", 0) else y) * o.get("alpha2", 1.0)
                if o.get("Z2") is not None:
                    m2 = o.get("mode2", 0)
                    y2 = y2 * (_act(o["Z2"], 1) if m2 == 0 else (o["Z2"] if m2 == 1 else _act(o["Z2"], 0)))
                y2 = y2 + source(o.get("add2"), o.get("Z2"))
                if o.get("out2") is not None:
                    o["out2"].copy_(y2)
            if o["slot"] >= 0:
                new = slots[o["slot"]].clone()
                new[:, :N] = y
                slots[o["slot"]] = new
            if y2 is not None and o.get("y2", -1) >= 0:
                new = slots[o["y2"]].clone()
                new[:, :N] = y2
                slots[o["y2"]] = new


def is_angle_form(Y, S):
    return S == 49 and Y.dim() == 2 and Y.shape[1] == 4


def _ang_to_Y(ang):
    """(Q,4) (sin, cos) pairs -> the (Q,49) harmonics (what the *_ang kernels rebuild in LDS)."""
    return B.real_sph_harm_full(7, torch.atan2(ang[:, 0], ang[:, 1]), torch.atan2(ang[:, 2], ang[:, 3]))


def bil_train_supported(S, C, I):
    return True          # the emulation has no shape restriction: the training form is exercised for every test model


def bil_reduce_project(Y, x, Bm, sp, Sm_init=None, B2=None, Sm2=None, want_P=True):
    if is_angle_form(Y, Bm.shape[1]):
        Y = _ang_to_Y(Y)
    Sm = bil_reduce(Y, x, sp)
    if Sm_init is not None:
        Sm = Sm + Sm_init
    if not want_P:
        return Sm, None
    P = torch.bmm(Bm.transpose(1, 2), Sm)
    if B2 is not None:
        P = P + torch.bmm(B2.transpose(1, 2), Sm2)
    return Sm, P


def bil_dy_multi(dSm_list, x_list, sp, ang=None):
    dY = sum(bil_dot(d, x, sp) for d, x in zip(dSm_list, x_list))
    if ang is None:
        return dY
    with torch.enable_grad():   # chain rule to the two angles
        th = torch.atan2(ang[:, 0], ang[:, 1]).detach().requires_grad_(True)
        ph = torch.atan2(ang[:, 2], ang[:, 3]).detach().requires_grad_(True)
        gt, gp = torch.autograd.grad((dY * B.real_sph_harm_full(7, th, ph)).sum(), (th, ph))
    out = torch.zeros_like(ang)
    out[:, 0], out[:, 1] = gt, gp
    return out


def quad_angles_fwd(R, qc, qa, qb, qd):
    phi, th = _quad_angles(R[qc.long()], R[qa.long()], R[qb.long()], R[qd.long()])
    return torch.stack([torch.sin(phi), torch.cos(phi), torch.sin(th), torch.cos(th)], dim=1)


def quad_angles_bwd(g_ang, R, qc, qa, qb, qd, packed=False):
    with torch.enable_grad():
        Rc = R[qc.long()].detach().clone().requires_grad_(True)
        Rb = R[qb.long()].detach().clone().requires_grad_(True)
        Rd = R[qd.long()].detach().clone().requires_grad_(True)
        Ra = R[qa.long()].detach()
        phi, th = _quad_angles(Rc, Ra, Rb, Rd)
        Gc, Gb, Gd = torch.autograd.grad((g_ang[:, 0] * phi + g_ang[:, 1] * th).sum(), (Rc, Rb, Rd))
    if not packed:
        return Gc, Gb, Gd
    Gbd = torch.zeros((Gb.shape[0], 8), dtype=Gb.dtype)
    Gbd[:, 0:3], Gbd[:, 4:7] = Gb, Gd
    return Gc, Gbd


def _jvp(f, x, t):
    """J t by the double-backward trick (plain autograd: `torch.func.jvp` leaves cyclic garbage behind, and the leak test of
    the training records counts what the cyclic collector finds)."""
    with torch.enable_grad():
        x = x.detach().clone().requires_grad_(True)
        y = f(x)
        v = torch.zeros_like(y, requires_grad=True)
        (g,) = torch.autograd.grad(y, x, v, create_graph=True)
        (jt,) = torch.autograd.grad(g, v, t.detach())
    return jt.detach()


def _ang_to_dY(ang, tang):
    """The tangent rows dY = Y_theta dtheta + Y_phi dphi (forward-mode through the oracle's harmonics)."""
    th = torch.atan2(ang[:, 0], ang[:, 1]).detach()
    ph = torch.atan2(ang[:, 2], ang[:, 3]).detach()
    return _jvp(lambda a: B.real_sph_harm_full(7, a[:, 0], a[:, 1]), torch.stack([th, ph], 1), tang[:, 0:2])


def quad_angles_jvp(R, tR, qc, qa, qb, qd):
    idx = [i.long() for i in (qc, qa, qb, qd)]
    t = _jvp(lambda Rx: torch.stack(_quad_angles(Rx[idx[0]], Rx[idx[1]], Rx[idx[2]], Rx[idx[3]]), dim=1), R, tR)
    out = torch.zeros((t.shape[0], 4), dtype=R.dtype)
    out[:, 0:2] = t
    return out


def bil_reduce_project_tan(ang, tang, x, tx, Bm, tB, Sm, sp, want_P=True):
    Smd = torch.zeros((sp.n_reduce, Bm.shape[1], x.shape[1]), dtype=x.dtype)
    if tang is not None:
        Smd = Smd + bil_reduce(_ang_to_dY(ang, tang), x, sp)
    if tx is not None:
        Smd = Smd + bil_reduce(_ang_to_Y(ang), tx, sp)
    if not want_P:
        return Smd, None
    Pd = torch.bmm(Bm.transpose(1, 2), Smd)
    if tB is not None:
        Pd = Pd + torch.bmm(tB.transpose(1, 2), Sm)
    return Smd, Pd


def bil_reduce_t_tan(ang, tang, D1, D2, sp):
    dx = bil_reduce_t(_ang_to_dY(ang, tang), D2, sp)
    if D1 is not None:
        dx = dx + bil_reduce_t(_ang_to_Y(ang), D1, sp)
    return dx


def bil_ang_train_supported(S, C, I):
    return S == 49       # the emulation takes any channel widths (the test models are small)


def bil_project_bwd(dP, Sm, Bm, x, sp, dY_accum=None, want_dY=True, gB_accum=None, dSm_accum=None):
    gB = torch.bmm(Sm, dP.transpose(1, 2))
    if gB_accum is not None:
        gB = gB_accum.add_(gB)
    dSm = torch.bmm(Bm, dP)
    if dSm_accum is not None:
        dSm = dSm_accum.add_(dSm)
    if not want_dY:
        return gB, dSm, None
    dY = bil_dot(dSm, x, sp)
    if dY_accum is not None:
        dY_accum += dY
        dY = dY_accum
    return gB, dSm, dY


def bil_fused_bwd_supported(S, C, I, O):
    return (S, C, I, O) == (7, 64, 16, 64)


def bil_fused_bwd(g, W2, Sm, Bm, alpha=1.0, gB_accum=None, W2_planes=None):
    E, S, C = Sm.shape
    I = Bm.shape[2]
    dP = (alpha * (g @ W2.t())).reshape(E, I, C)
    gB = torch.bmm(Sm, dP.transpose(1, 2))
    if gB_accum is not None:
        gB = gB_accum.add_(gB)
    return gB, torch.bmm(Bm, dP)


def _quad_angles(Rc, Ra, Rb, Rd):
    def ang(u, v):
        x = (u * v).sum(1)
        y = torch.linalg.cross(u, v, dim=-1).norm(dim=-1).clamp(min=1e-9)
        return torch.atan2(y, x)

    def rej(x, n):
        return x - ((x * n).sum(1) / (n * n).sum(1))[:, None] * n
    uac, uab, ubd = Rc - Ra, Rb - Ra, Rd - Rb
    return ang(uab, uac), ang(rej(uac, uab), rej(ubd, -uab))


def quad_basis_fwd(R, qc, qa, qb, qd, S):
    phi, th = _quad_angles(R[qc.long()], R[qa.long()], R[qb.long()], R[qd.long()])
    return B.real_sph_harm_full(S, phi, th)


def quad_basis_bwd(gY, R, qc, qa, qb, qd, S):
    with torch.enable_grad():
        Rc = R[qc.long()].detach().clone().requires_grad_(True)
        Rb = R[qb.long()].detach().clone().requires_grad_(True)
        Rd = R[qd.long()].detach().clone().requires_grad_(True)
        Ra = R[qa.long()].detach()
        phi, th = _quad_angles(Rc, Ra, Rb, Rd)
        Gc, Gb, Gd = torch.autograd.grad((gY * B.real_sph_harm_full(S, phi, th)).sum(), (Rc, Rb, Rd))
    return Gc, Gb, Gd


def quad_basis_bwd_packed(gY, R, qc, qa, qb, qd, S):
    Gc, Gb, Gd = quad_basis_bwd(gY, R, qc, qa, qb, qd, S)
    Gbd = torch.zeros((Gb.shape[0], 8), dtype=Gb.dtype)
    Gbd[:, 0:3], Gbd[:, 4:7] = Gb, Gd
    return Gc, Gbd


def bil_fused_fwd(Y, x, B, W2T, sp, alpha=1.0, W2T_planes=None):
    Sm, P = bil_reduce_project(Y, x, B, sp)
    return Sm, (P.reshape(P.shape[0], -1) @ W2T.t()) * alpha


def _angle_uv(u, v):
    x = (u * v).sum(1)
    y = torch.linalg.cross(u, v, dim=-1).norm(dim=-1).clamp(min=1e-9)
    return torch.atan2(y, x)


def dist_fwd(R, id_c, id_a):
    v = R[id_a.long()] - R[id_c.long()]
    return torch.sqrt((v * v).sum(1))


def dist_bwd(gD, R, id_c, id_a):
    v = R[id_a.long()] - R[id_c.long()]
    return gD[:, None] * v / torch.sqrt((v * v).sum(1))[:, None]


def dist_jvp(R, tR, gD, id_c, id_a, want_D=True, want_H=True):
    with torch.enable_grad():
        v = (R[id_a.long()] - R[id_c.long()]).detach().clone().requires_grad_(True)
        tv = (tR[id_a.long()] - tR[id_c.long()]).detach()
        gg = (torch.ones(v.shape[0], dtype=R.dtype) if gD is None else gD.detach().clone()).requires_grad_(True)
        D = torch.sqrt((v * v).sum(1))
        (W,) = torch.autograd.grad(D, v, gg, create_graph=True)
        s = (W * tv).sum()
        Dd, H = torch.autograd.grad(s, (gg, v))
    return (Dd if want_D else None), (H if want_H else None)


def angle_fwd(R, tc, ta, tb):
    Ra = R[ta.long()]
    return _angle_uv(R[tc.long()] - Ra, R[tb.long()] - Ra)


def angle_bwd(g, R, tc, ta, tb):
    with torch.enable_grad():
        Ra = R[ta.long()]
        u = (R[tc.long()] - Ra).detach().clone().requires_grad_(True)
        v = (R[tb.long()] - Ra).detach().clone().requires_grad_(True)
        Gc, Gb = torch.autograd.grad(_angle_uv(u, v), (u, v), g)
    return Gc, Gb


def angle_jvp(R, tR, g, tc, ta, tb, want_theta=True, want_H=True):
    with torch.enable_grad():
        Ra, tRa = R[ta.long()], tR[ta.long()]
        u = (R[tc.long()] - Ra).detach().clone().requires_grad_(True)
        v = (R[tb.long()] - Ra).detach().clone().requires_grad_(True)
        du, dv = (tR[tc.long()] - tRa).detach(), (tR[tb.long()] - tRa).detach()
        gg = (torch.ones(u.shape[0], dtype=R.dtype) if g is None else g.detach().clone()).requires_grad_(True)
        Gu, Gv = torch.autograd.grad(_angle_uv(u, v), (u, v), gg, create_graph=True)
        s = (Gu * du).sum() + (Gv * dv).sum()
        thd, Hc, Hb = torch.autograd.grad(s, (gg, u, v))
    return (thd if want_theta else None), (Hc if want_H else None), (Hb if want_H else None)


def gather_mul(x, idx32, m, scale=1.0):
    return x[idx32.long()] * m * scale


def rbf_aggregate_fwd(m, rbf, W, perm, seg_off, n_atoms, scale):
    return segsum(m * (rbf @ W.t()), perm, seg_off, n_atoms) * scale


def rbf_aggregate_bwd(g_out, m, rbf, W, id_a32, scale, want_m=True, want_rbf=True, acc_m=None, acc_rbf=None):
    g = g_out[id_a32.long()] * scale
    g_m = g * (rbf @ W.t()) if (want_m or acc_m is not None) else None
    g_rbf = (g * m) @ W if (want_rbf or acc_rbf is not None) else None
    if acc_m is not None:
        g_m = acc_m.add_(g_m)
    if acc_rbf is not None:
        g_rbf = acc_rbf.add_(g_rbf)
    return g_m, g_rbf


def cbf_project_supported(rad, y, W):
    return rad.dim() == 3 and rad.shape[1] * rad.shape[2] <= 64 and W.shape[0] <= 16


def cbf_project_fwd(rad, ie32, y, W):
    return (rad[ie32.long()] * y[:, :, None]).reshape(y.shape[0], -1) @ W.t()


def cbf_project_bwd(g, rad, seg_off, y, W):
    E, S, R = rad.shape
    t = (g @ W).reshape(-1, S, R)
    counts = (seg_off[1:] - seg_off[:-1]).long()
    e_of = torch.repeat_interleave(torch.arange(E), counts)
    g_y = (t * rad[e_of]).sum(dim=2)
    g_rad = torch.zeros_like(rad).index_add(0, e_of, t * y[:, :, None])
    return g_rad, g_y


_NAMES = ["cbf_project_supported", "cbf_project_fwd", "cbf_project_bwd", "quad_angles_jvp", "bil_reduce_project_tan", "bil_reduce_t_tan", "bil_ang_train_supported", "dist_fwd", "dist_bwd", "dist_jvp", "angle_fwd", "angle_bwd", "angle_jvp", "bil_train_supported", "gather_mul", "bil_fused_bwd", "bil_fused_bwd_supported", "segsum_multi", "is_angle_form", "quad_angles_fwd", "quad_angles_bwd", "rbf_aggregate_fwd", "rbf_aggregate_bwd", "bil_fused_fwd", "quad_basis_fwd", "quad_basis_bwd", "quad_basis_bwd_packed", "bil_reduce_project", "bil_project_bwd", "bil_dy_multi", "chain", "edge_basis_fwd", "edge_basis_bwd", "trip_basis_fwd", "trip_basis_bwd", "gemm", "dact_mul", "gather", "segsum", "bmm", "ssilu", "pm", "bil_reduce", "bil_reduce_t", "bil_dot",
          "bessel_rbf", "sph_radial", "ylm0", "ylm"]


@contextlib.contextmanager
def emulate():
    """Swap the HIP launchers for the CPU restatements above (tests only)."""
    saved = {n: getattr(K, n) for n in _NAMES}
    try:
        for n in _NAMES:
            setattr(K, n, globals()[n])
        yield
    finally:
        for n, f in saved.items():
            setattr(K, n, f)
