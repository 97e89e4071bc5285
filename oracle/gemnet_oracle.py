"""TEST INFRASTRUCTURE (oracle) — plain-PyTorch CPU restatement of GemNet's forward/energy/force path.

This is the checker the HIP path is diffed against and the CPU baseline timed by
bench.py.  It is functional (parameters come in as a {name: tensor} dict using the
reference's `named_parameters()` names), dtype-generic (float64 for goldens), and uses
index_add_ where the reference calls torch_scatter and CSR-free dense maths where the
reference builds zero-padded (E, Kmax, .) tensors (the results are identical: the padding
rows are zeros).

Reference followed (file:line under /root/reference/gemnet/model):
  gemnet.py:261-286 interatomic vectors, :288-311 neighbour angles, :313-332 rejection,
  :334-418 quadruplet angles, :420-451 triplet angles, :453-615 forward;
  layers/interaction_block.py:158-234, :363-422, :517-566, :653-696;
  layers/efficient.py:41-57, :159-189; layers/atom_update_block.py:55-72, :157-193;
  layers/embedding_block.py:27-34, :60-75; layers/base_layers.py:44-58, :84-89;
  layers/scaling.py:170-174.
Pinned by tests/golden/model_*.npz (E, F and parameter-gradient goldens produced by the
reference itself in float64 with the same deterministic weights).
"""
import json
import math

import numpy as np
import torch

from . import basis_oracle as B

INV_SQRT_2 = 1 / (2.0 ** 0.5)
INV_SQRT_3 = 1 / (3.0 ** 0.5)

DEFAULTS = dict(num_targets=1, direct_forces=False, cutoff=5.0, int_cutoff=10.0,
                envelope_exponent=5, extensive=True, forces_coupled=False)


# ----------------------------------------------------------------------------------------
# parameter inventory (the reference's named_parameters() order)
# ----------------------------------------------------------------------------------------
def param_spec(cfg):
    """[(name, shape, kind)], kind in {dense, emb, freq, scale, eff}."""
    c = {**DEFAULTS, **cfg}
    S, Rn = c["num_spherical"], c["num_radial"]
    ea, ee = c["emb_size_atom"], c["emb_size_edge"]
    et, eq = c["emb_size_trip"], c["emb_size_quad"]
    er, ec, es = c["emb_size_rbf"], c["emb_size_cbf"], c["emb_size_sbf"]
    bt, bq = c["emb_size_bil_trip"], c["emb_size_bil_quad"]
    T = c["triplets_only"]
    spec = [("rbf_basis.frequencies", (Rn,), "freq")]
    if not T:
        spec += [("mlp_rbf4.weight", (er, Rn), "dense"),
                 ("mlp_cbf4.weight", (ec, Rn * S), "dense"),
                 ("mlp_sbf4.weight", (S * S, Rn, es), "eff")]
    spec += [("mlp_rbf3.weight", (er, Rn), "dense"),
             ("mlp_cbf3.weight", (S, Rn, ec), "eff"),
             ("mlp_rbf_h.weight", (er, Rn), "dense"),
             ("mlp_rbf_out.weight", (er, Rn), "dense"),
             ("atom_emb.embeddings.weight", (93, ea), "emb"),
             ("edge_emb.dense.weight", (ee, 2 * ea + Rn), "dense")]

    def res(prefix, units):
        return [(f"{prefix}.dense_mlp.0.weight", (units, units), "dense"),
                (f"{prefix}.dense_mlp.1.weight", (units, units), "dense")]

    def atom_update(prefix, scale_name):
        s = [(f"{prefix}.dense_rbf.weight", (ee, er), "dense"),
             (f"{prefix}.scale_sum.scale_factor", (), f"scale:{scale_name}"),
             (f"{prefix}.layers.0.weight", (ea, ee), "dense")]
        for i in range(c["num_atom"]):
            s += res(f"{prefix}.layers.{i + 1}", ea)
        return s

    for i in range(c["num_blocks"] + 1):
        p = f"out_blocks.{i}"
        spec += atom_update(p, f"OutBlock_{i}_sum")
        spec += [(f"{p}.out_energy.weight", (c["num_targets"], ea), "dense")]
        if c["direct_forces"]:
            spec += [(f"{p}.scale_rbf.scale_factor", (), f"scale:OutBlock_{i}_had"),
                     (f"{p}.seq_forces.0.weight", (ee, ee), "dense")]
            for j in range(c["num_atom"]):
                spec += res(f"{p}.seq_forces.{j + 1}", ee)
            spec += [(f"{p}.out_forces.weight", (c["num_targets"], ee), "dense")]
    for i in range(c["num_blocks"]):
        p = f"int_blocks.{i}"
        n = i + 1
        spec += [(f"{p}.dense_ca.weight", (ee, ee), "dense")]
        if not T:
            q = f"{p}.quad_interaction"
            spec += [(f"{q}.dense_db.weight", (ee, ee), "dense"),
                     (f"{q}.mlp_rbf.weight", (ee, er), "dense"),
                     (f"{q}.scale_rbf.scale_factor", (), f"scale:QuadInteraction_{n}_had_rbf"),
                     (f"{q}.mlp_cbf.weight", (eq, ec), "dense"),
                     (f"{q}.scale_cbf.scale_factor", (), f"scale:QuadInteraction_{n}_had_cbf"),
                     (f"{q}.mlp_sbf.weight", (eq, es, bq), "eff"),
                     (f"{q}.scale_sbf_sum.scale_factor", (), f"scale:QuadInteraction_{n}_sum_sbf"),
                     (f"{q}.down_projection.weight", (eq, ee), "dense"),
                     (f"{q}.up_projection_ca.weight", (ee, bq), "dense"),
                     (f"{q}.up_projection_ac.weight", (ee, bq), "dense")]
        t = f"{p}.trip_interaction"
        spec += [(f"{t}.dense_ba.weight", (ee, ee), "dense"),
                 (f"{t}.mlp_rbf.weight", (ee, er), "dense"),
                 (f"{t}.scale_rbf.scale_factor", (), f"scale:TripInteraction_{n}_had_rbf"),
                 (f"{t}.mlp_cbf.weight", (et, ec, bt), "eff"),
                 (f"{t}.scale_cbf_sum.scale_factor", (), f"scale:TripInteraction_{n}_sum_cbf"),
                 (f"{t}.down_projection.weight", (et, ee), "dense"),
                 (f"{t}.up_projection_ca.weight", (ee, bt), "dense"),
                 (f"{t}.up_projection_ac.weight", (ee, bt), "dense")]
        for j in range(c["num_before_skip"]):
            spec += res(f"{p}.layers_before_skip.{j}", ee)
        for j in range(c["num_after_skip"]):
            spec += res(f"{p}.layers_after_skip.{j}", ee)
        spec += atom_update(f"{p}.atom_update", f"AtomUpdate_{n}_sum")
        spec += [(f"{p}.concat_layer.dense.weight", (ee, 2 * ea + ee), "dense")]
        for j in range(c["num_concat"]):
            spec += res(f"{p}.residual_m.{j}", ee)
    return spec


def make_params(cfg, seed, scale_factors=None, dtype=torch.float64):
    """Builder-owned deterministic weights (NOT reference-initialised): N(0, 1/fan_in) for
    dense/bilinear weights, U(-sqrt3, sqrt3) embeddings, frequencies n*pi*(1+small jitter),
    scale factors from the json dict (1.0 when the name is absent)."""
    rs = np.random.RandomState(seed)
    scale_factors = scale_factors or {}
    out = {}
    for name, shape, kind in param_spec(cfg):
        if kind == "dense":
            w = rs.standard_normal(shape) / math.sqrt(shape[1])
        elif kind == "eff":
            w = rs.standard_normal(shape) / math.sqrt(shape[0] * shape[1])
        elif kind == "emb":
            w = rs.uniform(-math.sqrt(3), math.sqrt(3), size=shape)
        elif kind == "freq":
            w = np.pi * np.arange(1, shape[0] + 1, dtype=np.float32).astype(np.float64)
            w = w * (1 + 0.01 * rs.standard_normal(shape))
        elif kind.startswith("scale:"):
            w = np.array(scale_factors.get(kind[6:], 1.0))
        else:
            raise ValueError(kind)
        out[name] = torch.tensor(np.asarray(w, dtype=np.float32).astype(np.float64), dtype=dtype)
    return out


def load_scale_factors(path):
    with open(path) as f:
        d = json.load(f)
    return {k: float(v) for k, v in d.items() if k != "comment"}


def expand_to_reference_state_dict(params):
    """Add the aliased duplicate keys the reference's state_dict carries
    (Dense.weight == Dense.linear.weight, base_layers.py:24-27; OutputBlock.seq_energy ==
    .layers, atom_update_block.py:130)."""
    out = {}
    for k, v in params.items():
        out[k] = v
        is_dense = k.endswith(".weight") and v.dim() == 2 and "embeddings" not in k
        if is_dense:
            out[k[:-len("weight")] + "linear.weight"] = v
    for k in list(out):
        if k.startswith("out_blocks.") and ".layers." in k:
            out[k.replace(".layers.", ".seq_energy.")] = out[k]
    return out


# ----------------------------------------------------------------------------------------
# layers
# ----------------------------------------------------------------------------------------
def ssilu(x):
    return torch.nn.functional.silu(x) * (1 / 0.6)


def dense(x, w, act):
    y = x @ w.t()
    return ssilu(y) if act else y


def residual(P, prefix, x):
    y = dense(x, P[f"{prefix}.dense_mlp.0.weight"], True)
    y = dense(y, P[f"{prefix}.dense_mlp.1.weight"], True)
    return (x + y) * INV_SQRT_2


def seg_sum(src, index, n):
    out = torch.zeros((n,) + tuple(src.shape[1:]), dtype=src.dtype)
    return out.index_add(0, index, src)


def atom_update(P, prefix, c, nA, m, rbf, id_a, head):
    """AtomUpdateBlock (atom_update_block.py:55-72); with head=True the OutputBlock (:157-193), which returns
    (E per atom, F per edge) — F is 0 unless direct_forces."""
    xe = m * dense(rbf, P[f"{prefix}.dense_rbf.weight"], False)
    x = seg_sum(xe, id_a, nA) * P[f"{prefix}.scale_sum.scale_factor"]
    x = dense(x, P[f"{prefix}.layers.0.weight"], True)
    for i in range(c["num_atom"]):
        x = residual(P, f"{prefix}.layers.{i + 1}", x)
    if not head:
        return x
    x = dense(x, P[f"{prefix}.out_energy.weight"], False)
    if not c["direct_forces"]:
        return x, 0.0
    f = xe * P[f"{prefix}.scale_rbf.scale_factor"]                      # atom_update_block.py:181-183
    f = dense(f, P[f"{prefix}.seq_forces.0.weight"], True)
    for i in range(c["num_atom"]):
        f = residual(P, f"{prefix}.seq_forces.{i + 1}", f)
    return x, dense(f, P[f"{prefix}.out_forces.weight"], False)            # (nEdges, num_targets)


def edge_embedding(w, h, m_rbf, id_c, id_a):
    # embedding_block.py:60-75 is called as (h, m, id_c, id_a): cat[h[id_c], h[id_a], m]
    return dense(torch.cat([h[id_c], h[id_a], m_rbf], dim=-1), w, True)


def bilinear(rbfW1, sph, x_t, id_reduce, W, nE):
    """efficient.py:159-189:  out[e,o] = sum_{t in seg(e)} sum_s sum_i sum_c sph[t,s] rbfW1[e,i,s] x[t,c] W[c,i,o].
    As the reference does it — the triplets / quadruplets of an edge zero-padded to (E, Kmax, .) by an index_put
    (efficient.py:173-182, basis_layers.py:153-159) and contracted by a batched matmul — in chunks of edges so that the
    padded tensors stay bounded (the segments are contiguous: `id_reduce` is sorted, data_container.py:324-328,369-375).
    (Until round 5 the (T, S, C) outer products were formed and index_add-ed: the same sums at twice the reference's time.)"""
    S = sph.shape[1]
    C = x_t.shape[1]
    T = sph.shape[0]
    if T == 0:
        sum_k = torch.zeros((nE, S, C), dtype=x_t.dtype)
    else:
        cnt = torch.bincount(id_reduce, minlength=nE)
        off = torch.zeros(nE + 1, dtype=torch.long)
        off[1:] = torch.cumsum(cnt, 0)
        kidx = torch.arange(T) - off[id_reduce]                          # position inside the segment (Kidx3 / Kidx4)
        parts = []
        kmax_all = int(cnt.max())
        step = max(1, (1 << 26) // (kmax_all * max(S, C)))               # edges per chunk: one padded operand <= 2^26 elements
        for e0 in range(0, nE, step):
            e1 = min(nE, e0 + step)
            lo, hi = int(off[e0]), int(off[e1])
            kmax = int(cnt[e0:e1].max())
            if kmax == 0:
                parts.append(torch.zeros((e1 - e0, S, C), dtype=x_t.dtype))
                continue
            rows, cols = id_reduce[lo:hi] - e0, kidx[lo:hi]
            sph2 = torch.zeros((e1 - e0, kmax, S), dtype=sph.dtype).index_put((rows, cols), sph[lo:hi])
            m2 = torch.zeros((e1 - e0, kmax, C), dtype=x_t.dtype).index_put((rows, cols), x_t[lo:hi])
            parts.append(torch.matmul(sph2.transpose(1, 2), m2))         # (e, S, C)
        sum_k = parts[0] if len(parts) == 1 else torch.cat(parts, 0)
    P = torch.matmul(rbfW1, sum_k)  # (E,I,S)@(E,S,C) -> (E,I,C)
    return torch.einsum("eic,cio->eo", P, W)


def neighbor_angles(u, v):
    x = torch.sum(u * v, dim=1)
    y = torch.linalg.cross(u, v).norm(dim=-1)
    y = torch.max(y, torch.tensor(1e-9, dtype=y.dtype))
    return torch.atan2(y, x)


def rejection(x, n):
    return x - (torch.sum(x * n, -1) / torch.sum(n * n, -1))[:, None] * n


def forward(cfg, P, inputs, need_forces=True, create_graph=False):
    """-> (E (nMol, num_targets), F (nAtoms, 3)).  inputs: reference batch dict (torch tensors);
    float tensors define the compute dtype via P."""
    c = {**DEFAULTS, **cfg}
    dt = P["rbf_basis.frequencies"].dtype
    T = c["triplets_only"]
    S, Rn = c["num_spherical"], c["num_radial"]
    Z = inputs["Z"]
    R = inputs["R"].to(dt).detach().clone()
    if need_forces and not c["direct_forces"]:
        R.requires_grad_(True)
    id_a, id_c, id_swap = inputs["id_a"], inputs["id_c"], inputs["id_swap"]
    id3_exp, id3_red = inputs["id3_expand_ba"], inputs["id3_reduce_ca"]
    batch_seg = inputs["batch_seg"]
    nA, nE = Z.shape[0], id_a.shape[0]
    nMol = int(inputs["N"].shape[0]) if "N" in inputs else int(batch_seg.max()) + 1
    p = c["envelope_exponent"]

    V = R[id_a] - R[id_c]
    D = torch.sqrt(torch.sum(V ** 2, dim=1))
    rbf = B.bessel_rbf(D, P["rbf_basis.frequencies"], c["cutoff"], p)

    ang3 = neighbor_angles(R[id_c[id3_red]] - R[id_a[id3_red]], R[id_c[id3_exp]] - R[id_a[id3_red]])
    rad3 = B.sph_bessel_radial(D, S, Rn, c["cutoff"], p)  # (E,S,R)
    sph3 = B.real_sph_harm_l0(S, ang3)  # (T,S)
    rbfW1_3 = torch.einsum("esr,sri->eis", rad3, P["mlp_cbf3.weight"])  # (E,I,S)

    if not T:
        i_b, i_a = inputs["id4_int_b"], inputs["id4_int_a"]
        red_ca, exp_abd, red_cab = inputs["id4_reduce_ca"], inputs["id4_expand_abd"], inputs["id4_reduce_cab"]
        red_i_ca, exp_i_db = inputs["id4_reduce_intm_ca"], inputs["id4_expand_intm_db"]
        red_i_ab, exp_i_ab = inputs["id4_reduce_intm_ab"], inputs["id4_expand_intm_ab"]
        D_ab = torch.sqrt(torch.sum((R[i_a] - R[i_b]) ** 2, dim=1))
        # a - b <- d   (gemnet.py:385-396)
        Ra, Rb, Rd = R[i_a[exp_i_ab]], R[i_b[exp_i_ab]], R[id_c[exp_i_db]]
        R_ba, R_bd = Ra - Rb, Rd - Rb
        phi_abd = neighbor_angles(R_ba, R_bd)
        R_bd_proj = rejection(R_bd, R_ba)[exp_abd]
        # c -> a <- b  (gemnet.py:399-411)
        Rc, Ra2, Rb2 = R[id_c[red_i_ca]], R[id_a[red_i_ca]], R[i_b[red_i_ab]]
        R_ac, R_ab = Rc - Ra2, Rb2 - Ra2
        phi_cab = neighbor_angles(R_ab, R_ac)[red_cab]
        R_ac_proj = rejection(R_ac, R_ab)[red_cab]
        theta_cabd = neighbor_angles(R_ac_proj, R_bd_proj)
        # cbf4: non-efficient SphericalBasisLayer with cutoff=int_cutoff (basis_layers.py:132-144)
        rad4 = B.sph_bessel_radial(D_ab, S, Rn, c["int_cutoff"], p)[exp_i_ab]  # (I,S,R)
        cbf4 = (rad4 * B.real_sph_harm_l0(S, phi_abd)[:, :, None]).reshape(-1, S * Rn)
        cbf4 = dense(cbf4, P["mlp_cbf4.weight"], False)
        # sbf4: TensorBasisLayer (basis_layers.py:239-295)
        deg = torch.arange(S) * 2 + 1
        rad_s = torch.repeat_interleave(rad3, deg, dim=1)  # (E,S^2,R)
        sph4 = B.real_sph_harm_full(S, phi_cab, theta_cabd)  # (Q,S^2)
        rbfW1_4 = torch.einsum("esr,sri->eis", rad_s, P["mlp_sbf4.weight"])
        rbf4 = dense(rbf, P["mlp_rbf4.weight"], False)

    h = P["atom_emb.embeddings.weight"][Z - 1]
    m = edge_embedding(P["edge_emb.dense.weight"], h, rbf, id_c, id_a)
    rbf3 = dense(rbf, P["mlp_rbf3.weight"], False)
    rbf_h = dense(rbf, P["mlp_rbf_h.weight"], False)
    rbf_out = dense(rbf, P["mlp_rbf_out.weight"], False)

    E_a, F_ca = atom_update(P, "out_blocks.0", c, nA, m, rbf_out, id_a, head=True)
    for i in range(c["num_blocks"]):
        pb = f"int_blocks.{i}"
        x_skip = dense(m, P[f"{pb}.dense_ca.weight"], True)
        # ---- triplet interaction (interaction_block.py:653-696)
        pt = f"{pb}.trip_interaction"
        x = dense(m, P[f"{pt}.dense_ba.weight"], True)
        x = x * dense(rbf3, P[f"{pt}.mlp_rbf.weight"], False) * P[f"{pt}.scale_rbf.scale_factor"]
        x = dense(x, P[f"{pt}.down_projection.weight"], True)
        x = bilinear(rbfW1_3, sph3, x[id3_exp], id3_red, P[f"{pt}.mlp_cbf.weight"], nE)
        x = x * P[f"{pt}.scale_cbf_sum.scale_factor"]
        x3 = (dense(x, P[f"{pt}.up_projection_ca.weight"], True)
              + dense(x, P[f"{pt}.up_projection_ac.weight"], True)[id_swap]) * INV_SQRT_2
        if T:
            x = (x_skip + x3) * INV_SQRT_2
        else:
            # ---- quadruplet interaction (interaction_block.py:517-566)
            pq = f"{pb}.quad_interaction"
            y = dense(m, P[f"{pq}.dense_db.weight"], True)
            y = y * dense(rbf4, P[f"{pq}.mlp_rbf.weight"], False) * P[f"{pq}.scale_rbf.scale_factor"]
            y = dense(y, P[f"{pq}.down_projection.weight"], True)
            y = y[exp_i_db]
            y = y * dense(cbf4, P[f"{pq}.mlp_cbf.weight"], False) * P[f"{pq}.scale_cbf.scale_factor"]
            y = bilinear(rbfW1_4, sph4, y[exp_abd], red_ca, P[f"{pq}.mlp_sbf.weight"], nE)
            y = y * P[f"{pq}.scale_sbf_sum.scale_factor"]
            x4 = (dense(y, P[f"{pq}.up_projection_ca.weight"], True)
                  + dense(y, P[f"{pq}.up_projection_ac.weight"], True)[id_swap]) * INV_SQRT_2
            x = (x_skip + x3 + x4) * INV_SQRT_3
        for j in range(c["num_before_skip"]):
            x = residual(P, f"{pb}.layers_before_skip.{j}", x)
        m = (m + x) * INV_SQRT_2
        for j in range(c["num_after_skip"]):
            m = residual(P, f"{pb}.layers_after_skip.{j}", m)
        h2 = atom_update(P, f"{pb}.atom_update", c, nA, m, rbf_h, id_a, head=False)
        h = (h + h2) * INV_SQRT_2
        m2 = edge_embedding(P[f"{pb}.concat_layer.dense.weight"], h, m, id_c, id_a)
        for j in range(c["num_concat"]):
            m2 = residual(P, f"{pb}.residual_m.{j}", m2)
        m = (m + m2) * INV_SQRT_2
        E_i, F_i = atom_update(P, f"out_blocks.{i + 1}", c, nA, m, rbf_out, id_a, head=True)
        E_a, F_ca = E_a + E_i, F_ca + F_i

    E_mol = seg_sum(E_a, batch_seg, nMol)
    if not c["extensive"]:
        cnt = seg_sum(torch.ones(nA, 1, dtype=dt), batch_seg, nMol).clamp(min=1)
        E_mol = E_mol / cnt
    if c["direct_forces"]:                                                   # gemnet.py:586-597
        if c["forces_coupled"]:
            id_undir = inputs["id_undir"]
            half = seg_sum(F_ca, id_undir, nE // 2) * 0.5                  # mean over the two directions
            F_ca = half[id_undir]
        F_ji = F_ca[:, :, None] * (V / D[:, None])[:, None, :]
        return E_mol, seg_sum(F_ji, id_a, nA)                               # (nAtoms, num_targets, 3)
    if not need_forces:
        return E_mol, None
    if c["num_targets"] > 1:                                                # gemnet.py:599-609
        F = torch.stack([-torch.autograd.grad(E_mol[:, t].sum(), R, create_graph=create_graph, retain_graph=True)[0]
                         for t in range(c["num_targets"])], dim=1)
        return E_mol, F
    F = -torch.autograd.grad(E_mol.sum(), R, create_graph=create_graph)[0]
    return E_mol, F


def training_loss(E, F, E_t, F_t, rho_force=0.999):
    """config 'rmse' loss of the reference trainer (training/trainer.py:284-292,338-343)."""
    e_mae = torch.nn.functional.l1_loss(E, E_t)
    f_rmse = torch.mean(torch.norm(F - F_t, p=2, dim=1))
    return (1 - rho_force) * e_mae + rho_force * f_rmse
