"""GemNet-Q: the quadruplets of one target atom form a (nearly) dense block — statistics and a float64 check (CPU).

A quadruplet q = (c -> a <- b <- d) reduces into the edge r(q) = (c -> a) and expands from the intermediate triplet
j(q) = (a <- b <- d); both end in atom a.  Per atom:  R_a = edges into a,  J_a = intermediate triplets of a,  and the
quadruplets of a are a subset of R_a x J_a.  If that subset is nearly all of it, the three heavy kernels are per-atom
GEMMs over operands that fit LDS:
    K1         Sm[r, s, :]  = sum_j Y[(r, j), s] x[j, :]         (49 |R_a|) x |J_a|  @  |J_a| x 32
    x-adjoint  gx[j, :]     = sum_{r, s} Y[(r, j), s] dSm[r, s, :]  |J_a| x (49 |R_a|) @ (49 |R_a|) x 32
    dY         dY[(r, j), s] = <dSm[r, s, :], x[j, :]>
instead of per-quadruplet VALU products with a 1.15 GB dxt round trip.

    python tools/exp/q_atom_blocks.py [n_mol n_atoms]      statistics of the bench batch + check on a small batch"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import bench as B
from gemnet_pytorch_amd.graph import GraphPlan
import cpu_kernels as CK


def blocks(plan):
    E, J = plan.n_edges, plan.n_intm
    atom_of_edge = plan.id_a.idx64                       # target atom a of the edge c -> a
    atom_of_int = plan.int_a.idx64                       # target atom a of the interaction edge b -> a
    atom_of_j = atom_of_int[plan.intm_ab.idx64]          # intermediate triplet j = (interaction edge ab, edge db)
    r, j = plan.quad.reduce.idx64, plan.quad.expand.idx64
    assert bool((atom_of_edge[r] == atom_of_j[j]).all()), "reduce edge and intermediate triplet end in the same atom"
    A = plan.n_atoms
    nR = torch.bincount(atom_of_edge, minlength=A)
    nJ = torch.bincount(atom_of_j, minlength=A)
    nQ = torch.bincount(atom_of_edge[r], minlength=A)
    return atom_of_edge, atom_of_j, nR, nJ, nQ


def stats(n_mol, n_atoms):
    cfg = dict(B.GEMNET_T, triplets_only=False)
    inputs, _ = B.make_batch(cfg, n_mol, n_atoms, first=0, device="cpu")
    plan = GraphPlan.from_inputs(inputs, False)
    _, _, nR, nJ, nQ = blocks(plan)
    dense = (nR * nJ).sum().item()
    print(f"{n_mol} x {n_atoms} atoms: {plan.n_edges} edges, {plan.n_intm} intermediate triplets, {plan.quad.size} quadruplets")
    print(f"  per atom: |R_a| mean {nR.float().mean():.1f} max {int(nR.max())};  |J_a| mean {nJ.float().mean():.0f} max {int(nJ.max())};"
          f"  quadruplets mean {nQ.float().mean():.0f}")
    print(f"  fill of the per-atom blocks: {plan.quad.size / dense:.3f}  (sum_a |R_a| |J_a| = {dense})")
    lds = 49 * 32 * 4 * nR.max().item()
    print(f"  dSm of one atom's reduce edges: up to {lds / 1024:.0f} KB in f32 ({lds / 2048:.0f} KB as one bf16 plane)")
    fl = 2.0 * dense * 49 * 32
    print(f"  dense-block flops per pass {fl / 1e9:.1f} GFLOP (per-quadruplet form: {2.0 * plan.quad.size * 49 * 32 / 1e9:.1f})")


def check():
    cfg = dict(B.GEMNET_T, triplets_only=False)
    inputs, _ = B.make_batch(cfg, 2, 10, first=0, device="cpu")
    plan = GraphPlan.from_inputs(inputs, False)
    aE, aJ, nR, nJ, nQ = blocks(plan)
    g = torch.Generator().manual_seed(0)
    Q, S, C = plan.quad.size, 49, 32
    Y = torch.randn(Q, S, generator=g, dtype=torch.float64)
    dSm = torch.randn(plan.n_edges, S, C, generator=g, dtype=torch.float64)
    ref = CK.bil_reduce_t(Y, dSm, plan.quad)
    out = torch.zeros(plan.n_intm, C, dtype=torch.float64)
    r, j = plan.quad.reduce.idx64, plan.quad.expand.idx64
    for a in range(plan.n_atoms):
        Ra = torch.nonzero(aE == a)[:, 0]
        Ja = torch.nonzero(aJ == a)[:, 0]
        if not len(Ra) or not len(Ja):
            continue
        posR = {int(e): i for i, e in enumerate(Ra)}
        posJ = {int(t): i for i, t in enumerate(Ja)}
        Ymat = torch.zeros(len(Ja), len(Ra) * S, dtype=torch.float64)      # zero where (r, j) is not a quadruplet
        for q in torch.nonzero(aE[r] == a)[:, 0].tolist():
            Ymat[posJ[int(j[q])], posR[int(r[q])] * S:(posR[int(r[q])] + 1) * S] = Y[q]
        out[Ja] = Ymat @ dSm[Ra].reshape(len(Ra) * S, C)
    err = float((out - ref).abs().max())
    print(f"x-adjoint as per-atom GEMMs vs the per-quadruplet form (2 x 10 atoms, {Q} quadruplets): max abs diff {err:.2e}")
    assert err < 1e-10


if __name__ == "__main__":
    n_mol, n_atoms = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (8, 32)
    check()
    stats(n_mol, n_atoms)
