"""Generate the golden fixtures under tests/golden/ by IMPORTING the reference.

Runs only in the build container (needs /root/reference); the GPU box never sees the
reference.  Only data (inputs + the reference's outputs) is written; weights are NOT
stored — they come from the builder-owned deterministic generator
`oracle.gemnet_oracle.make_params(cfg, seed)` and are pushed into the reference model
with load_state_dict.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

Oracle-side shims (never shipped as product):
  torch_scatter.scatter -> index_add based;  numba.njit -> identity;  np.math -> math;
  LambdaLR accepting (and dropping) the `verbose=` keyword torch >= 2.7 removed;
  a recording stand-in for `tf.train.load_checkpoint` (golden_tfnames only: TensorFlow is not installed).
"""
import math
import os
import sys
import types

sys.dont_write_bytecode = True
import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)


def install_shims():
    ts = types.ModuleType("torch_scatter")

    def scatter(src, index, dim=0, dim_size=None, reduce="add"):
        assert dim == 0
        dim_size = int(index.max()) + 1 if dim_size is None else int(dim_size)
        out = torch.zeros((dim_size,) + tuple(src.shape[1:]), dtype=src.dtype)
        out = out.index_add(0, index, src)
        if reduce == "mean":
            cnt = torch.zeros(dim_size, dtype=src.dtype).index_add(
                0, index, torch.ones(index.shape[0], dtype=src.dtype)).clamp(min=1)
            out = out / cnt.view((-1,) + (1,) * (src.dim() - 1))
        return out

    ts.scatter = scatter
    sys.modules["torch_scatter"] = ts
    nb = types.ModuleType("numba")

    def njit(*a, **k):
        if len(a) == 1 and callable(a[0]):
            return a[0]
        return lambda f: f

    nb.njit = njit
    sys.modules["numba"] = nb
    np.math = math
    import torch.optim.lr_scheduler as lrs

    class LambdaLR(lrs.LambdaLR):
        def __init__(self, optimizer, lr_lambda, last_epoch=-1, verbose=False):
            super().__init__(optimizer, lr_lambda, last_epoch=last_epoch)

    lrs.LambdaLR = LambdaLR
    sys.path.insert(0, REF)


install_shims()
from gemnet.model.gemnet import GemNet  # noqa: E402  (the reference)
import inspect  # noqa: E402
assert inspect.getsourcefile(GemNet).startswith(REF), "make_golden must import the REFERENCE GemNet"
from gemnet.model.layers import basis_utils as ref_bu  # noqa: E402
from gemnet.model.layers.basis_layers import (BesselBasisLayer, SphericalBasisLayer,  # noqa: E402
                                              TensorBasisLayer)
from gemnet.training.data_container import DataContainer  # noqa: E402

from gemnet_pytorch_amd.synthetic import make_dataset, make_molecule  # noqa: E402
from oracle import gemnet_oracle as GO  # noqa: E402

SCALE_FILE = os.path.join(REF, "scaling_factors.json")


def cfg_small(triplets_only, num_blocks=1):
    return dict(num_spherical=7, num_radial=6, num_blocks=num_blocks, emb_size_atom=64,
                emb_size_edge=64, emb_size_trip=32, emb_size_quad=32, emb_size_rbf=16,
                emb_size_cbf=16, emb_size_sbf=32, emb_size_bil_quad=32, emb_size_bil_trip=32,
                num_before_skip=1, num_after_skip=1, num_concat=1, num_atom=2,
                triplets_only=triplets_only)


def cfg_full(triplets_only, num_blocks=2):
    return dict(num_spherical=7, num_radial=6, num_blocks=num_blocks, emb_size_atom=128,
                emb_size_edge=128, emb_size_trip=64, emb_size_quad=32, emb_size_rbf=16,
                emb_size_cbf=16, emb_size_sbf=32, emb_size_bil_quad=32, emb_size_bil_trip=64,
                num_before_skip=1, num_after_skip=1, num_concat=1, num_atom=2,
                triplets_only=triplets_only)


# ------------------------------------------------------------------------------- G1 basis
def golden_basis():
    out = {}
    out["jn_zeros"] = ref_bu.Jn_zeros(7, 6)
    z = out["jn_zeros"]
    out["normalizer"] = np.array(
        [[1 / np.sqrt(0.5 * ref_bu.Jn(z[l, i], l + 1) ** 2) for i in range(6)] for l in range(7)])
    out["prefactor"] = np.array(
        [[ref_bu.sph_harm_prefactor(l, m) if m <= l else 0.0 for m in range(7)] for l in range(7)])
    rs = np.random.RandomState(7)
    d = torch.tensor(np.concatenate([np.linspace(0.9, 5.0, 40), rs.uniform(0.9, 5.2, 24)]),
                     dtype=torch.float64)
    d10 = torch.tensor(np.concatenate([np.linspace(0.9, 10.0, 40), rs.uniform(0.9, 10.3, 24)]),
                       dtype=torch.float64)
    theta = torch.tensor(np.concatenate([np.linspace(1e-3, np.pi - 1e-3, 40),
                                         rs.uniform(0, np.pi, 24)]), dtype=torch.float64)
    phi = torch.tensor(np.concatenate([np.linspace(1e-3, np.pi - 1e-3, 40)[::-1].copy(),
                                       rs.uniform(0, np.pi, 24)]), dtype=torch.float64)
    out.update(d=d.numpy(), d10=d10.numpy(), theta=theta.numpy(), phi=phi.numpy())
    n = d.shape[0]
    idx = torch.arange(n)
    zeros = torch.zeros(n, dtype=torch.long)

    bes = BesselBasisLayer(6, cutoff=5.0).double()
    freq = bes.frequencies.detach().clone()
    out["freq"] = freq.numpy()
    dd = d.clone().requires_grad_(True)
    y = bes(dd)
    out["bessel_rbf"] = y.detach().numpy()
    g = torch.stack([torch.autograd.grad(y[:, k].sum(), dd, retain_graph=True)[0] for k in range(6)], 1)
    out["bessel_rbf_dd"] = g.numpy()

    for name, cutoff, dist in (("c5", 5.0, d), ("c10", 10.0, d10)):
        lay = SphericalBasisLayer(7, 6, cutoff=cutoff, efficient=False).double()
        dd = dist.clone().requires_grad_(True)
        th = theta.clone().requires_grad_(True)
        o = lay(dd, th, idx, None)  # (n, 42) = rbf_env[l,n] * Y_l0
        out[f"cbf_{name}"] = o.detach().numpy()
        # separate factors: efficient branch returns (S,E,R) radial and padded sph
        lay_e = SphericalBasisLayer(7, 6, cutoff=cutoff, efficient=True).double()
        rad, sph2 = lay_e(dd, th, idx, zeros)
        out[f"radial_{name}"] = rad.detach().permute(1, 0, 2).numpy()  # (n,S,R)
        out[f"y_l0"] = sph2.detach()[:, 0, :].numpy()
        gr = torch.stack([torch.autograd.grad(rad[l, :, k].sum(), dd, retain_graph=True)[0]
                          for l in range(7) for k in range(6)], 1)
        out[f"radial_{name}_dd"] = gr.reshape(n, 7, 6).numpy()
        gy = torch.stack([torch.autograd.grad(sph2[:, 0, l].sum(), th, retain_graph=True,
                                              allow_unused=True)[0] if l > 0 else torch.zeros(n, dtype=torch.float64)
                          for l in range(7)], 1)
        out["y_l0_dtheta"] = gy.numpy()

    ten = TensorBasisLayer(7, 6, cutoff=5.0, efficient=True).double()
    th = theta.clone().requires_grad_(True)
    ph = phi.clone().requires_grad_(True)
    rad, sph2 = ten(d, th, ph, idx, zeros)
    out["radial_tensor_c5"] = rad.permute(1, 0, 2).numpy()  # (n,49,6)
    Y = sph2[:, 0, :]
    out["y_lm"] = Y.detach().numpy()
    out["y_lm_dtheta"] = torch.stack(
        [torch.autograd.grad(Y[:, k].sum(), th, retain_graph=True, allow_unused=True)[0]
         if k > 0 else torch.zeros(n, dtype=torch.float64) for k in range(49)], 1).numpy()
    gphi = []
    for k in range(49):
        g = torch.autograd.grad(Y[:, k].sum(), ph, retain_graph=True, allow_unused=True)[0]
        gphi.append(torch.zeros(n, dtype=torch.float64) if g is None else g)
    out["y_lm_dphi"] = torch.stack(gphi, 1).numpy()
    np.savez_compressed(os.path.join(HERE, "basis.npz"), **out)
    print("basis.npz", {k: v.shape for k, v in out.items()})


# ------------------------------------------------------------------------------ G2 indices
class _MemContainer(DataContainer):
    """The reference DataContainer fed from memory instead of an npz path."""

    def __init__(self, data, cutoff, int_cutoff, triplets_only):
        self._data = data
        super().__init__("<memory>", cutoff, int_cutoff, triplets_only=triplets_only)

    def _load_npz(self, path, keys):
        for k in keys:
            if k in self._data:
                setattr(self, k, self._data[k])


def special_molecules():
    mols = []
    # two atoms within cutoff; two atoms beyond the cutoff (no edge)
    mols.append(("pair", np.array([[0, 0, 0], [1.2, 0, 0]], np.float32)))
    mols.append(("noedge", np.array([[0, 0, 0], [7.5, 0, 0]], np.float32)))
    # linear (collinear triplets -> the 1e-9 clamp of gemnet.py:309)
    mols.append(("linear", np.array([[0, 0, 0], [1.1, 0, 0], [2.2, 0, 0], [3.3, 0, 0]], np.float32)))
    # one pair exactly at the cutoff (<=), one just beyond in float32
    mols.append(("atcut", np.array([[0, 0, 0], [5.0, 0, 0], [0, 3.0, 0], [5.0000005, 3.0, 0.0],
                                    [2.5, 1.5, 1.0]], np.float32)))
    for n, seed in ((3, 11), (5, 12), (8, 13), (12, 14), (14, 15)):
        mols.append((f"rand{n}", make_molecule(n, seed, box=max(3.0, 0.42 * n))["R"]))
    return mols


def run_container(Rlist, triplets_only, cutoff=5.0, int_cutoff=10.0):
    N = np.array([len(r) for r in Rlist], dtype=np.int32)
    R = np.concatenate(Rlist).astype(np.float32)
    Z = np.ones(len(R), np.int32)
    data = dict(N=N, Z=Z, R=R, E=np.zeros(len(N), np.float32), F=np.zeros_like(R))
    dc = _MemContainer(data, cutoff, int_cutoff, triplets_only)
    batch = dc[list(range(len(N)))]
    return R, N, {k: batch[k].numpy() for k in dc.index_keys}


def golden_indices():
    out = {}
    mols = special_molecules()
    names = []
    for name, R in mols:
        for to in (True, False):
            Rc, N, idx = run_container([R], to)
            tag = f"{name}.{'T' if to else 'Q'}"
            out[f"{tag}.R"] = Rc
            out[f"{tag}.N"] = N
            for k, v in idx.items():
                out[f"{tag}.{k}"] = v.astype(np.int32)
        names.append(name)
    # a batch of three molecules (block-diagonal offsets, _bmat_fast :115-151)
    batch = [mols[6][1], mols[4][1], mols[7][1]]
    for to in (True, False):
        Rc, N, idx = run_container(batch, to)
        tag = f"batch3.{'T' if to else 'Q'}"
        out[f"{tag}.R"], out[f"{tag}.N"] = Rc, N
        for k, v in idx.items():
            out[f"{tag}.{k}"] = v.astype(np.int32)
    names.append("batch3")
    out["names"] = np.array(names)
    np.savez_compressed(os.path.join(HERE, "indices.npz"), **out)
    print("indices.npz", len(out), "arrays")


# ------------------------------------------------------------------------------- G4 model

def grad_probes(name, numel, k=4):
    """k fixed +-1 probe vectors for the parameter `name` (seeded by the name): the dot products of a gradient with
    them pin its ELEMENTS (sign and position), which the norm alone does not, at 32 bytes per parameter."""
    import zlib
    rs = np.random.RandomState(zlib.crc32(name.encode()) & 0x7FFFFFFF)
    return rs.randint(0, 2, size=(k, numel)).astype(np.float64) * 2.0 - 1.0


def record_grads(tag, model, out):
    names, norms, proj = [], [], []
    for n, p in model.named_parameters():
        if p.grad is not None:
            names.append(n)
            norms.append(float(p.grad.norm()))
            proj.append(grad_probes(n, p.numel()) @ p.grad.detach().double().reshape(-1).numpy())
    out[f"{tag}.grad_names"], out[f"{tag}.grad_norms"] = np.array(names), np.array(norms)
    out[f"{tag}.grad_proj"] = np.array(proj)

def run_model(cfg, seed, Rlist, Zlist, tag, out, with_grads):
    to = cfg["triplets_only"]
    N = np.array([len(r) for r in Rlist], dtype=np.int32)
    R = np.concatenate(Rlist).astype(np.float32)
    Z = np.concatenate(Zlist).astype(np.int32)
    rs = np.random.RandomState(seed + 99)
    Et = rs.standard_normal(len(N)).astype(np.float32)
    Ft = rs.standard_normal(R.shape).astype(np.float32)
    dc = _MemContainer(dict(N=N, Z=Z, R=R, E=Et, F=Ft), 5.0, 10.0, to)
    batch = dc[list(range(len(N)))]
    sf = GO.load_scale_factors(SCALE_FILE)
    params = GO.make_params(cfg, seed, sf, dtype=torch.float64)
    model = GemNet(**cfg, scale_file=SCALE_FILE).double()
    missing = model.load_state_dict(GO.expand_to_reference_state_dict(params), strict=True)
    inputs = {k: v for k, v in batch.items() if k not in ("E", "F")}
    inputs["R"] = inputs["R"].double()
    model.train()
    E, F = model(inputs)
    out[f"{tag}.seed"] = np.array(seed)
    out[f"{tag}.cfg"] = np.array(repr(cfg))
    out[f"{tag}.Z"], out[f"{tag}.R"], out[f"{tag}.N"] = Z, R, N
    out[f"{tag}.Et"], out[f"{tag}.Ft"] = Et, Ft
    for k in dc.index_keys:
        out[f"{tag}.{k}"] = batch[k].numpy().astype(np.int32)
    out[f"{tag}.E"] = E.detach().numpy()
    out[f"{tag}.F"] = F.detach().numpy()
    if with_grads:
        loss = GO.training_loss(E, F, batch["E"].double(), batch["F"].double())
        out[f"{tag}.loss"] = loss.detach().numpy()
        loss.backward()
        record_grads(tag, model, out)
        # full gradients of three representative parameters
        for n in ("rbf_basis.frequencies", "mlp_cbf3.weight", "int_blocks.0.trip_interaction.mlp_cbf.weight",
                  "int_blocks.0.dense_ca.weight"):
            out[f"{tag}.grad.{n}"] = dict(model.named_parameters())[n].grad.numpy()
    print(tag, "E", E.detach().numpy().ravel()[:3], "|F|max", float(F.abs().max()),
          {k: int(batch[k].shape[0]) for k in ("id_a", "id3_reduce_ca")})


def golden_models():
    out = {}
    m12 = make_molecule(12, 1000 * 1 + 0)
    # config 1: GemNet-T, 1 block, emb 64, single 12-atom molecule
    run_model(cfg_small(True), 1, [m12["R"]], [m12["Z"]], "t1", out, with_grads=True)
    # GemNet-Q, 1 block, emb 64, same molecule
    run_model(cfg_small(False), 2, [m12["R"]], [m12["Z"]], "q1", out, with_grads=True)
    # GemNet-T full width, 2 blocks, batch of 2 (12 + 9 atoms)
    m9 = make_molecule(9, 77, box=4.5)
    run_model(cfg_full(True, 2), 3, [m12["R"], m9["R"]], [m12["Z"], m9["Z"]], "t2", out, with_grads=True)
    # GemNet-Q full width, 2 blocks, batch of 2
    run_model(cfg_full(False, 2), 4, [m12["R"], m9["R"]], [m12["Z"], m9["Z"]], "q2", out, with_grads=False)
    # GemNet-T full (4 blocks) on one 32-atom COLL-shaped molecule
    m32 = make_molecule(32, 2000)
    run_model(cfg_full(True, 4), 5, [m32["R"]], [m32["Z"]], "t4", out, with_grads=False)
    np.savez_compressed(os.path.join(HERE, "model.npz"), **out)
    print("model.npz", len(out), "arrays")


# ------------------------------------------------------------ G4b / G3: round-2 model + layer fixtures
HEAD_KEYS = ("out_energy.weight", "out_forces.weight")


def scale_heads(params, factor):
    """Multiply the output-head weights (E and direct F are linear in them) — used to bring mean|F| to 1 eV/A so the
    1e-5 eV/A force bar can be asserted literally on the deep models."""
    return {k: (v * factor if k.endswith(HEAD_KEYS) else v) for k, v in params.items()}


def _flatten(prefix, obj, out):
    if torch.is_tensor(obj):
        out[prefix] = obj.detach().numpy()
    elif isinstance(obj, (tuple, list)):
        for i, o in enumerate(obj):
            _flatten(f"{prefix}.{i}", o, out)
    elif isinstance(obj, dict):
        for k, o in obj.items():
            _flatten(f"{prefix}.{k}", o, out)


def run_model2(cfg, seed, Rlist, Zlist, tag, out, with_grads=False, unit_forces=True, layers=()):
    """Like run_model, plus: (1) the output heads are rescaled so that mean|F| = 1 (`<tag>.out_scale`, applied by the
    tests to the same generated weights); (2) direct-force / multi-target models; (3) forward hooks on `layers`
    record every tensor entering and leaving those reference modules (`<tag>.L.<module>.in|kw|out...`)."""
    to = cfg["triplets_only"]
    N = np.array([len(r) for r in Rlist], dtype=np.int32)
    R = np.concatenate(Rlist).astype(np.float32)
    Z = np.concatenate(Zlist).astype(np.int32)
    rs = np.random.RandomState(seed + 99)
    Et = rs.standard_normal(len(N)).astype(np.float32)
    Ft = rs.standard_normal(R.shape).astype(np.float32)
    dc = _MemContainer(dict(N=N, Z=Z, R=R, E=Et, F=Ft), 5.0, 10.0, to)
    batch = dc[list(range(len(N)))]
    sf = GO.load_scale_factors(SCALE_FILE)
    params = GO.make_params(cfg, seed, sf, dtype=torch.float64)
    inputs = {k: v for k, v in batch.items() if k not in ("E", "F")}
    inputs["R"] = inputs["R"].double()
    model = GemNet(**cfg, scale_file=SCALE_FILE).double()
    model.load_state_dict(GO.expand_to_reference_state_dict(params), strict=True)
    model.train()
    scale = 1.0
    if unit_forces:
        _, F0 = model(dict(inputs))
        scale = 1.0 / float(F0.detach().abs().mean())
        params = scale_heads(params, scale)
        model.load_state_dict(GO.expand_to_reference_state_dict(params), strict=True)
    store, hooks = {}, []
    for name in layers:
        def hook(mod, args, kwargs, output, name=name):
            _flatten(f"{tag}.L.{name}.in", args, store)
            _flatten(f"{tag}.L.{name}.kw", kwargs, store)
            _flatten(f"{tag}.L.{name}.out", output, store)
        hooks.append(model.get_submodule(name).register_forward_hook(hook, with_kwargs=True))
    E, F = model(dict(inputs))
    for h in hooks:
        h.remove()
    out.update(store)
    out[f"{tag}.seed"], out[f"{tag}.cfg"], out[f"{tag}.out_scale"] = np.array(seed), np.array(repr(cfg)), np.array(scale)
    out[f"{tag}.Z"], out[f"{tag}.R"], out[f"{tag}.N"] = Z, R, N
    out[f"{tag}.Et"], out[f"{tag}.Ft"] = Et, Ft
    for k in dc.index_keys:
        out[f"{tag}.{k}"] = batch[k].numpy().astype(np.int32)
    out[f"{tag}.E"], out[f"{tag}.F"] = E.detach().numpy(), F.detach().numpy()
    if with_grads:
        Fl = F[:, 0] if F.dim() == 3 else F
        loss = GO.training_loss(E[:, :1], Fl, batch["E"].double()[:, None] if batch["E"].dim() == 1 else batch["E"].double(),
                                batch["F"].double())
        out[f"{tag}.loss"] = loss.detach().numpy()
        loss.backward()
        record_grads(tag, model, out)
        named = dict(model.named_parameters())
        for n in ("mlp_cbf3.weight", "int_blocks.0.trip_interaction.mlp_cbf.weight", "int_blocks.0.dense_ca.weight",
                  "out_blocks.1.out_forces.weight", "out_blocks.0.seq_forces.0.weight", "mlp_sbf4.weight",
                  "int_blocks.3.quad_interaction.mlp_sbf.weight", "int_blocks.3.layers_after_skip.0.dense_mlp.1.weight"):
            if n in named and named[n].grad is not None:
                out[f"{tag}.grad.{n}"] = named[n].grad.numpy()
    print(tag, "E", E.detach().numpy().ravel()[:3], "mean|F|", float(F.detach().abs().mean()), "out_scale", scale,
          {k: int(batch[k].shape[0]) for k in ("id_a", "id3_reduce_ca")}, flush=True)


def golden_models2():
    """-> model2.npz: unit-force deep models (T/Q, 2 and 4 blocks), direct-force models (GemNet-dT/dQ), a two-target
    model, and per-layer inputs/outputs of the reference modules (P1/P2/P3/P4/P5/P10/P13)."""
    out = {}
    m12 = make_molecule(12, 1000 * 1 + 0)
    m9 = make_molecule(9, 77, box=4.5)
    m32 = make_molecule(32, 2000)
    pair = ([m12["R"], m9["R"]], [m12["Z"], m9["Z"]])
    Q_LAYERS = ("mlp_cbf3", "mlp_sbf4", "int_blocks.0", "int_blocks.0.trip_interaction",
                "int_blocks.0.quad_interaction", "int_blocks.0.trip_interaction.mlp_cbf",
                "int_blocks.0.quad_interaction.mlp_sbf", "int_blocks.0.atom_update", "out_blocks.1")
    T_LAYERS = ("mlp_cbf3", "int_blocks.1", "int_blocks.1.trip_interaction", "int_blocks.1.trip_interaction.mlp_cbf",
                "int_blocks.1.atom_update", "out_blocks.2")
    # per-layer captures: small GemNet-Q (both interactions) and full-width GemNet-T (second block)
    run_model2(cfg_small(False), 2, [m12["R"]], [m12["Z"]], "q1L", out, unit_forces=False, layers=Q_LAYERS)
    run_model2(cfg_full(True, 2), 3, *pair, "t2s", out, with_grads=True, layers=T_LAYERS)
    run_model2(cfg_full(False, 2), 4, *pair, "q2s", out, with_grads=True)
    # direct forces (gemnet.py:586-597, atom_update_block.py:181-188): small T uncoupled, small Q coupled, full-width T
    run_model2(dict(cfg_small(True), direct_forces=True, forces_coupled=False), 41, [m12["R"]], [m12["Z"]], "dt1", out,
               with_grads=True, unit_forces=False)
    run_model2(dict(cfg_small(False), direct_forces=True, forces_coupled=True), 42, [m12["R"]], [m12["Z"]], "dq1", out,
               with_grads=True, unit_forces=False)
    run_model2(dict(cfg_full(True, 2), direct_forces=True, forces_coupled=True), 43, *pair, "dt2s", out, with_grads=True)
    # two targets with autograd forces (one backward per target, gemnet.py:599-609)
    run_model2(dict(cfg_small(True), num_targets=2), 44, [m12["R"]], [m12["Z"]], "t1m", out, unit_forces=False)
    # the published configurations (pretrained/*/model_kwargs.json), 4 blocks, one 32-atom molecule, unit forces
    # ... with the second-order pass loss.backward() pinned on them too (gemnet.py:603-611, trainer.py:346)
    run_model2(cfg_full(True, 4), 5, [m32["R"]], [m32["Z"]], "t4s", out, with_grads=True)
    run_model2(cfg_full(False, 4), 6, [m32["R"]], [m32["Z"]], "q4s", out, with_grads=True)
    np.savez_compressed(os.path.join(HERE, "model2.npz"), **out)
    print("model2.npz", len(out), "arrays")


def golden_keys():
    import json
    out = {}
    for name, cfg in (("T", cfg_small(True)), ("Q", cfg_small(False)),
                      ("dT", dict(cfg_small(True), direct_forces=True))):
        m = GemNet(**cfg, scale_file=SCALE_FILE)
        out[name] = {"state_dict": {k: list(v.shape) for k, v in m.state_dict().items()},
                     "named_parameters": [n for n, _ in m.named_parameters()]}
    # the published model configurations (pretrained/*/model_kwargs.json): what a `model.pth` of the reference holds
    for name in ("GemNet-T", "GemNet-Q"):
        with open(os.path.join(REF, "pretrained", name, "model_kwargs.json")) as f:
            kw = json.load(f)
        m = GemNet(**dict(kw, scale_file=SCALE_FILE))
        out["pretrained/" + name] = {"model_kwargs": kw,
                                     "state_dict": {k: list(v.shape) for k, v in m.state_dict().items()},
                                     "named_parameters": [n for n, _ in m.named_parameters()]}
    with open(os.path.join(HERE, "state_dict_keys.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("state_dict_keys.json", {k: len(v["state_dict"]) for k, v in out.items()})


# ----------------------------------------------------------------------------- G5 trainer
TRAINER_CASES = {
    # tag: (model cfg, model seed, Trainer keywords)
    "rmse": (cfg_small(True), 21, dict(learning_rate=2e-3, decay_steps=3, decay_rate=0.5, warmup_steps=2,
                                       weight_decay=0.01, staircase=False, grad_clip_max=0.5, ema_decay=0.9,
                                       rho_force=0.99, loss="rmse")),
    "mae_agc": (cfg_small(True), 22, dict(learning_rate=1e-3, decay_steps=2, decay_rate=0.9, warmup_steps=0,
                                          weight_decay=0.0, staircase=True, grad_clip_max=0.05, ema_decay=0.99,
                                          rho_force=0.9, loss="mae", agc=True)),
    "quad": (cfg_small(False), 23, dict(learning_rate=1e-3, decay_steps=10, decay_rate=0.9, warmup_steps=3,
                                        weight_decay=0.001, grad_clip_max=10.0, ema_decay=0.999,
                                        rho_force=0.999, loss="rmse")),
}
TRAINER_STEPS = 4


def golden_trainer():
    """Four `Trainer.train_on_batch` steps + one `test_on_batch` with the EMA weights, run by the REFERENCE
    Trainer / Metrics (gemnet/training/trainer.py:325-420) in float64 on two fixed batches."""
    from gemnet.training.trainer import Trainer
    from gemnet.training.metrics import Metrics
    out = {}
    mols = [make_molecule(10, 500, box=4.2), make_molecule(7, 501, box=3.6), make_molecule(9, 502, box=4.0)]
    sf = GO.load_scale_factors(SCALE_FILE)
    for tag, (cfg, seed, kw) in TRAINER_CASES.items():
        to = cfg["triplets_only"]
        N = np.array([len(m["R"]) for m in mols], dtype=np.int32)
        R = np.concatenate([m["R"] for m in mols]).astype(np.float32)
        Z = np.concatenate([m["Z"] for m in mols]).astype(np.int32)
        rs = np.random.RandomState(seed)
        Et = rs.standard_normal(len(N)).astype(np.float32)
        Ft = (0.3 * rs.standard_normal(R.shape)).astype(np.float32)
        dc = _MemContainer(dict(N=N, Z=Z, R=R, E=Et, F=Ft), 5.0, 10.0, to)
        batches = [[0, 1], [2, 1], [0, 2], [1]]

        def stream():
            i = 0
            while True:
                b = dc[batches[i % len(batches)]]
                inputs = {k: v for k, v in b.items() if k not in ("E", "F")}
                inputs["R"] = inputs["R"].double()
                yield inputs, {"E": b["E"].double(), "F": b["F"].double()}
                i += 1

        model = GemNet(**cfg, scale_file=SCALE_FILE).double()
        model.load_state_dict(GO.expand_to_reference_state_dict(GO.make_params(cfg, seed, sf, dtype=torch.float64)),
                              strict=True)
        trainer = Trainer(model, **kw)
        metrics = Metrics("train", trainer.tracked_metrics)
        it = stream()
        losses, lrs_ = [], []
        for _ in range(TRAINER_STEPS):
            losses.append(float(trainer.train_on_batch(it, metrics)))
            lrs_.append([s.get_last_lr()[0] for s in trainer.schedulers.wrapped])
        unused = [n for n, p in model.named_parameters() if p.requires_grad and p.grad is None]
        assert not unused, unused
        res = metrics.result(append_tag=False)
        out[f"{tag}.N"], out[f"{tag}.Z"], out[f"{tag}.R"], out[f"{tag}.Et"], out[f"{tag}.Ft"] = N, Z, R, Et, Ft
        out[f"{tag}.cfg"], out[f"{tag}.kw"], out[f"{tag}.seed"] = np.array(repr(cfg)), np.array(repr(kw)), np.array(seed)
        out[f"{tag}.batches"] = np.array([",".join(map(str, b)) for b in batches])
        out[f"{tag}.losses"] = np.array(losses)
        out[f"{tag}.lrs"] = np.array(lrs_)
        out[f"{tag}.metric_names"] = np.array(sorted(res))
        out[f"{tag}.metric_values"] = np.array([float(res[k]) for k in sorted(res)])
        names = [n for n, _ in model.named_parameters()]
        out[f"{tag}.param_names"] = np.array(names)
        out[f"{tag}.param_norms"] = np.array([float(p.detach().norm()) for _, p in model.named_parameters()])
        out[f"{tag}.ema_norms"] = np.array([float(s.norm()) for s in trainer.exp_decay.shadow_params])
        # plateau decay + evaluation with the averaged weights (train.ipynb's validation block)
        for v in (1.0, 1.0, 1.0):
            trainer.decay_maybe(v)
        trainer.save_variable_backups()
        trainer.load_averaged_variables()
        val = Metrics("val", trainer.tracked_metrics)
        out[f"{tag}.val_loss"] = np.array(float(trainer.test_on_batch(it, val)))
        trainer.restore_variable_backups()
        out[f"{tag}.restored_norms"] = np.array([float(p.detach().norm()) for _, p in model.named_parameters()])
        print(tag, "losses", losses, "lr", lrs_[-1], "val", float(out[f"{tag}.val_loss"]))
    np.savez_compressed(os.path.join(HERE, "trainer.npz"), **out)
    print("trainer.npz", len(out), "arrays")


# ------------------------------------------------------------------------ G6 scale-factor fit
def golden_scaling():
    """fit_scaling.py:94-159 run with the REFERENCE GemNet / Trainer / AutomaticFit: every scale factor of a
    direct-force model fitted in creation order from two fixed batches (float64).  -> scaling_fit.json"""
    import json
    import tempfile
    from gemnet.model.layers.scaling import AutomaticFit
    from gemnet.model.utils import write_json
    from gemnet.training.trainer import Trainer
    from gemnet.training.metrics import Metrics
    result = {}
    mols = [make_molecule(10, 500, box=4.2), make_molecule(7, 501, box=3.6), make_molecule(9, 502, box=4.0)]
    for tag, to, seed in (("T", True, 31), ("Q", False, 32)):
        cfg = dict(cfg_small(to), direct_forces=True)
        N = np.array([len(m["R"]) for m in mols], dtype=np.int32)
        R = np.concatenate([m["R"] for m in mols]).astype(np.float32)
        Z = np.concatenate([m["Z"] for m in mols]).astype(np.int32)
        dc = _MemContainer(dict(N=N, Z=Z, R=R, E=np.zeros(len(N), np.float32), F=np.zeros_like(R)), 5.0, 10.0, to)
        batches = [[0, 1], [2]]

        def stream():
            i = 0
            while True:
                b = dc[batches[i % len(batches)]]
                inputs = {k: v for k, v in b.items() if k not in ("E", "F")}
                inputs["R"] = inputs["R"].double()
                yield inputs, {"E": b["E"].double(), "F": b["F"].double()}
                i += 1

        with tempfile.TemporaryDirectory() as tmp:
            scale_file = os.path.join(tmp, "scaling.json")
            write_json(scale_file, {"comment": "golden"})
            AutomaticFit.set2fitmode()
            model = GemNet(**cfg, scale_file=scale_file).double()
            model.load_state_dict(GO.expand_to_reference_state_dict(GO.make_params(cfg, seed, None, torch.float64)),
                                  strict=True)
            trainer = Trainer(model)
            metrics = Metrics("train", trainer.tracked_metrics, None)
            it = stream()
            order = []
            while not AutomaticFit.fitting_completed():
                for _ in range(len(batches)):
                    trainer.test_on_batch(it, metrics)
                order.append(AutomaticFit.activeVar._name)
                AutomaticFit.activeVar.fit()
            AutomaticFit.fitting_mode = False
            with open(scale_file) as f:
                fitted = json.load(f)
        fitted.pop("comment")
        result[tag] = dict(cfg=cfg, seed=seed, batches=batches, order=order, fitted=fitted,
                           N=N.tolist(), Z=Z.tolist(), R=[[float(x) for x in r] for r in R])
        print(tag, fitted)
    with open(os.path.join(HERE, "scaling_fit.json"), "w") as f:
        json.dump(result, f, indent=1)


# ----------------------------------------------------------------------------- N4: TF checkpoint variable names
def golden_tfnames():
    """Which TensorFlow checkpoint variable the reference's `GemNet.load_tfmodel` (gemnet.py:617-778) copies into
    which parameter.  TensorFlow is not installed: the reference module's `tf` is replaced by a recorder whose reader
    answers every `get_tensor(name)` with a 0-d array holding a serial number; after the call every parameter is
    filled with the serial number of the variable it was loaded from (0-d arrays broadcast in `.data.copy_`)."""
    import json
    import gemnet.model.gemnet as ref_mod
    out = {}
    for tag, cfg in (("T", cfg_small(True)), ("Q", cfg_small(False)), ("dT", dict(cfg_small(True), direct_forces=True)),
                     ("T3", cfg_full(True, 3))):
        m = GemNet(**cfg, scale_file=SCALE_FILE)
        for p_ in m.parameters():
            p_.data.fill_(-1.0)
        names = []

        class Reader:
            def get_tensor(self, name):
                names.append(name)
                return np.full((), len(names) - 1, dtype=np.float32)

        ref_mod.tf = types.SimpleNamespace(train=types.SimpleNamespace(load_checkpoint=lambda path: Reader()))
        error = None
        try:
            m.load_tfmodel("unused")
        except AttributeError as e:   # direct-force models: the reference reads out_forces/bias of a bias-free Dense
            error = f"{type(e).__name__}: {e}"
        finally:
            ref_mod.tf = None
        mapping, not_loaded = {}, []
        for n, p_ in m.named_parameters():
            v = p_.detach().flatten()
            assert bool((v == v[0]).all())
            if float(v[0]) < 0:
                not_loaded.append(n)
            else:
                mapping[n] = names[int(v[0])]
        assert len(set(mapping.values())) == len(mapping)
        assert error is not None or len(mapping) == len(names)
        out[tag] = dict(cfg={k: v for k, v in cfg.items() if isinstance(v, (int, float, bool, str))},
                        mapping=mapping, not_loaded=not_loaded, error=error,
                        last_request=names[-1] if error else None)
        print(tag, len(mapping), "variables;", "not loaded:", not_loaded)
    with open(os.path.join(HERE, "tf_names.json"), "w") as f:
        json.dump(out, f, indent=1)


# ------------------------------------------------------- G7: BASELINE-size fixtures (round 6)
def _index_digest(idx, triplets_only):
    """sizes + SHA-256 (int32 little-endian bytes) of the canonicalised reference index arrays."""
    import hashlib
    from oracle import index_oracle as IO
    can = IO.canonicalize({k: np.asarray(v) for k, v in idx.items()}, triplets_only)
    return {k: dict(n=int(np.asarray(v).shape[0]),
                    sha256=hashlib.sha256(np.ascontiguousarray(np.asarray(v).astype("<i4")).tobytes()).hexdigest())
            for k, v in sorted(can.items())}


def run_fullsize(cfg, seed, ds, tag, out, digests, with_grads=False):
    """Reference forward+force in float64 on a generated dataset `ds` (all its molecules in ONE batch); the output
    heads rescaled to mean|F| = 1 eV/A.  Only E, F, the head scale and the (seeded, regenerable) positions are stored;
    the reference's index arrays go into `digests` as sizes + SHA-256 of their canonical form."""
    import time
    to = cfg["triplets_only"]
    t0 = time.time()
    dc = _MemContainer(dict(ds), 5.0, 10.0, to)
    batch = dc[list(range(len(ds["N"])))]
    t_idx = time.time() - t0
    digests[tag] = _index_digest({k: batch[k].numpy() for k in dc.index_keys}, to)
    sf = GO.load_scale_factors(SCALE_FILE)
    params = GO.make_params(cfg, seed, sf, dtype=torch.float64)
    inputs = {k: v for k, v in batch.items() if k not in ("E", "F")}
    inputs["R"] = inputs["R"].double()
    model = GemNet(**cfg, scale_file=SCALE_FILE).double()
    model.load_state_dict(GO.expand_to_reference_state_dict(params), strict=True)
    model.train()
    t0 = time.time()
    _, F0 = model(dict(inputs))
    t_fwd = time.time() - t0
    scale = 1.0 / float(F0.detach().abs().mean())
    del F0
    model.load_state_dict(GO.expand_to_reference_state_dict(scale_heads(params, scale)), strict=True)
    E, F = model(dict(inputs))
    out[f"{tag}.seed"], out[f"{tag}.cfg"], out[f"{tag}.out_scale"] = np.array(seed), np.array(repr(cfg)), np.array(scale)
    out[f"{tag}.N"], out[f"{tag}.Z"], out[f"{tag}.R"] = ds["N"], ds["Z"], ds["R"]
    out[f"{tag}.E"], out[f"{tag}.F"] = E.detach().numpy(), F.detach().numpy()
    if with_grads:
        # the training step at this size (trainer.py:325-346): loss on the dataset's targets, loss.backward() THROUGH the force;
        # stored as the loss, every parameter's gradient norm and 4 fixed +-1 probe projections (record_grads)
        Et = torch.tensor(np.asarray(ds["E"], dtype=np.float64)).reshape(-1, 1)
        Ft = torch.tensor(np.asarray(ds["F"], dtype=np.float64))
        loss = GO.training_loss(E[:, :1], F, Et, Ft)
        out[f"{tag}.loss"] = loss.detach().numpy()
        loss.backward()
        record_grads(tag, model, out)
        out[f"{tag}.Et"], out[f"{tag}.Ft"] = Et.numpy().astype(np.float32), Ft.numpy().astype(np.float32)
        model.zero_grad(set_to_none=True)
    # the reference's OWN float32 path (its default dtype) on the same weights and inputs: what "the reference PyTorch CPU
    # path" returns, and how far fp32 rounding alone takes it from the float64 result on this fixture
    model32 = GemNet(**cfg, scale_file=SCALE_FILE)
    model32.load_state_dict(GO.expand_to_reference_state_dict({k: v.float() for k, v in scale_heads(params, scale).items()}),
                            strict=True)
    model32.train()
    in32 = dict(inputs)
    in32["R"] = in32["R"].float()
    E32, F32 = model32(in32)
    out[f"{tag}.E32"], out[f"{tag}.F32"] = E32.detach().numpy(), F32.detach().numpy()
    noise = float((F32.detach().double() - F.detach()).abs().mean())
    print(tag, "E", E.detach().numpy().ravel()[:3], "mean|F|", float(F.detach().abs().mean()), "out_scale", scale,
          {k: int(batch[k].shape[0]) for k in ("id_a", "id3_reduce_ca") + (() if to else ("id4_reduce_ca",))},
          f"index {t_idx:.1f} s, reference forward+force {t_fwd:.1f} s (float64); reference float32 vs float64 force MAE {noise:.3e}",
          flush=True)


def golden_fullsize():
    """-> fullsize.npz + fullsize_index.json: the workloads BASELINE.json names, run by the REFERENCE in float64 with
    the published 4-block configurations (gemnet/model/gemnet.py:453-615, training/data_container.py:244-489):
      t64s / q64s   one 64-atom molecule (configs[4]'s molecule size), GemNet-T and GemNet-Q
      tB32          the 32 x 32-atom GemNet-T batch of configs[1] (bench.py's rank-0 workload: make_dataset(32, 32, config=2))
      qB4           a 4 x 32-atom GemNet-Q batch (configs[2]'s per-molecule workload, batched)
    and index digests (sizes + SHA-256 of the canonical arrays) at 32 atoms, 64 atoms and for the B = 32 batch, T and Q."""
    import json
    out, digests = {}, {}
    m64 = make_molecule(64, 4000)
    one64 = dict(N=np.array([64], np.int32), Z=m64["Z"], R=m64["R"], E=np.zeros(1, np.float32), F=np.zeros_like(m64["R"]))
    run_fullsize(cfg_full(True, 4), 7, one64, "t64s", out, digests)
    run_fullsize(cfg_full(False, 4), 8, one64, "q64s", out, digests)
    run_fullsize(cfg_full(True, 4), 5, make_dataset(32, 32, config=2), "tB32", out, digests, with_grads=True)
    run_fullsize(cfg_full(False, 4), 6, make_dataset(4, 32, config=2), "qB4", out, digests, with_grads=True)
    # index-only digests: one 32-atom molecule (T and Q), the B = 32 batch as GemNet-Q sees it
    for tag, ds, to in (("idx32.T", make_dataset(1, 32, config=2), True), ("idx32.Q", make_dataset(1, 32, config=2), False),
                        ("idxB32.Q", make_dataset(32, 32, config=2), False)):
        dc = _MemContainer(dict(ds), 5.0, 10.0, to)
        batch = dc[list(range(len(ds["N"])))]
        digests[tag] = _index_digest({k: batch[k].numpy() for k in dc.index_keys}, to)
        print(tag, {k: v["n"] for k, v in digests[tag].items()}, flush=True)
    np.savez_compressed(os.path.join(HERE, "fullsize.npz"), **out)
    with open(os.path.join(HERE, "fullsize_index.json"), "w") as f:
        json.dump(digests, f, indent=1)
    print("fullsize.npz", len(out), "arrays; fullsize_index.json", len(digests), "digests")


if __name__ == "__main__":
    which = sys.argv[1:] or ["basis", "indices", "models", "models2", "keys", "trainer", "scaling", "tfnames", "fullsize"]
    if "tfnames" in which:
        golden_tfnames()
    if "scaling" in which:
        golden_scaling()
    if "trainer" in which:
        golden_trainer()
    if "basis" in which:
        golden_basis()
    if "indices" in which:
        golden_indices()
    if "models" in which:
        golden_models()
    if "models2" in which:
        golden_models2()
    if "keys" in which:
        golden_keys()
    if "fullsize" in which:
        golden_fullsize()
