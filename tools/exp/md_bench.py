"""The MD loop of the reference's calculator (ase_calculator.py:148-170: update positions -> molecule.get() -> model.predict)
on ONE molecule, per model kind: `md.DeviceMolecule` (device index build + padded replay of one hipGraph, round 5 also for
GemNet-Q) against the eager forward on device-built indices.   PYTHONPATH=. python tools/exp/md_bench.py [n_atoms=32] [steps=40]"""
import sys
import time

import numpy as np
import torch

from gemnet_pytorch_amd.index_device import DeviceGraphBuilder
from gemnet_pytorch_amd.md import DeviceMolecule
from gemnet_pytorch_amd.model.gemnet import GemNet
from gemnet_pytorch_amd.synthetic import make_molecule

CFG = dict(num_spherical=7, num_radial=6, num_blocks=4, emb_size_atom=128, emb_size_edge=128, emb_size_trip=64,
           emb_size_quad=32, emb_size_rbf=16, emb_size_cbf=16, emb_size_sbf=32, emb_size_bil_trip=64,
           emb_size_bil_quad=32, num_before_skip=1, num_after_skip=1, num_concat=1, num_atom=2)
n_atoms = int(sys.argv[1]) if len(sys.argv) > 1 else 32
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
import os  # noqa: E402
scale_file = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "gemnet_pytorch_amd",
                          "scaling_factors.json")
for kind in ("T", "Q"):
    torch.manual_seed(1)
    model = GemNet(**dict(CFG, triplets_only=kind == "T"), scale_file=scale_file).to("cuda").eval()
    mol = make_molecule(n_atoms, 3)
    R0, Z = mol["R"].astype(np.float32), mol["Z"]
    rng = np.random.RandomState(0)
    traj = [R0 + 0.002 * k * rng.randn(*R0.shape).astype(np.float32) for k in range(steps + 5)]
    dm = DeviceMolecule(R0, Z, 5.0, 10.0, triplets_only=kind == "T")
    dm.to("cuda")
    for R in traj[:5]:
        dm.update(R)
        model.predict(dm.get())
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for R in traj[5:]:
        dm.update(R)
        E, F = model.predict(dm.get())
    t_md = (time.perf_counter() - t0) / steps
    ff = next(iter(model.__dict__["_md_fields"].values()))
    bld = DeviceGraphBuilder(np.array([n_atoms]), 5.0, 10.0, kind == "T", device="cuda")
    Zd, Nd = torch.tensor(Z, device="cuda").long(), torch.tensor([n_atoms], device="cuda")

    def eager(R):
        Rd = torch.tensor(R, device="cuda")
        E, F = model(dict(Z=Zd, R=Rd, N=Nd, **bld(Rd)))
        return E.detach().cpu(), F.detach().cpu()
    for R in traj[:5]:
        eager(R)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for R in traj[5:]:
        E2, F2 = eager(R)
    t_eager = (time.perf_counter() - t0) / steps
    dev = float((F - F2).abs().max())
    print(f"GemNet-{kind}, {n_atoms} atoms: predict() {t_md * 1e3:.2f} ms/step (device index build + padded replay, "
          f"{ff.recaptures} re-captures), eager forward on device-built indices {t_eager * 1e3:.2f} ms/step; last-step force "
          f"deviation {dev:.1e}")
