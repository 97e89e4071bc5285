"""-m gpu: the MD entry point.  The reference's `GNNCalculator.calculate` (ase_calculator.py:148-170) is driven through a stub
of the `ase` calculator interface — `molecule.update(R=atoms.positions)`, `molecule.get()`, `model.predict(inputs)`,
`float(energy)`, `forces.numpy()` — once with a `Molecule` that follows the reference's class (host index construction
through the DataContainer) and once with `md.DeviceMolecule` (device index construction + one replayed hipGraph): the same
energies and forces along a short trajectory, for GemNet-T and GemNet-Q (both: device index build + the padded replay)."""
import numpy as np
import pytest
import torch

from conftest import SCALE_FILE
from gemnet_pytorch_amd.md import DeviceMolecule
from gemnet_pytorch_amd.model.gemnet import GemNet
from gemnet_pytorch_amd.synthetic import make_molecule
from gemnet_pytorch_amd.training.data_container import DataContainer

pytestmark = pytest.mark.gpu
CFG = dict(num_spherical=7, num_radial=6, num_blocks=2, emb_size_atom=128, emb_size_edge=128, emb_size_trip=64,
           emb_size_quad=32, emb_size_rbf=16, emb_size_cbf=16, emb_size_sbf=32, emb_size_bil_trip=64,
           emb_size_bil_quad=32, num_before_skip=1, num_after_skip=1, num_concat=1, num_atom=2)


class HostMolecule(DataContainer):
    """The reference's `Molecule` contract (ase_calculator.py:23-104): a one-molecule DataContainer with update / get / to."""

    def __init__(self, R, Z, cutoff, int_cutoff, triplets_only=False):
        data = dict(R=np.asarray(R, np.float32), Z=np.asarray(Z), N=np.array([len(Z)]), E=np.zeros((1, 1), np.float32),
                    F=np.zeros((len(Z), 3), np.float32))
        self._args = (cutoff, int_cutoff, triplets_only)
        self._data = data
        self._rebuild()
        self.device = "cpu"

    def _rebuild(self):
        self._dc = DataContainer.from_arrays(self._data, self._args[0], self._args[1], triplets_only=self._args[2])

    def update(self, R):
        self._data = dict(self._data, R=np.asarray(R, np.float32))
        self._rebuild()

    def to(self, device):
        self.device = device

    def get(self):
        b = self._dc[[0]]
        return {k: v.to(self.device) for k, v in b.items() if k not in ("E", "F")}


class Atoms:        # the two attributes of ase.Atoms the calculator reads
    def __init__(self, positions, numbers):
        self.positions, self.numbers = positions, numbers


class StubCalculator:
    """`GNNCalculator.calculate` without ase: the same five calls in the same order (ase_calculator.py:155-170)."""

    def __init__(self, molecule, model):
        self.molecule, self.model, self.results = molecule, model, {}

    def calculate(self, atoms):
        self.molecule.update(R=atoms.positions)
        inputs = self.molecule.get()
        energy, forces = self.model.predict(inputs)
        self.results["energy"] = float(energy)
        self.results["forces"] = forces.numpy()


@pytest.mark.parametrize("kind", ["T", "Q"])
def test_calculator_surface_reaches_the_device_path(kind):
    cfg = dict(CFG, triplets_only=kind == "T")
    torch.manual_seed(5)
    model = GemNet(**cfg, scale_file=SCALE_FILE).to("cuda").eval()
    mol = make_molecule(32, 7)
    R0, Z = mol["R"].astype(np.float32), mol["Z"]
    rng = np.random.RandomState(0)
    traj = [R0 + 0.02 * k * rng.randn(*R0.shape).astype(np.float32) for k in range(6)]   # neighbour lists change along it
    host = HostMolecule(R0, Z, 5.0, 10.0, triplets_only=cfg["triplets_only"])
    host.to("cuda")
    dev = DeviceMolecule(R0, Z, 5.0, 10.0, triplets_only=cfg["triplets_only"])
    dev.to("cuda")
    a, b = StubCalculator(host, model), StubCalculator(dev, model)
    sizes = set()
    for R in traj:
        atoms = Atoms(R, Z)
        a.calculate(atoms)
        b.calculate(atoms)
        sizes.add(int(host.get()["id_c"].shape[0]))
        scale = max(1.0, float(np.abs(a.results["forces"]).max()))
        assert abs(a.results["energy"] - b.results["energy"]) <= 2e-5 * max(1.0, abs(a.results["energy"]))
        np.testing.assert_allclose(b.results["forces"], a.results["forces"], atol=2e-5 * scale)
    assert len(sizes) > 1, "the trajectory was meant to change the edge count"
    fields = model.__dict__["_md_fields"]
    assert len(fields) == 1
    # one hipGraph served every step (a capacity margin above the first call) — GemNet-Q too since round 5
    ff = next(iter(fields.values()))
    assert ff.runner is not None and ff.runner.graph is not None and ff.recaptures <= 1
