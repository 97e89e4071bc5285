"""The MD adapter on the host: `md.DeviceMolecule` has the surface of the reference's `Molecule` (ase_calculator.py:23-104) and
`GemNet.predict` accepts what its `get()` returns.  (Host logic on the emulated launchers; the device path is
tests/test_gpu_md.py.)"""
import numpy as np
import pytest
import torch

from conftest import SCALE_FILE
import cpu_kernels
from gemnet_pytorch_amd.md import DeviceMolecule, MoleculeInputs
from gemnet_pytorch_amd.model.gemnet import GemNet
from gemnet_pytorch_amd.synthetic import make_molecule
from gemnet_pytorch_amd.training.data_container import DataContainer

CFG = dict(num_spherical=7, num_radial=6, num_blocks=1, emb_size_atom=32, emb_size_edge=32, emb_size_trip=16,
           emb_size_quad=16, emb_size_rbf=16, emb_size_cbf=16, emb_size_sbf=32, emb_size_bil_trip=16,
           emb_size_bil_quad=16, num_before_skip=1, num_after_skip=1, num_concat=1, num_atom=2, triplets_only=True)


def test_device_molecule_has_the_reference_molecule_surface():
    mol = make_molecule(10, 3)
    m = DeviceMolecule(mol["R"], mol["Z"], 5.0, 10.0, triplets_only=True)
    assert m.N.tolist() == [10] and m.device == "cpu"
    m.update(R=mol["R"] + 0.1)
    with pytest.raises(AssertionError):
        m.update(R=np.zeros((3, 3)))
    m.to("cpu")
    inputs = m.get()
    assert isinstance(inputs, MoleculeInputs) and set(inputs) == {"R", "Z", "N"}
    assert inputs["R"].dtype == torch.float32 and inputs["Z"].dtype == torch.int64
    np.testing.assert_allclose(inputs["R"].numpy(), (mol["R"] + 0.1).astype(np.float32))
    assert (inputs.cutoff, inputs.int_cutoff, inputs.triplets_only) == (5.0, 10.0, True)


def test_predict_accepts_molecule_inputs_and_matches_the_datacontainer_path():
    mol = make_molecule(10, 4)
    with cpu_kernels.emulate():
        torch.manual_seed(0)
        model = GemNet(**CFG, scale_file=SCALE_FILE).eval()
        model._check_inputs = lambda R: None
        m = DeviceMolecule(mol["R"], mol["Z"], 5.0, 10.0, triplets_only=True)
        E, F = model.predict(m.get())
        dc = DataContainer.from_arrays(dict(R=mol["R"].astype(np.float32), Z=mol["Z"], N=np.array([10]), E=np.zeros((1, 1), np.float32),
                                            F=np.zeros((10, 3), np.float32)), 5.0, 10.0, triplets_only=True)
        b = dc[[0]]
        E2, F2 = model.predict({k: v for k, v in b.items() if k not in ("E", "F")})
    assert not E.requires_grad and E.device.type == "cpu"
    torch.testing.assert_close(E, E2)
    torch.testing.assert_close(F, F2)
    with pytest.raises(ValueError):
        with cpu_kernels.emulate():
            model.predict(DeviceMolecule(mol["R"], mol["Z"], 5.0, 10.0, triplets_only=False).get())
