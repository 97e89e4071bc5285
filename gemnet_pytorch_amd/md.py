"""Molecular-dynamics entry point: the reference's `ase_calculator.Molecule` surface over the device-resident path.

The reference's MD loop (ase_calculator.py:148-170, `GNNCalculator.calculate`) does, every step,

    self.molecule.update(R=atoms.positions)        # new positions
    inputs = self.molecule.get()                   # index construction on the host (data_container.py:244-489) + H2D copies
    energy, forces = self.model.predict(inputs)    # forward + force, D2H

`Molecule` builds the neighbour lists / triplets / quadruplets in numpy and scipy on the CPU each time.  `DeviceMolecule`
has the same constructor and the same three methods the calculator uses (`update`, `get`, `to`) but `get()` hands back
only positions, atomic numbers and the molecule layout, tagged as `MoleculeInputs`; `GemNet.predict` recognises the tag
and serves the call from `runtime.DynamicForceField`: index construction on the device, padded to fixed capacities, ONE
captured hipGraph replayed per step (bit-identical to the eager run on the unpadded arrays).  The calculator class itself
is used unchanged:

    from gemnet_pytorch_amd.md import DeviceMolecule as Molecule          # the only edited line of an MD script
    calc = GNNCalculator(Molecule(R, Z, cutoff, int_cutoff, triplets_only), model=model, atoms=atoms)

Triplets-only and quadruplet models alike (GemNet-Q — the model of the reference's MD example — since round 5: the padding
scheme covers interaction edges, intermediate triplets and quadruplets, padded.py)."""
import numpy as np
import torch


class MoleculeInputs(dict):
    """What `DeviceMolecule.get()` returns: {"R", "Z", "N"} (no index arrays) + the cutoffs that define the graph."""

    def __init__(self, data, cutoff, int_cutoff, triplets_only, layout_key=None):
        super().__init__(data)
        self.cutoff, self.int_cutoff, self.triplets_only = float(cutoff), float(int_cutoff), bool(triplets_only)
        # (atomic numbers, molecule sizes) as a hashable HOST value — the key of the per-system force field; given by
        # DeviceMolecule (which holds both on the host), else derived from the tensors once per call (a device read-back)
        self.layout_key = layout_key


class DeviceMolecule:
    """Drop-in for the reference's `Molecule(R, Z, cutoff, int_cutoff, triplets_only)` (ase_calculator.py:23-104)."""

    def __init__(self, R, Z, cutoff, int_cutoff, triplets_only=False):
        R = np.asarray(R)
        Z = np.asarray(Z)
        assert R.shape == (len(Z), 3)
        self.cutoff, self.int_cutoff, self.triplets_only = cutoff, int_cutoff, triplets_only
        self.R = R
        self.Z = Z
        self.N = np.array([len(Z)], dtype=np.int32)
        self.device = "cpu"
        self._Z_dev = self._N_dev = None
        self._key = (np.asarray(Z, dtype=np.int64).tobytes(), (int(len(Z)),))

    def update(self, R):
        """New positions (ase_calculator.py:86-97)."""
        R = np.asarray(R) if not torch.is_tensor(R) else R
        assert tuple(self.R.shape) == tuple(R.shape)
        self.R = R

    def to(self, device):
        """Device of the tensors `get()` returns (ase_calculator.py:99-103)."""
        self.device = device
        self._Z_dev = self._N_dev = None

    def get(self):
        if self._Z_dev is None:   # constant over the trajectory: uploaded once
            self._Z_dev = torch.as_tensor(np.asarray(self.Z), dtype=torch.int64).to(self.device)
            self._N_dev = torch.as_tensor(self.N, dtype=torch.int64).to(self.device)
        R = self.R if torch.is_tensor(self.R) else torch.as_tensor(np.asarray(self.R, dtype=np.float32))
        return MoleculeInputs(dict(R=R.to(self.device, dtype=torch.float32), Z=self._Z_dev, N=self._N_dev),
                              self.cutoff, self.int_cutoff, self.triplets_only, layout_key=self._key)


def predict_molecule(model, inputs, to_host=False):
    """`GemNet.predict` for `MoleculeInputs`: -> (E, F) on the device (inference only); `to_host`: detached host copies, the
    replayed step's range flag checked after the copy (what `GemNet.predict` returns)."""
    R, Z, N = inputs["R"], inputs["Z"], inputs["N"]
    if inputs.triplets_only != model.triplets_only:
        raise ValueError("DeviceMolecule(triplets_only=...) does not match the model")
    if not R.is_cuda:
        # host tensors: the host index builder (include/gemnet_index.h) + the ordinary forward (raises without a device,
        # like every other entry point: there is no CPU compute path)
        from .training.data_container import DataContainer
        dc = DataContainer.from_arrays(dict(R=R.numpy(), Z=Z.numpy(), N=N.numpy(), E=np.zeros((1, 1), np.float32),
                                            F=np.zeros((R.shape[0], 3), np.float32)),
                                       inputs.cutoff, inputs.int_cutoff, triplets_only=model.triplets_only)
        b = dc[[0]]
        E, F = model({k: v for k, v in b.items() if k not in ("E", "F")})
        return (E.detach().cpu(), F.detach().cpu()) if to_host else (E, F)
    lk = inputs.layout_key
    if lk is None:
        lk = (Z.detach().cpu().numpy().astype(np.int64).tobytes(), tuple(int(n) for n in N.tolist()))
    key = (lk, inputs.cutoff, inputs.int_cutoff, R.device.index)
    cache = model.__dict__.setdefault("_md_fields", {})
    ff = cache.get(key)
    was_training = model.training
    if ff is None:
        model.eval()
        # (the caller's parameters are left as they are — `requires_grad` included: an eval-mode forward with forces by autograd
        #  treats the weights as constants anyway (ops.constant_weights), and a force field built in the middle of a training
        #  script must not freeze the model; the reference's predict() does not touch it either, gemnet.py:780-784)
        from .runtime import DynamicForceField
        ff = DynamicForceField(model, Z, N.cpu().numpy(), inputs.cutoff, inputs.int_cutoff)
        if len(cache) >= 8:
            cache.pop(next(iter(cache)))
        cache[key] = ff
    try:
        model.eval()
        E, F = ff(R)
        if to_host:
            # the MD loop reads its results on the host (ase_calculator.py:166-170): after this copy the replay has completed
            # and its device-side range check (runtime.RangeFlag) is exact — an overflow of the fp16-plane arithmetic in a
            # REPLAYED step warns, moves the model to the bf16 planes, captures anew and repeats the step
            Eh, Fh = E.detach().cpu(), F.detach().cpu()
            if hasattr(ff, "range_tripped") and ff.range_tripped():
                E, F = ff.recover(R)
                Eh, Fh = E.detach().cpu(), F.detach().cpu()
            return Eh, Fh
        return E, F
    finally:
        model.train(was_training)
