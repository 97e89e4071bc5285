"""-m gpu: batches of changing array sizes replayed from ONE captured hipGraph (gemnet_pytorch_amd/padded.py): the real
molecules of a batch padded to the capacities get the energies and forces of the plain eager run on the unpadded batch,
for several batches with different edge / triplet counts in turn, twice around (the second round replays only)."""
import pytest
import torch

from conftest import SCALE_FILE
from gemnet_pytorch_amd.index_device import DeviceGraphBuilder
from gemnet_pytorch_amd.model.gemnet import GemNet
from gemnet_pytorch_amd.padded import PaddedGraphRunner
from gemnet_pytorch_amd.synthetic import make_dataset
from test_gpu_fullsize import FULL

pytestmark = pytest.mark.gpu
DEV = "cuda"


def test_padded_graph_replay_equals_eager_on_changing_batches():
    cfg = dict(FULL, triplets_only=True)
    torch.manual_seed(5)
    model = GemNet(**cfg, scale_file=SCALE_FILE).to(DEV).eval()
    model.requires_grad_(False)
    n_mol, n_atoms = 8, 32
    batches = []
    for b in range(4):
        ds = make_dataset(n_mol, n_atoms, config=2, first=(b + 1) * n_mol)
        R = torch.tensor(ds["R"], device=DEV, dtype=torch.float32)
        Z = torch.tensor(ds["Z"], device=DEV).long()
        N = torch.tensor(ds["N"], device=DEV).long()
        idx = DeviceGraphBuilder(ds["N"], 5.0, 10.0, True, device=DEV)(R)
        batches.append((Z, R, N, idx))
    sizes = [(int(i["id_c"].shape[0]), int(i["id3_reduce_ca"].shape[0])) for _, _, _, i in batches]
    assert len(set(sizes)) > 1, sizes                     # the point of the exercise: the shapes differ
    Z, _, N, _ = batches[0]
    assert all(torch.equal(N, b[2]) for b in batches)     # one layout of molecule sizes; atoms and geometry change
    e_cap, t_cap = PaddedGraphRunner.suggest_capacities(sizes)
    runner = PaddedGraphRunner(model, Z, N, e_cap, t_cap)
    ref = []
    for Zb, R, Nb, idx in batches:
        E, F = model(dict(Z=Zb, R=R.clone(), N=Nb, **idx))
        ref.append((E.detach().clone(), F.detach().clone()))
    worst = 0.0
    for rnd in range(2):
        for (Zb, R, Nb, idx), (E0, F0) in zip(batches, ref):
            E, F = runner(R, idx, Z=Zb)
            torch.cuda.synchronize()
            assert E.shape == E0.shape and F.shape == F0.shape
            worst = max(worst, float((F - F0).abs().max()), float((E - E0).abs().max()))
            scale = float(F0.abs().max())
            assert float((F - F0).abs().max()) <= 1e-5 * scale and float((E - E0).abs().max()) <= 1e-5 * float(E0.abs().max())
    # index build + replay in one call, the build on its own stream
    builders = [DeviceGraphBuilder(N.cpu().numpy(), 5.0, 10.0, True, device=DEV) for _ in batches]
    for rnd in range(2):
        for (Zb, R, Nb, idx), (E0, F0), bld in zip(batches, ref, builders):
            E, F = runner.build_and_run(bld, R, Z=Zb, positions_ready=bool(rnd))
            torch.cuda.synchronize()
            assert torch.equal(E, E0) and torch.equal(F, F0)
    print(f"padded replay vs eager over {len(batches)} batches {sizes} at capacities ({runner.e_cap}, {runner.t_cap}), "
          f"{runner.G} dummy groups: max abs deviation {worst:.3e}")
    with pytest.raises(ValueError):
        small = PaddedGraphRunner(model, Z, N, sizes[0][0] - 8, sizes[0][1])
        small(batches[0][1], batches[0][3], Z=batches[0][0])
