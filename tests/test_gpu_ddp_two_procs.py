"""-m gpu: the N > 1 code path on the real device with the one GPU a box has — TWO processes on cuda:0, `gloo` process group
(its collectives stage the device tensors through the host), the HIP `TrainStep` on UNEVEN molecule shards: the flat `.grad`
views, the hipGraph-captured forward + force + double backward, the gradient all-reduce behind the replay, the fused
clip + AdamW + EMA reading the reduced buffer, the per-step OR of the range-flag word (runtime.RangeFlag.snapshot).
The all-reduced gradients and the parameters after two optimizer steps equal the single-process run on the union batch
(gemnet/training/trainer.py:325-360 is the single-process step being reproduced).  RCCL itself needs
one device per rank: it is exercised with world_size 1 in tests/test_gpu_trainer.py and by `bench.py --gpus N` on a node."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT, SCALE_FILE

pytestmark = pytest.mark.gpu

CFG = dict(num_spherical=7, num_radial=6, num_blocks=2, emb_size_atom=64, emb_size_edge=64, emb_size_trip=32,
           emb_size_quad=32, emb_size_rbf=16, emb_size_cbf=16, emb_size_sbf=32, emb_size_bil_quad=32, emb_size_bil_trip=32,
           num_before_skip=1, num_after_skip=1, num_concat=1, num_atom=2, triplets_only=True)
N_MOL = 6
SHARDS = [[0, 2, 3, 5], [1, 4]]      # uneven on purpose: the loss weights are global counts, not local ones
STEPS = 2


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _data():
    from gemnet_pytorch_amd.synthetic import make_molecule
    from gemnet_pytorch_amd.training.data_container import DataContainer
    mols = [make_molecule(n, 700 + i, box=max(3.5, 0.42 * n)) for i, n in enumerate((9, 12, 7, 11, 10, 8))]
    ds = dict(N=np.array([m["N"] for m in mols], np.int32), Z=np.concatenate([m["Z"] for m in mols]),
              R=np.concatenate([m["R"] for m in mols]), E=np.array([m["E"] for m in mols], np.float32),
              F=np.concatenate([m["F"] for m in mols]))
    return DataContainer.from_arrays(ds, 5.0, 10.0, triplets_only=True)


def _model():
    from gemnet_pytorch_amd.model.gemnet import GemNet
    torch.manual_seed(3)
    return GemNet(**CFG, scale_file=SCALE_FILE).to("cuda")


def _batch(dc, ids):
    b = {k: v.to("cuda") for k, v in dc[ids].items()}
    t = {"E": b.pop("E"), "F": b.pop("F")}
    return b, t


def _run(world, ids, fused, captured):
    from gemnet_pytorch_amd.training.ddp import TrainStep
    model = _model()
    ts = TrainStep(model, world_size=world, fused_optimizer=fused, grad_clip_max=10.0)
    inputs, targets = _batch(_data(), ids)
    if captured:
        inputs = model.with_indices(inputs)
        ts.capture(inputs, targets)
    losses, grads = [], None
    for _ in range(STEPS):
        losses.append(ts(inputs, targets).detach().clone())
        grads = ts.buf.flat.detach().clone() if grads is None else grads     # reduced gradient of the FIRST step
    torch.cuda.synchronize()
    params = torch.cat([p.detach().reshape(-1) for p in model.parameters()]).cpu().numpy()
    return [float(l) for l in losses], grads.cpu().numpy(), params, ts


def _worker(rank, world, port, out_dir, fused, captured):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        losses, grads, params, ts = _run(world, SHARDS[rank], fused, captured)
        lt = torch.tensor(losses, dtype=torch.float64)
        dist.all_reduce(lt)              # the ranks' loss shares add up to the loss of the union
        np.save(os.path.join(out_dir, f"grad_{rank}.npy"), grads)
        np.save(os.path.join(out_dir, f"param_{rank}.npy"), params)
        np.save(os.path.join(out_dir, f"loss_{rank}.npy"), lt.numpy())
        np.save(os.path.join(out_dir, f"flag_{rank}.npy"), np.array([ts.flag.trips, ts.flag._step if ts.flag is not None else -1]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("fused,captured", [(True, True), (True, False), (False, True)],
                         ids=["fused-optimizer+captured-graph", "fused-optimizer+eager", "torch-optimizer+captured-graph"])
def test_two_processes_on_one_gpu_equal_the_single_process_step(tmp_path, fused, captured):
    ref_losses, ref_grads, ref_params, _ = _run(1, sorted(sum(SHARDS, [])), fused, captured)
    mp.spawn(_worker, args=(2, _free_port(), str(tmp_path), fused, captured), nprocs=2, join=True)
    g0, g1 = np.load(tmp_path / "grad_0.npy"), np.load(tmp_path / "grad_1.npy")
    p0, p1 = np.load(tmp_path / "param_0.npy"), np.load(tmp_path / "param_1.npy")
    assert np.array_equal(g0, g1), "both ranks hold the same all-reduced gradient"
    assert np.array_equal(p0, p1), "both ranks hold the same parameters after the optimizer steps"
    gn = float(np.linalg.norm(ref_grads))
    gerr = float(np.linalg.norm(g0 - ref_grads)) / gn
    perr = float(np.abs(p0 - ref_params).max())
    prms = float(np.sqrt(np.mean((p0 - ref_params) ** 2)))
    lerr = float(np.abs(np.load(tmp_path / "loss_0.npy") - np.array(ref_losses)).max())
    flags = np.load(tmp_path / "flag_0.npy")
    print(f"two ranks vs one [{'fused' if fused else 'torch'} optimizer, {'captured' if captured else 'eager'}]: "
          f"|g - g_ref| / |g_ref| = {gerr:.2e}, |param - param_ref| max {perr:.2e} rms {prms:.2e}, loss err {lerr:.2e}; "
          f"flag trips {int(flags[0])}, flag snapshots {int(flags[1])}")
    # fp32 sums in a different order (two shards' partial gradients vs one pass over the union)
    assert gerr <= 2e-5 and lerr <= 1e-5 * max(1.0, max(abs(x) for x in ref_losses))
    # two AdamW steps at lr 1e-3 move a parameter by up to 2e-3, and Adam divides by sqrt(v): an element whose gradient is
    # itself at the rounding level (|g| ~ 1e-7 |g|_max) may step in another direction — bounded per element, tiny on average
    assert perr <= 2e-4 and prms <= 5e-6
    assert int(flags[0]) == 0 and int(flags[1]) == STEPS      # the per-step flag collective ran, nothing tripped


def _worker_overflow(rank, world, port, out_dir):
    """Two ranks, captured step + fused optimizer in the fp16-plane arithmetic; the atom embedding is blown past the fp16
    range after the capture.  Records, per call, whether THIS rank fell back at that call."""
    import warnings
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from gemnet_pytorch_amd import kernels as K
        from gemnet_pytorch_amd.training.ddp import TrainStep
        K.DEFAULT_CHAIN_MODE = "h3"
        model = _model().train()
        ts = TrainStep(model, world_size=world, fused_optimizer=True)      # counts exchanged per step (global_counts=None)
        inputs, targets = _batch(_data(), SHARDS[rank])
        inputs = model.with_indices(inputs)
        ts(inputs, targets)
        ts.capture(inputs, targets)
        ts(inputs, targets)
        with torch.no_grad():
            model.atom_emb.embeddings.weight.mul_(1.0e6)
        before = ts.fused.flat_p.clone()
        fell, losses = [], []
        for call in range(5):
            if rank == 1 and call == 1:
                torch.cuda.synchronize()          # the ranks' hosts run at different distances ahead of their device
            with warnings.catch_warnings(record=True) as w:
                warnings.simplefilter("always")
                losses.append(float(ts(inputs, targets)))
            fell.append(int(any("fp16-plane" in str(x.message) for x in w)))
        torch.cuda.synchronize()
        np.save(os.path.join(out_dir, f"fell_{rank}.npy"), np.array(fell))
        np.save(os.path.join(out_dir, f"state_{rank}.npy"),
                np.array([ts.flag.trips, int(model.matmul_precision == "split6"), int(bool(torch.isfinite(ts.fused.flat_p).all())),
                          ts.fused.steps, int(torch.equal(ts.fused.flat_p, before))]))
        np.save(os.path.join(out_dir, f"p_{rank}.npy"), ts.fused.flat_p.cpu().numpy())
        np.save(os.path.join(out_dir, f"l_{rank}.npy"), np.array(losses))
    finally:
        dist.destroy_process_group()


def test_two_ranks_leave_the_fp16_planes_at_the_same_call(tmp_path):
    """The range flag is OR-reduced over the ranks at a fixed point of every step and polled with a fixed lag
    (runtime.RangeFlag.snapshot / poll_lagged): both ranks recapture at the SAME call — a rank that recaptured alone would pair
    its collectives with the other rank's gradient all-reduce — skip the same steps on the device, correct Adam's step counter by
    exactly that many, and hold identical finite parameters afterwards."""
    mp.spawn(_worker_overflow, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    f0, f1 = np.load(tmp_path / "fell_0.npy"), np.load(tmp_path / "fell_1.npy")
    s0, s1 = np.load(tmp_path / "state_0.npy"), np.load(tmp_path / "state_1.npy")
    print("fell back at call:", f0.tolist(), f1.tolist(), "state [trips, split6, finite, optimizer steps, unchanged]:", s0.tolist(), s1.tolist())
    assert f0.tolist() == f1.tolist() and int(f0.sum()) == 1, "one fall-back, at the same call on both ranks"
    k = int(np.argmax(f0))
    assert k >= 2                      # lag: the word of step i is read at call i + 2
    assert s0.tolist() == s1.tolist() and s0[0] == 1 and s0[1] == 1 and s0[2] == 1 and s0[4] == 0
    # steps taken = 2 before the blow-up + the calls from the fall-back on (every earlier call was skipped on the device)
    assert int(s0[3]) == 2 + (5 - k), (int(s0[3]), k)
    assert np.array_equal(np.load(tmp_path / "p_0.npy"), np.load(tmp_path / "p_1.npy"))
    l0 = np.load(tmp_path / "l_0.npy")
    assert np.isfinite(l0[k:]).all()
