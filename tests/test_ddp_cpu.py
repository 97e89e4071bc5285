"""N>1 path on CPU: world_size-2 gloo processes run the data-parallel TrainStep on molecule shards
(launchers emulated on CPU, float64) and must reproduce the single-process gradient on the union
batch — loss weighting by global counts, flat-buffer all-reduce, shared-grad rescale, global clip."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT, SCALE_FILE
from gemnet_pytorch_amd.training.ddp import FlatGradBuffer, TrainStep, partition_molecules

CFG = dict(num_spherical=7, num_radial=6, num_blocks=2, emb_size_atom=16, emb_size_edge=16, emb_size_trip=16,
           emb_size_quad=16, emb_size_rbf=16, emb_size_cbf=16, emb_size_sbf=16, emb_size_bil_quad=16,
           emb_size_bil_trip=16, num_before_skip=1, num_after_skip=1, num_concat=1, num_atom=1,
           triplets_only=True)


def test_partition_is_balanced_and_complete():
    costs = [100, 1, 1, 1, 50, 49, 3, 97]
    shards = partition_molecules(costs, 2)
    assert sorted(sum(shards, [])) == list(range(8))
    loads = [sum(costs[i] for i in s) for s in shards]
    assert abs(loads[0] - loads[1]) <= 3


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _make(n_mol):
    from gemnet_pytorch_amd.synthetic import make_dataset
    from gemnet_pytorch_amd.training.data_container import DataContainer
    ds = make_dataset(n_mol, 8, config=9)
    return DataContainer.from_arrays(ds, 5.0, 10.0, triplets_only=True)


def _model():
    from gemnet_pytorch_amd.model.gemnet import GemNet
    torch.manual_seed(3)
    m = GemNet(**CFG, scale_file=SCALE_FILE).double()
    m._check_inputs = lambda R: None
    return m


def _batch(dc, ids):
    b = dc[ids]
    t = {"E": b.pop("E").double(), "F": b.pop("F").double()}
    b["R"] = b["R"].double()
    return b, t


def _worker(rank, world, port, shards, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import cpu_kernels
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        with cpu_kernels.emulate():
            model = _model()
            ts = TrainStep(model, world_size=world)
            inputs, targets = _batch(_make(4), shards[rank])
            loss = ts(inputs, targets, step_optimizer=False)
        lt = loss.clone()
        dist.all_reduce(lt)
        np.save(os.path.join(out_dir, f"grad_{rank}.npy"), ts.buf.flat.numpy())
        np.save(os.path.join(out_dir, f"loss_{rank}.npy"), np.array(float(lt)))
    finally:
        dist.destroy_process_group()


def test_two_rank_gradients_equal_single_process(tmp_path):
    import cpu_kernels
    dc = _make(4)
    with cpu_kernels.emulate():
        model = _model()
        ts = TrainStep(model, world_size=1)
        inputs, targets = _batch(dc, [0, 1, 2, 3])
        loss_ref = float(ts(inputs, targets, step_optimizer=False))
        ref = ts.buf.flat.clone().numpy()
    shards = [[0, 3], [1, 2]]
    mp.spawn(_worker, args=(2, _free_port(), shards, str(tmp_path)), nprocs=2, join=True)
    g0 = np.load(tmp_path / "grad_0.npy")
    g1 = np.load(tmp_path / "grad_1.npy")
    assert np.array_equal(g0, g1)  # both ranks hold the same all-reduced gradient
    np.testing.assert_allclose(g0, ref, rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(float(np.load(tmp_path / "loss_0.npy")), loss_ref, rtol=1e-10)


def _worker_padded(rank, world, port, shards, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import cpu_kernels
    from gemnet_pytorch_amd.padded import dummy_positions
    from gemnet_pytorch_amd.training.ddp import PaddedTrainStep
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        with cpu_kernels.emulate():
            model = _model()
            inputs, targets = _batch(_make(4), shards[rank])
            idx = {k: inputs[k] for k in ("id_c", "id_a", "id_swap", "id_undir", "id3_reduce_ca", "id3_expand_ba")}
            E, T = int(idx["id_c"].shape[0]), int(idx["id3_reduce_ca"].shape[0])
            # different capacities per rank on purpose: the padding is a local matter
            ts = PaddedTrainStep(model, inputs["Z"], inputs["N"], E + 20 + 8 * rank, T + 60 + 10 * rank, max_in_degree=64,
                                 n_groups=2, a_cap=int(inputs["Z"].shape[0]) + 5 * rank, world_size=world)
            ts.inputs["R"] = ts.pad.inputs["R"] = ts.pad.inputs["R"].double()
            ts.pad._R_fill = ts.pad._R_fill.double()
            ts.inputs["R"][ts.pad.a_cap:] = dummy_positions(2, ts.inputs["R"], offset=60.0)
            loss = ts.step(inputs["R"], idx, targets["E"], targets["F"], Z=inputs["Z"], N=inputs["N"], step_optimizer=False)
        lt = loss.clone()
        dist.all_reduce(lt)
        np.save(os.path.join(out_dir, f"pgrad_{rank}.npy"), ts.buf.flat.numpy())
        np.save(os.path.join(out_dir, f"ploss_{rank}.npy"), np.array(float(lt)))
    finally:
        dist.destroy_process_group()


def test_two_rank_padded_step_equals_single_process(tmp_path):
    """PaddedTrainStep under world_size 2 (gloo): each rank pads its own shard (to its own capacities); the all-reduced
    gradient and the summed loss equal the single-process TrainStep on the union batch."""
    import cpu_kernels
    dc = _make(4)
    with cpu_kernels.emulate():
        model = _model()
        ts = TrainStep(model, world_size=1)
        inputs, targets = _batch(dc, [0, 1, 2, 3])
        loss_ref = float(ts(inputs, targets, step_optimizer=False))
        ref = ts.buf.flat.clone().numpy()
    shards = [[0, 3], [1, 2]]
    mp.spawn(_worker_padded, args=(2, _free_port(), shards, str(tmp_path)), nprocs=2, join=True)
    g0 = np.load(tmp_path / "pgrad_0.npy")
    g1 = np.load(tmp_path / "pgrad_1.npy")
    assert np.array_equal(g0, g1)
    np.testing.assert_allclose(g0, ref, rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(float(np.load(tmp_path / "ploss_0.npy")), loss_ref, rtol=1e-10)


def test_flat_buffer_views():
    lin = torch.nn.Linear(3, 2)
    buf = FlatGradBuffer(lin.parameters())
    lin(torch.ones(1, 3)).sum().backward()
    assert buf.flat.numel() == 8 and float(buf.flat.abs().sum()) > 0
    assert lin.weight.grad.data_ptr() == buf.flat.data_ptr()


# ------------------------------------------------------------------ Trainer under world_size 2 (gloo)
def _trainer_worker(rank, world, port, shards, out_dir, mve):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import cpu_kernels
    from gemnet_pytorch_amd.training.metrics import Metrics
    from gemnet_pytorch_amd.training.trainer import Trainer
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        with cpu_kernels.emulate():
            model = _model_mve() if mve else _model()
            tr = Trainer(model, learning_rate=1e-3, weight_decay=0.0, grad_clip_max=1e9, loss="rmse", mve=mve)
            tr.dict2device = lambda d, device=None: d
            dc = _make(4)

            def it():
                while True:
                    yield _batch(dc, shards[rank])

            train = Metrics("train", tr.tracked_metrics)
            val = Metrics("val", tr.tracked_metrics)
            stream = it()
            loss = float(tr.train_on_batch(stream, train))
            grads = tr._grads.flat.clone().numpy()
            vloss = float(tr.test_on_batch(stream, val))
        np.save(os.path.join(out_dir, f"tg_{rank}.npy"), grads)
        np.save(os.path.join(out_dir, f"tl_{rank}.npy"), np.array([loss, vloss]))
    finally:
        dist.destroy_process_group()


def _model_mve():
    from gemnet_pytorch_amd.model.gemnet import GemNet
    torch.manual_seed(3)
    m = GemNet(**dict(CFG, num_targets=2), scale_file=SCALE_FILE).double()
    m._check_inputs = lambda R: None
    return m


@pytest.mark.parametrize("mve", [False, True])
def test_trainer_two_ranks_equal_single_process(tmp_path, mve):
    """Trainer.train_on_batch / test_on_batch on two gloo ranks (uneven shards 3 + 1 molecules): the all-reduced
    gradient is the single-process gradient of the union batch (MAE/RMSE and Gaussian-NLL objectives alike), and the
    loss every rank returns and logs is the global loss — identical on both ranks — for training and validation."""
    import cpu_kernels
    from gemnet_pytorch_amd.training.metrics import Metrics
    from gemnet_pytorch_amd.training.trainer import Trainer
    dc = _make(4)
    with cpu_kernels.emulate():
        model = _model_mve() if mve else _model()
        tr = Trainer(model, learning_rate=1e-3, weight_decay=0.0, grad_clip_max=1e9, loss="rmse", mve=mve)
        tr.dict2device = lambda d, device=None: d

        def it():
            while True:
                yield _batch(dc, [0, 1, 2, 3])

        stream = it()
        loss_ref = float(tr.train_on_batch(stream, Metrics("train", tr.tracked_metrics)))
        grad_ref = tr._grads.flat.clone().numpy()
        vloss_ref = float(tr.test_on_batch(stream, Metrics("val", tr.tracked_metrics)))
    shards = [[0, 1, 3], [2]]
    mp.spawn(_trainer_worker, args=(2, _free_port(), shards, str(tmp_path), mve), nprocs=2, join=True)
    g0, g1 = np.load(tmp_path / "tg_0.npy"), np.load(tmp_path / "tg_1.npy")
    l0, l1 = np.load(tmp_path / "tl_0.npy"), np.load(tmp_path / "tl_1.npy")
    assert np.array_equal(g0, g1)
    np.testing.assert_allclose(g0, grad_ref, rtol=1e-8, atol=1e-12)
    np.testing.assert_allclose(l0, l1, rtol=1e-12)                      # same value on every rank
    np.testing.assert_allclose(l0, [loss_ref, vloss_ref], rtol=1e-8)    # = the loss of the union batch


def test_train_step_counts_follow_the_batch():
    """Without `global_counts`, TrainStep exchanges the molecule/atom counts every step (batches of a real loader
    vary); the counts pinned by `capture()` are used only while capturing — an eager step with the same local shape
    must still run the exchange (a per-rank shortcut would unpair the collectives)."""
    ts = TrainStep.__new__(TrainStep)
    ts.global_counts, ts._pinned_counts, ts._use_pinned, ts.world_size = None, None, False, 1
    assert ts._counts(4, 32, "cpu") == (4.0, 32.0)
    assert ts._counts(3, 20, "cpu") == (3.0, 20.0)
    ts._pinned_counts = ((4, 32), (8.0, 64.0))
    assert ts._counts(4, 32, "cpu") == (4.0, 32.0) and ts._counts(3, 20, "cpu") == (3.0, 20.0)
    ts._use_pinned = True
    assert ts._counts(4, 32, "cpu") == (8.0, 64.0)
