"""-m gpu: H2 on the device — the REFERENCE Trainer's recorded trajectory (tests/golden/trainer.npz: four
train_on_batch steps, schedules, plateau decay, EMA evaluation; make_golden.py::golden_trainer) reproduced by
gemnet_pytorch_amd.training.Trainer with the HIP kernels in fp32."""
import ast
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, SCALE_FILE
from oracle import gemnet_oracle as GO
from gemnet_pytorch_amd.model.gemnet import GemNet
from gemnet_pytorch_amd.training.data_container import DataContainer
from gemnet_pytorch_amd.training.metrics import Metrics
from gemnet_pytorch_amd.training.trainer import Trainer

pytestmark = pytest.mark.gpu


def stream(dc, batches):
    i = 0
    while True:
        b = dc[batches[i % len(batches)]]
        yield {k: v for k, v in b.items() if k not in ("E", "F")}, {"E": b["E"], "F": b["F"]}
        i += 1


@pytest.mark.parametrize("tag", ["rmse", "mae_agc", "quad"])
def test_training_trajectory_on_the_gpu(tag):
    g = np.load(os.path.join(GOLDEN, "trainer.npz"))
    cfg, kw = ast.literal_eval(str(g[f"{tag}.cfg"])), ast.literal_eval(str(g[f"{tag}.kw"]))
    seed = int(g[f"{tag}.seed"])
    data = dict(N=g[f"{tag}.N"], Z=g[f"{tag}.Z"], R=g[f"{tag}.R"], E=g[f"{tag}.Et"], F=g[f"{tag}.Ft"])
    dc = DataContainer.from_arrays(data, 5.0, 10.0, triplets_only=cfg["triplets_only"])
    batches = [[int(i) for i in str(b).split(",")] for b in g[f"{tag}.batches"]]
    params = GO.make_params(cfg, seed, GO.load_scale_factors(SCALE_FILE))
    model = GemNet(**cfg, scale_file=SCALE_FILE)
    model.load_state_dict(GO.expand_to_reference_state_dict({k: v.float() for k, v in params.items()}), strict=True)
    model = model.to("cuda")
    trainer = Trainer(model, **kw)
    metrics = Metrics("train", trainer.tracked_metrics)
    it = stream(dc, batches)
    losses, lrs = [], []
    for _ in range(len(g[f"{tag}.losses"])):
        losses.append(float(trainer.train_on_batch(it, metrics)))
        lrs.append([s.get_last_lr()[0] for s in trainer.schedulers.wrapped])
    print(tag, "losses", losses, "reference", g[f"{tag}.losses"].tolist())
    # four optimizer steps in fp32 against the float64 reference run
    np.testing.assert_allclose(losses, g[f"{tag}.losses"], rtol=2e-3)
    np.testing.assert_allclose(lrs, g[f"{tag}.lrs"], rtol=1e-6)
    res = metrics.result(append_tag=False)
    np.testing.assert_allclose([float(res[k]) for k in sorted(res)], g[f"{tag}.metric_values"], rtol=2e-3)
    named = dict(model.named_parameters())
    names = [str(n) for n in g[f"{tag}.param_names"]]
    np.testing.assert_allclose([float(named[n].detach().norm()) for n in names], g[f"{tag}.param_norms"], rtol=1e-3)
    for v in (1.0, 1.0, 1.0):
        trainer.decay_maybe(v)
    trainer.save_variable_backups()
    trainer.load_averaged_variables()
    val_loss = float(trainer.test_on_batch(it, Metrics("val", trainer.tracked_metrics)))
    np.testing.assert_allclose(val_loss, float(g[f"{tag}.val_loss"]), rtol=2e-3)
    trainer.restore_variable_backups()
    np.testing.assert_allclose([float(named[n].detach().norm()) for n in names], g[f"{tag}.restored_norms"], rtol=1e-3)


def test_rccl_one_rank_all_reduce_inside_train_step():
    """First contact with RCCL (`nccl` backend on ROCm): a one-rank process group, the flat gradient buffer of the fused
    training step pushed through `dist.all_reduce` (TrainStep.always_reduce) — gradients and the parameters after the
    optimizer step are bit-identical to the step without the collective (a one-rank sum is the identity)."""
    import socket

    import torch.distributed as dist
    from gemnet_pytorch_amd.synthetic import make_dataset
    from gemnet_pytorch_amd.training.ddp import TrainStep
    from test_gpu_fullsize import FULL

    cfg = dict(FULL, triplets_only=True, num_blocks=2)
    ds = make_dataset(4, 12, config=7)
    dc = DataContainer.from_arrays(ds, 5.0, 10.0, triplets_only=True)
    b = dc[[0, 1, 2, 3]]
    targets = {k: b.pop(k).to("cuda") for k in ("E", "F")}
    inputs = {k: v.to("cuda") for k, v in b.items()}

    def run(with_rccl):
        torch.manual_seed(5)
        model = GemNet(**cfg, scale_file=SCALE_FILE).to("cuda")
        ts = TrainStep(model, world_size=1, fused_optimizer=True)
        ts.always_reduce = with_rccl
        losses = [float(ts(inputs, targets)) for _ in range(2)]
        torch.cuda.synchronize()
        return losses, ts.buf.flat.clone(), ts.fused.flat_p.clone()

    ref = run(False)
    created = False
    if not dist.is_initialized():
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1,
                                device_id=torch.device("cuda", torch.cuda.current_device()))
        created = True
    try:
        assert dist.get_backend() == "nccl"
        got = run(True)
        t = torch.ones(1 << 20, device="cuda")
        dist.all_reduce(t)                       # and a plain 4 MB all-reduce
        torch.cuda.synchronize()
        assert float(t.sum()) == float(1 << 20)
    finally:
        if created:
            dist.destroy_process_group()
    assert got[0] == ref[0]
    assert torch.equal(got[1], ref[1]) and torch.equal(got[2], ref[2])


@pytest.mark.parametrize("tag", ["rmse", "quad"])
def test_loader_batches_without_index_arrays_train_like_batches_with_them(tag):
    """`DataContainer(indices="device")`: the batches carry Z, R, N and the targets only; `Trainer.train_on_batch` builds the
    index arrays on the GPU (model.with_indices -> csrc/index_gpu.hip: the same arrays in the same canonical order as the host
    builder of data_container.py:244-489).  Same reports and the same parameters after three steps as with host-built arrays."""
    import copy
    g = np.load(os.path.join(GOLDEN, "trainer.npz"))
    cfg, kw = ast.literal_eval(str(g[f"{tag}.cfg"])), ast.literal_eval(str(g[f"{tag}.kw"]))
    data = dict(N=g[f"{tag}.N"], Z=g[f"{tag}.Z"], R=g[f"{tag}.R"], E=g[f"{tag}.Et"], F=g[f"{tag}.Ft"])
    batches = [[int(i) for i in str(b).split(",")] for b in g[f"{tag}.batches"]]
    params = GO.make_params(cfg, int(g[f"{tag}.seed"]), GO.load_scale_factors(SCALE_FILE))
    base = GemNet(**cfg, scale_file=SCALE_FILE)
    base.load_state_dict(GO.expand_to_reference_state_dict({k: v.float() for k, v in params.items()}), strict=True)
    out = {}
    for mode in ("host", "device"):
        dc = DataContainer.from_arrays(data, 5.0, 10.0, triplets_only=cfg["triplets_only"], indices=mode)
        b0 = dc[batches[0]]
        assert ("id_c" in b0) == (mode == "host") and {"Z", "R", "N", "E", "F"} <= set(b0)
        model = copy.deepcopy(base).to("cuda")
        trainer = Trainer(model, **kw)
        it, metrics = stream(dc, batches), Metrics("train", trainer.tracked_metrics)
        out[mode] = ([float(trainer.train_on_batch(it, metrics)) for _ in range(3)],
                     torch.cat([p.detach().reshape(-1) for p in model.parameters()]).cpu())
    assert out["host"][0] == out["device"][0], (out["host"][0], out["device"][0])
    assert torch.equal(out["host"][1], out["device"][1])


def test_device_mode_batches_use_the_containers_cutoffs_not_the_models():
    """A `DataContainer(indices="device")` built with cutoffs other than the model's: the graph built on the GPU is the one
    the host mode (and the reference, data_container.py:244-308) builds from the CONTAINER's cutoffs — same energies / forces."""
    from gemnet_pytorch_amd.synthetic import make_dataset
    cfg = dict(num_spherical=7, num_radial=6, num_blocks=1, emb_size_atom=64, emb_size_edge=64, emb_size_trip=32,
               emb_size_quad=32, emb_size_rbf=16, emb_size_cbf=16, emb_size_sbf=32, emb_size_bil_quad=32, emb_size_bil_trip=32,
               num_before_skip=1, num_after_skip=1, num_concat=1, num_atom=2, triplets_only=False)
    torch.manual_seed(3)
    model = GemNet(**cfg, scale_file=SCALE_FILE).to("cuda").eval()      # model cutoffs: 5 / 10 (the defaults)
    ds = make_dataset(3, 12, config=1)
    res = {}
    for mode in ("host", "device"):
        dc = DataContainer.from_arrays(ds, 4.0, 7.0, triplets_only=False, indices=mode)
        b = dc[[0, 1, 2]]
        inputs = {k: v.to("cuda") for k, v in b.items() if k not in ("E", "F")}
        full = model.with_indices(inputs)
        res[mode] = (int(full["id_c"].shape[0]), int(full["id4_reduce_ca"].shape[0]), *[t.detach().cpu() for t in model(inputs)])
    assert res["host"][:2] == res["device"][:2], (res["host"][:2], res["device"][:2])
    big = DataContainer.from_arrays(ds, 5.0, 10.0, triplets_only=False)[[0, 1, 2]]
    assert int(big["id_c"].shape[0]) > res["host"][0]          # (the model's own cutoffs would have given a larger graph)
    assert torch.allclose(res["host"][2], res["device"][2], rtol=0, atol=1e-5) and torch.allclose(res["host"][3], res["device"][3], rtol=0, atol=1e-5)
