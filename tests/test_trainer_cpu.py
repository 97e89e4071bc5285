"""H2 harness counterparts (Trainer / Metrics / schedules / EMA / DataProvider) against the reference's
own Trainer: tests/golden/trainer.npz holds 4 `train_on_batch` steps + plateau decay + a `test_on_batch`
with the averaged weights, produced by the REFERENCE classes in float64 (tests/golden/make_golden.py
::golden_trainer).  Here the same protocol runs through gemnet_pytorch_amd.training on the emulated
launchers (host logic only; the kernels themselves are covered by the -m gpu tests)."""
import ast
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, SCALE_FILE
from oracle import gemnet_oracle as GO
import cpu_kernels
from gemnet_pytorch_amd.model.gemnet import GemNet
from gemnet_pytorch_amd.training.data_container import DataContainer
from gemnet_pytorch_amd.training.data_provider import DataProvider
from gemnet_pytorch_amd.training.metrics import BestMetrics, Metrics
from gemnet_pytorch_amd.training.trainer import Trainer

CASES = ["rmse", "mae_agc", "quad"]


@pytest.fixture(scope="module")
def golden_trainer():
    return np.load(os.path.join(GOLDEN, "trainer.npz"))


def stream(dc, batches):
    i = 0
    while True:
        b = dc[batches[i % len(batches)]]
        inputs = {k: v for k, v in b.items() if k not in ("E", "F")}
        inputs["R"] = inputs["R"].double()
        yield inputs, {"E": b["E"].double(), "F": b["F"].double()}
        i += 1


@pytest.mark.parametrize("tag,padded", [(t, False) for t in CASES] + [("rmse", True), ("mae_agc", True)],
                         ids=CASES + ["rmse-padded", "mae_agc-padded"])
def test_training_trajectory_matches_reference_trainer(golden_trainer, tag, padded):
    """`padded`: the same protocol with `Trainer.enable_padded_graph` — every batch padded to fixed capacities (the
    captured-graph form of the step; eager on this CPU emulation) must reproduce the REFERENCE trainer's trajectory too."""
    g = golden_trainer
    cfg, kw = ast.literal_eval(str(g[f"{tag}.cfg"])), ast.literal_eval(str(g[f"{tag}.kw"]))
    seed, triplets_only = int(g[f"{tag}.seed"]), cfg["triplets_only"]
    data = dict(N=g[f"{tag}.N"], Z=g[f"{tag}.Z"], R=g[f"{tag}.R"], E=g[f"{tag}.Et"], F=g[f"{tag}.Ft"])
    dc = DataContainer.from_arrays(data, 5.0, 10.0, triplets_only=triplets_only)
    batches = [[int(i) for i in str(b).split(",")] for b in g[f"{tag}.batches"]]
    sf = GO.load_scale_factors(SCALE_FILE)
    with cpu_kernels.emulate():
        model = GemNet(**cfg, scale_file=SCALE_FILE)
        model.load_state_dict(GO.expand_to_reference_state_dict(GO.make_params(cfg, seed, sf, dtype=torch.float64)),
                              strict=True)
        model = model.double()
        model._check_inputs = lambda R: None
        trainer = Trainer(model, **kw)
        trainer.dict2device = lambda d, device=None: d  # float64 CPU emulation
        if padded:
            shapes = [dc[b] for b in batches]
            trainer.enable_padded_graph(a_cap=max(int(x["Z"].shape[0]) for x in shapes) + 3,
                                        e_cap=max(int(x["id_c"].shape[0]) for x in shapes) + 24,
                                        t_cap=max(int(x["id3_reduce_ca"].shape[0]) for x in shapes) + 80,
                                        max_in_degree=64, n_groups=8)
        metrics = Metrics("train", trainer.tracked_metrics)
        it = stream(dc, batches)
        losses, lrs = [], []
        for _ in range(len(g[f"{tag}.losses"])):
            losses.append(float(trainer.train_on_batch(it, metrics)))
            lrs.append([s.get_last_lr()[0] for s in trainer.schedulers.wrapped])
        np.testing.assert_allclose(losses, g[f"{tag}.losses"], rtol=1e-7)
        np.testing.assert_allclose(lrs, g[f"{tag}.lrs"], rtol=1e-12)
        if padded:
            assert trainer._pstep is not None and trainer._pstep.parts is not None     # the padded step really ran
            # train.ipynb saves trainer.state_dict() at every checkpoint: the padded step (model, lambdas, a hipGraph on
            # the device) must not be part of it, and loading such a state must not disturb a running trainer
            import io
            state = trainer.state_dict()
            state.pop("dict2device")     # this test's own instance-level override (a lambda)
            assert not {"_pstep", "_padded_caps", "_wgrad", "_grads", "model"} & set(state)
            blob = io.BytesIO()
            torch.save(state, blob)
            blob.seek(0)
            pstep = trainer._pstep
            trainer.load_state_dict(dict(torch.load(blob, weights_only=False), _pstep="stale", _padded_caps=None))
            assert trainer._pstep is pstep and trainer._padded_caps is not None
        res = metrics.result(append_tag=False)
        assert sorted(res) == [str(k) for k in g[f"{tag}.metric_names"]]
        np.testing.assert_allclose([float(res[k]) for k in sorted(res)], g[f"{tag}.metric_values"], rtol=1e-7)
        named = dict(model.named_parameters())
        names = [str(n) for n in g[f"{tag}.param_names"]]
        assert sorted(names) == sorted(named)
        np.testing.assert_allclose([float(named[n].detach().norm()) for n in names], g[f"{tag}.param_norms"],
                                   rtol=1e-8)
        trainable = [n for n in names if named[n].requires_grad]
        shadow = dict(zip([n for n, p in model.named_parameters() if p.requires_grad],
                          trainer.exp_decay.shadow_params))
        np.testing.assert_allclose([float(shadow[n].norm()) for n in trainable], g[f"{tag}.ema_norms"], rtol=1e-8)
        for v in (1.0, 1.0, 1.0):
            trainer.decay_maybe(v)
        trainer.save_variable_backups()
        trainer.load_averaged_variables()
        val = Metrics("val", trainer.tracked_metrics)
        val_loss = float(trainer.test_on_batch(it, val))
        np.testing.assert_allclose(val_loss, float(g[f"{tag}.val_loss"]), rtol=1e-7)
        trainer.restore_variable_backups()
        np.testing.assert_allclose([float(named[n].detach().norm()) for n in names], g[f"{tag}.restored_norms"],
                                   rtol=1e-8)


def test_trainer_state_dict_round_trip(golden_trainer):
    with cpu_kernels.emulate():
        model = GemNet(**ast.literal_eval(str(golden_trainer["rmse.cfg"])), scale_file=SCALE_FILE)
        a = Trainer(model, learning_rate=3e-3, warmup_steps=5, decay_steps=7, ema_decay=0.5)
        for _ in range(3):
            a.schedulers.step()
        a.decay_maybe(2.0)
        state = a.state_dict()
        b = Trainer(model, learning_rate=1e-3)
        b.load_state_dict(state)  # (the reference's load_state_dict raises: trainer.py:508)
    assert b.schedulers[0].get_last_lr() == a.schedulers[0].get_last_lr()
    assert b.plateau_callback.best == 2.0 and b.ema_decay == 0.5
    assert torch.equal(b.exp_decay.shadow, a.exp_decay.shadow)


def test_data_provider_split_and_batches(tmp_path):
    from gemnet_pytorch_amd.synthetic import make_dataset
    data = make_dataset(7, n_atoms=6)
    dc = DataContainer.from_arrays(data, 5.0, 10.0, triplets_only=True)
    dp = DataProvider(dc, ntrain=4, nval=2, batch_size=3, seed=1, random_split=True, shuffle=False)
    assert sorted(dp.nsamples.items()) == [("test", 1), ("train", 4), ("val", 2)]
    ids = np.concatenate([dp.idx[s] for s in ("train", "val", "test")])
    assert sorted(ids.tolist()) == list(range(7))
    it = dp.get_dataset("train")
    sizes = [int(next(it)[0]["N"].shape[0]) for _ in range(4)]
    assert sizes == [3, 1, 3, 1]  # 4 molecules in batches of 3, restarting forever
    inputs, targets = next(dp.get_dataset("val"))
    assert set(targets) == {"E", "F"} and "id3_reduce_ca" in inputs and "E" not in inputs
    dp.save_split(str(tmp_path / "split.npz"))
    saved = np.load(tmp_path / "split.npz")
    assert sorted(saved.files) == ["test", "train", "val"]


def test_best_metrics_round_trip(tmp_path):
    m = Metrics("val", ["loss", "energy_mae", "force_mae", "force_rmse"])
    m.update_state(nsamples=4, loss=torch.tensor(2.0), energy_mae=torch.tensor(1.0))
    m.update_state(nsamples=12, force_mae=torch.tensor(0.5), force_rmse=torch.tensor(0.75))
    m.update_state(nsamples=4, loss=torch.tensor(4.0), energy_mae=torch.tensor(3.0))
    assert m.loss == 3.0 and m.result()["energy_mae_val"] == 2.0
    best = BestMetrics(str(tmp_path), m)
    best.inititalize()
    assert best.loss == np.inf
    best.update(17, m)
    again = BestMetrics(str(tmp_path), m)
    again.restore()
    assert again.step == 17 and again.loss == 3.0


def test_reference_module_paths_resolve_to_native_classes():
    """The `gemnet/` namespace shims: every import the reference's callers make
    (train.ipynb, train_seml.py:12-18, fit_scaling.py:21-32, ase_calculator.py:5-6, predict.ipynb)."""
    import importlib
    wanted = {
        "gemnet.model.gemnet": ["GemNet"],
        "gemnet.model.utils": ["read_json", "write_json", "read_value_json", "update_json"],
        "gemnet.model.layers.scaling": ["AutomaticFit", "AutoScaleFit", "ScalingFactor"],
        "gemnet.training.trainer": ["Trainer"],
        "gemnet.training.metrics": ["Metrics", "BestMetrics"],
        "gemnet.training.data_container": ["DataContainer"],
        "gemnet.training.data_provider": ["DataProvider"],
        "gemnet.training.schedules": ["LinearWarmupExponentialDecay"],
        "gemnet.training.ema_decay": ["ExponentialMovingAverage"],
    }
    for mod, names in wanted.items():
        m = importlib.import_module(mod)
        for n in names:
            assert getattr(m, n).__module__.startswith("gemnet_pytorch_amd."), (mod, n)


def test_ema_checkpoint_layout_matches_reference():
    """ExponentialMovingAverage.state_dict uses the reference's keys (ema_decay.py:148-159: lists under
    `shadow_params` / `collected_params`), loads such a dict by COPY, and still reads round-1 flat checkpoints."""
    from gemnet_pytorch_amd.training.ema_decay import ExponentialMovingAverage
    torch.manual_seed(0)
    lin = torch.nn.Linear(3, 2)
    ema = ExponentialMovingAverage(lin.parameters(), 0.9)
    with torch.no_grad():
        lin.weight.add_(1.0)
    ema.update()
    ema.store()
    sd = ema.state_dict()
    assert set(sd) == {"decay", "num_updates", "shadow_params", "collected_params"}
    assert [tuple(t.shape) for t in sd["shadow_params"]] == [(2, 3), (2,)]
    assert [tuple(t.shape) for t in sd["collected_params"]] == [(2, 3), (2,)]
    # a dict in the reference's format (independent tensors) loads by copy into two different trainers
    ref_fmt = {"decay": 0.5, "num_updates": None, "shadow_params": [t.clone() for t in sd["shadow_params"]],
               "collected_params": [t.clone() for t in sd["collected_params"]]}
    a = ExponentialMovingAverage(torch.nn.Linear(3, 2).parameters(), 0.9)
    b = ExponentialMovingAverage(torch.nn.Linear(3, 2).parameters(), 0.9)
    a.load_state_dict(ref_fmt)
    b.load_state_dict(ref_fmt)
    assert a.decay == 0.5 and torch.equal(a.shadow, ema.shadow) and torch.equal(a.backup, ema.backup)
    a.shadow.add_(1.0)                                     # in-place update of one must not leak into the other
    assert torch.equal(b.shadow, ema.shadow) and torch.equal(ref_fmt["shadow_params"][0], sd["shadow_params"][0])
    b.load_state_dict({"decay": 0.9, "num_updates": None, "shadow": ema.shadow, "backup": None})   # round-1 layout
    assert torch.equal(b.shadow, ema.shadow) and b.shadow.data_ptr() != ema.shadow.data_ptr() and b.backup is None
    with pytest.raises(ValueError):
        a.load_state_dict(dict(ref_fmt, shadow_params=ref_fmt["shadow_params"][:1]))


def test_data_container_can_leave_the_index_arrays_to_the_device():
    """`DataContainer(indices="device")` (not in the reference): batches without the index arrays, everything else as before;
    the model refuses such a batch on the host (no CPU fallback) instead of guessing."""
    from gemnet_pytorch_amd.model.gemnet import GemNet
    from gemnet_pytorch_amd.synthetic import make_dataset
    from gemnet_pytorch_amd.training.data_container import DataContainer
    ds = make_dataset(3, 6, config=1)
    host = DataContainer.from_arrays(ds, 5.0, 10.0, triplets_only=True)
    dev = DataContainer.from_arrays(ds, 5.0, 10.0, triplets_only=True, indices="device")
    a, b = host[[0, 2]], dev[[0, 2]]
    assert set(b) == {"E", "N", "Z", "R", "F", "cutoffs"} and set(a) > set(b) - {"cutoffs"}
    assert b["cutoffs"].tolist() == [5.0, 10.0]        # the graph the batch stands for travels with it
    for k in set(b) - {"cutoffs"}:
        assert torch.equal(a[k], b[k]) and a[k].dtype == b[k].dtype
    with pytest.raises(ValueError):
        DataContainer.from_arrays(ds, 5.0, 10.0, triplets_only=True, indices="gpu")
    cfg = dict(num_spherical=7, num_radial=6, num_blocks=1, emb_size_atom=16, emb_size_edge=16, emb_size_trip=16,
               emb_size_quad=16, emb_size_rbf=16, emb_size_cbf=16, emb_size_sbf=16, emb_size_bil_quad=16, emb_size_bil_trip=16,
               num_before_skip=1, num_after_skip=1, num_concat=1, num_atom=1, triplets_only=True)
    from conftest import SCALE_FILE
    with pytest.raises(RuntimeError, match="HIP device only"):
        GemNet(**cfg, scale_file=SCALE_FILE)({k: v for k, v in b.items() if k not in ("E", "F")})
