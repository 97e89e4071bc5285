"""-m gpu: the arithmetic of a launch is per-call state (ABI 13: no library state; kernels.use_mode is thread-local and every
autograd node records the mode of its forward) — VERDICT r4 weak 5 / next 1a.

Two models with different `matmul_precision` (bf16 planes / fp16 planes), same weights:
  * interleaved on one thread, and run CONCURRENTLY from two Python threads on two streams, each must reproduce its solo run
    bit for bit, and every chain launch it issues — the forward's on the calling thread, the force pass's on the AUTOGRAD
    ENGINE's thread — must carry its own model's `nprod`;
  * force training: forwards interleaved, the two `loss.backward()` afterwards in the opposite order (the forwards'
    `chain_mode` blocks closed long ago): gradients equal the solo runs bit for bit."""
import threading

import pytest
import torch

from conftest import SCALE_FILE
from gemnet_pytorch_amd import _lib, ops
from gemnet_pytorch_amd import kernels as K
from gemnet_pytorch_amd.model.gemnet import GemNet
from gemnet_pytorch_amd.synthetic import make_dataset
from gemnet_pytorch_amd.training.data_container import DataContainer

pytestmark = pytest.mark.gpu
DEV = "cuda"
CFG = dict(num_spherical=7, num_radial=6, num_blocks=2, emb_size_atom=128, emb_size_edge=128, emb_size_trip=64,
           emb_size_quad=32, emb_size_rbf=16, emb_size_cbf=16, emb_size_sbf=32, emb_size_bil_trip=64,
           emb_size_bil_quad=32, num_before_skip=1, num_after_skip=1, num_concat=1, num_atom=2)
NPROD = {"split6": 6, "h3": 2}


def make(kind, precision, seed=4):
    torch.manual_seed(seed)
    m = GemNet(**dict(CFG, triplets_only=kind == "T"), scale_file=SCALE_FILE).to(DEV)
    m.matmul_precision = precision
    return m


def batch(kind, n_mol=4, n_atoms=20):
    ds = make_dataset(n_mol, n_atoms, config=2)
    dc = DataContainer.from_arrays(dict(ds), 5.0, 10.0, triplets_only=kind == "T")
    b = dc[list(range(n_mol))]
    return {k: v.to(DEV) for k, v in b.items() if k not in ("E", "F")}


class LaunchLog:
    """Wraps gn_chain_split_f32 of the loaded library: (thread name, nprod & 0xff, is it the autograd engine's thread?)."""

    def __init__(self):
        self.lib = _lib.load()
        self.orig = self.lib.gn_chain_split_f32
        self.rows = []
        self.main = threading.main_thread()

    def __enter__(self):
        def wrapped(args, nprod, stream):
            self.rows.append((threading.current_thread().name, int(nprod) & 0xff))
            return self.orig(args, nprod, stream)
        self.lib.gn_chain_split_f32 = wrapped
        return self

    def __exit__(self, *a):
        self.lib.gn_chain_split_f32 = self.orig


@pytest.mark.parametrize("kind", ["T", "Q"])
def test_interleaved_and_concurrent_models_keep_their_own_arithmetic(kind):
    inputs = batch(kind)
    A, B = make(kind, "split6").eval(), make(kind, None).eval()          # B: the process default
    assert K.DEFAULT_CHAIN_MODE == "h3"
    solo = {}
    with LaunchLog() as log:
        for name, m, want in (("A", A, 6), ("B", B, 2)):
            log.rows.clear()
            E, F = m(dict(inputs))
            torch.cuda.synchronize()
            solo[name] = (E.detach().clone(), F.detach().clone())
            threads = {t for t, _ in log.rows}
            assert {n for _, n in log.rows} == {want}, (name, set(log.rows))
            assert len(threads) >= 2, threads         # the forward's launches AND the force pass's (engine thread)
        assert not torch.equal(solo["A"][1], solo["B"][1])               # the two arithmetics do differ in the last bits
        # interleaved on one thread
        for _ in range(3):
            for name, m in (("A", A), ("B", B), ("B", B), ("A", A)):
                E, F = m(dict(inputs))
                assert torch.equal(E, solo[name][0]) and torch.equal(F, solo[name][1]), name
        # concurrently: two Python threads, two streams (the host-side passes serialise on ops.exclusive(); the kernels overlap)
        errors = []

        def worker(name, m, want):
            try:
                st = torch.cuda.Stream()
                with torch.cuda.stream(st):
                    for _ in range(6):
                        E, F = m(dict(inputs))
                        st.synchronize()
                        if not (torch.equal(E, solo[name][0]) and torch.equal(F, solo[name][1])):
                            errors.append(f"{name}: result differs from the solo run")
            except Exception as e:      # noqa: BLE001
                errors.append(f"{name}: {e!r}")

        log.rows.clear()
        ts = [threading.Thread(target=worker, args=a, name="worker-" + a[0]) for a in (("A", A, 6), ("B", B, 2))]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        assert not errors, errors
        by_thread = {}
        for t, n in log.rows:
            by_thread.setdefault(t, set()).add(n)
        assert by_thread["worker-A"] == {6} and by_thread["worker-B"] == {2}, by_thread
        # the engine's thread served both models: both arithmetics, never a third
        others = set().union(*[v for k, v in by_thread.items() if not k.startswith("worker-")])
        assert others == {2, 6}, by_thread


def test_training_backwards_after_both_forwards_run_in_their_own_arithmetic():
    inputs = batch("T")
    tgt = torch.randn(inputs["Z"].shape[0], 3, device=DEV, generator=torch.Generator(DEV).manual_seed(1))

    def grads(m):
        return torch.cat([p.grad.reshape(-1) for p in m.parameters() if p.grad is not None])

    def loss_of(m):
        E, F = m(dict(inputs))
        return ((F - tgt) ** 2).mean() + E.mean()

    solo = {}
    for name, prec in (("A", "split6"), ("B", None)):
        m = make("T", prec).train()
        with ops.position_second_order_grads(False):
            loss_of(m).backward()
        solo[name] = grads(m).clone()
    A, B = make("T", "split6").train(), make("T", None).train()
    with LaunchLog() as log:
        lA = loss_of(A)
        lB = loss_of(B)
        log.rows.clear()
        with ops.position_second_order_grads(False):
            lB.backward()                    # B first, then A: neither runs in the other's (or the default) arithmetic
            nB = {n for _, n in log.rows}
            log.rows.clear()
            lA.backward()
            nA = {n for _, n in log.rows}
    # loss-scaled sweeps of an "h3" stack run on the bf16 planes (kernels.linear_mode): 6 for both — the packed weights differ
    assert nA == {6} and nB == {NPROD[K.linear_mode("h3")]}, (nA, nB)
    assert torch.equal(grads(A), solo["A"]) and torch.equal(grads(B), solo["B"])
