"""The arithmetic mode of the Dense stacks is per-call state, not a process global (SURVEY.md §8(b)
"no global mutable state ... re-entrant because the autograd engine re-enters from its own thread").

On the device the autograd engine runs every backward on ITS thread: it inherits nothing thread-local from the thread that
ran the forward, and a training backward runs long after the forward's `chain_mode` block has closed — possibly after
another model with another `matmul_precision` has run.  Here (CPU emulation of the launchers, float64) the engine thread is
played by a real second thread: two models with different precisions, forwards interleaved on the main thread, each
`loss.backward()` on a thread of its own; every chain launch must arrive in the arithmetic of ITS model."""
import threading

import numpy as np
import torch

import cpu_kernels
import gemnet_pytorch_amd.kernels as K
from gemnet_pytorch_amd import ops
from oracle import gemnet_oracle as GO
from test_model_cpu import build
from test_oracle_model import load_case


def test_use_mode_is_thread_local():
    seen = {}

    def other():
        seen["other"] = K.current_mode()
        with K.use_mode("bf16"):
            seen["other_in"] = K.current_mode()

    with K.use_mode("split6"):
        t = threading.Thread(target=other)
        t.start()
        t.join()
        assert K.current_mode() == "split6"
    assert seen == {"other": K.DEFAULT_CHAIN_MODE, "other_in": "bf16"}
    assert K.current_mode() == K.DEFAULT_CHAIN_MODE


def test_two_models_two_precisions_backward_on_another_thread(golden_model, monkeypatch):
    g = golden_model
    monkeypatch.setattr(K, "DEFAULT_CHAIN_MODE", "h3")
    log = []
    with cpu_kernels.emulate():
        emu = K.chain

        def recording_chain(prog, mode=None):
            log.append((threading.current_thread().name, mode or K.current_mode()))
            return emu(prog, mode=mode)

        K.chain = recording_chain
        models, losses = {}, {}
        for name, prec in (("A", "bf16"), ("B", None)):       # B: the process default ("h3")
            cfg, params, inputs = load_case(g, "t2")
            m = build(cfg, params)
            m.matmul_precision = prec
            m._experimental_precision = True     # (the single-plane kernel mode: its loss-scaled sweeps differ from h3's)
            m.train()
            inputs["R"] = inputs["R"].double()
            log.clear()
            E, F = m(inputs)                                  # forward + first adjoint (S1, S2) on the main thread
            want = prec or "h3"
            assert log and {md for _, md in log} == {want}, (name, set(log))
            losses[name] = GO.training_loss(E, F, torch.tensor(g["t2.Et"]).double()[:, None], torch.tensor(g["t2.Ft"]).double())
            models[name] = m
        assert K.current_mode() == "h3"                       # the forwards' blocks have closed

        def backward(name):
            with ops.position_second_order_grads(False):
                losses[name].backward()

        # S3 / S4 of each model on a thread of its own (thread-local state = the defaults), B first: A's backward must not
        # run in B's arithmetic nor in the process default.  Loss-scaled sweeps run in kernels.linear_mode(mode).
        for name, want in (("B", K.linear_mode("h3")), ("A", K.linear_mode("bf16"))):
            log.clear()
            t = threading.Thread(target=backward, args=(name,), name="engine-" + name)
            t.start()
            t.join()
            assert log and {th for th, _ in log} == {"engine-" + name}
            assert {md for _, md in log} == {want}, (name, set(log))
    # and the gradients are the reference's (the emulation is exact in every mode: this checks the plumbing, not rounding)
    named = dict(models["A"].named_parameters())
    names = [str(n) for n in g["t2.grad_names"]]
    norms = np.array([0.0 if named[n].grad is None else float(named[n].grad.norm()) for n in names])
    np.testing.assert_allclose(norms, g["t2.grad_norms"], rtol=1e-7, atol=1e-12)
