"""The fused force-training path (gemnet_pytorch_amd/ops_train.py: Dense stacks as twice-differentiable chain programs,
sweeps S1..S4) on the CPU emulation of the launchers, float64:
  * loss.backward() through the force reproduces the REFERENCE's parameter gradients (goldens: norms, +-1 probe
    projections of every parameter, stored tensors) for GemNet-T / GemNet-Q, 1, 2 and 4 blocks;
  * the composite closure (GEMNET_TRAIN2=0) still does — it remains the fallback for the f32 chain kernel;
  * the path really runs the chain programs (launch counts), and `kernels.fuse_program` leaves the adjoint programs with
    source terms unchanged in value."""
from collections import Counter

import numpy as np
import pytest
import torch

import cpu_kernels
from conftest import check_grad_probes
from oracle import gemnet_oracle as GO
from gemnet_pytorch_amd import kernels as K
from gemnet_pytorch_amd import ops
from test_model_cpu import build
from test_oracle_model import load_case


def _run(g, tag, train2, count=None):
    cfg, params, inputs = load_case(g, tag)
    old = ops.USE_TRAIN2
    ops.USE_TRAIN2 = train2
    try:
        with cpu_kernels.emulate():
            saved = {}
            if count is not None:
                for n in cpu_kernels._NAMES:
                    f = getattr(K, n)
                    saved[n] = f
                    setattr(K, n, (lambda *a, _f=f, _n=n, **k: (count.update([_n]), _f(*a, **k))[1]))
            try:
                model = build(cfg, params).train()
                inputs["R"] = inputs["R"].double()
                E, F = model(inputs)
                loss = GO.training_loss(E[:, :1], F[:, 0] if F.dim() == 3 else F,
                                        torch.tensor(g[f"{tag}.Et"]).double()[:, None], torch.tensor(g[f"{tag}.Ft"]).double())
                loss.backward()
            finally:
                for n, f in saved.items():
                    setattr(K, n, f)
    finally:
        ops.USE_TRAIN2 = old
    return model, loss


@pytest.mark.parametrize("train2", [True, False], ids=["train2", "composite"])
@pytest.mark.parametrize("tag", ["t1", "q1", "t2", "t2s", "q2s"])
def test_training_gradients_against_reference(golden_model, golden_model2, tag, train2):
    g = golden_model2 if f"{tag}.E" in golden_model2 else golden_model
    cnt = Counter()
    model, loss = _run(g, tag, train2, cnt)
    np.testing.assert_allclose(loss.item(), float(g[f"{tag}.loss"]), rtol=1e-9)
    named = dict(model.named_parameters())
    names = [str(n) for n in g[f"{tag}.grad_names"]]
    norms = np.array([0.0 if named[n].grad is None else float(named[n].grad.norm()) for n in names])
    np.testing.assert_allclose(norms, g[f"{tag}.grad_norms"], rtol=1e-7, atol=1e-12)
    check_grad_probes(g, tag, {n: named[n].grad for n in names}, rtol=1e-7)
    for n in names:
        key = f"{tag}.grad.{n}"
        if key in g:
            np.testing.assert_allclose(named[n].grad.numpy(), g[key], rtol=1e-6, atol=1e-10)
    # the stacks run as chain programs in the training form, and only there
    assert (cnt["chain"] > 0) == train2, cnt
    if train2:
        assert cnt["pm"] < 0.5 * (cnt["pm"] + cnt["chain"] * 6), cnt
    if tag.startswith("q"):
        # GemNet-Q: with the fused form the quadruplet interaction runs on the angle-form twins (ops_train._QuadAngles2 /
        # _BilinearAng2: tangent rows rebuilt in-kernel) — no (Q, 49) harmonics, no scalar reduce / dot kernels over them
        nb = cfg_blocks(g, tag)
        if train2:
            assert cnt["ylm"] == 0 and cnt["bil_reduce"] == 0 and cnt["bil_dot"] == 0, cnt
            assert cnt["quad_angles_fwd"] == 1 and cnt["quad_angles_jvp"] == 1, cnt
            assert cnt["bil_reduce_project_tan"] == nb and cnt["bil_reduce_t_tan"] == nb, cnt      # S3 / S4: one launch per block
        else:
            assert cnt["ylm"] > 0 and cnt["bil_reduce_project_tan"] == 0 and cnt["quad_angles_jvp"] == 0, cnt


def cfg_blocks(g, tag):
    cfg, _, _ = load_case(g, tag)
    return int(cfg["num_blocks"])


def test_published_gemnet_t_second_order_gradients(golden_model2):
    """t4s = the published 4-block GemNet-T configuration on a 32-atom molecule, fused training form."""
    g = golden_model2
    model, loss = _run(g, "t4s", True)
    np.testing.assert_allclose(loss.item(), float(g["t4s.loss"]), rtol=1e-9)
    named = dict(model.named_parameters())
    names = [str(n) for n in g["t4s.grad_names"]]
    check_grad_probes(g, "t4s", {n: named[n].grad for n in names}, rtol=1e-6)


def test_fused_adjoint_programs_with_sources_equal_unfused():
    """kernels.fuse_program folds SCALE ops — now also those that carry a second-order source term — into their
    producers; the fused program computes the same values (float64 interpreter)."""
    gen = torch.Generator().manual_seed(4)
    M, w = 37, 128

    def mk(*shape):
        return torch.randn(*shape, generator=gen, dtype=torch.float64)
    g, W1, W2 = mk(M, w), mk(w, w) / 11, mk(w, w) / 11
    z = [mk(M, w) for _ in range(3)]
    mu = [mk(M, w) for _ in range(3)]
    zd = [mk(M, w) for _ in range(3)]
    S = K.ChainProgram.source

    def program():
        outs = [torch.zeros(M, w, dtype=torch.float64) for _ in range(5)]
        p = K.ChainProgram(M)
        p.load(0, g)
        p.scale(0, 0, 0.7, width=w)
        p.scale(1, 0, 1.0, Z=z[2], add=S(mu[2], zd[2], d2=True), out=outs[0], width=w)
        p.gemm(W2, a_slot=1, y_slot=1)
        p.scale(1, 1, 1.0, Z=z[1], add=S(mu[1], zd[1], d2=True), out=outs[1], width=w)
        p.gemm(W1, a_slot=1, y_slot=0, res=0, beta=1.0)
        p.scale(0, 0, 0.5, out=outs[2], width=w)
        p.scale(1, 0, 1.0, Z=z[0], add=S(mu[0], zd[0], d2=True), out=outs[3], width=w)
        p.gemm(W2, a_slot=1, y_slot=-1, out=outs[4])
        return p, outs
    p0, o0 = program()
    cpu_kernels.chain(p0, mode="split6")      # programs with source terms run in the loss-scaled mode (kernels.linear_mode)
    p1, o1 = program()
    fused = K.fuse_program(p1)
    assert len(fused.ops) < len(p1.ops)
    assert any(o.get("add") is not None or o.get("add2") is not None for o in fused.ops if o["kind"] != "scale")
    cpu_kernels.chain(fused, mode="split6")
    for a, b in zip(o0, o1):
        torch.testing.assert_close(a, b, rtol=1e-12, atol=1e-12)


@pytest.mark.parametrize("tag", ["t2", "t1"])
def test_position_gradient_of_the_force_loss(golden_model, tag):
    """d loss / d R through the force (the second-order POSITION terms of the fused geometry ops, ops_train._Dist2B /
    _Angle2B: d/dR [J^T g] dR by dual numbers on the GPU) equals the composite closure's; the caller's leaf R is used as
    it is when it requires grad (as the reference does, gemnet.py:494)."""
    g = golden_model
    grads = {}
    for train2 in (True, False):
        cfg, params, inputs = load_case(g, tag)
        old = ops.USE_TRAIN2
        ops.USE_TRAIN2 = train2
        try:
            with cpu_kernels.emulate():
                model = build(cfg, params).train()
                R = inputs["R"].double().requires_grad_(True)
                inputs["R"] = R
                E, F = model(inputs)
                loss = GO.training_loss(E[:, :1], F, torch.tensor(g[f"{tag}.Et"]).double()[:, None], torch.tensor(g[f"{tag}.Ft"]).double())
                loss.backward()
                grads[train2] = R.grad.clone()
        finally:
            ops.USE_TRAIN2 = old
    assert float(grads[False].abs().max()) > 0
    torch.testing.assert_close(grads[True], grads[False], rtol=1e-8, atol=1e-10 * float(grads[False].abs().max()))
    # (GemNet-Q is not in this list: d loss / d R through the force is NaN there on the composite closure already — the
    #  second derivative of the clamped |u x v| of its collinear intermediate triplets; a caller whose R requires grad is still
    #  routed to the composite closure for the quadruplet geometry, ops.position_graph: the fused twins carry no Hessian of
    #  the two angles)


def test_multi_target_models_train_on_the_composite_closure(golden_model2):
    """num_targets = 2: one force pass per target over a shared graph (gemnet.py:599-609).  The fused training form keeps
    ONE record of its sweeps per stack, so such models must not take it: the step runs on the composite closure and the
    parameter gradients equal the ones with the fused form switched off."""
    g = golden_model2
    grads = {}
    for train2 in (True, False):
        cfg, params, inputs = load_case(g, "t1m")
        old = ops.USE_TRAIN2
        ops.USE_TRAIN2 = train2
        cnt = Counter()
        try:
            with cpu_kernels.emulate():
                f = K.chain
                K.chain = lambda *a, _f=f, **k: (cnt.update(["chain"]), _f(*a, **k))[1]
                try:
                    model = build(cfg, params).train()
                    inputs["R"] = inputs["R"].double()
                    E, F = model(inputs)
                    assert F.shape[1] == 2
                    ((E ** 2).sum() + (F ** 2).sum()).backward()
                finally:
                    K.chain = f
        finally:
            ops.USE_TRAIN2 = old
        assert cnt["chain"] == 0
        grads[train2] = {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}
    for n in grads[False]:
        torch.testing.assert_close(grads[True][n], grads[False][n], rtol=1e-10, atol=1e-12)


@pytest.mark.parametrize("tag", ["t2s", "q2s"])
def test_eager_training_steps_free_their_records_without_the_cyclic_collector(golden_model2, tag):
    """The records shared by the four sweeps hold cotangents whose autograd history leads back — through C++ edges the cyclic
    garbage collector cannot follow — to the node that owns the record; and a stack with tail projections used to record
    its own output.  Unless the final sweep drops the former and the record holds an alias of the latter, an eager training
    step's tensors stay alive for good (without the release: 24, 48, 72, ... live records) or until the collector's next
    pass (~1 GiB per step at B = 32).  With the collector OFF, nothing of a step may survive it."""
    import gc
    from gemnet_pytorch_amd import ops_train
    g = golden_model2
    cfg, params, inputs = load_case(g, tag)
    with cpu_kernels.emulate():
        model = build(cfg, params).train()
        inputs["R"] = inputs["R"].double()
        Et, Ft = torch.tensor(g[f"{tag}.Et"]).double()[:, None], torch.tensor(g[f"{tag}.Ft"]).double()

        def step():
            model.zero_grad(set_to_none=True)
            E, F = model(inputs)
            GO.training_loss(E[:, :1], F, Et, Ft).backward()

        step()
        gc.collect()
        gc.collect()
        was_on = gc.isenabled()
        gc.disable()
        try:
            for _ in range(3):
                step()
            live = sum(isinstance(o, ops_train._Rec) for o in gc.get_objects())
            garbage = gc.collect()
        finally:
            if was_on:
                gc.enable()
    assert live == 0, live
    assert garbage == 0, garbage
    # ... and a train-mode forward whose outputs are dropped WITHOUT a backward (the final sweep never runs) frees them too:
    # the records keep values, not autograd history (cotangents and returned adjoints are stored as detached aliases)
    with cpu_kernels.emulate():
        for _ in range(2):
            E, F = model(inputs)
            del E, F
        gc.collect()
        gc.collect()
        assert sum(isinstance(o, ops_train._Rec) for o in gc.get_objects()) == 0
