"""Shared by test_layers_cpu.py / test_gpu_layers.py: replay the per-layer fixtures recorded from the
REFERENCE modules by forward hooks (tests/golden/make_golden.py::run_model2 -> model2.npz, keys
`<case>.L.<module path>.in.<i>[.<j>] / .kw.<name> / .out[.<i>]`) through the same-named modules of this package.

The reference hands its layers zero-padded / transposed tensors; the adapters below restate them in this package's
CSR form: rbf_W1 (E,I,S) -> (E,S,I); padded sph (E,S,Kmax) -> per-triplet rows sph[id_reduce[t], :, Kidx[t]]."""
import numpy as np
import torch

from gemnet_pytorch_amd import ops
from gemnet_pytorch_amd.graph import GraphPlan, RowIndex, SegmentPlan

LAYER_CASES = [
    ("q1L", "mlp_cbf3"), ("q1L", "mlp_sbf4"), ("q1L", "int_blocks.0.trip_interaction.mlp_cbf"),
    ("q1L", "int_blocks.0.quad_interaction.mlp_sbf"), ("q1L", "int_blocks.0.trip_interaction"),
    ("q1L", "int_blocks.0.quad_interaction"), ("q1L", "int_blocks.0.atom_update"), ("q1L", "out_blocks.1"),
    ("q1L", "int_blocks.0"),
    ("t2s", "mlp_cbf3"), ("t2s", "int_blocks.1.trip_interaction.mlp_cbf"), ("t2s", "int_blocks.1.trip_interaction"),
    ("t2s", "int_blocks.1.atom_update"), ("t2s", "out_blocks.2"), ("t2s", "int_blocks.1"),
]


class Rec:
    """The recorded tensors of one module call."""

    def __init__(self, g, case, layer, device, dtype):
        self.pre = f"{case}.L.{layer}."
        self.g, self.device, self.dtype = g, device, dtype

    def __call__(self, key):
        v = torch.tensor(self.g[self.pre + key])
        return v.to(self.device, self.dtype) if v.is_floating_point() else v.to(self.device)

    def has(self, key):
        return self.pre + key in self.g


def per_entry_sph(sph_padded_T, id_reduce, Kidx):
    """(E,S,Kmax) transposed zero-padded harmonics of the reference -> (T,S) rows."""
    return sph_padded_T[id_reduce.long(), :, Kidx.long()].contiguous()


def run_layer(model, plan, inputs, rec, layer):
    """Call `model.<layer>` on the recorded reference inputs; returns (ours, reference) lists of tensors."""
    mod = model.get_submodule(layer)
    leaf = layer.split(".")[-1]
    if leaf in ("mlp_cbf3", "mlp_sbf4"):                       # P5 EfficientInteractionDownProjection
        rbf_env = rec("in.0.0")                                 # (S, E, R); tensor basis: rows repeated (2l+1)x
        rad = rbf_env.permute(1, 0, 2)
        if leaf == "mlp_sbf4":
            L = int(round(rad.shape[1] ** 0.5))
            rad = rad[:, [l * l for l in range(L)], :]
        out = mod(rad.contiguous())                             # (E, S, I)
        return [out], [rec("out.0").permute(0, 2, 1)]
    if leaf in ("mlp_cbf", "mlp_sbf") and "interaction" in layer:   # P4 EfficientInteractionBilinear
        rbf_W1, sphT, x_t, id_reduce, Kidx = rec("in.0.0"), rec("in.0.1"), rec("in.1"), rec("in.2"), rec("in.3")
        T = x_t.shape[0]
        sp = SegmentPlan(id_reduce, torch.arange(T, device=x_t.device), rbf_W1.shape[0], T)
        out = mod(rbf_W1.permute(0, 2, 1).contiguous(), per_entry_sph(sphT, id_reduce, Kidx), x_t.contiguous(), sp)
        return [out], [rec("out")]
    if leaf == "trip_interaction":                              # P2
        m, rbf3, rbf_W1, sphT = rec("in.0"), rec("in.1"), rec("in.2.0"), rec("in.2.1")
        sph = per_entry_sph(sphT, inputs["id3_reduce_ca"], inputs["Kidx3"])
        out = mod(m, rbf3, (rbf_W1.permute(0, 2, 1).contiguous(), sph), plan)
        return [out], [rec("out")]
    if leaf == "quad_interaction":                              # P3
        m, rbf, cbf, rbf_W1, sphT = rec("in.0"), rec("in.1"), rec("in.2"), rec("in.3.0"), rec("in.3.1")
        sph = per_entry_sph(sphT, inputs["id4_reduce_ca"], inputs["Kidx4"])
        out = mod(m, rbf, cbf, (rbf_W1.permute(0, 2, 1).contiguous(), sph), plan)
        return [out], [rec("out")]
    if leaf == "atom_update":                                   # P10
        out = mod(rec("in.0"), rec("in.1"), rec("in.2"), plan.id_a)
        return [out], [rec("out")]
    if layer.startswith("out_blocks."):                         # P10 + P13 head
        E, F = mod(rec("in.0"), rec("in.1"), rec("in.2"), plan.id_a)
        ours, ref = [E], [rec("out.0")]
        if rec.has("out.1"):
            ours.append(F), ref.append(rec("out.1"))
        return ours, ref
    if layer.startswith("int_blocks.") and layer.count(".") == 1:   # P1 whole InteractionBlock[TripletsOnly]
        kw = dict(h=rec("kw.h"), m=rec("kw.m"), rbf3=rec("kw.rbf3"), rbf_h=rec("kw.rbf_h"), plan=plan,
                  rbf4=None, cbf4=None, sbf4=None)
        sph3 = per_entry_sph(rec("kw.cbf3.1"), inputs["id3_reduce_ca"], inputs["Kidx3"])
        kw["cbf3"] = (rec("kw.cbf3.0").permute(0, 2, 1).contiguous(), sph3)
        if rec.has("kw.sbf4.0"):
            sph4 = per_entry_sph(rec("kw.sbf4.1"), inputs["id4_reduce_ca"], inputs["Kidx4"])
            kw.update(rbf4=rec("kw.rbf4"), cbf4=rec("kw.cbf4"),
                      sbf4=(rec("kw.sbf4.0").permute(0, 2, 1).contiguous(), sph4))
        h, m = mod(**kw)
        return [h, m], [rec("out.0"), rec("out.1")]
    raise KeyError(layer)


def replay(model, g, case, layer, inputs, device, dtype, fused):
    """Run one recorded layer call in the fused first-order mode (what eval / inference uses: single-launch Dense,
    LDS-resident stacks, fused bilinear) or the composite mode (what force training differentiates twice)."""
    plan = GraphPlan.from_inputs(inputs, model.triplets_only)
    rec = Rec(g, case, layer, device, dtype)
    with torch.no_grad(), ops.weight_cache({}), ops.fused_first_order(fused), ops.param_grads(not fused):
        ours, ref = run_layer(model, plan, inputs, rec, layer)
    return ours, ref
