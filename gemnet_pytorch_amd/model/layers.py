"""Layer modules of the MI355X-native GemNet.

Module/parameter NAMES mirror the reference so `state_dict()` has the reference's key set
(including the aliased duplicates `Dense.weight` == `Dense.linear.weight`
and `OutputBlock.seq_energy` == `.layers`) and `pretrained/*/model.pth` style checkpoints load.
The COMPUTE is different: every forward goes through the HIP ops of `gemnet_pytorch_amd.ops`
(f32-MFMA GEMMs, CSR-segmented gather/reduce kernels, closed-form basis kernels); nothing here
builds the reference's zero-padded (E,Kmax,.) tensors or calls sympy/torch_scatter.

Reference counterparts (file:line under /root/reference/gemnet/model/layers):
  Dense/ScaledSiLU/ResidualLayer  base_layers.py:5-89
  AtomEmbedding/EdgeEmbedding     embedding_block.py:7-75
  AtomUpdateBlock/OutputBlock     atom_update_block.py:9-193
  EfficientInteractionDownProjection/Bilinear  efficient.py:5-57,120-189
  BesselBasisLayer/SphericalBasisLayer/TensorBasisLayer  basis_layers.py:9-295
  TripletInteraction/QuadrupletInteraction/InteractionBlock(TripletsOnly)  interaction_block.py:11-696
"""
import math
import os

import numpy as np
import torch
from scipy import special as _sp
from scipy.optimize import brentq as _brentq

from .. import ops
from .initializers import he_orthogonal_init
from .scaling import AutomaticFit, ScalingFactor

INV_SQRT_2 = 1 / (2.0 ** 0.5)
INV_SQRT_3 = 1 / (3.0 ** 0.5)


# ------------------------------------------------------------------------------ dense stack
class Dense(torch.nn.Module):
    """Bias-free linear layer + optional ScaledSiLU; y = ssilu(x W^T)."""

    def __init__(self, in_features, out_features, bias=False, activation=None, name=None):
        super().__init__()
        self.linear = torch.nn.Linear(in_features, out_features, bias=bias)
        he_orthogonal_init(self.linear.weight)
        if self.linear.bias is not None:
            self.linear.bias.data.fill_(0)
        self.weight = self.linear.weight  # alias: both keys appear in state_dict like the reference
        self.bias = self.linear.bias
        if isinstance(activation, str):
            activation = activation.lower()
        if activation in ("swish", "silu"):
            self.act = True
        elif activation is None:
            self.act = False
        else:
            raise NotImplementedError("Activation function not implemented for GemNet (yet).")

    def forward(self, x, **epilogue):
        """`epilogue`: optional fused stages of ops.dense (mul/alpha/res/res_rows/beta/res2/beta2/g1..)."""
        if self.bias is not None:
            y = ops.linear(x, self.weight) + self.bias
            assert not epilogue, "fused epilogues are not supported together with a bias"
            return ops.ssilu(y) if self.act else y
        return ops.dense(x, self.weight, self.act, **epilogue)


class ResidualLayer(torch.nn.Module):
    """(x + MLP(x)) / sqrt(2)."""

    def __init__(self, units: int, nLayers: int = 2, activation=None, name=None):
        super().__init__()
        self.dense_mlp = torch.nn.Sequential(
            *[Dense(units, units, activation=activation, bias=False) for _ in range(nLayers)])

    def stackable(self):
        return len(self.dense_mlp) == 2 and all(d.act and d.bias is None for d in self.dense_mlp)

    def as_stack_layer(self, skip=None, skip_beta=1.0):
        return dict(W1=self.dense_mlp[0].weight, W2=self.dense_mlp[1].weight, skip=skip, skip_beta=skip_beta)

    def forward(self, inputs, res2=None, beta2=1.0):
        """(inputs + MLP(inputs))/sqrt(2), optionally followed by (. + res2)*beta2 — both residual adds
        ride in the epilogue of the last GEMM."""
        x = inputs
        n = len(self.dense_mlp)
        for i, layer in enumerate(self.dense_mlp):
            if i + 1 < n:
                x = layer(x)
            else:
                x = layer(x, res=inputs, beta=INV_SQRT_2, res2=res2, beta2=beta2)
        return x


class AtomEmbedding(torch.nn.Module):
    def __init__(self, emb_size, name=None):
        super().__init__()
        self.emb_size = emb_size
        self.embeddings = torch.nn.Embedding(93, emb_size)
        torch.nn.init.uniform_(self.embeddings.weight, a=-np.sqrt(3), b=np.sqrt(3))

    def forward(self, z_rows):
        """z_rows: RowIndex of Z-1 into the 93-row table."""
        return ops.gather_rows(self.embeddings.weight, z_rows)


class EdgeEmbedding(torch.nn.Module):
    """Dense(cat[h[id_c], h[id_a], m]) evaluated WITHOUT materialising the concatenation:
    (h @ W_c^T)[id_c] + (h @ W_a^T)[id_a] + m @ W_m^T — the two atom terms are GEMMs over nAtoms
    rows instead of nEdges rows (embedding_block.py:60-75 does cat + one (E, 2A+F) GEMM)."""

    def __init__(self, atom_features, edge_features, out_features, activation=None, name=None):
        super().__init__()
        self.atom_features = atom_features
        self.dense = Dense(2 * atom_features + edge_features, out_features, activation=activation, bias=False)

    def atom_terms(self, h):
        """(h W_c^T, h W_a^T): the two atom-row GEMMs of forward(), callable ahead of the edge term (other stream)."""
        A = self.atom_features
        W = self.dense.weight
        return ops.dense(h, W[:, :A]), ops.dense(h, W[:, A:2 * A])

    def forward(self, h, m_rbf, id_c, id_a, terms=None):
        A = self.atom_features
        W = self.dense.weight
        g1, g2 = terms if terms is not None else self.atom_terms(h)
        return ops.dense(m_rbf, W[:, 2 * A:], self.dense.act, g1=g1, i1=id_c, g2=g2, i2=id_a)


class AtomUpdateBlock(torch.nn.Module):
    def __init__(self, emb_size_atom, emb_size_edge, emb_size_rbf, nHidden, activation=None,
                 scale_file=None, name="atom_update"):
        super().__init__()
        self.name = name
        self.emb_size_edge = emb_size_edge
        self.dense_rbf = Dense(emb_size_rbf, emb_size_edge, activation=None, bias=False)
        self.scale_sum = ScalingFactor(scale_file=scale_file, name=name + "_sum")
        self.layers = self.get_mlp(emb_size_atom, nHidden, activation)
        # one-pass Dense(rbf) (.) m -> atom sum (csrc/aggregate.hip); GEMNET_OUT_FUSE=0 keeps the GEMM + segmented-sum
        # form in the OutputBlocks (the A/B switch of docs/HISTORY.md section 9)
        self.fuse_aggregate = True

    def get_mlp(self, units, nHidden, activation):
        dense1 = Dense(self.emb_size_edge, units, activation=activation, bias=False)
        res = [ResidualLayer(units, nLayers=2, activation=activation) for _ in range(nHidden)]
        return torch.nn.ModuleList([dense1] + res)

    def _aggregate(self, m, rbf, id_a, want_x=False):
        """scale * sum_{edges into atom} m * dense_rbf(rbf)   (atom_update_block.py:60-68)."""
        if ops.is_fused() or not AutomaticFit.fitting_mode:  # Hadamard and scale in the GEMM epilogue (linear: scale before the sum)
            if self.fuse_aggregate and not want_x and self.dense_rbf.bias is None and not self.dense_rbf.act:
                fused = ops.rbf_aggregate(m, rbf, self.dense_rbf.weight, id_a, self.scale_sum.value())
                if fused is not None:        # Dense + Hadamard + segmented sum in one pass (csrc/aggregate.hip)
                    return fused, None
            x = self.dense_rbf(rbf, mul=m, alpha=self.scale_sum.value())
            return ops.segsum_rows(x, id_a), x
        x = m * self.dense_rbf(rbf)
        return self.scale_sum(m, ops.segsum_rows(x, id_a)), x

    def _mlp_stack(self, x, layers, res2=None, beta2=1.0, tails=()):
        """Dense + ResidualLayers of the atom MLP as one LDS-resident launch (ops.stack)."""
        first = dict(W=layers[0].weight, act=layers[0].act)
        res_layers = [l.as_stack_layer() for l in layers[1:]]
        if res2 is not None:
            if res_layers:
                res_layers[-1].update(skip=res2, skip_beta=beta2)
            else:
                first.update(res=res2, beta=beta2)
        return ops.stack(x, first=first, layers=res_layers, s=INV_SQRT_2, tails=tails)

    @staticmethod
    def _stackable(layers):
        return (ops.stacks_enabled() and isinstance(layers[0], Dense) and layers[0].bias is None
                and all(isinstance(l, ResidualLayer) and l.stackable() for l in layers[1:]))

    def forward(self, h, m, rbf, id_a, res2=None, beta2=1.0, tails=()):
        """`tails`: weights Wt whose projections h_new @ Wt^T are wanted too (returned after h_new)."""
        x, _ = self._aggregate(m, rbf, id_a)
        if self._stackable(self.layers):
            return self._mlp_stack(x, self.layers, res2, beta2, tails)
        assert not tails, "tail projections ride on the stacked path only"
        n = len(self.layers)
        for i, layer in enumerate(self.layers):
            if i + 1 == n and res2 is not None and isinstance(layer, ResidualLayer):
                x = layer(x, res2=res2, beta2=beta2)
                res2 = None
            else:
                x = layer(x)
        if res2 is not None:
            x = (x + res2) * beta2
        return x


class OutputBlock(AtomUpdateBlock):
    def __init__(self, emb_size_atom, emb_size_edge, emb_size_rbf, nHidden, num_targets,
                 activation=None, direct_forces=True, output_init="HeOrthogonal", scale_file=None,
                 name="output", **kwargs):
        super().__init__(name=name, emb_size_atom=emb_size_atom, emb_size_edge=emb_size_edge,
                         emb_size_rbf=emb_size_rbf, nHidden=nHidden, activation=activation,
                         scale_file=scale_file)
        assert isinstance(output_init, str)
        self.output_init = output_init
        self.direct_forces = direct_forces
        self.dense_rbf = Dense(emb_size_rbf, emb_size_edge, activation=None, bias=False)
        self.fuse_aggregate = os.environ.get("GEMNET_OUT_FUSE", "1") == "1"    # see AtomUpdateBlock.__init__
        self.seq_energy = self.layers  # alias (reference atom_update_block.py:130)
        self.out_energy = Dense(emb_size_atom, num_targets, bias=False, activation=None)
        if self.direct_forces:
            self.scale_rbf = ScalingFactor(scale_file=scale_file, name=name + "_had")
            self.seq_forces = self.get_mlp(emb_size_edge, nHidden, activation)
            self.out_forces = Dense(emb_size_edge, num_targets, bias=False, activation=None)
        self.reset_parameters()

    def reset_parameters(self):
        mode = self.output_init.lower()
        heads = [self.out_energy] + ([self.out_forces] if self.direct_forces else [])
        if mode == "heorthogonal":
            for d in heads:
                he_orthogonal_init(d.weight)
        elif mode == "zeros":
            for d in heads:
                torch.nn.init.zeros_(d.weight)
        else:
            raise UserWarning(f"Unknown output_init: {self.output_init}")

    def aggregate(self, m, rbf, id_a):
        """The edge -> atom part of forward() (atom_update_block.py:160-166), callable ahead of the rest."""
        return self._aggregate(m, rbf, id_a, want_x=self.direct_forces)

    def forward(self, h, m, rbf, id_a, agg=None, E_sum=None):
        """`agg`: the result of `aggregate(m, rbf, id_a)` when the caller computed it already (on another stream).
        `E_sum`: running sum of the energies of the earlier output blocks; returned energy = E_sum + this block's
        (added in the epilogue of the energy head instead of by one elementwise launch per block)."""
        x_E, x = agg if agg is not None else self._aggregate(m, rbf, id_a, want_x=self.direct_forces)
        if self._stackable(self.seq_energy):
            x_E = self._mlp_stack(x_E, self.seq_energy)
        else:
            for layer in self.seq_energy:
                x_E = layer(x_E)
        if E_sum is not None and ops.is_fused() and self.out_energy.bias is None:
            x_E = self.out_energy(x_E, res=E_sum)
        else:
            x_E = self.out_energy(x_E)
            if E_sum is not None:
                x_E = E_sum + x_E
        if self.direct_forces:
            if ops.is_fused() or not AutomaticFit.fitting_mode:  # x already carries scale_sum: rescale to scale_rbf
                x_F = x * (self.scale_rbf.value() / self.scale_sum.value())
            else:
                x_F = self.scale_rbf(m, x)
            for layer in self.seq_forces:
                x_F = layer(x_F)
            x_F = self.out_forces(x_F)
        else:
            x_F = 0
        return x_E, x_F


# ---------------------------------------------------------------------- efficient bilinear
class EfficientInteractionDownProjection(torch.nn.Module):
    """rbf_W1[e,s,i] = sum_r rad[e,l(s),r] W[s,r,i]  (efficient.py:41-57).

    `rad` arrives as (E, L, R) with one row per degree l.  For the circular basis s == l; for the
    tensor basis (num_spherical = L^2 slots) the reference first repeats row l (2l+1) times
    (basis_layers.py:254-256) — here the repeat is folded into a block-structured weight so the
    (E, L^2, R) tensor never exists.  Output layout is (E, S, I); the reference's (E, I, S) and its
    transposed zero-padded harmonics are not needed."""

    def __init__(self, num_spherical, num_radial, emb_size_interm, name="EfficientDownProj"):
        super().__init__()
        self.num_spherical = num_spherical
        self.num_radial = num_radial
        self.emb_size_interm = emb_size_interm
        self.weight = torch.nn.Parameter(torch.empty((num_spherical, num_radial, emb_size_interm)))
        he_orthogonal_init(self.weight)

    def forward(self, rad):
        S, R, I = self.weight.shape
        L = rad.shape[1]
        if L == S:
            sizes = [1] * L
        else:
            assert L * L == S, "tensor basis expects num_spherical**2 weight slots"
            sizes = [2 * l + 1 for l in range(L)]
        def make(W=self.weight):
            blocks, start = [], 0
            for n in sizes:  # rows (r) x cols (slot-in-l, i)
                blocks.append(W[start:start + n].permute(1, 0, 2).reshape(R, n * I))
                start += n
            return torch.block_diag(*blocks)                          # (L*R, S*I)
        # frozen weight (inference): the block-diagonal form is built once, not with 2 S + 1 small launches per step
        Wbd = ops.cached_form("bd", self.weight, lambda: make(self.weight.detach())) if ops._frozen(self.weight) else make()
        out = ops.mm(rad.reshape(-1, L * R), Wbd, False, True)        # (E, S*I)
        return out.reshape(-1, S, I)


class EfficientInteractionBilinear(torch.nn.Module):
    """out[e,o] = sum_{t in seg(e)} sum_s sum_i sum_c sph[t,s] rbfW1[e,i,s] x[g(t),c] W[c,i,o]
    (efficient.py:159-189) as K1 segmented reduce -> K2 per-edge (I,S)x(S,C) -> K3 GEMM."""

    def __init__(self, emb_size, emb_size_interm, units_out, name="EfficientBilinear"):
        super().__init__()
        self.emb_size = emb_size
        self.emb_size_interm = emb_size_interm
        self.units_out = units_out
        self.weight = torch.nn.Parameter(torch.empty((emb_size, emb_size_interm, units_out)))
        he_orthogonal_init(self.weight)

    def forward(self, rbf_W1, sph, x, seg_plan, alpha=1.0):
        return ops.bilinear(rbf_W1, sph, x, self.weight, seg_plan, alpha)


# ------------------------------------------------------------------------------ basis layers
def _jn_zeros(n, k):
    """float32 roots of j_l (l<n), found between the float32-rounded roots of j_{l-1}
    (same bracketing scheme as the reference, basis_utils.py:14-29)."""
    zerosj = np.zeros((n, k), dtype="float32")
    zerosj[0] = np.arange(1, k + 1) * np.pi
    points = np.arange(1, k + n) * np.pi
    racines = np.zeros(k + n - 1, dtype="float32")
    for i in range(1, n):
        for j in range(k + n - 1 - i):
            racines[j] = _brentq(lambda r: _sp.spherical_jn(i, r), points[j], points[j + 1])
        points = racines
        zerosj[i][:k] = racines[:k]
    return zerosj


def _sph_normalizer(z):
    n, k = z.shape
    return np.array([[1.0 / math.sqrt(0.5 * _sp.spherical_jn(l + 1, z[l, i]) ** 2) for i in range(k)]
                     for l in range(n)], dtype=np.float64)


class BesselBasisLayer(torch.nn.Module):
    def __init__(self, num_radial, cutoff, envelope_exponent=5, name="bessel_basis"):
        super().__init__()
        self.num_radial = num_radial
        self.cutoff = float(cutoff)
        self.p = int(envelope_exponent)
        self.frequencies = torch.nn.Parameter(
            torch.tensor(np.pi * np.arange(1, num_radial + 1, dtype=np.float32)), requires_grad=True)

    def forward(self, d):
        return ops.bessel_rbf(d, self.frequencies, self.cutoff, self.p)


class _RadialTables(torch.nn.Module):
    def __init__(self, num_spherical, num_radial, cutoff, envelope_exponent):
        super().__init__()
        assert num_radial <= 64
        self.num_radial = num_radial
        self.num_spherical = num_spherical
        self.cutoff = float(cutoff)
        self.p = int(envelope_exponent)
        z = _jn_zeros(num_spherical, num_radial)
        self.register_buffer("z_ln", torch.tensor(z), persistent=False)
        self.register_buffer("n_ln", torch.tensor(_sph_normalizer(z)), persistent=False)

    def _apply(self, fn, *a, **k):
        super()._apply(fn, *a, **k)
        # .double()/.float()/.half() must not touch the tables the kernels expect
        self.z_ln = self.z_ln.to(torch.float32)
        self.n_ln = self.n_ln.to(torch.float64)
        return self

    def radial(self, d):
        return ops.sph_radial(d, self.z_ln, self.n_ln, self.cutoff, self.p)  # (E,S,R)


class SphericalBasisLayer(_RadialTables):
    """Radial (E,S,R) + Y_l0 (T,S).  efficient=True returns the pair for the bilinear layer;
    efficient=False returns the (T, S*R) product (basis_layers.py:132-144)."""

    def __init__(self, num_spherical, num_radial, cutoff, envelope_exponent=5, efficient=False,
                 name="spherical_basis"):
        super().__init__(num_spherical, num_radial, cutoff, envelope_exponent)
        self.efficient = efficient

    def forward(self, D, angle, reduce_rows=None):
        rad = self.radial(D)
        sph = ops.ylm0(angle, self.num_spherical)
        if self.efficient:
            return rad, sph
        rad_t = ops.gather_rows(rad, reduce_rows)                      # (T,S,R)
        return (rad_t * sph[:, :, None]).reshape(-1, self.num_spherical * self.num_radial)


class TensorBasisLayer(_RadialTables):
    """Radial repeated (2l+1)x -> (E,S^2,R) and real Y_lm (Q,S^2) (basis_layers.py:239-295)."""

    def __init__(self, num_spherical, num_radial, cutoff, envelope_exponent=5, efficient=False,
                 name="tensor_basis"):
        super().__init__(num_spherical, num_radial, cutoff, envelope_exponent)
        self.efficient = efficient

    def forward(self, D, theta, phi):
        # radial stays (E,S,R); the (2l+1)x repeat happens inside the down projection's weight
        return self.radial(D), ops.ylm(theta, phi, self.num_spherical)


# ------------------------------------------------------------------------ interaction blocks
def _up_pair_train(inter, x, plan):
    """(up_ca(x) + up_ac(x)[id_swap]) / sqrt2 (interaction_block.py:696-705) in the training form: each up projection one
    twice-differentiable launch (ops.dense -> ops_train.stack), the swap a row gather, the sum in the second epilogue."""
    x = ops.accumulate_gradient(x)          # two fused consumers: one running gradient
    y_sw = ops.gather_rows(inter.up_projection_ac(x), plan.id_swap)
    return inter.up_projection_ca(x, res=y_sw, beta=INV_SQRT_2)


class TripletInteraction(torch.nn.Module):
    def __init__(self, emb_size_edge, emb_size_trip, emb_size_bilinear, emb_size_rbf, emb_size_cbf,
                 activation=None, scale_file=None, name="TripletInteraction", **kwargs):
        super().__init__()
        self.name = name
        self.dense_ba = Dense(emb_size_edge, emb_size_edge, activation=activation, bias=False)
        self.mlp_rbf = Dense(emb_size_rbf, emb_size_edge, activation=None, bias=False)
        self.scale_rbf = ScalingFactor(scale_file=scale_file, name=name + "_had_rbf")
        self.mlp_cbf = EfficientInteractionBilinear(emb_size_trip, emb_size_cbf, emb_size_bilinear)
        self.scale_cbf_sum = ScalingFactor(scale_file=scale_file, name=name + "_sum_cbf")
        self.down_projection = Dense(emb_size_edge, emb_size_trip, activation=activation, bias=False)
        self.up_projection_ca = Dense(emb_size_bilinear, emb_size_edge, activation=activation, bias=False)
        self.up_projection_ac = Dense(emb_size_bilinear, emb_size_edge, activation=activation, bias=False)

    def _head_ok(self):
        ws = (self.dense_ba, self.mlp_rbf, self.down_projection)
        return (ops.stacks_enabled() and not AutomaticFit.fitting_mode
                and all(d.bias is None and d.weight.shape[0] <= 128 and d.weight.shape[1] <= 128
                        and d.weight.shape[1] % 16 == 0 for d in ws) and not self.mlp_rbf.act)

    def _pair_ok(self):
        ups = (self.up_projection_ca, self.up_projection_ac)
        return all(d.bias is None and d.weight.shape[0] % 16 == 0 and d.weight.shape[0] <= 128
                   and d.weight.shape[1] % 16 == 0 and d.weight.shape[1] <= 128 for d in ups) \
            and self.up_projection_ca.act == self.up_projection_ac.act

    def forward(self, m, rbf3, cbf3, plan, pair=False):
        """`pair`: return the two up projections as ops.SwappedPair (y_ca, y_ac, id_swap) instead of
        x3 = (y_ca + y_ac[id_swap]) / sqrt2 — the caller's stack adds them in its first epilogue."""
        rbf_W1, sph = cbf3
        if self._head_ok():  # dense_ba, radial Hadamard, down projection: one LDS-resident launch
            x_ba = ops.dense_hadamard_down(m, rbf3, self.dense_ba.weight, self.mlp_rbf.weight,
                                           self.down_projection.weight, self.dense_ba.act,
                                           self.down_projection.act, self.scale_rbf.value())
            x = self.mlp_cbf(rbf_W1, sph, x_ba, plan.trip, alpha=self.scale_cbf_sum.value())
            if pair and ops.constant_weights() and self._pair_ok() and plan.id_swap.inverse is not None:
                # both up projections in one launch (and one adjoint launch); the swap gather and the sum move into
                # the epilogue of the stack that consumes them
                y_ac, y_ca = ops.up_project_pair(x, self.up_projection_ac.weight, self.up_projection_ca.weight,
                                                 plan.id_swap, self.up_projection_ca.act, INV_SQRT_2)
                return ops.SwappedPair(y_ca, y_ac, plan.id_swap)
            if ops.train2_enabled():
                return _up_pair_train(self, x, plan)
            x = ops.accumulate_gradient(x)
            # (up_ca(x) + up_ac(x)[id_swap]) / sqrt2 with the factor on both activations (alpha): the adjoint of the
            # swapped term is then a pure row gather of the incoming gradient, no scaling pass
            return self.up_projection_ca(x, alpha=INV_SQRT_2, res=self.up_projection_ac(x, alpha=INV_SQRT_2),
                                         res_rows=plan.id_swap)
        x_ba = self.dense_ba(m)
        if ops.is_fused() or not AutomaticFit.fitting_mode:
            x_ba = self.mlp_rbf(rbf3, mul=x_ba, alpha=self.scale_rbf.value())
            x_ba = self.down_projection(x_ba)
            x = self.mlp_cbf(rbf_W1, sph, x_ba, plan.trip, alpha=self.scale_cbf_sum.value())
            # (up_ca(x) + up_ac(x)[id_swap]) / sqrt2: the swapped rows are gathered in the epilogue
            return self.up_projection_ca(x, res=self.up_projection_ac(x), res_rows=plan.id_swap, beta=INV_SQRT_2)
        x_ba = self.scale_rbf(x_ba, x_ba * self.mlp_rbf(rbf3))
        x_ba = self.down_projection(x_ba)
        # gather by id3_expand_ba is fused into the segmented reduce (no (T,C) tensor)
        x = self.mlp_cbf(rbf_W1, sph, x_ba, plan.trip)
        # the reference observes the variance of the GATHERED rows (interaction_block.py:678-682)
        x = self.scale_cbf_sum(ops.gather_rows(x_ba, plan.trip.expand) if AutomaticFit.fitting_mode else x_ba, x)
        x_ca = self.up_projection_ca(x)
        x_ac = ops.gather_rows(self.up_projection_ac(x), plan.id_swap)
        return (x_ca + x_ac) * INV_SQRT_2


class QuadrupletInteraction(torch.nn.Module):
    def __init__(self, emb_size_edge, emb_size_quad, emb_size_bilinear, emb_size_rbf, emb_size_cbf,
                 emb_size_sbf, activation=None, scale_file=None, name="QuadrupletInteraction", **kwargs):
        super().__init__()
        self.name = name
        self.dense_db = Dense(emb_size_edge, emb_size_edge, activation=activation, bias=False)
        self.mlp_rbf = Dense(emb_size_rbf, emb_size_edge, activation=None, bias=False)
        self.scale_rbf = ScalingFactor(scale_file=scale_file, name=name + "_had_rbf")
        self.mlp_cbf = Dense(emb_size_cbf, emb_size_quad, activation=None, bias=False)
        self.scale_cbf = ScalingFactor(scale_file=scale_file, name=name + "_had_cbf")
        self.mlp_sbf = EfficientInteractionBilinear(emb_size_quad, emb_size_sbf, emb_size_bilinear)
        self.scale_sbf_sum = ScalingFactor(scale_file=scale_file, name=name + "_sum_sbf")
        self.down_projection = Dense(emb_size_edge, emb_size_quad, activation=activation, bias=False)
        self.up_projection_ca = Dense(emb_size_bilinear, emb_size_edge, activation=activation, bias=False)
        self.up_projection_ac = Dense(emb_size_bilinear, emb_size_edge, activation=activation, bias=False)

    def forward(self, m, rbf, cbf, sbf, plan):
        rbf_W1, sph = sbf
        ws = (self.dense_db, self.mlp_rbf, self.down_projection)
        if (ops.stacks_enabled() and not AutomaticFit.fitting_mode and not self.mlp_rbf.act
                and all(d.bias is None and d.weight.shape[0] <= 128 and d.weight.shape[1] <= 128
                        and d.weight.shape[1] % 16 == 0 for d in ws)):
            x_db = ops.dense_hadamard_down(m, rbf, self.dense_db.weight, self.mlp_rbf.weight,
                                           self.down_projection.weight, self.dense_db.act,
                                           self.down_projection.act, self.scale_rbf.value())
            x_db = ops.gather_rows(x_db, plan.intm_db)
            x_db = self.mlp_cbf(cbf, mul=x_db, alpha=self.scale_cbf.value())
            x = ops.accumulate_gradient(self.mlp_sbf(rbf_W1, sph, x_db, plan.quad, alpha=self.scale_sbf_sum.value()))
            if ops.train2_enabled():
                return _up_pair_train(self, x, plan)
            return self.up_projection_ca(x, alpha=INV_SQRT_2, res=self.up_projection_ac(x, alpha=INV_SQRT_2),
                                         res_rows=plan.id_swap)
        x_db = self.dense_db(m)
        if ops.is_fused() or not AutomaticFit.fitting_mode:
            x_db = self.mlp_rbf(rbf, mul=x_db, alpha=self.scale_rbf.value())
            x_db = ops.gather_rows(self.down_projection(x_db), plan.intm_db)
            x_db = self.mlp_cbf(cbf, mul=x_db, alpha=self.scale_cbf.value())
            x = self.mlp_sbf(rbf_W1, sph, x_db, plan.quad, alpha=self.scale_sbf_sum.value())
            return self.up_projection_ca(x, res=self.up_projection_ac(x), res_rows=plan.id_swap, beta=INV_SQRT_2)
        x_db = self.scale_rbf(x_db, x_db * self.mlp_rbf(rbf))
        x_db = self.down_projection(x_db)
        x_db = ops.gather_rows(x_db, plan.intm_db)                 # (I, emb_quad)
        x_db = self.scale_cbf(x_db, x_db * self.mlp_cbf(cbf))
        x = self.mlp_sbf(rbf_W1, sph, x_db, plan.quad)             # gather by id4_expand_abd fused
        x = self.scale_sbf_sum(ops.gather_rows(x_db, plan.quad.expand) if AutomaticFit.fitting_mode else x_db, x)
        x_ca = self.up_projection_ca(x)
        x_ac = ops.gather_rows(self.up_projection_ac(x), plan.id_swap)
        return (x_ca + x_ac) * INV_SQRT_2


class _InteractionBase(torch.nn.Module):
    def _build_common(self, emb_size_atom, emb_size_edge, emb_size_rbf, num_before_skip, num_after_skip,
                      num_concat, num_atom, activation, scale_file, block_nr):
        self.layers_before_skip = torch.nn.ModuleList(
            [ResidualLayer(emb_size_edge, activation=activation) for _ in range(num_before_skip)])
        self.layers_after_skip = torch.nn.ModuleList(
            [ResidualLayer(emb_size_edge, activation=activation) for _ in range(num_after_skip)])
        self.atom_update = AtomUpdateBlock(emb_size_atom=emb_size_atom, emb_size_edge=emb_size_edge,
                                           emb_size_rbf=emb_size_rbf, nHidden=num_atom, activation=activation,
                                           scale_file=scale_file, name=f"AtomUpdate_{block_nr}")
        self.concat_layer = EdgeEmbedding(emb_size_atom, emb_size_edge, emb_size_edge, activation=activation)
        self.residual_m = torch.nn.ModuleList(
            [ResidualLayer(emb_size_edge, activation=activation) for _ in range(num_concat)])

    @staticmethod
    def _chain_then_skip(layers, x, skip):
        """(skip + layers(x)) / sqrt2 with the skip add folded into the last residual layer's epilogue."""
        n = len(layers)
        if n == 0:
            return (skip + x) * INV_SQRT_2
        for i, layer in enumerate(layers):
            x = layer(x, res2=skip, beta2=INV_SQRT_2) if i + 1 == n else layer(x)
        return x

    def _stack_ok(self):
        res = list(self.layers_before_skip) + list(self.layers_after_skip) + list(self.residual_m)
        return (ops.stacks_enabled() and self.dense_ca.bias is None and self.concat_layer.dense.bias is None
                and all(l.stackable() for l in res) and len(self.layers_before_skip) > 0
                and len(self.residual_m) > 0)

    def _update_stacked(self, h, m, x3, x4, rbf_h, plan):
        """Edge update as two LDS-resident stacks (+ the atom stack in between):
        {dense_ca(+x3[,x4]) -> residuals before skip (+m) -> residuals after skip} and
        {concat-Dense (atom terms gathered in its epilogue) -> residual_m (+m)}."""
        first = dict(W=self.dense_ca.weight, act=self.dense_ca.act)
        if isinstance(x3, ops.SwappedPair):   # (dense_ca(m) + y_ca + y_ac[id_swap]) / sqrt2, one gradient for the pair
            first.update(res=x3.y_ac, res_rows=x3.swap, beta=1.0, res2=x3.y_ca, beta2=INV_SQRT_2, tied=True)
        elif x4 is None:
            first.update(res=x3, beta=INV_SQRT_2)
        else:
            first.update(res=x3, beta=1.0, res2=x4, beta2=INV_SQRT_3)
        layers = [l.as_stack_layer() for l in self.layers_before_skip]
        layers[-1].update(skip=m, skip_beta=INV_SQRT_2)
        layers += [l.as_stack_layer() for l in self.layers_after_skip]
        # consumed by the atom update's aggregation and by the second stack: one running gradient (ops.accumulate_gradient)
        m = ops.accumulate_gradient(ops.stack(m, first=first, layers=layers, s=INV_SQRT_2))
        A = self.concat_layer.atom_features
        W = self.concat_layer.dense.weight
        if self.atom_update._stackable(self.atom_update.layers):
            # the two atom terms of the concat-Dense are tail projections of the atom stack (same launch)
            h, g1, g2 = self.atom_update(h, m, rbf_h, plan.id_a, res2=h, beta2=INV_SQRT_2,
                                         tails=(W[:, :A], W[:, A:2 * A]))
        else:
            h = self.atom_update(h, m, rbf_h, plan.id_a, res2=h, beta2=INV_SQRT_2)
            g1, g2 = ops.dense(h, W[:, :A]), ops.dense(h, W[:, A:2 * A])
        first = dict(W=W[:, 2 * A:], act=self.concat_layer.dense.act, g1=g1, i1=plan.id_c, g2=g2, i2=plan.id_a)
        layers = [l.as_stack_layer() for l in self.residual_m]
        layers[-1].update(skip=m, skip_beta=INV_SQRT_2)
        m = ops.stack(m, first=first, layers=layers, s=INV_SQRT_2)
        return h, m

    def _update(self, h, m, x, rbf_h, plan):
        m = self._chain_then_skip(self.layers_before_skip, x, m)
        for layer in self.layers_after_skip:
            m = layer(m)
        h = self.atom_update(h, m, rbf_h, plan.id_a, res2=h, beta2=INV_SQRT_2)
        m2 = self.concat_layer(h, m, plan.id_c, plan.id_a)
        m = self._chain_then_skip(self.residual_m, m2, m)
        return h, m


class InteractionBlockTripletsOnly(_InteractionBase):
    def __init__(self, emb_size_atom, emb_size_edge, emb_size_trip, emb_size_quad, emb_size_rbf,
                 emb_size_cbf, emb_size_bil_trip, num_before_skip, num_after_skip, num_concat, num_atom,
                 activation=None, scale_file=None, name="Interaction", **kwargs):
        super().__init__()
        self.name = name
        block_nr = name.split("_")[-1]
        self.dense_ca = Dense(emb_size_edge, emb_size_edge, activation=activation, bias=False)
        self.trip_interaction = TripletInteraction(
            emb_size_edge=emb_size_edge, emb_size_trip=emb_size_trip, emb_size_bilinear=emb_size_bil_trip,
            emb_size_rbf=emb_size_rbf, emb_size_cbf=emb_size_cbf, activation=activation,
            scale_file=scale_file, name=f"TripInteraction_{block_nr}")
        self._build_common(emb_size_atom, emb_size_edge, emb_size_rbf, num_before_skip, num_after_skip,
                           num_concat, num_atom, activation, scale_file, block_nr)

    def forward(self, h, m, rbf3, cbf3, rbf_h, plan, **kwargs):
        m = ops.accumulate_gradient(m)   # dense_ca stack + the interaction heads (+ the output block, through autograd)
        x3 = self.trip_interaction(m, rbf3, cbf3, plan, pair=self._stack_ok())
        if self._stack_ok():
            return self._update_stacked(h, m, x3, None, rbf_h, plan)
        x = self.dense_ca(m, res=x3, beta=INV_SQRT_2)
        return self._update(h, m, x, rbf_h, plan)


class InteractionBlock(_InteractionBase):
    def __init__(self, emb_size_atom, emb_size_edge, emb_size_trip, emb_size_quad, emb_size_rbf,
                 emb_size_cbf, emb_size_sbf, emb_size_bil_trip, emb_size_bil_quad, num_before_skip,
                 num_after_skip, num_concat, num_atom, activation=None, scale_file=None, name="Interaction"):
        super().__init__()
        self.name = name
        block_nr = name.split("_")[-1]
        self.dense_ca = Dense(emb_size_edge, emb_size_edge, activation=activation, bias=False)
        self.quad_interaction = QuadrupletInteraction(
            emb_size_edge=emb_size_edge, emb_size_quad=emb_size_quad, emb_size_bilinear=emb_size_bil_quad,
            emb_size_rbf=emb_size_rbf, emb_size_cbf=emb_size_cbf, emb_size_sbf=emb_size_sbf,
            activation=activation, scale_file=scale_file, name=f"QuadInteraction_{block_nr}")
        self.trip_interaction = TripletInteraction(
            emb_size_edge=emb_size_edge, emb_size_trip=emb_size_trip, emb_size_bilinear=emb_size_bil_trip,
            emb_size_rbf=emb_size_rbf, emb_size_cbf=emb_size_cbf, activation=activation,
            scale_file=scale_file, name=f"TripInteraction_{block_nr}")
        self._build_common(emb_size_atom, emb_size_edge, emb_size_rbf, num_before_skip, num_after_skip,
                           num_concat, num_atom, activation, scale_file, block_nr)

    def forward(self, h, m, rbf4, cbf4, sbf4, rbf3, cbf3, rbf_h, plan, **kwargs):
        m = ops.accumulate_gradient(m)
        x4 = self.quad_interaction(m, rbf4, cbf4, sbf4, plan)
        x3 = self.trip_interaction(m, rbf3, cbf3, plan)
        if self._stack_ok():
            return self._update_stacked(h, m, x3, x4, rbf_h, plan)
        x = self.dense_ca(m, res=x3, beta=1.0, res2=x4, beta2=INV_SQRT_3)
        return self._update(h, m, x, rbf_h, plan)
