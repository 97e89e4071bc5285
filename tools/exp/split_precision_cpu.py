"""CPU experiment (no GPU): force error of GemNet-T when every Dense contraction runs as a split-bf16 product
(operands split into 2 or 3 bf16 planes, products accumulated in fp32), against the float64 run of the same model.
Decides which operand split the bf16-MFMA chain/GEMM kernels may use under the 1e-5 eV/A force bar."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import cpu_kernels
import gemnet_pytorch_amd.kernels as K
from oracle import gemnet_oracle as GO
from gemnet_pytorch_amd.model.gemnet import GemNet
from gemnet_pytorch_amd.synthetic import make_dataset
from gemnet_pytorch_amd.training.data_container import DataContainer

SCALE_FILE = os.path.join(ROOT, "gemnet_pytorch_amd", "scaling_factors.json")
MODE = {"planes": 0, "terms": 0}


def split(x, n):
    out, r = [], x
    for _ in range(n):
        h = r.to(torch.bfloat16).to(torch.float32)
        out.append(h); r = r - h
    return out


def smm(a, b):
    """a @ b with both operands split into MODE['planes'] bf16 planes; keep the MODE['terms'] largest cross terms."""
    if MODE["planes"] == 0 or a.dtype != torch.float32:
        return a @ b
    A, B = split(a, MODE["planes"]), split(b, MODE["planes"])
    pairs = sorted(((i + j, i, j) for i in range(len(A)) for j in range(len(B))))[:MODE["terms"]]
    acc = None
    for _, i, j in reversed(pairs):      # small terms first
        t = A[i] @ B[j]
        acc = t if acc is None else acc + t
    return acc


class _MM:
    """wraps a tensor so `x @ y` inside the emulated launchers goes through smm"""


def patched_emulation():
    import types
    src_gemm, src_chain, src_fused = cpu_kernels.gemm, cpu_kernels.chain, cpu_kernels.bil_fused_fwd
    # re-exec the three emulation functions with `@` replaced by smm(...)
    import inspect, re
    ns = dict(cpu_kernels.__dict__); ns["smm"] = smm
    code = inspect.getsource(src_gemm).replace("z = a @ b", "z = smm(a, b)")
    exec(code, ns); g = ns["gemm"]
    code = inspect.getsource(src_chain).replace('z = slots[o["a_slot"]][:, :Kd] @ W.t()', 'z = smm(slots[o["a_slot"]][:, :Kd], W.t())')
    exec(code, ns); c = ns["chain"]
    code = inspect.getsource(src_fused).replace("(P.reshape(P.shape[0], -1) @ W2T.t())", "smm(P.reshape(P.shape[0], -1), W2T.t())")
    exec(code, ns); f = ns["bil_fused_fwd"]
    return g, c, f


def main():
    n_mol, n_atoms = int(sys.argv[1]) if len(sys.argv) > 1 else 2, 32
    cfg = dict(num_spherical=7, num_radial=6, num_blocks=4, emb_size_atom=128, emb_size_edge=128, emb_size_trip=64,
               emb_size_quad=32, emb_size_rbf=16, emb_size_cbf=16, emb_size_sbf=32, emb_size_bil_trip=64,
               emb_size_bil_quad=32, num_before_skip=1, num_after_skip=1, num_concat=1, num_atom=2, triplets_only=True)
    ds = make_dataset(n_mol, n_atoms, config=2, first=0)
    dc = DataContainer.from_arrays(ds, 5.0, 10.0, triplets_only=True)
    batch = dc[list(range(n_mol))]
    inputs = {k: v for k, v in batch.items() if k not in ("E", "F")}
    params = GO.make_params(cfg, 5, GO.load_scale_factors(SCALE_FILE), dtype=torch.float64)

    def run(dtype):
        model = GemNet(**cfg, scale_file=SCALE_FILE)
        model.load_state_dict(GO.expand_to_reference_state_dict(params), strict=True)
        model = model.to(dtype).eval(); model._check_inputs = lambda R: None
        inp = dict(inputs); inp["R"] = inputs["R"].to(dtype)
        E, F = model(inp)
        return E.detach().double(), F.detach().double()

    g, c, f = patched_emulation()
    with cpu_kernels.emulate():
        K.gemm, K.chain, K.bil_fused_fwd = g, c, f
        MODE.update(planes=0, terms=0)
        E64, F64 = run(torch.float64)
        s = 1.0 / float(F64.abs().mean())   # forces are linear in the output heads: rescale to mean|F| = 1
        print(f"mean|F| raw {1/s:.3e}; errors below are for mean|F| = 1 (eV/A), energies scaled alike")
        for name, planes, terms in (("fp32", 0, 0), ("bf16x1", 1, 1), ("bf16x3 (2 planes, 3 terms)", 2, 3),
                                    ("bf16x4 (2 planes, 4 terms)", 2, 4), ("bf16x6 (3 planes, 6 terms)", 3, 6),
                                    ("bf16x9", 3, 9)):
            MODE.update(planes=planes, terms=terms)
            E, F = run(torch.float32)
            print(f"{name:30s} force MAE {float((F - F64).abs().mean()) * s:.3e}  max {float((F - F64).abs().max()) * s:.3e}"
                  f"   E err {float((E - E64).abs().max()) * s:.3e}")


if __name__ == "__main__":
    main()
