"""hipGraph replay vs eager, bit for bit, for one full-size model under A/B switches (the T 8x64 mismatch of mode "h3").
    PYTHONPATH=. python tools/exp/t_graph_race.py            all variants, each in its own process
    PYTHONPATH=. python tools/exp/t_graph_race.py <variant>"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
VARIANTS = {
    "base": {},
    "split6": {"GEMNET_CHAIN_MODE": "split6"},
    "no_overlap": {"T_NO_OVERLAP": "1"},
    "no_out_fuse": {"GEMNET_OUT_FUSE": "0"},
    "no_aggregate": {"GEMNET_AGGREGATE": "0"},
    "no_fork": {"T_NO_FORK": "1"},
    "rbf_out_on_side": {"T_RBF_OUT_SIDE": "1"},
    "32x32": {"T_SIZE": "32x32"},
    "16x48": {"T_SIZE": "16x48"},
    "4x96": {"T_SIZE": "4x96"},
    "64x16": {"T_SIZE": "64x16"},
    "2x128": {"T_SIZE": "2x128"},
    "12x40_split6": {"T_SIZE": "12x40", "GEMNET_CHAIN_MODE": "split6"},
    "12x40": {"T_SIZE": "12x40"},
}
if len(sys.argv) == 1:
    for name, env in VARIANTS.items():
        e = dict(os.environ, **env)
        p = subprocess.run([sys.executable, __file__, name], env=e, capture_output=True, text=True, timeout=600)
        print(p.stdout.strip() or p.stderr[-1500:], flush=True)
    sys.exit(0)

sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
from conftest import SCALE_FILE  # noqa: E402
from gemnet_pytorch_amd import kernels as K  # noqa: E402
from gemnet_pytorch_amd.model.gemnet import GemNet  # noqa: E402
from gemnet_pytorch_amd.synthetic import make_dataset  # noqa: E402
from test_gpu_fullsize import FULL, batch_of  # noqa: E402

KEEP = []
if os.environ.get("T_KEEP"):
    names = ["rbf_aggregate_fwd", "rbf_aggregate_bwd"] if os.environ["T_KEEP"] == "agg" else \
        ["rbf_aggregate_fwd", "rbf_aggregate_bwd", "chain", "gemm", "segsum", "gather", "bil_fused_fwd", "bil_fused_bwd",
         "bil_reduce_t", "bil_dy_multi", "trip_basis_fwd", "trip_basis_bwd", "edge_basis_fwd", "edge_basis_bwd", "segsum_multi"]
    for n in names:
        f = getattr(K, n)

        def w(*a, _f=f, **k):
            out = _f(*a, **k)
            KEEP.append((a, k, out))       # every operand and result of these launches stays allocated
            return out
        setattr(K, n, w)
if os.environ.get("T_NO_FORK"):
    from gemnet_pytorch_amd import ops as _ops
    _isf = _ops.is_fused
dev = "cuda"
nm, na = (int(v) for v in os.environ.get("T_SIZE", "8x64").split("x"))
cfg = dict(FULL, triplets_only=True)
torch.manual_seed(11)
model = GemNet(**cfg, scale_file=SCALE_FILE).to(dev).eval()
model.requires_grad_(False)
if os.environ.get("T_NO_OVERLAP"):
    model.overlap_output_blocks = False
if os.environ.get("T_NO_FORK") or os.environ.get("T_RBF_OUT_SIDE"):
    import gemnet_pytorch_amd.model.gemnet as _G
    _src = open(_G.__file__).read()
    if os.environ.get("T_NO_FORK"):
        _src = _src.replace("fork = side is not None and ops.is_fused() and T", "fork = False")
    else:   # the form before the fix: the output blocks' radial projection produced on the side stream
        _a = _src.index("                rbf_out = self.mlp_rbf_out(rbf)\n                if _RBF_OUT_ACC:")
        _src = (_src[:_a] + "                with torch.cuda.stream(side):\n                    rbf_out = self.mlp_rbf_out(rbf)\n"
                + "                    ev_b = torch.cuda.Event(); ev_b.record(side)\n" + _src[_a + len("                rbf_out = self.mlp_rbf_out(rbf)\n"):])
    exec(compile(_src, _G.__file__, "exec"), _G.__dict__)
    model.__class__ = _G.GemNet
ds = make_dataset(nm, na, config=2)
inputs = batch_of(ds, range(nm), True)
for _ in range(3):
    E0, F0 = model(inputs)
torch.cuda.synchronize()
E0, F0 = E0.detach().clone(), F0.detach().clone()
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    model(inputs)
torch.cuda.current_stream().wait_stream(side)
graph = torch.cuda.CUDAGraph()
with torch.cuda.graph(graph):
    Eg, Fg = model(inputs)
worst, nbad = 0.0, 0
for rep in range(10):
    graph.replay()
    torch.cuda.synchronize()
    d = float((Fg - F0).abs().max())
    worst, nbad = max(worst, d), nbad + int(d != 0.0)
E1, F1 = model(inputs)
print(f"{sys.argv[1]:14s} [{K.CHAIN_MODE}] {nm}x{na}: graph replays differing from eager: {nbad}/10, worst max|dF| = {worst:.3e}; "
      f"eager again vs eager {float((F1 - F0).abs().max()):.3e}; mean|F| {float(F0.abs().mean()):.3f}")
