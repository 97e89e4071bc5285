"""GPU diagnosis of the GemNet-Q bf16 + side-stream run-to-run difference (tools/exp/bf16_determinism.py): E (forward
only) vs F (forward + adjoint), under the A/B switches.   python tools/exp/q_side_race.py [variant-name]"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
VARIANTS = {
    "base": {},
    "no_grad_acc": {"GEMNET_GRAD_ACC": "0"},
    "no_out_fuse": {"GEMNET_OUT_FUSE": "0"},
    "no_aggregate": {"GEMNET_AGGREGATE": "0"},
    "serialize": {"AMD_SERIALIZE_KERNEL": "3"},
    "no_quad_angles": {"GEMNET_QUAD_ANGLES": "0"},
}
if len(sys.argv) == 1:
    for name, env in VARIANTS.items():
        e = dict(os.environ, **env)
        p = subprocess.run([sys.executable, __file__, name], env=e, capture_output=True, text=True, timeout=600)
        print(p.stdout.strip() or p.stderr[-800:], flush=True)
    sys.exit(0)

sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
from conftest import SCALE_FILE  # noqa: E402
from gemnet_pytorch_amd.model.gemnet import GemNet  # noqa: E402
from gemnet_pytorch_amd.synthetic import make_dataset  # noqa: E402
from gemnet_pytorch_amd.training.data_container import DataContainer  # noqa: E402
from test_gpu_fullsize import FULL  # noqa: E402

dev = "cuda"
cfg = dict(FULL, triplets_only=False)
torch.manual_seed(11)
model = GemNet(**cfg, scale_file=SCALE_FILE).to(dev).eval()
model.requires_grad_(False)
ds = make_dataset(8, 64, config=4)
dc = DataContainer.from_arrays(ds, 5.0, 10.0, triplets_only=False)
b = dc[list(range(8))]
inputs = {k: v.to(dev) for k, v in b.items() if k not in ("E", "F")}
model.matmul_precision = "bf16"
out = []
for rep in range(6):
    E, F = model(inputs)
    out.append((E.detach().clone(), F.detach().clone()))
torch.cuda.synchronize()
dE = max(float((e - out[0][0]).abs().max()) for e, _ in out[1:])
dF = max(float((f - out[0][1]).abs().max()) for _, f in out[1:])
dF_late = max(float((f - out[2][1]).abs().max()) for _, f in out[3:])
print(f"{sys.argv[1]:16s}: max|dE| = {dE:.3e}  max|dF| = {dF:.3e}  (runs 3..5 vs run 2: {dF_late:.3e})  mean|F| {float(out[0][1].abs().mean()):.3f}")
