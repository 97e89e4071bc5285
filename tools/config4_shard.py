"""BASELINE.json configs[4], one GPU's shard: GemNet-Q, 64 molecules x 64 atoms (batch 512 over 8 GPUs), forward+force,
Dense stacks with plain bf16 MFMA operands (`model.matmul_precision = "bf16"`, fp32 accumulate, fp32 everywhere else)
against the default arithmetic (six split-bf16 products, fp32-equivalent) on the same batch.  Reports sizes, ms/step of
both and the force / energy deviation of the bf16 run — as measured (the reference's own bf16 autocast is at 1e-2,
SURVEY.md section 7).  Runs on the GPU box:  python tools/config4_shard.py [n_mol] [n_atoms]"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from gemnet_pytorch_amd import kernels as K  # noqa: E402
from gemnet_pytorch_amd.graph import GraphPlan  # noqa: E402
from gemnet_pytorch_amd.index_device import DeviceGraphBuilder  # noqa: E402
from gemnet_pytorch_amd.model.gemnet import GemNet  # noqa: E402
from gemnet_pytorch_amd.synthetic import make_dataset  # noqa: E402

n_mol = int(sys.argv[1]) if len(sys.argv) > 1 else 64
n_atoms = int(sys.argv[2]) if len(sys.argv) > 2 else 64
dev = torch.device("cuda", 0)
cfg = dict(bench.GEMNET_T, triplets_only=False)
torch.manual_seed(1234)
model = GemNet(**cfg, scale_file=bench.SCALE_FILE).to(dev).eval()
model.requires_grad_(False)
t0 = time.time()
ds = make_dataset(n_mol, n_atoms, config=4)
R = torch.tensor(ds["R"], device=dev)
builder = DeviceGraphBuilder(ds["N"], cfg["cutoff"], cfg["int_cutoff"], False, device=dev)
idx = builder(R)
inputs = dict(Z=torch.tensor(ds["Z"], device=dev).long(), R=R, N=torch.tensor(ds["N"], device=dev).long(), **idx)
plan = GraphPlan.from_inputs(inputs, False).warm()
torch.cuda.synchronize()
sizes = dict(atoms=plan.n_atoms, edges=plan.n_edges, triplets=plan.trip.size, interaction_edges=plan.n_int,
             intermediate_triplets=plan.n_intm, quadruplets=plan.quad.size)
print("[config4] sizes", sizes, f"(index build + plan {time.time() - t0:.1f} s)", flush=True)
# unit forces: scale the output heads so that mean|F| = 1 eV/A (forces are linear in them)
E, F = model(inputs)
s = 1.0 / float(F.abs().mean())
with torch.no_grad():
    for ob in model.out_blocks:
        ob.out_energy.weight.mul_(s)
model._wcache.clear()
res = {}
model._experimental_precision = True    # kernel-level experiment (not a model option since round 5)
for mode in (None, "bf16"):
    model.matmul_precision = mode
    for _ in range(2):
        E, F = model(inputs)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    steps = 3
    for _ in range(steps):
        E, F = model(inputs)
    torch.cuda.synchronize()
    res[mode or "default"] = dict(ms_per_step=(time.perf_counter() - t1) / steps * 1e3, E=E.detach().clone(), F=F.detach().clone())
    print(f"[config4] {mode or K.DEFAULT_CHAIN_MODE + ' (default)'}: {res[mode or 'default']['ms_per_step']:.1f} ms/step (eager), "
          f"peak memory {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB", flush=True)
ref, b = res["default"], res["bf16"]
out = dict(config="GemNet-Q, %d molecules x %d atoms, forward+force, 1 GPU (shard of BASELINE configs[4])" % (n_mol, n_atoms),
           per_gpu=sizes, ms_per_step={k: round(v["ms_per_step"], 2) for k, v in res.items()},
           molecules_per_s={k: round(n_mol / v["ms_per_step"] * 1e3, 1) for k, v in res.items()},
           bf16_vs_default=dict(force_mae_eV_per_A=float((b["F"] - ref["F"]).abs().mean()),
                                force_max_abs=float((b["F"] - ref["F"]).abs().max()),
                                mean_abs_force=float(ref["F"].abs().mean()),
                                energy_max_abs=float((b["E"] - ref["E"]).abs().max()),
                                max_abs_energy=float(ref["E"].abs().max())),
           peak_memory_gib=round(torch.cuda.max_memory_allocated() / 2**30, 1),
           note="bf16 = Dense stacks with bf16 MFMA operands, fp32 accumulate; default = kernels.DEFAULT_CHAIN_MODE (two fp16 planes, three products). "
                "The default run itself is covered by the golden / property tests; forces scaled to mean|F| = 1 eV/A.")
print(json.dumps(out))
