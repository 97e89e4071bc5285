"""Index construction time for the headline batch: device builder (csrc/index_gpu.hip) vs host builder."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gemnet_pytorch_amd.synthetic import make_dataset
from gemnet_pytorch_amd.index_device import DeviceGraphBuilder
from gemnet_pytorch_amd.training.data_container import build_indices

for n_atoms, B in ((32, 32), (64, 8)):
    data = make_dataset(B, n_atoms=n_atoms)
    R = torch.tensor(data["R"], device="cuda")
    for to in (True, False):
        bld = DeviceGraphBuilder(data["N"], 5.0, 10.0, to)
        out = bld(R, dtype=torch.int32); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            out = bld(R, dtype=torch.int32)
        torch.cuda.synchronize()
        tg = (time.perf_counter() - t0) / 10
        t0 = time.perf_counter()
        ref = build_indices(data["R"], data["N"], 5.0, 10.0, to)
        th = time.perf_counter() - t0
        same = all(np.array_equal(out[k].cpu().numpy(), ref[k]) for k in ref)
        sizes = {k: int(out[k].numel()) for k in ("id_a", "id3_reduce_ca") + (() if to else ("id4_reduce_ca",))}
        print(f"B={B} x {n_atoms} atoms {'T' if to else 'Q'}: device {tg*1e3:.3f} ms  host {th*1e3:.1f} ms  identical={same}  {sizes}", flush=True)
