"""Which step of bench.extra_train_dynamic's padded loop trips the range flag, and what is non-finite there?"""
import torch
import bench as B
from gemnet_pytorch_amd.index_device import DeviceGraphBuilder
from gemnet_pytorch_amd.model.gemnet import GemNet
from gemnet_pytorch_amd.padded import PaddedGraphRunner
from gemnet_pytorch_amd.synthetic import make_dataset
from gemnet_pytorch_amd.training.ddp import PaddedTrainStep

dev = torch.device("cuda")
cfg = dict(B.GEMNET_T)
g = torch.Generator().manual_seed(1)
data = []
for b in range(4):
    ds = make_dataset(32, 32, config=2, first=(b + 1) * 32)
    R = torch.tensor(ds["R"], device=dev, dtype=torch.float32)
    idx = DeviceGraphBuilder(ds["N"], cfg["cutoff"], cfg["int_cutoff"], True, device=dev)(R)
    data.append(dict(R=R, Z=torch.tensor(ds["Z"], device=dev).long(), N=torch.tensor(ds["N"], device=dev).long(), idx=idx,
                     E=torch.randn(32, 1, generator=g).to(dev), F=torch.randn(32 * 32, 3, generator=g).to(dev)))
sizes = [(int(d["idx"]["id_c"].shape[0]), int(d["idx"]["id3_reduce_ca"].shape[0])) for d in data]
torch.manual_seed(1234)
model = GemNet(**cfg, scale_file=B.SCALE_FILE).to(dev)
ts = PaddedTrainStep(model, data[0]["Z"], data[0]["N"], *PaddedGraphRunner.suggest_capacities(sizes), fused_optimizer=True)
for i in range(14):
    d = data[i % 4]
    loss = ts.step(d["R"], d["idx"], d["E"], d["F"], Z=d["Z"])
    torch.cuda.synchronize()
    gflat = ts.buf.flat if hasattr(ts.buf, "flat") else None
    gn = float(gflat.norm()) if gflat is not None else float("nan")
    bad = int((~torch.isfinite(gflat)).sum()) if gflat is not None else -1
    print(f"step {i}: loss {float(loss):.5f} flag {ts.flag.tripped()} |g| {gn:.4e} non-finite grads {bad} mode {model.matmul_precision} "
          f"params finite {bool(torch.isfinite(ts.fused.flat_p).all())}")
    if bad > 0:
        off = 0
        for n, p in model.named_parameters():
            k = p.numel()
            seg = gflat[off:off + k]
            if not bool(torch.isfinite(seg).all()):
                print("   non-finite:", n, tuple(p.shape), int((~torch.isfinite(seg)).sum()))
            off += k
        break
