"""First chain launch of the padded training step (fp16-plane arithmetic) whose outputs are non-finite: program, mode, operand ranges."""
import torch
import bench as B
from gemnet_pytorch_amd import kernels as K
from gemnet_pytorch_amd.index_device import DeviceGraphBuilder
from gemnet_pytorch_amd.model.gemnet import GemNet
from gemnet_pytorch_amd.padded import PaddedGraphRunner
from gemnet_pytorch_amd.synthetic import make_dataset
from gemnet_pytorch_amd.training.ddp import PaddedTrainStep

dev = torch.device("cuda")
cfg = dict(B.GEMNET_T)
g = torch.Generator().manual_seed(1)
ds = make_dataset(32, 32, config=2, first=32)
R = torch.tensor(ds["R"], device=dev, dtype=torch.float32)
idx = DeviceGraphBuilder(ds["N"], cfg["cutoff"], cfg["int_cutoff"], True, device=dev)(R)
Z, N = torch.tensor(ds["Z"], device=dev).long(), torch.tensor(ds["N"], device=dev).long()
Et, Ft = torch.randn(32, 1, generator=g).to(dev), torch.randn(1024, 3, generator=g).to(dev)
E_, T_ = int(idx["id_c"].shape[0]), int(idx["id3_reduce_ca"].shape[0])
torch.manual_seed(1234)
model = GemNet(**cfg, scale_file=B.SCALE_FILE).to(dev)
import sys
caps = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else PaddedGraphRunner.suggest_capacities([(E_, T_)])
ts = PaddedTrainStep(model, Z, N, *caps, fused_optimizer=True)
real_chain = K.chain
state = {"n": 0, "found": False}


def rng(t):
    t = t.float()
    fin = torch.isfinite(t)
    return (f"shape {tuple(t.shape)} finite {bool(fin.all())} absmax {float(t[fin].abs().max()) if bool(fin.any()) else float('nan'):.3e} "
            f"zero rows {int((t.abs().amax(dim=1) == 0).sum()) if t.dim() == 2 else -1}")


def chain(prog, mode=None):
    out = real_chain(prog, mode=mode)
    state["n"] += 1
    if state["found"]:
        return out
    torch.cuda.synchronize()
    for i, o in enumerate(prog.ops):
        for key in ("out", "pre_out", "out2"):
            t = o.get(key)
            if t is not None and not bool(torch.isfinite(t).all()):
                state["found"] = True
                print(f"launch {state['n']} mode {mode or K.current_mode()} M {prog.M}: op {i} ({o['kind']}) writes non-finite {key}: {rng(t)}")
                bad_rows = (~torch.isfinite(t)).any(dim=1).nonzero().flatten()
                print("   bad rows:", bad_rows[:8].tolist(), "... count", int(bad_rows.numel()), "of", t.shape[0], "; first real-row count:", E_, T_)
                for j, oo in enumerate(prog.ops):
                    desc = {k: (rng(v) if torch.is_tensor(v) and v.dtype == torch.float32 and v.dim() == 2 else v) for k, v in oo.items()
                            if v is not None and k not in ("W", "packed") and not (isinstance(v, (int, float)) and v in (0, 1, -1, 1.0, 0.0, False))}
                    print("   op", j, desc)
                return out
    return out


K.chain = chain
ts.capture = lambda *a, **k: ts       # stay eager
ts._captured = True
loss = ts.step(R, idx, Et, Ft, Z=Z)
torch.cuda.synchronize()
print("loss", float(loss), "launches", state["n"], "found", state["found"], "caps", caps, "sizes", (E_, T_))
