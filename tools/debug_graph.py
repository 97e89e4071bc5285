"""Stepwise hipGraph capture diagnostics (run on the GPU box)."""
import faulthandler
import os
import sys

faulthandler.enable()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from gemnet_pytorch_amd import kernels as K
from gemnet_pytorch_amd.graph import GraphPlan
from gemnet_pytorch_amd.model.gemnet import GemNet

dev = torch.device("cuda", 0)
print("torch", torch.__version__, flush=True)


def capture(fn, label):
    print(f"-- capture: {label}", flush=True)
    fn(); fn()
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn(); fn()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = fn()
    torch.cuda.synchronize()
    g.replay()
    torch.cuda.synchronize()
    print(f"   ok: {label}", flush=True)
    return g, out


x = torch.randn(1000, 128, device=dev)
W = torch.randn(128, 128, device=dev)
capture(lambda: K.gemm(x, W, act=True), "single gemm launcher")
capture(lambda: K.ssilu(K.gemm(x, W), 0), "two launchers")

xr = x.clone().requires_grad_(True)
def fb():
    y = bench.__dict__  # noqa
    from gemnet_pytorch_amd import ops
    z = ops.ssilu(ops.linear(xr, W)).sum()
    return torch.autograd.grad(z, xr)[0]
capture(fb, "autograd.grad through 2 custom Functions")

cfg = dict(bench.GEMNET_T)
cfg["num_blocks"] = 1
model = GemNet(**cfg, scale_file=bench.SCALE_FILE).to(dev).eval()
inputs, _ = bench.make_batch(cfg, 4, 32, 0, dev)
GraphPlan.from_inputs(inputs, True).warm()
def fwd_nograd():
    with torch.no_grad():
        plan = GraphPlan.from_inputs(inputs, True)
        return model._energy(inputs["R"], plan)[0]
capture(fwd_nograd, "GemNet forward energy only (no_grad)")
model.requires_grad_(False)
capture(lambda: model(inputs), "GemNet forward+force, params frozen")
model.requires_grad_(True)
capture(lambda: model(inputs), "GemNet forward+force, params require grad")
print("ALL OK", flush=True)
