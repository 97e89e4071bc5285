"""Accuracy of the two-plane fp16 chain arithmetic ("h3") against float64 as a function of the operand scale: one
128 x 128 layer and a LOAD -> STORE round trip, x ~ N(0, 1) * scale.  Run on the GPU box:
    python tools/exp/h3_accuracy.py
"""
import torch

from gemnet_pytorch_amd import kernels as K

DEV = "cuda"


def main():
    g = torch.Generator().manual_seed(0)
    M = 8192
    x0 = torch.randn(M, 128, generator=g)
    W = (torch.randn(128, 128, generator=g) / 11).to(DEV)
    print(f"{'scale':>8s} {'mode':>7s} {'rel err of x W^T':>18s} {'round trip rel':>16s}")
    for scale in (1.0, 1e-2, 1e-3, 1e-4, 1e-5, 1e-6, 1e-8, 1e2, 1e4):
        x = (x0 * scale).to(DEV)
        ref = x.double() @ W.double().t()
        for mode in ("f32", "split6", "h3"):
            y = torch.empty(M, 128, device=DEV)
            p = K.ChainProgram(M)
            p.load(0, x)
            p.gemm(W, a_slot=0, y_slot=1, out=y)
            K.chain(p, mode=mode)
            err = float((y.double() - ref).norm() / ref.norm())
            rt = torch.empty(M, 128, device=DEV)
            p = K.ChainProgram(M)
            p.load(0, x)
            p.store(0, rt)
            K.chain(p, mode=mode)
            rte = float((rt.double() - x.double()).norm() / x.double().norm())
            print(f"{scale:8.0e} {mode:>7s} {err:18.3e} {rte:16.3e}")


if __name__ == "__main__":
    main()
