"""Is a chain launch bit-reproducible when another chain kernel shares the GPU (two streams)?  Serial result vs. the result
with a second stream hammering small chain programs next to it.   PYTHONPATH=. python tools/exp/h3_concurrency.py [mode]"""
import sys
import torch
from gemnet_pytorch_amd import kernels as K

DEV = "cuda"
mode = sys.argv[1] if len(sys.argv) > 1 else "h3"


def program(M, g, n=3, adj=True):
    x = torch.randn(M, 128, generator=g).to(DEV)
    Ws = [(torch.randn(128, 128, generator=g) / 11).to(DEV) for _ in range(n)]
    zs = [torch.randn(M, 128, generator=g).to(DEV) for _ in range(n)]
    outs = [torch.empty(M, 128, device=DEV) for _ in range(n)]
    p = K.ChainProgram(M)
    p.load(0, x)
    cur, oth = 0, 1
    for i in range(n):
        if adj:
            p.gemm(Ws[i], a_slot=cur, y_slot=oth, mul=zs[i], mul_mode=1, res=cur, beta=1.0, out=outs[i])
        else:
            p.gemm(Ws[i], a_slot=cur, y_slot=oth, act=True, pre_out=zs[i], pre_deriv=True, out=outs[i])
        cur, oth = oth, cur
    return p, outs


def main():
    g = torch.Generator().manual_seed(0)
    for M, adj in ((9000, True), (9000, False), (18122, True), (512, True), (4000, True), (13000, True)):
        p, outs = program(M, g, adj=adj)
        K.chain(p, mode=mode)
        torch.cuda.synchronize()
        ref = [o.clone() for o in outs]
        q, _ = program(512, g, adj=True)
        q2, _ = program(512, g, adj=False)
        side = torch.cuda.Stream()
        bad = 0
        for it in range(200):
            with torch.cuda.stream(side):
                for _ in range(4):
                    K.chain(q, mode=mode)
                    K.chain(q2, mode=mode)
            K.chain(p, mode=mode)
            K.chain(p, mode=mode)
            torch.cuda.synchronize()
            bad += int(any(not torch.equal(a, b) for a, b in zip(ref, outs)))
        print(f"[{mode}] M={M} adj={adj}: {bad} of 200 concurrent runs differ from the serial result")


if __name__ == "__main__":
    main()
