"""Chain programs (mode h3 / split6) on one stream, fused aggregate kernels on another: are both bit-reproducible?
   PYTHONPATH=. python tools/exp/h3_concurrency2.py [mode]"""
import sys
import torch
from gemnet_pytorch_amd import kernels as K

sys.path.insert(0, "tools/exp")
from h3_concurrency import program  # noqa: E402

DEV = "cuda"
mode = sys.argv[1] if len(sys.argv) > 1 else "h3"


def main():
    g = torch.Generator().manual_seed(0)
    A, E = 512, 9000
    m = torch.randn(E, 128, generator=g).to(DEV)
    rbf = torch.randn(E, 16, generator=g).to(DEV)
    W = (torch.randn(128, 16, generator=g) / 4).to(DEV)
    id_a = torch.sort(torch.randint(0, A, (E,), generator=g))[0].to(DEV)
    perm = torch.arange(E, dtype=torch.int32, device=DEV)
    seg = torch.searchsorted(id_a, torch.arange(A + 1, device=DEV)).to(torch.int32)
    id32 = id_a.to(torch.int32)
    gout = torch.randn(A, 128, generator=g).to(DEV)

    def agg():
        o = K.rbf_aggregate_fwd(m, rbf, W, perm, seg, A, 0.3)
        gm, gr = K.rbf_aggregate_bwd(gout, m, rbf, W, id32, 0.3)
        return o, gm, gr

    ref_a = [t.clone() for t in agg()]
    for M, adj in ((9000, True), (9000, False), (512, True), (512, False)):
        p, outs = program(M, g, adj=adj)
        K.chain(p, mode=mode)
        torch.cuda.synchronize()
        ref = [o.clone() for o in outs]
        side = torch.cuda.Stream()
        bad_c = bad_a = 0
        for it in range(200):
            with torch.cuda.stream(side):
                res = [agg() for _ in range(3)]
            K.chain(p, mode=mode)
            K.chain(p, mode=mode)
            torch.cuda.synchronize()
            bad_c += int(any(not torch.equal(a, b) for a, b in zip(ref, outs)))
            bad_a += int(any(not torch.equal(a, b) for r in res for a, b in zip(ref_a, r)))
        print(f"[{mode}] chain M={M} adj={adj} next to the aggregate kernels: chain differs {bad_c}/200, aggregate differs {bad_a}/200")


if __name__ == "__main__":
    main()
