"""Minimal co-run experiment for the round-4 replay finding (docs/HISTORY.md section 11): NO model, two kernels of the library,
constant inputs.  A two-branch hipGraph is captured — branch A: a string of Dense-stack chain programs (edge rows),
branch B: a string of fused aggregation kernels (forward + adjoint), every launch writing its own output buffer — and
replayed; every output of every replay is compared bit for bit with the eager result of the same launch.

   PYTHONPATH=. python tools/exp/graph_corun.py [mode=h3|split6] [replays=50]

Each kernel on its own is deterministic and reads only constant buffers here, so ANY mismatch is produced below the
library: by the graph runtime or the hardware."""
import os
import sys

import torch

from gemnet_pytorch_amd import kernels as K

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from h3_concurrency import program  # noqa: E402

DEV = "cuda"


def main():
    mode = sys.argv[1] if len(sys.argv) > 1 else "h3"
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
    g = torch.Generator().manual_seed(0)
    A, E = 512, 11824
    m = torch.randn(E, 128, generator=g).to(DEV)
    rbf = torch.randn(E, 16, generator=g).to(DEV)
    W = (torch.randn(128, 16, generator=g) / 4).to(DEV)
    id_a = torch.sort(torch.randint(0, A, (E,), generator=g))[0].to(DEV)
    perm = torch.arange(E, dtype=torch.int32, device=DEV)
    seg = torch.searchsorted(id_a, torch.arange(A + 1, device=DEV)).to(torch.int32)
    id32 = id_a.to(torch.int32)
    gout = torch.randn(A, 128, generator=g).to(DEV)
    progs = [program(M, g, adj=adj) for M, adj in ((E, True), (E, False), (A, True), (E, True), (E, False), (A, False))]

    def branch_a():
        for p, _ in progs:
            K.chain(p, mode=mode)

    def branch_b(n=6):
        out = []
        for _ in range(n):
            out.append(K.rbf_aggregate_fwd(m, rbf, W, perm, seg, A, 0.3))
            out.extend(K.rbf_aggregate_bwd(gout, m, rbf, W, id32, 0.3))
        return out

    branch_a()
    ref_b = [t.clone() for t in branch_b()]
    torch.cuda.synchronize()
    ref_a = [[o.clone() for o in outs] for _, outs in progs]

    for label, two_streams in (("two branches", True), ("one branch", False)):
        side = torch.cuda.Stream()
        cap = torch.cuda.Stream()
        cap.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(cap):
            branch_a(), branch_b()           # warm-up on the capture stream
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=cap):
            if two_streams:
                side.wait_stream(cap)
                with torch.cuda.stream(side):
                    res_b = branch_b()
                branch_a()
                cap.wait_stream(side)
            else:
                res_b = branch_b()
                branch_a()
        bad_a = bad_b = 0
        detail = []
        for it in range(reps):
            graph.replay()
            torch.cuda.synchronize()
            wrong_a = [i for i, ((_, outs), ref) in enumerate(zip(progs, ref_a)) if any(not torch.equal(a, b) for a, b in zip(outs, ref))]
            wrong_b = [i for i, (a, b) in enumerate(zip(res_b, ref_b)) if not torch.equal(a, b)]
            bad_a += bool(wrong_a)
            bad_b += bool(wrong_b)
            if wrong_b and len(detail) < 3:
                i = wrong_b[0]
                d = (res_b[i] != ref_b[i])
                rows = d.reshape(d.shape[0], -1).any(dim=1).nonzero().flatten()
                cols = d.reshape(d.shape[0], -1)[int(rows[0])].nonzero().flatten().tolist()
                detail.append(f"replay {it}: aggregation output {i} ({tuple(res_b[i].shape)}): {int(d.sum())} elements in rows "
                              f"{rows.tolist()[:8]}, columns of the first row {cols[:20]}")
        print(f"[{mode}] {label}: chain outputs differ in {bad_a}/{reps} replays, aggregation outputs in {bad_b}/{reps}")
        for line in detail:
            print("    " + line)
        sys.stdout.flush()


if __name__ == "__main__":
    main()
