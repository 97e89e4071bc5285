"""Does a kernel of a replayed multi-queue hipGraph see its own registers change?  (round-4 replay finding, docs/HISTORY.md section 11)

The aggregation adjoint (csrc/aggregate.hip) loads 32 values of a CONSTANT weight per lane at kernel start and keeps them
in registers.  The -DGN_AGG_SELFCHECK build re-reads the weight (volatile) right after the first load and at every edge,
compares with the registers and logs mismatches.  This script captures the configuration whose replays misbehave
(GemNet-T 8 x 64 atoms, output-block radial projection on the side stream), replays it, and prints the log:
kind 1 = the two loads at kernel start disagree; kind 2 / 3 = register != memory at the time of use (3: memory stable).

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -DGN_AGG_SELFCHECK -I include gemnet_pytorch_amd/csrc/*.hip \\
          -o tools/exp/bin/libgemnet_hip_aggcheck.so
    PYTHONPATH=.:tests python tools/exp/agg_selfcheck.py [replays]"""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gemnet_pytorch_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, "tools", "exp", "bin", "libgemnet_hip_aggcheck.so")
lib = _lib.load()
lib.gn_agg_selfcheck_read.argtypes = [ctypes.c_void_p, ctypes.c_int]
lib.gn_agg_selfcheck_read.restype = ctypes.c_int

sys.path.insert(0, os.path.join(ROOT, "tools"))
import hbcheck_run as H          # noqa: E402  (model / batch builders)
from gemnet_pytorch_amd.model import gemnet as G   # noqa: E402
from gemnet_pytorch_amd.model.gemnet import GemNet  # noqa: E402


def read_log(reset=True):
    buf = np.zeros((4096, 10), dtype=np.uint32)
    n = lib.gn_agg_selfcheck_read(buf.ctypes.data_as(ctypes.c_void_p), int(reset))
    return n, buf[:min(n, 4096)]


def show(tag, n, rec):
    print(f"--- {tag}: {n} mismatch records")
    for r in rec[:40]:
        kind, first = int(r[0]) & 0xff, int(r[0]) >> 8
        hw = int(r[7])
        print(f"   kind {kind} block {r[1]:5d} wave {r[2]} lane {r[3]:2d} first k {first:2d} mask {int(r[4]):08x} "
              f"reg {np.uint32(r[5]).view(np.float32):+.6e} mem {np.uint32(r[6]).view(np.float32):+.6e} "
              f"hw_id {hw:08x} (wave {hw & 15} simd {(hw >> 4) & 3} cu {(hw >> 8) & 15} sh {(hw >> 12) & 1} se {(hw >> 13) & 7}) "
              f"xcc {int(r[8]) & 15} edge {int(r[9])}")
    sys.stdout.flush()


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    G._RBF_OUT_SIDE = os.environ.get("AGG_RBF_OUT_SIDE", "1") == "1"
    cfg = dict(H.FULL, triplets_only=True)
    torch.manual_seed(11)
    model = GemNet(**cfg, scale_file=H.SCALE_FILE).to("cuda").eval()
    model.requires_grad_(False)
    inputs = H.batch(8, 64, True)
    E0, F0 = (t.detach().clone() for t in model(inputs))
    torch.cuda.synchronize()
    show("eager warm-up run", *read_log())
    for _ in range(5):
        model(inputs)
    torch.cuda.synchronize()
    show("5 eager runs", *read_log())
    H.warm(lambda: model(inputs))
    read_log()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        Eg, Fg = model(inputs)
    bad = 0
    for _ in range(reps):
        graph.replay()
        torch.cuda.synchronize()
        bad += int(not torch.equal(Fg, F0))
    print(f"{bad} of {reps} replays differ from the eager run")
    show(f"{reps} graph replays", *read_log())


if __name__ == "__main__":
    main()
