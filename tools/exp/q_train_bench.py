"""GemNet-Q (BASELINE configs[2]) forward+force and TRAINING step alone (bench.extra_gemnet_q), for A/B runs:
   PYTHONPATH=. python tools/exp/q_train_bench.py [n_mol=32] [n_atoms=32]
   GEMNET_TRAIN2_QUAD=0 ... : the round-4 form (quadruplet layer on the composite closure, eager)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

import bench  # noqa: E402

n_mol = int(sys.argv[1]) if len(sys.argv) > 1 else 32
n_atoms = int(sys.argv[2]) if len(sys.argv) > 2 else 32
torch.cuda.set_device(0)
out = bench.extra_gemnet_q(n_mol, n_atoms, 0)
print(json.dumps(out))
