"""Find the first gn_chain_f32 launch that disagrees with the CPU interpreter (tests/cpu_kernels.chain)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["GEMNET_STACKS"] = "1"
import numpy as np, torch
import cpu_kernels as CK
from oracle import gemnet_oracle as GO
from gemnet_pytorch_amd import kernels as K
from gemnet_pytorch_amd.model.gemnet import GemNet
from test_oracle_model import load_case
from conftest import GOLDEN, SCALE_FILE

tag = sys.argv[1] if len(sys.argv) > 1 else "q2"
g = np.load(os.path.join(GOLDEN, "model.npz"))
cfg, params, inputs = load_case(g, tag)
model = GemNet(**cfg, scale_file=SCALE_FILE)
model.load_state_dict(GO.expand_to_reference_state_dict({k: v.float() for k, v in params.items()}), strict=True)
model = model.to("cuda").eval()
real_chain = K.chain
n_call = [0]

def checked(prog):
    n_call[0] += 1
    # CPU copy of the program (inputs cloned before the launch; outputs fresh)
    memo = {}
    def cpu(t):
        if t is None or isinstance(t, int):
            return t
        if id(t) not in memo:
            memo[id(t)] = (t, t.detach().cpu().double() if t.is_floating_point() else t.detach().cpu())
        return memo[id(t)][1]
    ref = K.ChainProgram(prog.M)
    for o in prog.ops:
        ref.ops.append({k: (cpu(v) if torch.is_tensor(v) else v) for k, v in o.items()})
    real_chain(prog)
    torch.cuda.synchronize()
    CK.chain(ref)
    kinds = [o["kind"] + (f"[N{o['W'].shape[0]}K{o['W'].shape[1]} a{o['a_slot']} y{o['slot']}]" if o["kind"] == "gemm" else "") for o in prog.ops]
    worst = 0.0
    for i, (o, r) in enumerate(zip(prog.ops, ref.ops)):
        for key in ("pre_out", "out"):
            if torch.is_tensor(o.get(key)):
                d = float((o[key].detach().cpu().double() - r[key]).abs().max())
                sc = float(r[key].abs().max()) + 1e-30
                worst = max(worst, d / sc)
                if d / sc > 1e-3:
                    print(f"  call {n_call[0]} M={prog.M} op {i} {kinds[i]} {key}: max err {d:.3e} (scale {sc:.3e})")
    print(f"call {n_call[0]}: M={prog.M} n_ops={len(prog.ops)} worst rel err {worst:.2e}  {' '.join(kinds) if worst > 1e-3 else ''}", flush=True)

K.chain = checked
dev_inputs = {k: v.to("cuda") for k, v in inputs.items()}
E, F = model(dev_inputs)
print("F mae vs golden", float(np.abs(F.detach().cpu().numpy() - g[f"{tag}.F"]).mean()))
