"""GPU box: the 64-atom fixtures (q64s / t64s of tests/golden/fullsize.npz) per chain arithmetic: force MAE against the
reference's float64 forces, and the largest |activation| leaving each interaction block (fp16-plane range check)."""
import os, sys, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from conftest import SCALE_FILE
from fullsize_common import dataset, load_fullsize, params_of, triplets_only
from oracle import gemnet_oracle as GO
from gemnet_pytorch_amd.model.gemnet import GemNet
from gemnet_pytorch_amd.training.data_container import DataContainer

g = load_fullsize()
for tag in sys.argv[1:] or ["q64s", "t64s", "qB4"]:
    cfg, params = params_of(g, tag)
    ds, to = dataset(tag), triplets_only(tag)
    dc = DataContainer.from_arrays(ds, 5.0, 10.0, triplets_only=to)
    batch = dc[list(range(len(ds["N"])))]
    inputs = {k: v.to("cuda") for k, v in batch.items() if k not in ("E", "F")}
    for mode in (None, "split6", "f32"):
        model = GemNet(**cfg, scale_file=SCALE_FILE)
        model.load_state_dict(GO.expand_to_reference_state_dict(params), strict=True)
        model = model.to("cuda").eval()
        model.matmul_precision = mode
        peaks = []
        hooks = [b.register_forward_hook(lambda mod, a, out, peaks=peaks: peaks.append(
            tuple(float(o.detach().abs().max()) for o in (out if isinstance(out, tuple) else (out,)) if torch.is_tensor(o))))
                 for b in model.int_blocks]
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            E, F = model(inputs)
        fell = [str(x.message)[:60] for x in w if "fp16-plane" in str(x.message)]
        mae = float(np.abs(F.detach().cpu().numpy() - g[f"{tag}.F"]).mean())
        print(f"{tag} mode={mode or 'h3(default)'} -> now {model.matmul_precision}: force MAE {mae:.3e}; fallback={bool(fell)}; "
              f"block output peaks {peaks[-len(model.int_blocks):]}", flush=True)
        for h in hooks:
            h.remove()
