"""Per-kernel HBM-side traffic from the two rocprofv3 PMC passes of tools/gpu_artifacts.sh (runs anywhere: pandas only).

    python tools/pmc_summary.py gpurun_out/<tag> [r2]  ->  profiles/<round>_pmc_traffic.txt, profiles/<round>_traffic.json

gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE under-reports wide coalesced reads by 2x, so
bytes_per_launch = (2 * fetch_kb + write_kb) * 1024; WRITE_SIZE is taken at face value."""
import glob, json, os, re, sys
import pandas as pd

tag = sys.argv[1]
rnd = sys.argv[2] if len(sys.argv) > 2 else "r2"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"at::native::", "", name)
    return name.split("(")[0][:64]


tab = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    fs = glob.glob(os.path.join(tag, "pmc_" + c, "**", "*counter_collection.csv"), recursive=True)
    df = pd.read_csv(fs[0])
    df = df[df["Counter_Name"] == c]
    df["k"] = df["Kernel_Name"].map(short)
    tab[c] = df.groupby("k")["Counter_Value"].agg(["mean", "count"])
t = pd.DataFrame({"n": tab["FETCH_SIZE"]["count"], "fetch_kb": tab["FETCH_SIZE"]["mean"],
                  "write_kb": tab["WRITE_SIZE"]["mean"]}).fillna(0.0)
t["bytes_per_launch"] = (2 * t["fetch_kb"] + t["write_kb"]) * 1024
t["total"] = t["bytes_per_launch"] * t["n"]
t = t.sort_values("total", ascending=False).drop(columns="total")
head = """# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on `python bench.py --no-graph --steps 2 --warmup 1
# --no-cpu-baseline --no-roofline` (tools/gpu_artifacts.sh, summarised by tools/pmc_summary.py), MI355X.
# Raw counter values are KB per dispatch (mean over `n` dispatches).
# gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE under-reports wide coalesced reads by 2x ->
# bytes_per_launch = (2 * fetch_kb + write_kb) * 1024.  WRITE_SIZE is uncalibrated for 4-byte strided stores (taken at face value).
# The working set of this batch is Infinity-Cache resident: these are L2 memory-side requests, not DRAM bytes.
"""
with open(os.path.join(root, "profiles", rnd + "_pmc_traffic.txt"), "w") as f:
    f.write(head + t.head(40).round(1).to_string() + "\n")
fam = {"chain": [k for k in t.index if k.startswith("chain_kernel") or k.startswith("chain_split_kernel")],
       "gemm": [k for k in t.index if k.startswith("gemm_")],
       "bil_fused_fwd": [k for k in t.index if k.startswith("bil_fused_fwd")],
       "bil_project_bwd": [k for k in t.index if k.startswith("bil_project_bwd")],
       "bil_dy_multi": [k for k in t.index if k.startswith("bil_dy_multi")],
       "bil_reduce_t": [k for k in t.index if k.startswith("bil_reduce_t") or k.startswith("bil_expand")],
       "bil_reduce_project": [k for k in t.index if k.startswith("bil_reduce_project")]}
fam = {k: v for k, v in fam.items() if v}
out = {}
for name, ks in fam.items():
    sub = t.loc[ks]
    n = sub["n"].sum()
    out[name] = {"kernels": ks, "launches_sampled": int(n),
                 "fetch_kb_raw": round(float((sub["fetch_kb"] * sub["n"]).sum() / n), 1),
                 "write_kb_raw": round(float((sub["write_kb"] * sub["n"]).sum() / n), 1),
                 "bytes_per_launch": int((sub["bytes_per_launch"] * sub["n"]).sum() / n),
                 "source": "profiles/%s_pmc_traffic.txt" % rnd}
with open(os.path.join(root, "profiles", rnd + "_traffic.json"), "w") as f:
    json.dump(out, f, indent=1)
print(t.head(16).round(1).to_string())
print(json.dumps({k: v["bytes_per_launch"] for k, v in out.items()}))
