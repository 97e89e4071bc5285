"""Per-kernel HBM-side traffic and MFMA-busy figures from the rocprofv3 PMC passes of tools/gpu_artifacts.sh (runs
anywhere: pandas only).

    python tools/pmc_summary.py gpurun_out/<tag> <mode: T|Q|train> [r3]
      -> profiles/<round>_pmc_traffic_<mode>.txt, profiles/<round>_traffic_<mode>.json     (FETCH_SIZE / WRITE_SIZE passes)
      -> profiles/<round>_pmc_sq_<mode>.txt, profiles/<round>_mfma_busy.json[mode]         (SQ pass)

gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE under-reports wide coalesced reads by 2x, so
bytes_per_launch = (2 * fetch_kb + write_kb) * 1024; WRITE_SIZE is taken at face value.
MFMA busy: SQ_VALU_MFMA_BUSY_CYCLES (summed over the SIMDs) / (dispatch duration x 2.4 GHz x 1024 SIMDs), durations from
the rocprofv3 --kernel-trace --stats run of the same workload (average per kernel; the counter pass's own Start/End
timestamps when that run is absent)."""
import glob, json, os, re, sys
import pandas as pd

tag = sys.argv[1]
mode = sys.argv[2]
rnd = sys.argv[3] if len(sys.argv) > 3 else "r3"
# optional: the algorithmic bytes / flops per launch of every launcher family of the same workload (bench.py with
# GEMNET_DUMP_FAMILIES=<prefix>: <prefix>_<title>.json) — written next to the counted bytes so that the ratio can be formed
# from ONE file
alg = {}
if len(sys.argv) > 4 and os.path.exists(sys.argv[4]):
    with open(sys.argv[4]) as f:
        alg = json.load(f)
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLK_HZ, N_SIMD = 2.4e9, 1024


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"at::native::", "", name)
    return name.split("(")[0][:64]


FAMILIES = {"bil_reduce_project_tan": ("bil_reduce_project_ang_tan",), "bil_reduce_t_tan": ("bil_expand_ang_tan", "bil_expand_rows_ang_tan"),
            "chain": ("chain_kernel", "chain_split_kernel"), "gemm": ("gemm_nt", "gemm_generic", "gemm_smallk", "gemm_n1"),
            "gemm_tn": ("gemm_tn",), "bil_fused_fwd": ("bil_fused_fwd",), "bil_fused_bwd": ("bil_fused_bwd",),
            "bil_project_bwd": ("bil_project_bwd",), "bil_dy_multi": ("bil_dy_multi",),
            "bil_reduce_t": ("bil_reduce_t", "bil_expand"), "bil_reduce_project": ("bil_reduce_project",),
            "rbf_aggregate_fwd": ("rbf_aggregate_fwd",), "rbf_aggregate_bwd": ("rbf_aggregate_bwd",)}


def family_of(k):
    for fam, prefixes in FAMILIES.items():
        if any(k.startswith(p) for p in prefixes):
            return fam
    return None


def load(counter_dir):
    fs = glob.glob(os.path.join(tag, counter_dir, "**", "*counter_collection.csv"), recursive=True)
    if not fs:
        return None
    df = pd.read_csv(fs[0])
    df["k"] = df["Kernel_Name"].map(short)
    return df


# ---------------------------------------------------------------- traffic
tab = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    df = load(f"pmc_{mode}_{c}")
    if df is not None:
        df = df[df["Counter_Name"] == c]
        tab[c] = df.groupby("k")["Counter_Value"].agg(["mean", "count"])
if len(tab) == 2:
    t = pd.DataFrame({"n": tab["FETCH_SIZE"]["count"], "fetch_kb": tab["FETCH_SIZE"]["mean"],
                      "write_kb": tab["WRITE_SIZE"]["mean"]}).fillna(0.0)
    t["bytes_per_launch"] = (2 * t["fetch_kb"] + t["write_kb"]) * 1024
    t["total"] = t["bytes_per_launch"] * t["n"]
    t = t.sort_values("total", ascending=False).drop(columns="total")
    head = f"""# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, no trace domains) on the {mode} workload of bench.py
# (tools/gpu_artifacts.sh, summarised by tools/pmc_summary.py), MI355X.  Raw counter values are KB per dispatch (mean over n).
# gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE under-reports wide coalesced reads by 2x ->
# bytes_per_launch = (2 * fetch_kb + write_kb) * 1024.  WRITE_SIZE is uncalibrated for 4-byte strided stores (face value).
# The working set of this batch is Infinity-Cache resident: these are L2 memory-side requests, not DRAM bytes.
"""
    with open(os.path.join(root, "profiles", f"{rnd}_pmc_traffic_{mode}.txt"), "w") as f:
        f.write(head + t.head(45).round(1).to_string() + "\n")
    out = {}
    for fam in FAMILIES:
        ks = [k for k in t.index if family_of(k) == fam]
        if not ks:
            continue
        sub = t.loc[ks]
        n = sub["n"].sum()
        out[fam] = {"kernels": ks, "launches_sampled": int(n),
                    "fetch_kb_raw": round(float((sub["fetch_kb"] * sub["n"]).sum() / n), 1),
                    "write_kb_raw": round(float((sub["write_kb"] * sub["n"]).sum() / n), 1),
                    "bytes_per_launch": int((sub["bytes_per_launch"] * sub["n"]).sum() / n),
                    "source": f"profiles/{rnd}_pmc_traffic_{mode}.txt"}
        if fam in alg:
            a = alg[fam]
            out[fam].update(algorithmic_bytes_per_launch=a["algorithmic_bytes_per_launch"],
                            algorithmic_flops_per_launch=a["algorithmic_flops_per_launch"], launches_per_step=a["launches"],
                            counted_over_algorithmic=round(out[fam]["bytes_per_launch"] / max(a["algorithmic_bytes_per_launch"], 1), 3))
    with open(os.path.join(root, "profiles", f"{rnd}_traffic_{mode}.json"), "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps({k: v["bytes_per_launch"] for k, v in out.items()}))

# ---------------------------------------------------------------- SQ counters: MFMA busy, wait / issue fractions
df = load(f"pmc_{mode}_sq")
if df is not None:
    df["dur_ns"] = df["End_Timestamp"] - df["Start_Timestamp"]
    # kernel durations of the UNINSTRUMENTED run of the same workload (rocprofv3 --kernel-trace --stats, hipGraph replay),
    # when it is there: counter collection serialises and stretches the dispatches (chain: 46 vs 30 us)
    sdir = {"T": "prof", "train": "prof_train", "Q": "prof_Q", "Qtrain": "prof_Qtrain"}.get(mode, "prof_" + mode)
    sf = glob.glob(os.path.join(tag, sdir, "**", "*kernel_stats.csv"), recursive=True)
    if sf:
        ks = pd.read_csv(sf[0])
        avg = {short(n): float(a) for n, a in zip(ks["Name"], ks["AverageNs"])}
        df["dur_ns"] = [avg.get(k, d) for k, d in zip(df["k"], df["dur_ns"])]
    piv = df.pivot_table(index=["Dispatch_Id", "k"], columns="Counter_Name", values="Counter_Value", aggfunc="sum").reset_index()
    dur = df.groupby("Dispatch_Id")["dur_ns"].first()
    piv["dur_ns"] = piv["Dispatch_Id"].map(dur)
    piv["fam"] = piv["k"].map(family_of)
    rows, busy = [], {}
    for k, gk in piv.groupby("k"):
        if "SQ_VALU_MFMA_BUSY_CYCLES" not in gk or gk["SQ_WAVE_CYCLES"].sum() == 0:
            continue
        wc = gk["SQ_WAVE_CYCLES"].mean()
        rows.append((k, len(gk), gk["dur_ns"].mean() / 1e3, 100 * gk["SQ_WAIT_ANY"].mean() / wc, 100 * gk["SQ_WAIT_INST_ANY"].mean() / wc,
                     100 * gk["SQ_ACTIVE_INST_ANY"].mean() / wc,
                     100 * gk["SQ_VALU_MFMA_BUSY_CYCLES"].mean() / (gk["dur_ns"].mean() * 1e-9 * CLK_HZ * N_SIMD)))
    rows.sort(key=lambda r: -r[1] * r[2])
    with open(os.path.join(root, "profiles", f"{rnd}_pmc_sq_{mode}.txt"), "w") as f:
        f.write(f"# rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY\n"
                f"# (one pass, no trace domains) on the {mode} workload of bench.py, MI355X.  Means per dispatch; the three wave fractions\n"
                f"# are of SQ_WAVE_CYCLES; MFMA busy % = SQ_VALU_MFMA_BUSY_CYCLES / (duration x 2.4 GHz x 1024 SIMDs), duration = the\n"
                f"# kernel's average in the uninstrumented kernel-trace run of the same workload.\n"
                f"# {'kernel':58s} {'n':>5s} {'us':>8s} {'wait_any%':>9s} {'wait_inst%':>10s} {'active%':>8s} {'mfma_busy%':>10s}\n")
        for r in rows[:40]:
            f.write(f"  {r[0]:58s} {r[1]:5d} {r[2]:8.1f} {r[3]:9.1f} {r[4]:10.1f} {r[5]:8.1f} {r[6]:10.1f}\n")
    for fam, gf in piv[piv["fam"].notna()].groupby("fam"):
        if "SQ_VALU_MFMA_BUSY_CYCLES" in gf:
            busy[fam] = {"mfma_busy_pct": round(float(100 * gf["SQ_VALU_MFMA_BUSY_CYCLES"].sum() / (gf["dur_ns"].sum() * 1e-9 * CLK_HZ * N_SIMD)), 2),
                         "dispatches": int(len(gf)), "source": f"profiles/{rnd}_pmc_sq_{mode}.txt"}
    path = os.path.join(root, "profiles", f"{rnd}_mfma_busy.json")
    allb = json.load(open(path)) if os.path.exists(path) else {}
    allb[mode] = busy
    with open(path, "w") as f:
        json.dump(allb, f, indent=1)
    print("mfma busy %:", {k: v["mfma_busy_pct"] for k, v in busy.items()})
