"""Per-kernel time of ONE step from a `rocprofv3 --kernel-trace --stats` summary of `bench.py` (runs anywhere).

    python tools/step_breakdown.py profiles/r1_final_kernel_stats.csv > profiles/r1_final_step_breakdown.txt

The number of profiled steps is inferred from the most frequent kernel whose per-step count is known (24 launches of
chain_kernel<5> per GemNet-T forward+force step)."""
import csv, re, sys

rows = list(csv.DictReader(open(sys.argv[1])))
per_step = {"chain_split_kernel<5, 3, true>": 16, "chain_kernel<5>": 24}
steps = None
for r in rows:
    for k, n in per_step.items():
        if k in r["Name"]:
            steps = int(r["Calls"]) / n
assert steps, "no anchor kernel found"


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "").replace("at::native::", "")
    return re.sub(r"\(.*", "", n)[:58]


tot = sum(float(r["TotalDurationNs"]) for r in rows) / steps / 1e3
print(f"# {sys.argv[1]}: {steps:.0f} steps profiled; sum of kernel time per step {tot:.0f} us "
      f"(rocprof-inflated durations; the hipGraph step itself is timed by bench.py)")
print(f"# {'kernel':58s} {'launches/step':>13s} {'avg us':>8s} {'us/step':>9s} {'share':>6s}")
acc = 0.0
for r in rows:
    us = float(r["TotalDurationNs"]) / steps / 1e3
    if us < 1.0:
        continue
    acc += us
    print(f"  {short(r['Name']):58s} {int(r['Calls']) / steps:13.1f} {float(r['AverageNs']) / 1e3:8.1f} {us:9.1f} {100 * us / tot:5.1f}%")
print(f"# listed: {acc:.0f} us of {tot:.0f} us")
