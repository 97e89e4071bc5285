"""Timeline of one hipGraph replay from a rocprofv3 --kernel-trace CSV: which kernels are on the critical path, how
much of the step no kernel is running at all, per-queue busy time.   python tools/timeline.py <kernel_trace.csv> [n_kernels_per_step]"""
import csv, sys, collections, re

path = sys.argv[1]
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r["Queue_Id"]))
rows.sort()
# a step = the kernels between two launches of the first kernel of the forward (edge_basis_fwd; --marker=NAME for another
# one: the training step starts with pack_weight_split_grouped)
marker = next((a.split("=", 1)[1] for a in sys.argv if a.startswith("--marker=")), "edge_basis_fwd")
starts = [i for i, r in enumerate(rows) if marker in r[2]]
if len(starts) < 3:
    sys.exit("no steps found")
a, b = starts[-2], starts[-1]
step = rows[a:b]
t0, t1 = step[0][0], rows[b][0]
print(f"# step of {len(step)} kernels, {1e-3 * (t1 - t0):.1f} us from first start to next step's first start")
def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n); n = re.sub(r"void ", "", n)
    return n[:60]
# union of busy intervals
busy = 0; cur_s, cur_e = None, None
for s, e, _, _ in sorted(step):
    if cur_e is None or s > cur_e:
        if cur_e is not None: busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
print(f"# some kernel running: {1e-3 * busy:.1f} us; idle (launch gaps): {1e-3 * (t1 - t0 - busy):.1f} us")
perq = collections.Counter()
for s, e, n, q in step: perq[q] += e - s
print("# busy per queue:", {q: round(1e-3 * v, 1) for q, v in perq.items()})
# gap before each kernel (to the latest end of anything earlier), biggest first
gaps = []
latest = step[0][0]
for s, e, n, q in step:
    gaps.append((s - latest, n, q, e - s)); latest = max(latest, e)
tot_pos = sum(g for g, *_ in gaps if g > 0)
print(f"# sum of positive gaps {1e-3 * tot_pos:.1f} us; by following kernel:")
byk = collections.defaultdict(lambda: [0, 0, 0])
for g, n, q, d in gaps:
    k = short(n); byk[k][0] += max(g, 0); byk[k][1] += 1; byk[k][2] += d
for k, (g, c, d) in sorted(byk.items(), key=lambda kv: -(kv[1][0] + kv[1][2]))[:40]:
    print(f"  {k:60s} n={c:3d}  run {1e-3 * d:8.1f} us  gap-before {1e-3 * g:7.1f} us  ({1e-3 * g / c:4.1f}/launch)")
if "--list" in sys.argv:
    for s, e, n, q in step:
        print(f"{1e-3 * (s - t0):9.1f} {1e-3 * (e - s):7.1f} q{q} {short(n)}")
# exclusive time: parts of the step during which exactly one kernel runs, attributed to it (a proxy for the critical path)
ev = []
for i, (s, e, n, q) in enumerate(step):
    ev.append((s, 1, i)); ev.append((e, -1, i))
ev.sort()
active = set(); excl = collections.Counter(); conc = collections.Counter(); last_t = ev[0][0]
for t, d, i in ev:
    if t > last_t:
        conc[len(active)] += t - last_t
        if len(active) == 1:
            excl[short(step[next(iter(active))][2])] += t - last_t
    last_t = t
    if d > 0: active.add(i)
    else: active.discard(i)
print("# time by number of kernels running concurrently (us):", {k: round(1e-3 * v, 1) for k, v in sorted(conc.items())})
print("# exclusive (alone on the chip) time by kernel:")
for k, v in excl.most_common(25):
    print(f"  {k:60s} {1e-3 * v:8.1f} us")
