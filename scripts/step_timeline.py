"""Timeline of one benchmark step from a rocprofv3 kernel trace (csv): per kernel of the step its duration and the idle gap in front of it, averaged over the steps
found.  usage: step_timeline.py <kernel_trace.csv>"""
import csv, sys, re, collections
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def short(nm):
    m = re.search(r"\bk_\w+", nm)
    if m: return m.group(0)
    m = re.search(r"(\w+)<", nm) or re.search(r"(\w+)\(", nm)
    return (m.group(1) if m else nm)[:48]
# steps start at a k_transpose followed (within a few kernels) by k_fwd2
names = [short(r["Kernel_Name"]) for r in rows]
starts = [i for i, nm in enumerate(names) if nm.startswith("k_fwd2")]
steps = []
for a, b in zip(starts[:-1], starts[1:]):
    steps.append((a, b))
steps = steps[len(steps) // 2:]          # the timed half
agg = collections.OrderedDict(); gaps = collections.defaultdict(list); durs = collections.defaultdict(list); tot = []
for a, b in steps:
    t_prev_end = None
    for k in range(a, b):
        r = rows[k]; s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        key = (k - a, names[k])
        if t_prev_end is not None: gaps[key].append(s - t_prev_end)
        durs[key].append(e - s); t_prev_end = e
    tot.append(int(rows[b]["Start_Timestamp"]) - int(rows[a]["Start_Timestamp"]))
common = [k for k in durs if len(durs[k]) >= 0.8 * len(steps)]
print(f"{len(steps)} steps, mean step (k_fwd2 start to next k_fwd2 start) {sum(tot) / len(tot) / 1e3:.1f} us")
sk = sg = 0.0
for k in sorted(common):
    d = sum(durs[k]) / len(durs[k]) / 1e3; g = (sum(gaps[k]) / len(gaps[k]) / 1e3) if gaps[k] else 0.0
    sk += d; sg += g
    print(f"  {k[0]:2d} {k[1]:48s} gap before {g:8.1f} us   kernel {d:8.1f} us")
print(f"  sum of kernels {sk:.1f} us, sum of gaps inside the step {sg:.1f} us (the gap in front of k_fwd2 closes the cycle: {sum(tot) / len(tot) / 1e3 - sk - sg:.1f} us)")
