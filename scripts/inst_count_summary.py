"""usage: inst_count_summary.py <dir>   (dir holds it<mi>_<pass>/ trees made by scripts/_run_n.sh)"""
import sys, os, glob, csv, collections, json, re
out = sys.argv[1]
res = collections.defaultdict(dict)
for d in sorted(glob.glob(os.path.join(out, "it*_*"))):
    mi = int(re.match(r"it(\d+)_", os.path.basename(d)).group(1))
    per = collections.defaultdict(lambda: collections.defaultdict(float)); names = {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_fwd2" not in r["Kernel_Name"]: continue
            per[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
    acc = collections.defaultdict(list)
    for disp, cs in per.items():
        for c, v in cs.items(): acc[c].append(v)
    for c, v in acc.items(): res[mi][c] = sum(v) / len(v)
W = 4096 * 4
mis = sorted(res)
print(json.dumps({str(k): v for k, v in res.items()}, indent=1))
for c in sorted(res[mis[0]]):
    row = [res[mi].get(c, float("nan")) / W for mi in mis]
    line = f"{c:28s} per wave: " + "  ".join(f"it{mi}={v:10.1f}" for mi, v in zip(mis, row))
    if len(mis) >= 2: line += f"   per wave-iteration ({mis[0]}->{mis[-1]}): {(row[-1] - row[0]) / (mis[-1] - mis[0]):8.2f}"
    print(line)
