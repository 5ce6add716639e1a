"""Summarises rocprofv3 output of scripts/gpu_round.sh: per-kernel mean duration from the kernel trace and
HBM bytes per launch from the FETCH_SIZE / WRITE_SIZE passes (MI355X_MICROARCH.md 'HBM': both counters are in
KiB-like units of the TCC EA request counters; on gfx950 FETCH_SIZE under-reports wide coalesced reads by 2x, so the
read side is doubled before it is compared with a byte count)."""
import collections
import csv
import glob
import json
import os
import sys

out = sys.argv[1]


def short(name):
    for k in ("k_forward_rt", "k_forward", "k_backward", "k_transpose", "k_fwd", "k_bwd"):
        if k in name:
            return k
    return name[:48]


def kernel_trace(d):
    res = collections.defaultdict(list)
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            res[short(r["Kernel_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6)
    return res


def pmc(d, counter):
    res = collections.defaultdict(list)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        per = collections.defaultdict(float)
        names = {}
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                per[r["Dispatch_Id"]] += float(r["Counter_Value"])
                names[r["Dispatch_Id"]] = short(r["Kernel_Name"])
        for k, v in per.items():
            res[names[k]].append(v)
    return res


summary = {}
kt = kernel_trace(os.path.join(out, "prof"))
for k, v in kt.items():
    if k.startswith("k_"):
        summary.setdefault(k, {})["mean_ms"] = sum(v) / len(v)
        summary[k]["launches"] = len(v)
for counter, d, mult in (("FETCH_SIZE", "pmc_fetch", 2.0), ("WRITE_SIZE", "pmc_write", 1.0)):
    for k, v in pmc(os.path.join(out, d), counter).items():
        if k.startswith("k_"):
            # counters are reported in KiB (rocprofv3 derived metric: requests * 64 B / 1024)
            summary.setdefault(k, {})[counter + "_bytes_per_launch"] = mult * 1024.0 * sum(v) / len(v)
            summary[k][counter + "_raw_mean"] = sum(v) / len(v)
print(json.dumps(summary, indent=1))
json.dump(summary, open(os.path.join(out, "pmc_summary.json"), "w"), indent=1)
