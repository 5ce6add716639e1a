"""Summarises the SQ-counter passes of scripts/gpu_r2.sh (sq1 / sq2): per kernel, counters summed over the chip, per launch.
Units (MI355X_MICROARCH.md): SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* are quad-cycles summed over waves;
SQ_BUSY_CYCLES per SE; SQ_VALU_MFMA_BUSY_CYCLES in cycles; GRBM_GUI_ACTIVE = shader-clock cycles of the launch.
Derived: valu_busy = ACTIVE_INST_VALU / WAVE_CYCLES (share of wave time issuing VALU), wait_any = WAIT_ANY / WAVE_CYCLES (parked at
s_waitcnt / barrier), lds_conflict = LDS_BANK_CONFLICT / LDS_IDX_ACTIVE, mfma_util = VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE * 4 SIMD * 256 CU)."""
import collections
import csv
import glob
import json
import os
import sys

out = sys.argv[1]


def short(name):
    for k in ("k_forward_rt", "k_forward", "k_backward_rt", "k_backward", "k_transpose", "k_fwd2", "k_sa_", "k_ca_"):
        if k in name:
            return name[name.index(k):].split("<")[0].split("(")[0]
    return name[:40]


agg = collections.defaultdict(lambda: collections.defaultdict(list))
for d in ("sq1", "sq2"):
    for f in glob.glob(os.path.join(out, d, "**", "*counter_collection.csv"), recursive=True):
        per = collections.defaultdict(lambda: collections.defaultdict(float))
        names = {}
        for r in csv.DictReader(open(f)):
            per[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
            names[r["Dispatch_Id"]] = short(r["Kernel_Name"])
        for disp, cs in per.items():
            for c, v in cs.items():
                agg[names[disp]][c].append(v)
summary = {}
for k, cs in agg.items():
    if not k.startswith("k_"):
        continue
    s = {c: sum(v) / len(v) for c, v in cs.items()}
    s["launches"] = max(len(v) for v in cs.values())
    wc = s.get("SQ_WAVE_CYCLES")
    if wc:
        for name, c in (("valu_busy", "SQ_ACTIVE_INST_VALU"), ("wait_any", "SQ_WAIT_ANY"), ("wait_inst_any", "SQ_WAIT_INST_ANY"), ("active_inst_any", "SQ_ACTIVE_INST_ANY")):
            if c in s:
                s[name] = s[c] / wc
    if s.get("SQ_LDS_IDX_ACTIVE"):
        s["lds_conflict"] = s.get("SQ_LDS_BANK_CONFLICT", 0.0) / s["SQ_LDS_IDX_ACTIVE"] if "SQ_LDS_BANK_CONFLICT" in s else None
    if s.get("GRBM_GUI_ACTIVE") and "SQ_VALU_MFMA_BUSY_CYCLES" in s:
        s["mfma_util"] = s["SQ_VALU_MFMA_BUSY_CYCLES"] / (s["GRBM_GUI_ACTIVE"] * 4 * 256)
    summary[k] = s
print(json.dumps(summary, indent=1))
json.dump(summary, open(os.path.join(out, "sq_summary.json"), "w"), indent=1)
