"""Summarises the SQ-counter passes of scripts/gpu_r2.sh (sq1 / sq2): per kernel, counters summed over the chip, per launch.
Units (MI355X_MICROARCH.md): SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* are quad-cycles summed over waves;
SQ_BUSY_CYCLES per SE; SQ_VALU_MFMA_BUSY_CYCLES in cycles; GRBM_GUI_ACTIVE = shader-clock cycles of the launch.
Derived: valu_busy = ACTIVE_INST_VALU / WAVE_CYCLES (share of WAVE time issuing VALU), wait_any = WAIT_ANY / WAVE_CYCLES (parked at
s_waitcnt / barrier), lds_conflict = LDS_BANK_CONFLICT / LDS_IDX_ACTIVE.
Per-unit utilisations (the counters are sums over the chip; GRBM_GUI_ACTIVE is summed over the 8 XCDs, so the launch lasts kernel_cycles = GUI_ACTIVE / 8 -- checked
against SQ_INSTS_MFMA x 64 cycles = SQ_VALU_MFMA_BUSY_CYCLES and against the traced duration x clock):
  simd_valu_util = 4 ACTIVE_INST_VALU / (kernel_cycles x 1024 SIMDs)      share of a SIMD's cycles with a VALU instruction in its pipe
  cu_lds_inst_util = 4 ACTIVE_INST_LDS / (kernel_cycles x 256 CUs)         share of a CU's cycles with an LDS instruction in flight
  cu_lds_array_util = LDS_IDX_ACTIVE / (kernel_cycles x 256 CUs)            share of a CU's cycles with the LDS array busy
  resident_waves_per_cu = 4 WAVE_CYCLES / (kernel_cycles x 256)
  mfma_util = VALU_MFMA_BUSY_CYCLES / (kernel_cycles x 1024 SIMDs)  (rounds 2-4 and profiles/r05/k_* divided by GUI_ACTIVE x 1024: 8 x too small)."""
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
    if s.get("GRBM_GUI_ACTIVE"):
        kc = s["GRBM_GUI_ACTIVE"] / 8.0
        s["kernel_cycles"] = kc
        if "SQ_VALU_MFMA_BUSY_CYCLES" in s:
            s["mfma_util"] = s["SQ_VALU_MFMA_BUSY_CYCLES"] / (kc * 1024)
        if "SQ_ACTIVE_INST_LDS" in s:
            s["cu_lds_inst_util"] = 4 * s["SQ_ACTIVE_INST_LDS"] / (kc * 256)
        if "SQ_LDS_IDX_ACTIVE" in s:
            s["cu_lds_array_util"] = s["SQ_LDS_IDX_ACTIVE"] / (kc * 256)
        # (sq1 and sq2 are separate passes: the cycle count of the sq2 pass serves the sq1 counters too -- same command, same clocks to a few per cent)
        if "SQ_ACTIVE_INST_VALU" in s:
            s["simd_valu_util"] = 4 * s["SQ_ACTIVE_INST_VALU"] / (kc * 1024)
        if wc:
            s["resident_waves_per_cu"] = 4 * wc / (kc * 256)
    summary[k] = s
print(json.dumps(summary, indent=1))
json.dump(summary, open(os.path.join(out, "sq_summary.json"), "w"), indent=1)
