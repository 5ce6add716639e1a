#!/bin/bash
# PMC passes over scripts/pmc_probe.py (forward kernel at 101 and 201 iterations): per-iteration instruction / stall counts
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/pmc_${1:-x}
mkdir -p $OUT
(cd /tmp && rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES -d $OUT/p1 -o p --output-format csv -- python $OLDPWD/scripts/pmc_probe.py > $OUT/p1.log 2>&1)
(cd /tmp && rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_FLAT -d $OUT/p2 -o p --output-format csv -- python $OLDPWD/scripts/pmc_probe.py > $OUT/p2.log 2>&1)
(cd /tmp && rocprofv3 --kernel-trace --pmc SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_INSTS_VMEM GRBM_GUI_ACTIVE SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA -d $OUT/p3 -o p --output-format csv -- python $OLDPWD/scripts/pmc_probe.py > $OUT/p3.log 2>&1)
python - <<PY
import csv, glob, collections
for d in ("p1","p2","p3"):
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % d, recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(float))
        for r in csv.DictReader(open(f)):
            if "k_f" in r["Kernel_Name"]:
                agg[int(r["Dispatch_Id"])][r["Counter_Name"]] += float(r["Counter_Value"])
        ks = sorted(agg)
        if len(ks) >= 2:
            a, b = agg[ks[-2]], agg[ks[-1]]
            print(d, "per instance-iteration (B=4096, 100 iterations):")
            for c in sorted(b): print("   %-26s %12.1f   (launch@201: %.3e)" % (c, (b[c] - a[c]) / (4096 * 100.0), b[c]))
PY
find $OUT -name "*.db" -delete
