#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/pmcb_${1:-x}
mkdir -p $OUT
(cd /tmp && rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES -d $OUT/p1 -o p --output-format csv -- python $OLDPWD/scripts/pmc_probe_bwd.py > $OUT/p1.log 2>&1)
(cd /tmp && rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_FLAT -d $OUT/p2 -o p --output-format csv -- python $OLDPWD/scripts/pmc_probe_bwd.py > $OUT/p2.log 2>&1)
tail -2 $OUT/p1.log
python - <<PY
import csv, glob, collections
for d in ("p1","p2"):
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % d, recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(float))
        for r in csv.DictReader(open(f)):
            if "k_backward" in r["Kernel_Name"]:
                agg[int(r["Dispatch_Id"])][r["Counter_Name"]] += float(r["Counter_Value"])
        ks = sorted(agg)
        b = agg[ks[-1]]
        print(d, "backward per instance (B=4096):")
        for c in sorted(b): print("   %-26s %12.1f" % (c, b[c] / 4096.0))
PY
find $OUT -name "*.db" -delete
