#!/bin/bash
# copies what should be judged from a gpurun_out/<tag> pass (scripts/gpu_r5.sh) into profiles/<round>/<prefix>_*:   scripts/collect_profiles.sh k r05 k
TAG=$1; ROUND=$2; PRE=$3
S=gpurun_out/$TAG; D=profiles/$ROUND; mkdir -p $D
cp $S/summary.txt $D/${PRE}_summary.txt
for f in bench.json configs.json pmc_summary.json sq_summary.json c3_sq_summary.json c4_mfma_summary.json c3_phase_cycles.log; do [ -f $S/$f ] && cp $S/$f $D/${PRE}_$f; done
[ -f gpurun_out/$TAG/frontend_layer.json ] && cp gpurun_out/$TAG/frontend_layer.json $D/${PRE}_frontend_layer.json
for p in prof c3_prof c4_prof c5_prof; do
  f=$(find $S/$p -name "*kernel_stats.csv" 2>/dev/null | head -1)
  [ -n "$f" ] && head -14 "$f" > $D/${PRE}_$(echo $p | sed 's/_prof//; s/prof/step/')_kernel_stats.csv
done
ls -la $D | grep " ${PRE}_"
