cd /tmp && export TMPDIR=/tmp
for aa in 0 1; do
  mkdir -p $GRAFT_REPO_ROOT/gpurun_out/c4ab_$aa
  timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/c4ab_$aa -o t --output-format csv -- python $GRAFT_REPO_ROOT/scripts/sdp_c4_probe.py 1024 1e-4 $aa 2>&1 | grep -E "^path|^bwd" 
  grep -E "k_sa_" $(find $GRAFT_REPO_ROOT/gpurun_out/c4ab_$aa -name "*kernel_stats.csv" | head -1) | cut -d, -f1-4 | sed 's/(DevT.*)"//' | cut -c1-80
done
find $GRAFT_REPO_ROOT/gpurun_out -name "*.db" -delete
