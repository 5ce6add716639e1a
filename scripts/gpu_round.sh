#!/bin/bash
# One GPU-box pass: parity tests, smoke, bench line, rocprofv3 kernel-trace stats and (separately) the HBM PMC passes.
# usage (from the repo root on the GPU box):  bash scripts/gpu_round.sh <tag>
# Everything lands in gpurun_out/<tag>/ ; copy what should be judged into profiles/.
TAG=${1:-r01}
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
make -C oracle -B >/dev/null 2>&1
echo "== pytest -m gpu" | tee "$OUT/summary.txt"
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 | tee -a "$OUT/summary.txt"
echo "== smoke" | tee -a "$OUT/summary.txt"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -3 | tee -a "$OUT/summary.txt"
echo "== bench" | tee -a "$OUT/summary.txt"
timeout 300 python bench.py 2>"$OUT/bench.err" | tee "$OUT/bench.json" | cut -c1-600 | tee -a "$OUT/summary.txt"
echo "== bench eps=1e-8" | tee -a "$OUT/summary.txt"
timeout 300 python bench.py --eps 1e-8 --no-cpu 2>>"$OUT/bench.err" | tee "$OUT/bench_eps8.json" | cut -c1-300 | tee -a "$OUT/summary.txt"
echo "== bench --accel 1 (Anderson acceleration on both sides)" | tee -a "$OUT/summary.txt"
timeout 300 python bench.py --accel 1 2>>"$OUT/bench.err" | tee "$OUT/bench_accel1.json" | cut -c1-300 | tee -a "$OUT/summary.txt"
(cd /tmp && timeout 240 rocprofv3 --kernel-trace --stats -d "$OUT/prof_accel1" -o trace --output-format csv -- python "$OLDPWD/bench.py" --no-cpu --accel 1 > "$OUT/prof_accel1.log" 2>&1)
find "$OUT/prof_accel1" -name "*kernel_stats.csv" | head -1 | xargs -r head -4 | tee -a "$OUT/summary.txt"
echo "== rocprofv3 kernel-trace stats" | tee -a "$OUT/summary.txt"
(cd /tmp && timeout 240 rocprofv3 --kernel-trace --stats -d "$OUT/prof" -o trace --output-format csv -- python "$OLDPWD/bench.py" --no-cpu > "$OUT/prof.log" 2>&1)
find "$OUT/prof" -name "*kernel_stats.csv" | head -1 | xargs -r head -12 | tee -a "$OUT/summary.txt"
echo "== pmc FETCH_SIZE / WRITE_SIZE (separate passes)" | tee -a "$OUT/summary.txt"
(cd /tmp && timeout 240 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$OUT/pmc_fetch" -o p --output-format csv -- python "$OLDPWD/bench.py" --no-cpu --steps 2 --warmup 1 > "$OUT/pmc_fetch.log" 2>&1)
(cd /tmp && timeout 240 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$OUT/pmc_write" -o p --output-format csv -- python "$OLDPWD/bench.py" --no-cpu --steps 2 --warmup 1 > "$OUT/pmc_write.log" 2>&1)
python scripts/pmc_summary.py "$OUT" 2>&1 | tee -a "$OUT/summary.txt"
# keep the merge small: drop raw databases
find "$OUT" -name "*.db" -delete
du -sh "$OUT" | tee -a "$OUT/summary.txt"
