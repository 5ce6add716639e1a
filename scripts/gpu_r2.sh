#!/bin/bash
# Round-2 GPU-box pass.  usage: bash scripts/gpu_r2.sh <tag> [stages...]   stages: tests bench prof pmc sq list configs
# Everything lands in gpurun_out/<tag>/ ; copy what should be judged into profiles/r02/.
TAG=${1:-r02}; shift
STAGES="${*:-tests bench}"
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
has() { [[ " $STAGES " == *" $1 "* ]]; }
BENCH_ARGS=${BENCH_ARGS:-}
if has tests; then
  echo "== pytest -m gpu" | tee "$OUT/summary.txt"
  timeout ${TEST_TIMEOUT:-1500} python -m pytest tests -m gpu -q ${PYTEST_ARGS:--x} --durations=15 2>&1 | tail -40 | tee "$OUT/pytest.log" | tail -25 | tee -a "$OUT/summary.txt"
  echo "== smoke" | tee -a "$OUT/summary.txt"
  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -3 | tee -a "$OUT/summary.txt"
fi
if has bench; then
  echo "== bench $BENCH_ARGS" | tee -a "$OUT/summary.txt"
  timeout 400 python bench.py $BENCH_ARGS 2>"$OUT/bench.err" | tee "$OUT/bench.json" | cut -c1-700 | tee -a "$OUT/summary.txt"
fi
if has prof; then
  echo "== rocprofv3 kernel-trace stats" | tee -a "$OUT/summary.txt"
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d "$OUT/prof" -o trace --output-format csv -- python "$OLDPWD/bench.py" --no-cpu $BENCH_ARGS > "$OUT/prof.log" 2>&1)
  find "$OUT/prof" -name "*kernel_stats.csv" | head -1 | xargs -r head -12 | tee -a "$OUT/summary.txt"
fi
if has pmc; then
  echo "== pmc FETCH_SIZE / WRITE_SIZE (separate passes)" | tee -a "$OUT/summary.txt"
  (cd /tmp && timeout 240 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$OUT/pmc_fetch" -o p --output-format csv -- python "$OLDPWD/bench.py" --no-cpu --steps 2 --warmup 1 $BENCH_ARGS > "$OUT/pmc_fetch.log" 2>&1)
  (cd /tmp && timeout 240 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$OUT/pmc_write" -o p --output-format csv -- python "$OLDPWD/bench.py" --no-cpu --steps 2 --warmup 1 $BENCH_ARGS > "$OUT/pmc_write.log" 2>&1)
  python scripts/pmc_summary.py "$OUT" 2>&1 | tee -a "$OUT/summary.txt"
fi
if has sq; then
  echo "== SQ counters (two passes over bench.py --steps 2)" | tee -a "$OUT/summary.txt"
  (cd /tmp && timeout 240 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT -d "$OUT/sq1" -o p --output-format csv -- python "$OLDPWD/bench.py" --no-cpu --steps 2 --warmup 1 $BENCH_ARGS > "$OUT/sq1.log" 2>&1)
  (cd /tmp && timeout 240 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE -d "$OUT/sq2" -o p --output-format csv -- python "$OLDPWD/bench.py" --no-cpu --steps 2 --warmup 1 $BENCH_ARGS > "$OUT/sq2.log" 2>&1)
  python scripts/sq_summary.py "$OUT" 2>&1 | tee -a "$OUT/summary.txt"
fi
if has list; then
  (cd /tmp && timeout 60 rocprofv3 -L > "$OUT/counters_all.txt" 2>&1); grep -i -o "SQ_[A-Z_0-9]*MFMA[A-Z_0-9]*\|SQ_INSTS_VALU_[A-Z0-9_]*" "$OUT/counters_all.txt" | sort -u | tee "$OUT/counters_mfma.txt" | head -40
fi
if has configs; then
  echo "== BASELINE configs" | tee -a "$OUT/summary.txt"
  timeout 900 python scripts/bench_configs.py "$OUT/configs.json" 2>&1 | grep -v amdgpu.ids | tail -12 | tee -a "$OUT/summary.txt"
fi
find "$OUT" -name "*.db" -delete
du -sh "$OUT" | tee -a "$OUT/summary.txt"
