#!/bin/bash
# Round-6 GPU-box pass (same stages as gpu_r2.sh).  usage: bash scripts/gpu_r6.sh <tag> [stages...]   stages: tests bench prof pmc sq list configs
# Everything lands in gpurun_out/<tag>/ ; copy what should be judged into profiles/r06/.
TAG=${1:-r02}; shift
STAGES="${*:-tests bench}"
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
has() { [[ " $STAGES " == *" $1 "* ]]; }
# a broken box (first kernel of every process faults, then hangs until its timeout) once ate 25 GPU-minutes: check the device first and bail out
if ! timeout 180 python -c "import torch; assert float(torch.ones(1024, device='cuda').sum().item()) == 1024.0" > "$OUT/device_check.log" 2>&1; then echo "device check FAILED: aborting the pass" | tee "$OUT/summary.txt"; tail -3 "$OUT/device_check.log"; exit 3; fi
BENCH_ARGS=${BENCH_ARGS:-}
if has tests; then
  echo "== pytest -m gpu" | tee "$OUT/summary.txt"
  timeout ${TEST_TIMEOUT:-600} python -m pytest tests -m gpu -q ${PYTEST_ARGS--x} --tb=short --durations=8 2>&1 | grep -v "Warning\|warnings.warn\|note_ignored_args" | tail -400 > "$OUT/pytest.log"; grep -E "^E  |^FAILED|passed|failed" "$OUT/pytest.log" | cut -c1-600 | tail -40 | tee -a "$OUT/summary.txt"
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
if has frontend; then
  echo "== whole layer (frontend: parameter maps + plugin + one-launch recovery)" | tee -a "$OUT/summary.txt"
  timeout 300 python scripts/bench_frontend.py "$OUT/frontend_layer.json" 2>&1 | grep -v amdgpu.ids | tail -4 | tee -a "$OUT/summary.txt"
fi
if has c3prof; then
  echo "== C3 (SOCP n=100, 10 x SOC(11), B=4096): kernel trace, SQ counters, in-kernel phase cycles" | tee -a "$OUT/summary.txt"
  (cd /tmp && CONFIGS=C3 timeout 300 rocprofv3 --kernel-trace --stats -d "$OUT/c3_prof" -o trace --output-format csv -- python "$OLDPWD/scripts/bench_configs.py" "$OUT/c3_configs.json" > "$OUT/c3_prof.log" 2>&1)
  find "$OUT/c3_prof" -name "*kernel_stats.csv" | head -1 | xargs -r head -6 | cut -c1-260 | tee -a "$OUT/summary.txt"
  (cd /tmp && CONFIGS=C3 timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT -d "$OUT/c3_sq1" -o p --output-format csv -- python "$OLDPWD/scripts/bench_configs.py" "$OUT/c3_configs_sq.json" > "$OUT/c3_sq1.log" 2>&1)
  python - "$OUT" <<'PY' | tee -a "$OUT/summary.txt"
import csv, glob, collections, json, os, re, sys
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(out, "c3_sq1", "**", "*counter_collection.csv"), recursive=True):
    per = collections.defaultdict(lambda: collections.defaultdict(float)); names = {}
    for r in csv.DictReader(open(f)):
        per[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"]); mm = re.search(r"\bk_\w+", r["Kernel_Name"]); names[r["Dispatch_Id"]] = mm.group(0) if mm else r["Kernel_Name"][:40]
    for d, cs in per.items():
        for c, v in cs.items(): agg[names[d]][c].append(v)
res = {}
for k, cs in agg.items():
    if not any(t in k for t in ("k_fwd2", "k_backward")): continue
    s = {c: sum(v) / len(v) for c, v in cs.items()}
    s["launches"] = max(len(v) for v in cs.values())
    if s.get("SQ_WAVE_CYCLES"):
        s["valu_busy"] = s.get("SQ_ACTIVE_INST_VALU", 0.0) / s["SQ_WAVE_CYCLES"]; s["wait_any"] = s.get("SQ_WAIT_ANY", 0.0) / s["SQ_WAVE_CYCLES"]
    res[k] = s
print(json.dumps(res, indent=1))
json.dump(res, open(os.path.join(out, "c3_sq_summary.json"), "w"), indent=1)
PY
  CE_ENGINE_SO=$PWD/cvxpylayers_amd/csrc/libcone_engine_timing.so timeout 300 python scripts/timing_probe.py C3 4096 2>&1 | grep -v amdgpu.ids | head -40 | tee "$OUT/c3_phase_cycles.log" | head -30 | tee -a "$OUT/summary.txt"
fi
if has c5prof; then
  echo "== C5 (portfolio n=501, shared A, B=16384): kernel trace" | tee -a "$OUT/summary.txt"
  (cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d "$OUT/c5_prof" -o trace --output-format csv -- python "$OLDPWD/scripts/shared_a_probe.py" C5 16384 default > "$OUT/c5_prof.log" 2>&1)
  find "$OUT/c5_prof" -name "*kernel_stats.csv" | head -1 | xargs -r head -6 | cut -c1-260 | tee -a "$OUT/summary.txt"
  grep variant "$OUT/c5_prof.log" | tee -a "$OUT/summary.txt"
fi
find "$OUT" -name "*.db" -delete
du -sh "$OUT" | tee -a "$OUT/summary.txt"
if has c4prof; then
  echo "== C4 (SDP 20x20, shared A): kernel trace + MFMA counters" | tee -a "$OUT/summary.txt"
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d "$OUT/c4_prof" -o trace --output-format csv -- python "$OLDPWD/scripts/sdp_c4_probe.py" 1024 1e-4 > "$OUT/c4_prof.log" 2>&1)
  find "$OUT/c4_prof" -name "*kernel_stats.csv" | head -1 | xargs -r head -8 | tee -a "$OUT/summary.txt"
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_MFMA SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d "$OUT/c4_mfma" -o p --output-format csv -- python "$OLDPWD/scripts/sdp_c4_probe.py" 1024 1e-4 > "$OUT/c4_mfma.log" 2>&1)
  python - "$OUT" <<'PY' | tee -a "$OUT/summary.txt"
import csv, glob, collections, json, os, re, sys
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(out, "c4_mfma", "**", "*counter_collection.csv"), recursive=True):
    per = collections.defaultdict(lambda: collections.defaultdict(float)); names = {}
    for r in csv.DictReader(open(f)):
        per[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"]); mm = re.search(r"\bk_\w+", r["Kernel_Name"]); names[r["Dispatch_Id"]] = mm.group(0) if mm else r["Kernel_Name"][:40]
    for d, cs in per.items():
        for c, v in cs.items(): agg[names[d]][c].append(v)
res = {}
for k, cs in agg.items():
    if not any(t in k for t in ("k_sa_", "k_ca_", "k_fwd2", "k_backward")): continue
    s = {c: sum(v) / len(v) for c, v in cs.items()}
    s["launches"] = max(len(v) for v in cs.values())
    # GRBM_GUI_ACTIVE is summed over the 8 XCDs, SQ_VALU_MFMA_BUSY_CYCLES over the 1024 SIMDs of the device: busy share of one SIMD =
    # MFMA-busy / (1024 x kernel cycles); flops: SQ_INSTS_MFMA x 2048 (v_mfma_f64_16x16x4_f64)
    if s.get("GRBM_GUI_ACTIVE"): s["kernel_cycles"] = s["GRBM_GUI_ACTIVE"] / 8; s["mfma_util_per_simd"] = s.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (1024 * s["kernel_cycles"])
    s["mfma_f64_flops_per_launch"] = s.get("SQ_INSTS_MFMA", 0.0) * 2048
    if s.get("SQ_WAVE_CYCLES"): s["valu_busy_share_of_wave_time"] = s.get("SQ_ACTIVE_INST_VALU", 0.0) / s["SQ_WAVE_CYCLES"]
    res[k] = s
print(json.dumps(res, indent=1))
json.dump(res, open(os.path.join(out, "c4_mfma_summary.json"), "w"), indent=1)
PY
fi
find "$OUT" -name "*.db" -delete
