#!/bin/bash
# quick GPU check: parity tests + kernel timing probe
cd "${GRAFT_REPO_ROOT:-/root/repo}"
make -C oracle -B >/dev/null 2>&1
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -8
timeout 600 python scripts/perf_probe.py 2>&1 | grep -v amdgpu.ids
if [ -n "$AB" ]; then CE_FWD=rt timeout 600 python scripts/perf_probe.py 2>&1 | grep -v amdgpu.ids | tail -4; fi
