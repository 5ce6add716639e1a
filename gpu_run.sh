cd $GRAFT_REPO_ROOT
make -C oracle -B >/dev/null 2>&1
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -5
python scripts/perf_probe.py 2>&1 | grep -v amdgpu.ids
