#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export PYTHONPATH="$PWD"
timeout 600 python scripts/debug_eig.py 2>&1 | tail -32 | tee gpurun_out/debug_eig.log
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -40 | tee gpurun_out/pytest_gpu.log
timeout 900 python bench.py --cpu-sample 0 2>&1 | tail -1 > gpurun_out/bench_r1d.log
