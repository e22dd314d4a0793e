#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export PYTHONPATH="$PWD"
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -40 | tee gpurun_out/pytest_gpu.log
timeout 900 python bench.py 2>&1 | tail -1 > gpurun_out/bench_r1e.log
timeout 600 python scripts/kernel_sweep.py 512 precond 2>&1 | tee gpurun_out/sweep10.log | tail -5
