#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export PYTHONPATH="$PWD"
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -60 | tee gpurun_out/pytest_gpu.log
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -3 | tee gpurun_out/smoke.log
timeout 600 python scripts/kernel_sweep.py 512 precond 2>&1 | tee gpurun_out/sweep_precond.log | tail
timeout 900 python bench.py --size 512 --steps 3 --warmup 1 --cpu-sample 64 2>&1 | tail -2 | tee gpurun_out/bench512.log
