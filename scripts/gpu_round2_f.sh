#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export PYTHONPATH="$PWD"
timeout 900 python -m pytest tests/test_distributed.py tests/test_gpu_configs.py -m gpu -q --durations=8 2>&1 | tail -40 | cut -c1-400 | tee gpurun_out/pytest_gpu_r2f.log
