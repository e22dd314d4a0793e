#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export PYTHONPATH="$PWD"
timeout 900 python -m pytest tests/test_gpu_configs.py -m gpu -q -x 2>&1 | tail -30 | tee gpurun_out/pytest_configs.log
BK_FULLSIZE=256 timeout 600 python -m pytest tests/test_gpu_fullsize.py -m gpu -q -x 2>&1 | tail -5 | tee gpurun_out/pytest_c4.log
