#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export PYTHONPATH="$PWD"
timeout 600 python -m pytest tests/test_distributed.py -m gpu -q 2>&1 | tail -40 | cut -c1-600 | tee gpurun_out/pytest_gpu_r2g.log
