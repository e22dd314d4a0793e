#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export PYTHONPATH="$PWD"
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "hopf or shift_invert" 2>&1 | tail -30 | tee gpurun_out/pytest_hopf.log
timeout 600 python scripts/kernel_sweep.py 512 axpy 2>&1 | tee gpurun_out/sweep7.log | tail -30
