#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export PYTHONPATH="$PWD"
timeout 900 python -m pytest tests -m gpu -q -x -k "complex_shift" 2>&1 | tail -15 | tee gpurun_out/pytest_cs.log
