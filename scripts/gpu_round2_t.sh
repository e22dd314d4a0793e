#!/bin/bash
# checkpoints + block preconditioner + continuation tests after the host-logic changes
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export PYTHONPATH="$PWD"
timeout 1200 python -m pytest tests/test_gpu_parity.py -q -x -k "continuation or checkpoint or cgl or hopf or bisect or branch" 2>&1 | tail -4
