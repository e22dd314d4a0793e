#!/bin/bash
# z round trip with the hoisted / folded symbol: parity, the C2 corrector test, per-kernel durations
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export PYTHONPATH="$PWD"
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -q -x -k "dct or precond or c2_sh2d" 2>&1 | tail -3
SKIPTEST=1 VARIANTS="dct_xcd_map=3" bash scripts/gpu_round2_o.sh
