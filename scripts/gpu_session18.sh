#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export PYTHONPATH="$PWD"
timeout 600 python -m pytest tests -m gpu -q -x -k "dct or precond or fullsize or dist" 2>&1 | tail -5 | tee gpurun_out/pytest_dct.log
timeout 300 python scripts/dct_trace.py 512 2>&1 | grep -v amdgpu.ids | tee gpurun_out/dct_trace.log
for o in "dct_fused_ax0=1" "dct_fused_ax0=0"; do
echo "== $o"
BK_OPTS="$o" BK_SWEEP_FAST=1 timeout 300 python scripts/kernel_sweep.py 512 precond 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-200
done | tee gpurun_out/sweep18.log
