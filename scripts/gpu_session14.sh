#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export PYTHONPATH="$PWD"
timeout 600 python -m pytest tests -m gpu -q -k "dct or precond or fullsize or dist" 2>&1 | tail -15 | tee gpurun_out/pytest_dct.log
rm -f gpurun_out/sweep14.log
for f in 256 0; do
  echo "== dct_fused=$f" | tee -a gpurun_out/sweep14.log
  BK_OPTS="dct_fused=$f" timeout 300 python scripts/kernel_sweep.py 512 precond 2>&1 | grep -v amdgpu.ids | tail -4 | tee -a gpurun_out/sweep14.log
done
timeout 600 python bench.py --steps 3 --warmup 1 --cpu-sample 0 2>&1 | tail -1 | cut -c1-400 | tee gpurun_out/bench14.log
