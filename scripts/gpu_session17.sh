#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export PYTHONPATH="$PWD"
timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -5 | tee gpurun_out/pytest_gpu.log
timeout 300 python scripts/dct_trace.py 512 2>&1 | grep -v amdgpu.ids | tee gpurun_out/dct_trace.log
BK_SWEEP_FAST=1 timeout 300 python scripts/kernel_sweep.py 512 precond 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-200 | tee gpurun_out/sweep17.log
timeout 600 python bench.py --steps 3 --warmup 1 --cpu-sample 0 2>&1 | tail -1 > gpurun_out/bench17.log
python - <<'PY'
import json
d = json.loads(open('gpurun_out/bench17.log').read())
print(d['value'], d['ms_per_step'], d['roofline'])
for k, v in d['kernels'].items(): print(k, v)
PY
