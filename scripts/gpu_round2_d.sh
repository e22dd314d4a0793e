#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export PYTHONPATH="$PWD"
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "linesearch or thick_start or dparam" 2>&1 | tail -30 | cut -c1-300 | tee gpurun_out/pytest_gpu_r2d.log
timeout 400 python scripts/kernel_sweep.py 512 variants > gpurun_out/sweep_r2d.jsonl 2> gpurun_out/sweep_r2d.err
tail -c 300 gpurun_out/sweep_r2d.err
python - <<'PY'
import json
for l in open('gpurun_out/sweep_r2d.jsonl'):
    d=json.loads(l)
    print(d['kernel'], {k:v for k,v in d.items() if k not in('kernel','n','frac_of_8TBs','ms','gbs')}, 'ms %.3f'%d['ms'], 'GB/s %.0f'%d['gbs'], '%.1f%%'%(100*d['frac_of_8TBs']))
PY
