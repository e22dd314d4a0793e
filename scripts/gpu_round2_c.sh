#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export PYTHONPATH="$PWD"
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "linesearch or thick_start" 2>&1 | tail -80 | cut -c1-400 | tee gpurun_out/pytest_gpu_r2c.log
timeout 400 python scripts/c3_fullsize.py 1024 eig,hopf > gpurun_out/c3_full.jsonl 2> gpurun_out/c3_full.err
tail -c 600 gpurun_out/c3_full.err; cut -c1-1800 gpurun_out/c3_full.jsonl
timeout 600 python bench.py --workload branch --steps 2 --cpu-sample 0 --eig-dim 60 2> gpurun_out/branch512_r2c.err | tail -1 > gpurun_out/branch512_r2c.json
tail -c 300 gpurun_out/branch512_r2c.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/branch512_r2c.json'))
print(d['config']['workload'][60:220]); print('  init', d['config']['initialisation'])
for p in d['per_step']: print('  ', {k:(round(v,4) if isinstance(v,float) else v) for k,v in p.items() if k!='rightmost'})
PY
