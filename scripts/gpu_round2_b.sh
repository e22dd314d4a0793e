#!/bin/bash
# remaining GPU tests (after the first failure of call a) + eigensolver experiments at 256^3
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export PYTHONPATH="$PWD"
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -25 | tee gpurun_out/pytest_gpu_r2b.log
for cfg in "1 0 0" "1 1 0" "1 1 60" "0 1 60"; do
  set -- $cfg
  timeout 300 python bench.py --size 256 --workload branch --steps 3 --cpu-sample 0 --eig-thick $1 --eig-inexact $2 --eig-dim $3 2>> gpurun_out/branch256_r2b.err | tail -1 >> gpurun_out/branch256_r2b.jsonl
done
python - <<'PY'
import json
for l in open('gpurun_out/branch256_r2b.jsonl'):
    d=json.loads(l)
    print(d['config']['workload'][60:200])
    print('  init', d['config']['initialisation'])
    for p in d['per_step']: print('  ', {k:(round(v,4) if isinstance(v,float) else v) for k,v in p.items() if k!='rightmost'}, [round(x,7) for x in p['rightmost']])
PY
tail -c 300 gpurun_out/branch256_r2b.err
