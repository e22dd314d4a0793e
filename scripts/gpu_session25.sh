#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export PYTHONPATH="$PWD"
timeout 1200 python bench.py 2>&1 | tail -1 > gpurun_out/bench_r1k.log
python - <<'PY'
import json
d = json.loads(open('gpurun_out/bench_r1k.log').read())
print(d['value'], d['ms_per_step']); print(d['roofline']); print(d['cpu_baseline'])
PY
