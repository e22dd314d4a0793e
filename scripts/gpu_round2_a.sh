#!/bin/bash
# GPU tests, default bench line (with the steady-state record), config-5 branch workload at 512^3.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export PYTHONPATH="$PWD"
timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -25 | tee gpurun_out/pytest_gpu_r2a.log
timeout 400 python bench.py --steps 5 --warmup 1 2> gpurun_out/bench_r2a.err | tail -1 > gpurun_out/bench_r2a.json
tail -c 400 gpurun_out/bench_r2a.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_r2a.json'))
print('bench', d['value'], d['ms_per_step'], d['roofline']['frac'], d['inner_loop']['frac_of_peak'])
print('steady', json.dumps(d['steady_state'])[:600])
PY
timeout 900 python bench.py --workload branch --steps 3 --cpu-sample 0 2> gpurun_out/branch512_r2a.err | tail -1 > gpurun_out/branch512_r2a.json
tail -c 400 gpurun_out/branch512_r2a.err
cut -c1-1500 gpurun_out/branch512_r2a.json
