#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export PYTHONPATH="$PWD"
python scripts/_dbg.py 2>&1 | grep -E "^(1|4) (20|63) " | cut -c1-200
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q 2>&1 | tail -8 | cut -c1-300
timeout 400 python bench.py --steps 5 --warmup 1 --cpu-sample 0 2>/dev/null | tail -1 > gpurun_out/bench_r2m.json
python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_r2m.json'))
print('bench', d['value'], d['ms_per_step'], d['config']['itlinear_per_step'], d['roofline']['frac'], d['inner_loop']['frac_of_peak'], 'steady', d['steady_state']['ms_per_corrector'], d['steady_state']['itlinear'])
print({k:(round(v['ms_total']/d['steps'],2), v['calls']//d['steps']) for k,v in d['kernels'].items()})
PY
