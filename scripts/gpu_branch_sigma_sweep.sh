#!/bin/bash
# config 5 at 256^3: eigensolve cost against the shift of the shift-invert transformation
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export PYTHONPATH="$PWD"
: > gpurun_out/branch_sigma.jsonl
for s in ${SIGMAS:-0.1 0.03 0.01 0.003}; do
timeout 400 python bench.py --workload branch --size ${SIZE:-256} --steps ${STEPS:-3} --cpu-sample 0 --eig-sigma $s 2> gpurun_out/branch_sigma_$s.err | tail -1 >> gpurun_out/branch_sigma.jsonl
done
python - <<'PY'
import json
for l in open('gpurun_out/branch_sigma.jsonl'):
    d=json.loads(l)
    print(d['config']['workload'][:90])
    for p in d['per_step']: print('  step', p['step'], '%.2fs'%p['seconds'], 'itlin', p['itlinear'], 'solves', p['eig_solves'], 'inner its', p['eig_inner_iterations'], 'conv', p['eig_converged'], ['%.6f'%x for x in p['rightmost'][:3]])
PY
