#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export PYTHONPATH="$PWD"
timeout 900 python -m pytest tests -m gpu -q -k "eig or shiftinvert or shift_invert or hopf or krylovkit or branch or continuation or cont or bisection or c3" 2>&1 | tail -8 | cut -c1-300
OUT=gpurun_out/r3p_branch.jsonl
: > $OUT
for g in 0 1; do
timeout 600 python bench.py --workload branch --size 256 --steps 4 --opt eig_gram=$g 2>/dev/null | tail -1 >> $OUT
done
python - <<'PY'
import json
for l in open('gpurun_out/r3p_branch.jsonl'):
    d = json.loads(l)
    print('s/step %.3f' % (d['ms_per_step'] / 1e3), 'init %.2f' % d['config']['initialisation']['seconds'], d['config']['initialisation']['eig_solves'],
          [(p['eig_solves'], p['eig_inner_iterations'], round(p['seconds'], 3), [round(x, 8) for x in p['rightmost'][:3]]) for p in d['per_step']])
PY
