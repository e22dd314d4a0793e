#!/bin/bash
# two-lane bordered solve: suite, then A/B on the configs that do not saturate HBM by themselves
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export PYTHONPATH="$PWD"
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -15 > gpurun_out/r3q_pytest.log
tail -8 gpurun_out/r3q_pytest.log | cut -c1-300
OUT=gpurun_out/r3q_two_lanes.jsonl
: > $OUT
for tl in 0 1; do
  timeout 300 python bench.py --size 256 --steps 10 --warmup 3 --cpu-sample 0 --opt two_lanes=$tl 2>/dev/null | tail -1 >> $OUT
  timeout 300 python bench.py --size 512 --size-z 64 --steps 20 --warmup 5 --cpu-sample 0 --no-steady --opt two_lanes=$tl 2>/dev/null | tail -1 >> $OUT
  timeout 300 python bench.py --size 512 --steps 5 --warmup 2 --cpu-sample 0 --no-steady --opt two_lanes=$tl 2>/dev/null | tail -1 >> $OUT
done
python - <<'PY'
import json
for l in open('gpurun_out/r3q_two_lanes.jsonl'):
    try:
        d = json.loads(l); c = d['config']; s = d.get('steady_state') or {}
        print(c['grid'], 'ms %.2f' % d['ms_per_step'], 'itlin', c['itlinear_per_step'], 'p', c['full_corrector']['p'], 'steady %.2f' % s.get('ms_per_corrector', 0),
              {k: round(v['ms_total'] / d['steps'], 2) for k, v in d['kernels'].items()})
    except Exception as e:
        print('unparsed', e, l[:200])
PY
for tl in 0 1; do BK_OPTS_TWO=$tl timeout 300 python - <<PY 2>/dev/null | tail -2 | cut -c1-400
import os, sys
sys.argv = ['x']
os.environ['BK_GMRES_CHUNK'] = ''
exec(open('scripts/bench_configs.py').read().replace('ctx = hip.Context(0)', 'ctx = hip.Context(0); ctx.set_option("two_lanes", $tl)'))
PY
done
