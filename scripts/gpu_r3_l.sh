#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export PYTHONPATH="$PWD"
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -40 > gpurun_out/r3l_pytest.log
tail -12 gpurun_out/r3l_pytest.log | cut -c1-300
OUT=gpurun_out/r3l_bench.jsonl
: > $OUT
run() { timeout 400 python bench.py --steps 8 --warmup 2 --cpu-sample 0 "$@" 2>/dev/null | tail -1 >> $OUT; }
run --size 512 --opt gmres_chunk=1
run --size 512
run --size 512 --size-z 64 --no-steady --opt gmres_chunk=1
run --size 512 --size-z 64 --no-steady
python - <<'PY'
import json
for l in open('gpurun_out/r3l_bench.jsonl'):
    try:
        d = json.loads(l); c = d['config']; s = d.get('steady_state') or {}
        print(c['grid'], 'ms %.2f' % d['ms_per_step'], 'itlin', c['itlinear_per_step'], 'ms/app %.3f' % (d['ms_per_step'] / c['itlinear_per_step']), 'p', c['full_corrector']['p'], c['full_corrector']['residuals'],
              'steady ms %.1f it %s' % (s.get('ms_per_corrector', 0), s.get('itlinear')), {k: (round(v['ms_total'] / d['steps'], 2), round(v['gbs'] / 8000, 3)) for k, v in d['kernels'].items()})
    except Exception as e:
        print('unparsed', e, l[:300])
PY
