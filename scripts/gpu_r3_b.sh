#!/bin/bash
# round 3, GPU call B: full GPU suite (no -x), then the single-pass Gram-Schmidt budget (orth_tol) x device chunks at 512^3
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export PYTHONPATH="$PWD"
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -60 > gpurun_out/r3b_pytest.log
tail -25 gpurun_out/r3b_pytest.log | cut -c1-400
OUT=gpurun_out/r3b_orth.jsonl
: > $OUT
for ot in 1e-8 1e-6 1e-5 1e-4; do
  for ch in 1 4; do
    timeout 300 python bench.py --size 512 --steps 5 --warmup 2 --cpu-sample 0 --opt gmres_chunk=$ch --opt orth_tol=$ot 2>/dev/null | tail -1 >> $OUT
  done
done
for ot in 1e-8 1e-5; do
  timeout 300 python bench.py --size 512 --size-z 64 --steps 20 --warmup 5 --cpu-sample 0 --no-steady --opt gmres_chunk=4 --opt orth_tol=$ot 2>/dev/null | tail -1 >> $OUT
done
python - <<'PY'
import json
for l in open('gpurun_out/r3b_orth.jsonl'):
    try:
        d = json.loads(l); c = d['config']; s = d.get('steady_state') or {}
        print(c['grid'], 'ms %.2f' % d['ms_per_step'], 'itlin', c['itlinear_per_step'], 'res', '%.2e' % c['residual_after_step'], 'full', c['full_corrector']['itlinear'], c['full_corrector']['residuals'][-1],
              'steady ms %.1f it %s' % (s.get('ms_per_corrector', 0), s.get('itlinear')), {k: round(v['ms_total'] / d['steps'], 2) for k, v in d['kernels'].items()})
    except Exception as e:
        print('unparsed', e, l[:300])
PY
