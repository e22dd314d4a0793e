#!/bin/bash
# round 3: full GPU suite + burst-kernel A/B + bench on the adopted defaults
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export PYTHONPATH="$PWD"
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -60 > gpurun_out/r3h_pytest.log
tail -8 gpurun_out/r3h_pytest.log | cut -c1-400
timeout 600 python scripts/kernel_sweep.py 512 burst > gpurun_out/r3h_burst.jsonl 2>/dev/null
python - <<'PY'
import json, collections
t = collections.defaultdict(list)
for l in open('gpurun_out/r3h_burst.jsonl'):
    try:
        d = json.loads(l)
    except Exception:
        continue
    t[(d['kernel'], d['k'], d['burst'])].append(d['frac_of_8TBs'])
for kern in ('multidot', 'multiaxpy'):
    print(kern, ' '.join('k=%d %.3f->%.3f' % (k, sum(t[(kern, k, 0)]) / 2, sum(t[(kern, k, 1)]) / 2) for k in sorted({key[1] for key in t if key[0] == kern})))
PY
OUT=gpurun_out/r3h_bench.jsonl
: > $OUT
run() { timeout 400 python bench.py --size 512 --steps 8 --warmup 2 --cpu-sample 0 "$@" 2>/dev/null | tail -1 >> $OUT; }
run --opt krylov_burst=0
run
run --opt gmres_chunk=4
run --opt gmres_chunk=4 --opt orth_tol=1e-5
python - <<'PY'
import json
for l in open('gpurun_out/r3h_bench.jsonl'):
    try:
        d = json.loads(l); c = d['config']; s = d.get('steady_state') or {}
        print('ms %.2f' % d['ms_per_step'], 'itlin', c['itlinear_per_step'], 'ms/app %.3f' % (d['ms_per_step'] / c['itlinear_per_step']), 'roof', d['roofline']['kernel'], round(d['roofline']['frac'], 3), 'inner', round(d['inner_loop']['frac_of_peak'], 3),
              'steady ms %.1f it %s' % (s.get('ms_per_corrector', 0), s.get('itlinear')), {k: (round(v['ms_total'] / d['steps'], 2), round(v['gbs'] / 8000, 3)) for k, v in d['kernels'].items()})
    except Exception as e:
        print('unparsed', e, l[:300])
PY
