#!/bin/bash
# round 3, GPU call C: exact-K branch-free Krylov kernels (suite + A/B sweep), DGKS eta x orth_tol x chunk at 512^3
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export PYTHONPATH="$PWD"
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -60 > gpurun_out/r3c_pytest.log
tail -8 gpurun_out/r3c_pytest.log | cut -c1-400
timeout 600 python scripts/kernel_sweep.py 512 exact > gpurun_out/r3c_exact_sweep.jsonl 2> gpurun_out/r3c_exact_sweep.err
python - <<'PY'
import json, collections
t = collections.defaultdict(list)
for l in open('gpurun_out/r3c_exact_sweep.jsonl'):
    try:
        d = json.loads(l)
    except Exception:
        continue
    t[(d['kernel'], d['k'], d['exact'])].append(d['frac_of_8TBs'])
for kern in ('multidot', 'multiaxpy'):
    for k in sorted({key[1] for key in t if key[0] == kern}):
        a, b = t[(kern, k, 0)], t[(kern, k, 1)]
        print(kern, 'k=%2d' % k, 'bucketed %.3f' % (sum(a) / len(a)), 'exact %.3f' % (sum(b) / len(b)))
PY
OUT=gpurun_out/r3c_policy.jsonl
: > $OUT
run() { timeout 300 python bench.py --size 512 --steps 5 --warmup 2 --cpu-sample 0 "$@" 2>/dev/null | tail -1 >> $OUT; }
run --opt krylov_exact=0 --opt gmres_chunk=1
run --opt krylov_exact=1 --opt gmres_chunk=1
run --opt krylov_exact=1 --opt gmres_chunk=4
for eta in 0.01 0.001; do
  for ot in 1e-8 1e-5 1e-4; do
    run --opt gmres_chunk=1 --dgks-eta $eta --opt orth_tol=$ot
    run --opt gmres_chunk=4 --dgks-eta $eta --opt orth_tol=$ot
  done
done
python - <<'PY'
import json
for l in open('gpurun_out/r3c_policy.jsonl'):
    try:
        d = json.loads(l); c = d['config']; s = d.get('steady_state') or {}
        print('ms %.2f' % d['ms_per_step'], 'itlin', c['itlinear_per_step'], 'res', '%.2e' % c['residual_after_step'], 'roof', d['roofline']['kernel'], round(d['roofline']['frac'], 3), 'inner', round(d['inner_loop']['frac_of_peak'], 3),
              'steady ms %.1f it %s' % (s.get('ms_per_corrector', 0), s.get('itlinear')), {k: (round(v['ms_total'] / d['steps'], 2), round(v['gbs'] / 8000, 3)) for k, v in d['kernels'].items()})
    except Exception as e:
        print('unparsed', e, l[:300])
PY
