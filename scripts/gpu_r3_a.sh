#!/bin/bash
# round 3, GPU call A: full GPU test suite + smoke, default bench line, device-resident Arnoldi chunks at slab / full
# size (gmres_chunk 1 / 2 / 4 with the convergence-predicted speculation), bare-shell 8-rank launch over the host communicator
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export PYTHONPATH="$PWD"
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -40 > gpurun_out/r3a_pytest.log
tail -5 gpurun_out/r3a_pytest.log
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -2
timeout 600 python bench.py --steps 10 --warmup 3 2> gpurun_out/r3a_bench.err | tail -1 > gpurun_out/r3a_bench_512.json
OUT=gpurun_out/r3a_chunks.jsonl
: > $OUT
for ch in 1 2 4; do
  timeout 300 python bench.py --size 512 --size-z 64 --steps 20 --warmup 5 --cpu-sample 0 --no-steady --opt gmres_chunk=$ch 2>/dev/null | tail -1 >> $OUT
done
for ch in 1 4; do
  timeout 300 python bench.py --size 512 --steps 6 --warmup 2 --cpu-sample 0 --no-steady --opt gmres_chunk=$ch 2>/dev/null | tail -1 >> $OUT
  timeout 300 python bench.py --size 256 --steps 10 --warmup 3 --cpu-sample 0 --no-steady --opt gmres_chunk=$ch 2>/dev/null | tail -1 >> $OUT
done
BK_BENCH_HOSTCOMM=1 timeout 600 python bench.py --gpus 8 --size 256 --steps 1 --warmup 1 --cpu-sample 0 --no-steady 2> gpurun_out/r3a_hostcomm8.err | tail -1 > gpurun_out/r3a_hostcomm8_bare_shell.json
python - <<'PY'
import json
def show(path):
    for l in open(path):
        try:
            d = json.loads(l); c = d['config']
            print(c['grid'], 'ranks', d['n_gpus'], 'ms %.2f' % d['ms_per_step'], 'itlin', c['itlinear_per_step'], 'p', c['full_corrector']['p'],
                  'roof', d['roofline'] and (d['roofline']['kernel'], round(d['roofline']['frac'], 3)), 'inner', d['inner_loop'] and round(d['inner_loop']['frac_of_peak'], 3),
                  {k: round(v['ms_total'] / d['steps'], 2) for k, v in d['kernels'].items()}, 'comm', d.get('comm'))
        except Exception as e:
            print('unparsed', e, l[:300])
for p in ('gpurun_out/r3a_bench_512.json', 'gpurun_out/r3a_chunks.jsonl', 'gpurun_out/r3a_hostcomm8_bare_shell.json'):
    print(p); show(p)
PY
tail -3 gpurun_out/r3a_hostcomm8.err
