#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export PYTHONPATH="$PWD"
timeout 900 python -m pytest tests/test_distributed.py -m gpu -q 2>&1 | tail -25 | cut -c1-400
OUT=gpurun_out/r3r_hostcomm_lanes.jsonl
: > $OUT
for tl in 0 1; do
BK_BENCH_HOSTCOMM=1 timeout 600 python bench.py --gpus 4 --size 256 --steps 1 --warmup 1 --cpu-sample 0 --no-steady --opt two_lanes=$tl 2> gpurun_out/r3r_$tl.err | tail -1 >> $OUT
done
python - <<'PY'
import json
for l in open('gpurun_out/r3r_hostcomm_lanes.jsonl'):
    try:
        d = json.loads(l); c = d['config']
        print(c['grid'], 'ranks', d['n_gpus'], 'ms %.1f' % d['ms_per_step'], 'itlin', c['itlinear_per_step'], 'p', c['full_corrector']['p'], c['full_corrector']['residuals'])
    except Exception as e:
        print('unparsed', e, l[:300])
PY
tail -3 gpurun_out/r3r_1.err | cut -c1-300
