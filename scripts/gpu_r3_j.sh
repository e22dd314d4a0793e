#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export PYTHONPATH="$PWD"
timeout 900 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_parity.py -m gpu -q -k "precond or roundtrip or fullsize or dct" 2>&1 | tail -15 | cut -c1-300
OUT=gpurun_out/r3j_bench.jsonl
: > $OUT
run() { timeout 400 python bench.py --steps 8 --warmup 2 --cpu-sample 0 --no-steady "$@" 2>/dev/null | tail -1 >> $OUT; }
run --size 512 --opt dct_zblock=0
run --size 512 --opt dct_zblock=1
run --size 256 --opt dct_zblock=0
run --size 256 --opt dct_zblock=1
python - <<'PY'
import json
for l in open('gpurun_out/r3j_bench.jsonl'):
    try:
        d = json.loads(l); c = d['config']
        print(c['grid'], 'ms %.2f' % d['ms_per_step'], 'itlin', c['itlinear_per_step'], 'ms/app %.3f' % (d['ms_per_step'] / c['itlinear_per_step']), 'p', c['full_corrector']['p'],
              {k: (round(v['ms_total'] / d['steps'], 2), round(v['gbs'] / 8000, 3)) for k, v in d['kernels'].items()})
    except Exception as e:
        print('unparsed', e, l[:300])
PY
cd /tmp && export TMPDIR=/tmp
R="$OLDPWD"
for zb in 0 1; do
timeout 600 rocprofv3 --kernel-trace --stats -d "$R/gpurun_out/prof_zb$zb" -- python "$R/bench.py" --steps 3 --warmup 1 --cpu-sample 0 --no-steady --opt dct_zblock=$zb > "$R/gpurun_out/prof_zb$zb.log" 2>&1
python "$R/scripts/prof_summary.py" "$R/gpurun_out/prof_zb$zb" 300 2>/dev/null | grep -i "dct\|TOTAL\|Name" | head -12 | cut -c1-200
done
