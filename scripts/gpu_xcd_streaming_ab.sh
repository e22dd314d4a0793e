#!/bin/bash
# XCD-blocked streaming of the Krylov BLAS kernels + XCD-contiguous DCT tiles: parity, A/B sweep, bench
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export PYTHONPATH="$PWD"
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "gmres or krylov or dct or arnoldi or multidot or vec" 2>&1 | tail -2
timeout 300 python scripts/kernel_sweep.py 512 xcd > gpurun_out/sweep_xcd.jsonl 2> gpurun_out/sweep_xcd.err
python - <<'PY'
import json
for l in open('gpurun_out/sweep_xcd.jsonl'):
    d=json.loads(l)
    if 'xcd_map' in d: print(d['kernel'], 'k', d['k'], 'xcd', d['xcd_map'], 'rep', d['rep'], 'ms %.3f'%d['ms'], 'frac %.3f'%d['frac_of_8TBs'])
PY
for o in "vec_xcd_map=1" "vec_xcd_map=0"; do
timeout 300 python bench.py --steps 5 --warmup 2 --cpu-sample 0 --no-steady --opt $o 2>/dev/null | tail -1 > gpurun_out/bench_p_$o.json
python - <<PY
import json
d=json.load(open('gpurun_out/bench_p_$o.json'))
print('$o', 'ms/step %.2f'%d['ms_per_step'], 'it', d['config']['itlinear_per_step'], round(d['roofline']['frac'],3), {k:round(v['ms_total']/d['steps'],2) for k,v in d['kernels'].items()})
PY
done
