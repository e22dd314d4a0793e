#!/bin/bash
# persistent DCT workgroups with next-tile prefetch: parity first, then the 512^3 bench at several grid sizes

set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export PYTHONPATH="$PWD"
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "dct or precond or spectral" 2>&1 | tail -4 | tee gpurun_out/pytest_dct_n.log
timeout 900 python -m pytest tests/test_gpu_fullsize.py tests/test_distributed.py -q -x 2>&1 | tail -4 | tee -a gpurun_out/pytest_dct_n.log
for g in 512 1024 2048; do
timeout 300 python bench.py --steps 5 --warmup 2 --cpu-sample 0 --no-steady --opt dct_grid=$g 2> gpurun_out/bench_n_$g.err | tail -1 > gpurun_out/bench_n_$g.json
done
timeout 300 python bench.py --size 256 --steps 10 --warmup 3 --cpu-sample 0 --no-steady 2>/dev/null | tail -1 > gpurun_out/bench_n_256.json
python - <<'PY'
import json
for g in (512, 1024, 2048, "256"):
    try:
        d=json.load(open('gpurun_out/bench_n_%s.json'%g))
    except Exception as e:
        print(g, 'failed', e); continue
    print(g, 'ms/step %.2f'%d['ms_per_step'], 'it', d['config']['itlinear_per_step'], 'roofline', d['roofline']['kernel'] if 'kernel' in d['roofline'] else '', round(d['roofline']['frac'],3), {k:round(v['ms_total']/d['steps'],2) for k,v in d['kernels'].items()})
PY
