#!/bin/bash
# 2-rank bench over the host communicator (slab z-solve default) vs 1 rank at the same size; new surface tests
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export PYTHONPATH="$PWD"
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "eigarpack or linesearch" 2>&1 | tail -15 | cut -c1-300
timeout 300 python bench.py --size 256 --steps 2 --warmup 1 --cpu-sample 0 --no-steady 2>/dev/null | tail -1 > gpurun_out/bench256_1rank.json
for slab in 1 0; do
BK_BENCH_HOSTCOMM=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --size 256 --steps 2 --warmup 1 --cpu-sample 0 --no-steady --opt dct_dist_slab=$slab 2> gpurun_out/bench256_2rank_slab$slab.err | tail -1 > gpurun_out/bench256_2rank_slab$slab.json
tail -c 300 gpurun_out/bench256_2rank_slab$slab.err
done
python - <<'PY'
import json
for f in ('bench256_1rank','bench256_2rank_slab1','bench256_2rank_slab0'):
    try:
        d=json.load(open('gpurun_out/%s.json'%f))
        print(f, d['n_gpus'], 'ms %.1f'%d['ms_per_step'], 'itlinear', d['config']['itlinear_per_step'], 'res', d['config']['residual_after_step'], 'p', d['config']['full_corrector']['p'], {k: round(v['ms_total']/d['steps'],2) for k,v in d['kernels'].items()})
    except Exception as e: print(f, 'failed', e)
PY
