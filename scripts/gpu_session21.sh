#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export PYTHONPATH="$PWD"
timeout 900 python -m pytest tests -m gpu -q -x -k "dist or two_rank or share" 2>&1 | tail -30 | tee gpurun_out/pytest_dist.log
BK_BENCH_HOSTCOMM=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 1 --warmup 1 --cpu-sample 0 2>&1 | tail -1 > gpurun_out/bench_hostcomm2.log
python - <<'PY'
import json
d = json.loads(open('gpurun_out/bench_hostcomm2.log').read())
print(d['value'], d['ms_per_step'], d['config'].get('full_corrector'))
for k, v in d['kernels'].items(): print(k, round(v['ms_total'],1), v['calls'], round(v['gbs']))
PY
