#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export PYTHONPATH="$PWD"
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -60 | tee gpurun_out/pytest_gpu.log
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -2 | tee gpurun_out/smoke.log
timeout 900 python bench.py 2>&1 | tail -1 > gpurun_out/bench_default.log
BK_FORCE_DIST=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --size 128 --steps 1 --warmup 0 --cpu-sample 0 2>&1 | tail -3 > gpurun_out/bench_forcedist.log
tail -c 600 gpurun_out/bench_forcedist.log
