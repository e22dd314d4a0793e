#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export PYTHONPATH="$PWD"
BK_BENCH_HOSTCOMM=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29519 bench.py --gpus 2 --size 128 --steps 1 --warmup 0 --cpu-sample 0 2>&1 | tail -4 > gpurun_out/bench_2rank_hostcomm.log
tail -c 1800 gpurun_out/bench_2rank_hostcomm.log
timeout 900 python bench.py --size 128 --steps 1 --warmup 0 --cpu-sample 0 2>&1 | tail -1 > gpurun_out/bench_128_1rank.log
timeout 1500 python scripts/run_branch.py 256 3 15 2>&1 | tail -12 | tee gpurun_out/branch256.log
