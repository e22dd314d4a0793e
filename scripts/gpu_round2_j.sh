#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export PYTHONPATH="$PWD"
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_distributed.py -m gpu -q -x 2>&1 | tail -15 | cut -c1-300 | tee gpurun_out/pytest_gpu_r2j.log
for c in 1 4 8; do
BK_GMRES_CHUNK=$c timeout 300 python scripts/bench_configs.py 2> gpurun_out/configs_chunk$c.err | tee gpurun_out/configs_chunk$c.jsonl | cut -c1-420
done
