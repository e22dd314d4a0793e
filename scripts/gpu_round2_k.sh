#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export PYTHONPATH="$PWD"
for c in 1 2 4 8; do
BK_GMRES_CHUNK=$c timeout 300 python scripts/bench_configs.py 2> gpurun_out/configs_chunk$c.err | tee gpurun_out/configs_chunk$c.jsonl | cut -c1-330
done
