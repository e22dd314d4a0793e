#!/bin/bash
# one-off GPU command runner: bash scripts/gpu_one.sh '<command>' (output also lands in gpurun_out/one.log)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export PYTHONPATH="$PWD"; mkdir -p gpurun_out
bash -c "$1" 2>&1 | tee gpurun_out/one.log | tail -${TAILN:-40}
