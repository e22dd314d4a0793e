#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export PYTHONPATH="$PWD"
rm -f gpurun_out/branch20.log
for nat in 0 1; do
timeout 300 python scripts/run_branch.py 64 8 15 $nat 2>&1 | tail -1 | tee -a gpurun_out/branch20.log
done
timeout 600 python scripts/run_branch.py 256 3 15 1 2>&1 | tail -4 | tee -a gpurun_out/branch20.log
