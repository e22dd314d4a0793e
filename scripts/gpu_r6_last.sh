#!/bin/bash
# round 6, last run on the final tree: the whole GPU suite + smoke, then rocprofv3 kernel stats / PMC / SQ and the bench lines (scripts/gpu_r6_profile.sh)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export PYTHONPATH="$PWD"; mkdir -p gpurun_out
bash scripts/gpu_tests.sh 2>&1 | tail -8 | cut -c1-200
bash scripts/gpu_r6_profile.sh 2>&1 | tail -12 | cut -c1-400
