#!/bin/bash
# stability of the whole GPU suite on the final tree: N consecutive runs (the driver's round-end run must not meet a flaky test)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export PYTHONPATH="$PWD"; mkdir -p gpurun_out
for i in $(seq 1 ${REPS:-3}); do
  timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -3 | tee gpurun_out/suite_rep_$i.log
done
