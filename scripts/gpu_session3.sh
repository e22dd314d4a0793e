#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export PYTHONPATH="$PWD"
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -80 | tee gpurun_out/pytest_gpu.log
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -3 | tee gpurun_out/smoke.log
timeout 600 python scripts/kernel_sweep.py 512 precond,axpy 2>&1 | tee gpurun_out/sweep3.log | tail -40
for eta in 0.7071 0.1 0.01; do
  timeout 600 python bench.py --size 512 --steps 2 --warmup 1 --cpu-sample 0 --dgks-eta $eta 2>&1 | tail -1 > gpurun_out/bench512_eta$eta.log
done
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d "$OLDPWD/gpurun_out/prof_r1" -- python "$OLDPWD/bench.py" --size 512 --steps 2 --warmup 1 --cpu-sample 0 > "$OLDPWD/gpurun_out/prof_bench.log" 2>&1
cd "$OLDPWD"; find gpurun_out/prof_r1 -name "*stats*" | head; ls -la gpurun_out/prof_r1/* | head -20
