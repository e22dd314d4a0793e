#!/bin/bash
# One GPU-box session: parity tests, smoke, kernel sweep, bench, rocprof.  Everything lands in gpurun_out/.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export PYTHONPATH="$PWD"
echo "== rocminfo"; rocminfo | grep -E "Marketing Name|Compute Unit|Max Clock" | head -6
echo "== pytest gpu"
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -40 | tee gpurun_out/pytest_gpu.log
echo "== smoke"
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -5 | tee gpurun_out/smoke.log
echo "== kernel sweep"
timeout 900 python scripts/kernel_sweep.py ${SWEEP_SIZE:-512} ${SWEEP_WHAT:-jvp,krylov,blas} 2>&1 | tee gpurun_out/sweep.log | tail -60
echo "== bench"
timeout 900 python bench.py --size ${BENCH_SIZE:-256} --steps 2 --warmup 1 --cpu-sample ${CPU_SAMPLE:-64} 2>&1 | tail -3 | tee gpurun_out/bench.log
