#!/bin/bash
# Full evidence run: GPU tests, smoke, default bench line, rocprofv3 kernel stats and PMC (FETCH_SIZE / WRITE_SIZE) passes.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
TAG=${1:-r3}
mkdir -p gpurun_out
export PYTHONPATH="$PWD"
[ "${SKIP_TESTS:-0}" = 1 ] || timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -30 | tee gpurun_out/pytest_gpu.log
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -2 | tee gpurun_out/smoke.log
timeout 900 python bench.py 2> gpurun_out/bench_${TAG}.err | tail -1 > gpurun_out/bench_${TAG}.log
cd /tmp && export TMPDIR=/tmp
R="$OLDPWD"
timeout 900 rocprofv3 --kernel-trace --stats -d "$R/gpurun_out/prof_${TAG}" -- python "$R/bench.py" --steps 3 --warmup 1 --cpu-sample 0 --no-steady > "$R/gpurun_out/prof_${TAG}.log" 2>&1
timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d "$R/gpurun_out/pmc_fetch_${TAG}" -- python "$R/bench.py" --steps 1 --warmup 0 --cpu-sample 0 --no-steady > "$R/gpurun_out/pmc_fetch_${TAG}.log" 2>&1
timeout 900 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d "$R/gpurun_out/pmc_write_${TAG}" -- python "$R/bench.py" --steps 1 --warmup 0 --cpu-sample 0 --no-steady > "$R/gpurun_out/pmc_write_${TAG}.log" 2>&1
cd "$R"
python scripts/prof_summary.py gpurun_out/prof_${TAG} 300 > gpurun_out/prof_${TAG}_summary.txt 2>&1
head -16 gpurun_out/prof_${TAG}_summary.txt | cut -c1-170
python scripts/pmc_summary.py gpurun_out/pmc_fetch_${TAG} gpurun_out/pmc_write_${TAG} gpurun_out/${TAG} | cut -c1-150
