#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export PYTHONPATH="$PWD"
timeout 2000 python -m pytest tests -m gpu -q 2>&1 | tail -40 | tee gpurun_out/pytest_gpu.log
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -3 | tee gpurun_out/smoke.log
timeout 600 python scripts/kernel_sweep.py 512 precond 2>&1 | tee gpurun_out/sweep5.log | tail -6
timeout 900 python bench.py --size 512 --steps 3 --warmup 1 --cpu-sample 0 2>&1 | tail -1 > gpurun_out/bench512_r1c.log
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d "$OLDPWD/gpurun_out/prof_r1c" -- python "$OLDPWD/bench.py" --size 512 --steps 2 --warmup 1 --cpu-sample 0 > "$OLDPWD/gpurun_out/prof_bench_c.log" 2>&1
timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d "$OLDPWD/gpurun_out/pmc_fetch" -- python "$OLDPWD/bench.py" --size 512 --steps 1 --warmup 0 --cpu-sample 0 > "$OLDPWD/gpurun_out/pmc_fetch.log" 2>&1
timeout 900 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d "$OLDPWD/gpurun_out/pmc_write" -- python "$OLDPWD/bench.py" --size 512 --steps 1 --warmup 0 --cpu-sample 0 > "$OLDPWD/gpurun_out/pmc_write.log" 2>&1
cd "$OLDPWD"
python scripts/prof_summary.py gpurun_out/prof_r1c 300 > gpurun_out/prof_r1c_summary.txt 2>&1
python scripts/prof_summary.py gpurun_out/pmc_fetch 300 > gpurun_out/pmc_fetch_summary.txt 2>&1
python scripts/prof_summary.py gpurun_out/pmc_write 300 > gpurun_out/pmc_write_summary.txt 2>&1
rm -rf gpurun_out/prof_r1 gpurun_out/prof_r1b
head -30 gpurun_out/prof_r1c_summary.txt; tail -30 gpurun_out/pmc_fetch_summary.txt
