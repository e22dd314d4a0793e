#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export PYTHONPATH="$PWD"
timeout 2000 python -m pytest tests -m gpu -q 2>&1 | tail -60 | tee gpurun_out/pytest_gpu.log
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -3 | tee gpurun_out/smoke.log
timeout 600 python scripts/kernel_sweep.py 512 precond 2>&1 | tee gpurun_out/sweep4.log | tail -4
timeout 900 python bench.py --size 512 --steps 3 --warmup 1 --cpu-sample 96 2>&1 | tail -1 > gpurun_out/bench512_r1b.log
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d "$OLDPWD/gpurun_out/prof_r1b" -- python "$OLDPWD/bench.py" --size 512 --steps 2 --warmup 1 --cpu-sample 0 > "$OLDPWD/gpurun_out/prof_bench_b.log" 2>&1
cd "$OLDPWD"; python - <<'PY'
import sqlite3, glob
f = glob.glob('gpurun_out/prof_r1b/**/*.db', recursive=True)[0]
cur = sqlite3.connect(f).cursor()
rows = cur.execute("select start, end-start, grid_x, lds_size, vgpr_count from kernels where name like '%dct_fft%' and (end-start) > 300000 order by start limit 12").fetchall()
print("dct passes us:", [round(r[1]/1e3) for r in rows], rows[0][2:] if rows else None)
PY
