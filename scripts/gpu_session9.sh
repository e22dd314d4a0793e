#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export PYTHONPATH="$PWD"
for o in "axpy_nt=1" "axpy_nt=0" "axpy_nt=0 --opt axpy_blocks=4096"; do
  echo "== $o"
  timeout 600 python bench.py --size 256 --steps 1 --warmup 0 --cpu-sample 0 --opt $o 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print(d['ms_per_step'], d['config']['itlinear_per_step'], d['config']['full_corrector']['residuals'], d['config']['cell_corrector']['itlinear'])"
done
timeout 1200 python -m pytest tests -m gpu -q -x -k "dct or fullsize or distributed" 2>&1 | tail -8
