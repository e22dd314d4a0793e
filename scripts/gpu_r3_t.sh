#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export PYTHONPATH="$PWD"
timeout 900 python -m pytest tests -m gpu -q -k "cgl or c3 or hopf or lap or dst or cshift or complex" 2>&1 | tail -6 | cut -c1-300
for mf in 0 1; do
echo "dst_mfma=$mf"
timeout 300 python - <<PY 2>/dev/null | tail -3 | cut -c1-420
exec(open('scripts/bench_configs.py').read().replace('ctx = hip.Context(0)', 'ctx = hip.Context(0); ctx.set_option("dst_mfma", $mf)'))
PY
done
timeout 600 python scripts/c3_fullsize.py 1024 eig block 2>/dev/null | tail -3 | cut -c1-500
