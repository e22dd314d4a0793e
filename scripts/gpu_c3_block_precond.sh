#!/bin/bash
# cGL 2x2-block spectral preconditioner: parity tests, C3 at full size (eigensolve + Hopf run) with both preconditioners
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export PYTHONPATH="$PWD"
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "cgl" 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_configs.py -q -x -k "c3" 2>&1 | tail -3
timeout 600 python scripts/c3_fullsize.py 1024 eig,hopf block > gpurun_out/c3_block.jsonl 2> gpurun_out/c3_block.err
timeout 300 python scripts/bench_configs.py > gpurun_out/configs_s.jsonl 2> gpurun_out/configs_s.err
python - <<'PY'
import json
for f in ('gpurun_out/c3_block.jsonl', 'gpurun_out/configs_s.jsonl'):
    for l in open(f):
        d=json.loads(l)
        print({k:(round(v,4) if isinstance(v,float) else v) for k,v in d.items() if k not in ('vals','exact_real','param','steps','rstar','residuals','n_unstable','n_imag')})
PY
