#!/bin/bash
# fused MINRES passes (Lanczos axpy + dot in the stencil kernel, r . M^-1 r from the spectrum): parity tests, then config 5 A/B
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export PYTHONPATH="$PWD"
timeout 900 python -m pytest tests -m gpu -x -q -k "symmetric_krylov or fused_minres or minres or hermitian or branch or eig" 2>&1 | tail -8
for f in 0 1; do
timeout 600 python bench.py --workload branch --size 256 --steps 4 --opt minres_fused=$f 2> gpurun_out/r3u_branch256_f$f.err | tail -1 > gpurun_out/r3u_branch256_fused$f.json
python - <<PY
import json
d=json.load(open('gpurun_out/r3u_branch256_fused$f.json'))
print('minres_fused=$f ms/step %.1f'%d['ms_per_step'], [ (s['eig_solves'], s['eig_inner_iterations'], round(s['seconds'],3)) for s in d['per_step']], {k:(round(v['ms_total']/d['steps'],1), round(v.get('frac_of_peak',0),2)) for k,v in d.get('kernels',{}).items()})
PY
done
