#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export PYTHONPATH="$PWD"
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_distributed.py -m gpu -q -k "dct or precond or distributed or host_comm or krylov or gmres or device_resident" 2>&1 | tail -8 | cut -c1-300
timeout 300 python scripts/kernel_sweep.py 512 slabemu 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print('wide1', d['R'], d['nzl'], d['slab_zsolve'], 'ms %.3f'%d['ms'])"
BK_OPTS=dct_lt_wide=0 timeout 300 python scripts/kernel_sweep.py 512 slabemu 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print('wide0', d['R'], d['nzl'], d['slab_zsolve'], 'ms %.3f'%d['ms'])"
for w in 1 0; do BK_OPTS=dct_lt_wide=$w BK_SWEEP_FAST=1 timeout 300 python scripts/kernel_sweep.py 256 precond 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print('256^3 wide$w', {k:d[k] for k in ('roundtrip','threads','ms')})"; done
