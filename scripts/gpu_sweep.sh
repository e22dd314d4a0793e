#!/bin/bash
# bash scripts/gpu_sweep.sh <what> [size]: one kernel_sweep.py selection, printed compactly
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export PYTHONPATH="$PWD"
W=${1:-variants}; S=${2:-512}
timeout 400 python scripts/kernel_sweep.py $S $W > gpurun_out/sweep_$W.jsonl 2> gpurun_out/sweep_$W.err
tail -c 300 gpurun_out/sweep_$W.err
python - "$W" <<'PY'
import json,sys
for l in open('gpurun_out/sweep_%s.jsonl'%sys.argv[1]):
    d=json.loads(l)
    print(d['kernel'], {k:v for k,v in d.items() if k not in('kernel','n','frac_of_8TBs','ms','gbs')}, 'ms %.3f'%d['ms'], 'GB/s %.0f'%d['gbs'], '%.1f%%'%(100*d['frac_of_8TBs']))
PY
