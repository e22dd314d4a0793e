#!/bin/bash
# The exact geometry of the driver's N = 8 / 4 scaling runs (512^3 on 8 / 4 z-slabs, slab z-solve of the preconditioner) and of
# BASELINE config 4 (256^3 on 8 slabs), with the ranks sharing ONE GPU over the host-staged communicator: everything except
# the RCCL calls themselves.  Each line must reproduce the 1-rank corrector (itlinear, p, residual).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export PYTHONPATH="$PWD"
OUT=gpurun_out/hostcomm_ranks.jsonl
: > $OUT
one() {   # size ranks
    timeout ${3:-600} python -m torch.distributed.run --nnodes=1 --nproc-per-node $2 --master-addr 127.0.0.1 --master-port $((29600 + $2 + $1 / 64)) \
        bench.py --gpus $2 --size $1 --steps 1 --warmup 1 --cpu-sample 0 --no-steady 2> gpurun_out/hostcomm_$1_$2.err | tail -1 >> $OUT
    tail -c 400 gpurun_out/hostcomm_$1_$2.err | grep -v "^$" | tail -3
}
timeout 300 python bench.py --size 256 --steps 3 --warmup 1 --cpu-sample 0 2>/dev/null | tail -1 > gpurun_out/bench256_1rank.json
cat gpurun_out/bench256_1rank.json >> $OUT
export BK_BENCH_HOSTCOMM=1
one 256 8 400
one 512 8 900
# one 512 4 900     (4 ranks: covered at small size by tests/test_distributed.py)
python - <<'PY'
import json
for l in open('gpurun_out/hostcomm_ranks.jsonl'):
    try:
        d = json.loads(l)
        c = d['config']
        print(c['grid'], 'ranks', d['n_gpus'], 'ms %.1f' % d['ms_per_step'], 'itlinear', c['itlinear_per_step'], 'res', c['residual_after_step'],
              'p', c['full_corrector']['p'], 'full itlinear', c['full_corrector']['itlinear'], 'cell p', c['cell_corrector']['p'])
    except Exception as e:
        print('unparsed line', e, l[:200])
PY
