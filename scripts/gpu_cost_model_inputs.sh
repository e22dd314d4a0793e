#!/bin/bash
# measured local costs for the multi-GPU cost model (scripts/multigpu_cost_model.py): a 512^3 bench first (reference line and
# warm clocks), then the corrector step on the z-slab one of 8 / 4 / 2 ranks owns, then the local part of one preconditioner
# application on those slabs with and without the slab z-solve
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export PYTHONPATH="$PWD"
timeout 300 python bench.py --steps 5 --warmup 2 --cpu-sample 0 --no-steady 2>/dev/null | tail -1 > gpurun_out/bench_slab512.json
for nz in 64 128 256; do
# one lane (what an RCCL run does by default) and two lanes (option two_lanes = 1: opt-in on ranks)
timeout 300 python bench.py --size 512 --size-z $nz --steps 20 --warmup 5 --cpu-sample 0 --no-steady --opt two_lanes=0 2> gpurun_out/bench_slab$nz.err | tail -1 > gpurun_out/bench_slab$nz.json
[ -z "${COST_NO_LANES:-}" ] && timeout 300 python bench.py --size 512 --size-z $nz --steps 20 --warmup 5 --cpu-sample 0 --no-steady --opt two_lanes=1 2>> gpurun_out/bench_slab$nz.err | tail -1 > gpurun_out/bench_slab${nz}_two_lanes.json
done
timeout 300 python scripts/kernel_sweep.py 512 slabemu > gpurun_out/slabemu.jsonl 2> gpurun_out/slabemu.err
python - <<'PY'
import json
import os
for nz in (512, 64, '64_two_lanes', 128, '128_two_lanes', 256, '256_two_lanes'):
    if not os.path.exists('gpurun_out/bench_slab%s.json'%nz): continue
    d=json.load(open('gpurun_out/bench_slab%s.json'%nz))
    print('slab', nz, 'ms/step %.2f'%d['ms_per_step'], 'itlinear', d['config']['itlinear_per_step'], {k:round(v['ms_total']/d['steps'],2) for k,v in d['kernels'].items()})
for l in open('gpurun_out/slabemu.jsonl'):
    d=json.loads(l); print(d['R'], d['nzl'], d['slab_zsolve'], 'ms %.3f'%d['ms'])
PY
