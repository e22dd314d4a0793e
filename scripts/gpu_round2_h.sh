#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export PYTHONPATH="$PWD"
timeout 300 python scripts/kernel_sweep.py 512 slabemu > gpurun_out/slabemu.jsonl 2> gpurun_out/slabemu.err
tail -c 300 gpurun_out/slabemu.err; cat gpurun_out/slabemu.jsonl
for nz in 64 128 256; do
timeout 300 python bench.py --size 512 --size-z $nz --steps 5 --warmup 1 --cpu-sample 0 --no-steady 2> gpurun_out/bench_slab$nz.err | tail -1 > gpurun_out/bench_slab$nz.json
tail -c 200 gpurun_out/bench_slab$nz.err
python - $nz <<'PY'
import json,sys
d=json.load(open('gpurun_out/bench_slab%s.json'%sys.argv[1]))
print('slab', sys.argv[1], 'ms/step', d['ms_per_step'], 'itlinear', d['config']['itlinear_per_step'])
for k,v in sorted(d['kernels'].items(), key=lambda kv:-kv[1]['ms_total']):
    print(f"  {k:10s} {v['ms_total']/d['steps']:8.2f} ms/step calls/step {v['calls']/d['steps']:.0f} avg {v['avg_ms']:.3f} ms  {v['gbs']:.0f} GB/s")
PY
done
