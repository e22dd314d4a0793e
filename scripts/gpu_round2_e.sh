#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export PYTHONPATH="$PWD"
timeout 300 python scripts/kernel_sweep.py 512 nt2 > gpurun_out/sweep_r2e.jsonl 2> gpurun_out/sweep_r2e.err
tail -c 300 gpurun_out/sweep_r2e.err
python - <<'PY'
import json
for l in open('gpurun_out/sweep_r2e.jsonl'):
    d=json.loads(l)
    print(d['kernel'], {k:v for k,v in d.items() if k not in('kernel','n','frac_of_8TBs','ms','gbs')}, 'ms %.3f'%d['ms'], 'GB/s %.0f'%d['gbs'], '%.1f%%'%(100*d['frac_of_8TBs']))
PY
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -12 | cut -c1-300 | tee gpurun_out/pytest_gpu_r2e.log
timeout 400 python bench.py --steps 5 --warmup 1 --cpu-sample 0 2> gpurun_out/bench_r2e.err | tail -1 > gpurun_out/bench_r2e.json
python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_r2e.json'))
print('bench', d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['frac'], d['inner_loop']['frac_of_peak'])
for k,v in sorted(d['kernels'].items(), key=lambda kv:-kv[1]['ms_total']):
    print(f"  {k:10s} {v['ms_total']/d['steps']:8.2f} ms/step avg {v['avg_ms']:.3f} ms  {v['gbs']:.0f} GB/s ({v['gbs']/80:.1f}%)")
print('steady', d['steady_state']['ms_per_corrector'], d['steady_state']['itlinear'])
PY
