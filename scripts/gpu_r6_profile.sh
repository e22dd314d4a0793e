#!/bin/bash
# round 6, final kernel sources: rocprofv3 kernel stats / PMC traffic / SQ counters, then -- with that PMC file in profiles/ -- the bench lines
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export PYTHONPATH="$PWD"; mkdir -p gpurun_out
export TAG=r6
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "minres or krylovls or symmetric or eig" 2>&1 | tail -3
bash scripts/gpu_evidence.sh prof sq 2>&1 | tail -24 | cut -c1-200
cp gpurun_out/r6_pmc_hbm_traffic.json gpurun_out/r6_pmc_hbm_traffic.txt profiles/          # (the box's scratch copy: bench.py reads it for roofline.traffic)
timeout 900 python bench.py --block-log 2> gpurun_out/r6_bench.err | tail -1 > gpurun_out/r6_bench_512_1gpu.json
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 2> gpurun_out/r6_bench_driver.err | tail -1 > gpurun_out/r6_bench_512_1gpu_driver_args.json
timeout 600 python bench.py --size 256 2> gpurun_out/r6_bench256.err | tail -1 > gpurun_out/r6_bench_256_1gpu.json
python scripts/bench_brief.py gpurun_out/r6_bench_512_1gpu.json gpurun_out/r6_bench_512_1gpu_driver_args.json gpurun_out/r6_bench_256_1gpu.json | cut -c1-380
python - <<'PY'
import json
o = json.loads(open("gpurun_out/r6_bench_512_1gpu_driver_args.json").read().strip().splitlines()[-1])
print(o["roofline"]); print(o["inner_loop"]); print(o["fixed_input"]["itlinear"], o["fixed_input"]["ms_per_step"], o["steady_state"]["ms_per_corrector"], o["steady_state"]["itlinear"], o["cpu_baseline"]["value"])
PY
