#!/bin/bash
# round 6: the two bench lines once more, now that profiles/ holds the PMC traffic of exactly these kernel sources (roofline.traffic)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export PYTHONPATH="$PWD"; mkdir -p gpurun_out
timeout 900 python bench.py --block-log 2> gpurun_out/r6_bench.err | tail -1 > gpurun_out/r6_bench_512_1gpu.json
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 2> gpurun_out/r6_bench_driver.err | tail -1 > gpurun_out/r6_bench_512_1gpu_driver_args.json
timeout 600 python bench.py --size 256 2> gpurun_out/r6_bench256.err | tail -1 > gpurun_out/r6_bench_256_1gpu.json
python scripts/bench_brief.py gpurun_out/r6_bench_512_1gpu.json gpurun_out/r6_bench_512_1gpu_driver_args.json gpurun_out/r6_bench_256_1gpu.json | cut -c1-420
python - <<'PY'
import json
o = json.loads(open("gpurun_out/r6_bench_512_1gpu_driver_args.json").read().strip().splitlines()[-1])
print(o["roofline"]); print(o["fixed_input"]["itlinear"], o["fixed_input"]["ms_per_step"], o["steady_state"]["ms_per_corrector"], o["steady_state"]["itlinear"], o["cpu_baseline"]["value"])
PY
