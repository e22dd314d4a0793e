#!/bin/bash
# round 6, final sources: the whole GPU suite + smoke, rocprofv3 kernel stats / PMC traffic / SQ counters, then -- with the PMC traffic of exactly
# these kernel sources in profiles/ -- the bench lines, BASELINE config 5, the cache-resident configs
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export PYTHONPATH="$PWD"; mkdir -p gpurun_out; R="$PWD"
export TAG=r6
rm -f gpurun_out/stencil_free_probe.jsonl gpurun_out/fullsize_phases.jsonl gpurun_out/tolerance_probe.jsonl
( time timeout 1800 python -m pytest tests -m gpu -q --durations=8 2>&1 | tail -30 ) > gpurun_out/r6_pytest_gpu.log 2>&1
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -2 > gpurun_out/r6_smoke.log
bash scripts/gpu_evidence.sh prof sq 2>&1 | tail -30 | cut -c1-200
cp gpurun_out/r6_pmc_hbm_traffic.json gpurun_out/r6_pmc_hbm_traffic.txt profiles/          # (the box's scratch copy: bench.py reads it for roofline.traffic)
timeout 900 python bench.py --block-log 2> gpurun_out/r6_bench.err | tail -1 > gpurun_out/r6_bench_512_1gpu.json
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 2> gpurun_out/r6_bench_driver.err | tail -1 > gpurun_out/r6_bench_512_1gpu_driver_args.json
timeout 600 python bench.py --size 256 2> gpurun_out/r6_bench256.err | tail -1 > gpurun_out/r6_bench_256_1gpu.json
timeout 900 python bench.py --workload branch --steps 10 --cpu-sample 0 2> gpurun_out/r6_branch512.err | tail -1 > gpurun_out/r6_branch_512_10steps.json
timeout 400 python bench.py --workload branch --steps 1 --cpu-sample 0 --eig-tol 1e-12 2> gpurun_out/r6_branch512_tol.err | tail -1 > gpurun_out/r6_branch_512_1step_eigtol_1e-12.json
timeout 300 python bench.py --workload branch --size 256 --steps 5 --cpu-sample 0 2> gpurun_out/r6_branch256.err | tail -1 > gpurun_out/r6_branch_256_5steps.json
timeout 300 python scripts/bench_configs.py > gpurun_out/r6_configs_c2_c3.jsonl 2> gpurun_out/r6_configs.err
tail -14 gpurun_out/r6_pytest_gpu.log | cut -c1-200; cat gpurun_out/r6_smoke.log
python scripts/bench_brief.py gpurun_out/r6_bench_512_1gpu.json gpurun_out/r6_bench_512_1gpu_driver_args.json gpurun_out/r6_bench_256_1gpu.json | cut -c1-380
python - <<'PY'
import json
o = json.loads(open("gpurun_out/r6_bench_512_1gpu_driver_args.json").read().strip().splitlines()[-1])
print(o["roofline"]); print(o["inner_loop"]); print(o["fixed_input"]["itlinear"], o["fixed_input"]["ms_per_step"], o["steady_state"]["ms_per_corrector"], o["steady_state"]["itlinear"], o["cpu_baseline"]["value"])
for f in ("r6_branch_512_10steps", "r6_branch_512_1step_eigtol_1e-12", "r6_branch_256_5steps"):
    try:
        o = json.loads(open(f"gpurun_out/{f}.json").read().strip().splitlines()[-1])
        print(f, "s/step %.2f" % (o["ms_per_step"] / 1e3), "all converged", o["config"]["all_eigensolves_converged"],
              [(round(p["seconds"], 1), p["eig_solves"], p["eig_inner_iterations"], p["eig_converged"], "%.7f" % p["rightmost"][0]) for p in o["per_step"]])
    except Exception as e:
        print(f, "failed", repr(e))
PY
tail -4 gpurun_out/r6_configs_c2_c3.jsonl | cut -c1-300
