#!/bin/bash
# round 6: the evidence set on the final sources -- bench lines (default / driver arguments / fixed input / block log), rocprofv3 kernel stats,
# PMC traffic, SQ counters, the multi-GPU cost model's inputs, 8 ranks on one GPU (host-staged communicator), BASELINE config 5 (10 steps)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export PYTHONPATH="$PWD"; mkdir -p gpurun_out; R="$PWD"
export TAG=r6
timeout 400 python -m pytest tests/test_gpu_stencil_free.py -m gpu -q 2>&1 | tail -4 > gpurun_out/r6_pytest_stencil_free.log
timeout 900 python bench.py --block-log 2> gpurun_out/r6_bench.err | tail -1 > gpurun_out/r6_bench_512_1gpu.json
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 2> gpurun_out/r6_bench_driver.err | tail -1 > gpurun_out/r6_bench_512_1gpu_driver_args.json
timeout 600 python bench.py --size 256 --block-log 2> gpurun_out/r6_bench256.err | tail -1 > gpurun_out/r6_bench_256_1gpu.json
timeout 300 python bench.py --size 256 --shift 0 --ls-maxiter 4 --no-full --steps 2 --warmup 1 --cpu-sample 0 --no-steady --no-fixed --block-log 2> gpurun_out/r6_bench256_s0.err | tail -1 > gpurun_out/r6_bench_256_shift0_bounded.json
BK_FORCE_DIST=1 timeout 600 python bench.py --steps 5 --warmup 2 --cpu-sample 0 --no-steady 2> gpurun_out/r6_bench_rccl1.err | tail -1 > gpurun_out/r6_bench_512_rccl_bootstrap_1rank.json
python scripts/bench_brief.py gpurun_out/r6_bench_512_1gpu.json gpurun_out/r6_bench_512_1gpu_driver_args.json gpurun_out/r6_bench_256_1gpu.json gpurun_out/r6_bench_512_rccl_bootstrap_1rank.json | cut -c1-400
bash scripts/gpu_evidence.sh prof sq 2>&1 | tail -40 | cut -c1-200
bash scripts/gpu_evidence.sh cost 2>&1 | tail -14 | cut -c1-250
bash scripts/gpu_hostcomm_8rank_bench.sh 2>&1 | tail -6 | cut -c1-300
timeout 900 python bench.py --workload branch --steps 10 --cpu-sample 0 2> gpurun_out/r6_branch512.err | tail -1 > gpurun_out/r6_branch_512_10steps.json
timeout 400 python bench.py --workload branch --steps 1 --cpu-sample 0 --eig-tol 1e-12 2> gpurun_out/r6_branch512_tol.err | tail -1 > gpurun_out/r6_branch_512_1step_eigtol_1e-12.json
timeout 300 python bench.py --workload branch --size 256 --steps 5 --cpu-sample 0 2> gpurun_out/r6_branch256.err | tail -1 > gpurun_out/r6_branch_256_5steps.json
python - <<'PY'
import json
for f in ("r6_branch_512_10steps", "r6_branch_512_1step_eigtol_1e-12", "r6_branch_256_5steps"):
    try:
        o = json.loads(open(f"gpurun_out/{f}.json").read().strip().splitlines()[-1])
        print(f, "s/step %.2f" % (o["ms_per_step"] / 1e3), "all converged", o["config"]["all_eigensolves_converged"],
              [(round(p["seconds"], 1), p["eig_solves"], p["eig_inner_iterations"], p["eig_converged"], "%.7f" % p["rightmost"][0]) for p in o["per_step"]])
    except Exception as e:
        print(f, "failed", repr(e))
PY
