#!/bin/bash
# GPU-box runs (rounds 4-5), one script, parts selected by name:   bash scripts/gpu_evidence.sh <part> [<part> ...]   (TAG=r5 ...)
# (the box shows 256 logical CPUs under a 16-CPU cgroup quota: anything OpenMP must be pinned to the quota -- scripts/cpu_diag.sh,
#  bench.available_cpus(); the full-size parity tests and bench.py's CPU baseline do that themselves)
#   leads     start-offset / LDS-layout sweep of the hot kernels (both layouts), then bench.py with the winning options
#   tests     pytest -m gpu + smoke   (6 min serial; `-n 3 --dist loadfile` was tried: ~5 min, tests/test_gpu_parity.py keeps one worker busy)
#   newtests  only the tests added this round (TESTS_K = pytest -k expression)
#   bench     default bench line + the driver's arguments (--gpus 1 --steps 20 --warmup 5)
#   prof      rocprofv3 kernel stats of the bench + the two PMC passes (FETCH_SIZE / WRITE_SIZE)   [TAG, default r4]
#   hostcomm  N ranks sharing this GPU over the host-staged communicator with the RCCL-default code path (RANKS, SIZE)
#   ab        bench.py A/B on the same box: AB_A / AB_B = extra bench.py arguments of the two runs (e.g. "--opt gmres_sstep=0")
#   sq        one SQ-counter pass over a bench step (LDS bank-conflict share, stall breakdown per kernel)
#   cost      inputs of the multi-GPU cost model (bench.py --size-z slabs, slab z-solve emulation)
# Everything lands in gpurun_out/ (scratch); summaries that are judged get copied to profiles/ by hand.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export PYTHONPATH="$PWD"
TAG=${TAG:-r4}
R="$PWD"
for part in "$@"; do
case "$part" in
leads)
    timeout 600 python scripts/micro/start_offsets.py ${SIZE:-512} ${LEADS:-dct,axpy,jvp} > gpurun_out/${TAG}_leads.jsonl 2> gpurun_out/${TAG}_leads.err
    BKHIP_LIB="$R/bifurcationkit.jl_amd/lib/libbkhip_layout0.so" timeout 300 python scripts/micro/start_offsets.py ${SIZE:-512} dct > gpurun_out/${TAG}_leads_layout0.jsonl 2>> gpurun_out/${TAG}_leads.err
    cut -c1-420 gpurun_out/${TAG}_leads.jsonl gpurun_out/${TAG}_leads_layout0.jsonl
    ARGS=$(tail -1 gpurun_out/${TAG}_leads.jsonl | python -c "import json,sys; print(json.loads(sys.stdin.read())['bench_args'])")
    echo "bench with: $ARGS"
    timeout 600 python bench.py --steps 10 --warmup 3 --cpu-sample 0 --no-steady 2> gpurun_out/${TAG}_leads_bench0.err | tail -1 > gpurun_out/${TAG}_leads_bench_baseline.json
    timeout 600 python bench.py --steps 10 --warmup 3 --cpu-sample 0 --no-steady $ARGS 2> gpurun_out/${TAG}_leads_bench1.err | tail -1 > gpurun_out/${TAG}_leads_bench_tuned.json
    python scripts/bench_brief.py gpurun_out/${TAG}_leads_bench_baseline.json gpurun_out/${TAG}_leads_bench_tuned.json
    ;;
tests)
    timeout 1500 python -m pytest tests -m gpu -q --timeout=400 --durations=12 > gpurun_out/${TAG}_pytest_gpu.log 2>&1
    tail -${TAILN:-45} gpurun_out/${TAG}_pytest_gpu.log | cut -c1-300
    timeout 300 python __graft_entry__.py smoke 2>&1 | tail -2 | tee gpurun_out/${TAG}_smoke.log
    ;;
newtests)
    # every test under its own clock (pytest-timeout), slowest tests listed, the whole log kept
    timeout ${TESTS_LIMIT:-900} python -m pytest tests -m gpu -q -x --timeout=${TEST_TIMEOUT:-240} --durations=8 -k "${TESTS_K}" > gpurun_out/${TAG}_pytest_new.log 2>&1
    tail -${TAILN:-40} gpurun_out/${TAG}_pytest_new.log | cut -c1-400
    ;;
bench)
    timeout 900 python bench.py 2> gpurun_out/${TAG}_bench.err | tail -1 > gpurun_out/${TAG}_bench_512_1gpu.json
    timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 2> gpurun_out/${TAG}_bench_driver.err | tail -1 > gpurun_out/${TAG}_bench_512_1gpu_driver_args.json
    python scripts/bench_brief.py gpurun_out/${TAG}_bench_512_1gpu.json gpurun_out/${TAG}_bench_512_1gpu_driver_args.json
    ;;
prof)
    cd /tmp && export TMPDIR=/tmp
    timeout 900 rocprofv3 --kernel-trace --stats -d "$R/gpurun_out/prof_${TAG}" -- python "$R/bench.py" --steps 3 --warmup 1 --cpu-sample 0 --no-steady > "$R/gpurun_out/prof_${TAG}.log" 2>&1
    timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d "$R/gpurun_out/pmc_fetch_${TAG}" -- python "$R/bench.py" --steps 1 --warmup 0 --cpu-sample 0 --no-steady > "$R/gpurun_out/pmc_fetch_${TAG}.log" 2>&1
    timeout 900 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d "$R/gpurun_out/pmc_write_${TAG}" -- python "$R/bench.py" --steps 1 --warmup 0 --cpu-sample 0 --no-steady > "$R/gpurun_out/pmc_write_${TAG}.log" 2>&1
    cd "$R"
    python scripts/prof_summary.py gpurun_out/prof_${TAG} 300 > gpurun_out/${TAG}_rocprofv3_kernel_stats_bench512.txt 2>&1
    head -18 gpurun_out/${TAG}_rocprofv3_kernel_stats_bench512.txt | cut -c1-170
    python scripts/pmc_summary.py gpurun_out/pmc_fetch_${TAG} gpurun_out/pmc_write_${TAG} gpurun_out/${TAG} | cut -c1-150
    ;;
hostcomm)
    OUT=gpurun_out/${TAG}_hostcomm_ranks.jsonl
    for lanes in ${LANES:-0 1}; do
        BK_BENCH_HOSTCOMM=1 timeout 900 python bench.py --gpus ${RANKS:-8} --size ${SIZE:-256} --steps 2 --warmup 1 --cpu-sample 0 --no-steady --opt two_lanes=$lanes 2> gpurun_out/${TAG}_hostcomm_${lanes}.err | tail -1 >> $OUT
    done
    timeout 300 python bench.py --size ${SIZE:-256} --steps 2 --warmup 1 --cpu-sample 0 --no-steady 2>/dev/null | tail -1 >> $OUT
    python scripts/bench_brief.py $OUT
    ;;
ab)
    timeout 600 python bench.py --steps ${AB_STEPS:-10} --warmup 3 --cpu-sample 0 ${AB_COMMON:---no-steady} ${AB_A:-} 2> gpurun_out/${TAG}_ab_a.err | tail -1 > gpurun_out/${TAG}_ab_a.json
    timeout 600 python bench.py --steps ${AB_STEPS:-10} --warmup 3 --cpu-sample 0 ${AB_COMMON:---no-steady} ${AB_B:-} 2> gpurun_out/${TAG}_ab_b.err | tail -1 > gpurun_out/${TAG}_ab_b.json
    [ -n "${AB_C:-}" ] && timeout 600 python bench.py --steps ${AB_STEPS:-10} --warmup 3 --cpu-sample 0 ${AB_COMMON:---no-steady} ${AB_C} 2> gpurun_out/${TAG}_ab_c.err | tail -1 > gpurun_out/${TAG}_ab_c.json
    tail -3 gpurun_out/${TAG}_ab_a.err gpurun_out/${TAG}_ab_b.err | cut -c1-300
    python scripts/bench_brief.py gpurun_out/${TAG}_ab_a.json gpurun_out/${TAG}_ab_b.json $([ -n "${AB_C:-}" ] && echo gpurun_out/${TAG}_ab_c.json)
    ;;
sq)
    bash scripts/gpu_sq_counters.sh ${TAG} 2>&1 | tail -14 | cut -c1-200
    ;;
cost)
    bash scripts/gpu_cost_model_inputs.sh > gpurun_out/${TAG}_cost_inputs.out 2>&1
    tail -12 gpurun_out/${TAG}_cost_inputs.out | cut -c1-250
    ;;
*)
    echo "unknown part $part"
    ;;
esac
done
