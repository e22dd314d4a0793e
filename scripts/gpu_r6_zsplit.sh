#!/bin/bash
# round 6, items 1c / 3: (a) parity of the 512-lane z round trip (lane-pair merged middle), (b) A/B of the round trip's lane count and
# of the first block (powers of T vs powers of W) on identical inputs, (c) per-kernel durations of the preconditioner under rocprofv3
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export PYTHONPATH="$PWD"; mkdir -p gpurun_out
R="$PWD"
rm -f gpurun_out/stencil_free_probe.jsonl gpurun_out/fullsize_phases.jsonl
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "dct or precond or Precond or spectral" 2>&1 | tail -8 > gpurun_out/z_pytest_dct.log
timeout 300 python -m pytest tests/test_gpu_stencil_free.py -m gpu -q 2>&1 | tail -15 > gpurun_out/z_pytest_stencil_free.log
timeout 900 python -m pytest tests/test_gpu_fullsize.py -m gpu -q -k "preconditioner_roundtrip or c5_512 or (generic_state and hex)" 2>&1 | tail -12 > gpurun_out/z_pytest_fullsize.log
B="python bench.py --steps 10 --warmup 3 --cpu-sample 0 --no-steady --block-log"
for v in "dct_rt_lanes=256" "dct_rt_lanes=512"; do
  timeout 400 $B --opt $v > gpurun_out/z_bench_${v}.json 2> gpurun_out/z_bench_${v}.err
done
for v in "gmres_monomial_shift=1" "gmres_monomial_shift=0"; do
  timeout 400 $B --opt $v > gpurun_out/z_bench_${v}.json 2> gpurun_out/z_bench_${v}.err
  timeout 300 python bench.py --size 256 --shift 0 --ls-maxiter 4 --no-full --steps 2 --warmup 1 --cpu-sample 0 --no-steady --no-fixed --block-log --opt $v > gpurun_out/z_bench_s0_256_${v}.json 2> gpurun_out/z_bench_s0_256_${v}.err
done
cd /tmp && export TMPDIR=/tmp
for v in 256 512; do
  timeout 300 rocprofv3 --kernel-trace --stats -d "$R/gpurun_out/z_prof_$v" -- python "$R/scripts/micro/precond_loop.py" 512 dct_rt_lanes=$v > "$R/gpurun_out/z_prof_$v.log" 2>&1
done
cd "$R"
for v in 256 512; do python scripts/prof_summary.py gpurun_out/z_prof_$v 100 2>&1 | head -12 | cut -c1-200 > gpurun_out/z_prof_${v}_summary.txt; done
tail -4 gpurun_out/z_pytest_dct.log gpurun_out/z_pytest_stencil_free.log gpurun_out/z_pytest_fullsize.log
python scripts/bench_brief.py gpurun_out/z_bench_*.json | cut -c1-420
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/z_bench_*.json")):
    try:
        o = json.loads(open(f).read().strip().splitlines()[-1]); bl = o["config"]["block_log"]; fx = o.get("fixed_input") or {}
        print(f.split("/")[-1], "first-block ratios", ["%.1e" % r for r in bl["first_block_last_pivot_ratio"]], "min ratio %.1e" % bl["min_last_pivot_ratio"],
              "truncated", bl["truncated_blocks"], "fixed_input", fx.get("itlinear"), "%.2f ms" % fx.get("ms_per_step", 0.0))
    except Exception as e:
        print(f, "failed", repr(e))
PY
cat gpurun_out/z_prof_256_summary.txt gpurun_out/z_prof_512_summary.txt
