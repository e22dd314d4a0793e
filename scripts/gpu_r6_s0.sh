#!/bin/bash
# round 6, item 1: the reference SH3d example's own pairing Pl = cholesky(L1) (shift 0) through the default (stencil-free) path --
# the new parity cases, then bench lines at 256^3 / 512^3 with the block log
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export PYTHONPATH="$PWD"; mkdir -p gpurun_out
rm -f gpurun_out/stencil_free_probe.jsonl gpurun_out/fullsize_phases.jsonl
timeout 900 python -m pytest tests/test_gpu_stencil_free.py -m gpu -q 2>&1 | tail -40 > gpurun_out/s0_pytest_stencil_free.log
timeout 1500 python -m pytest tests/test_gpu_fullsize.py -m gpu -q -k "tiled_corrector or generic_state" 2>&1 | tail -60 > gpurun_out/s0_pytest_fullsize.log
for n in 256 512; do
  timeout 600 python bench.py --size $n --shift 0 --steps 10 --warmup 3 --cpu-sample 0 --opt gmres_block_log=1 \
      > gpurun_out/s0_bench_${n}.json 2> gpurun_out/s0_bench_${n}_blocklog.txt
done
timeout 600 python bench.py --size 512 --steps 10 --warmup 3 --cpu-sample 0 > gpurun_out/s1_bench_512.json 2> gpurun_out/s1_bench_512.err
tail -5 gpurun_out/s0_pytest_stencil_free.log gpurun_out/s0_pytest_fullsize.log
python - <<'PY'
import json
for f in ("s0_bench_256", "s0_bench_512", "s1_bench_512"):
    try:
        o = json.loads(open(f"gpurun_out/{f}.json").read().strip().splitlines()[-1])
        c = o["config"]
        print(f, o["value"], o["ms_per_step"], c["itlinear_per_step"], c["full_corrector"], c["cell_corrector"], c["gmres_blocks"], o["roofline"]["frac"])
    except Exception as e:
        print(f, "failed", e)
PY
