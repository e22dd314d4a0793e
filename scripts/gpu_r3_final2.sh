#!/bin/bash
# after the shift-invert operator forwards the fused Lanczos step: solver tests, PMC passes + kernel stats + bench line on
# these sources (the PMC file is put where bench.py looks for it before the bench line is taken), config 5 at 256^3
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export PYTHONPATH="$PWD"
R="$PWD"
timeout 300 python -m pytest tests -m gpu -x -q -k "fused_minres or symmetric_krylov or hermitian or shift_invert or branch_native" 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d "$R/gpurun_out/pmc_fetch_r3" -- python "$R/bench.py" --steps 1 --warmup 0 --cpu-sample 0 --no-steady > "$R/gpurun_out/pmc_fetch_r3.log" 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d "$R/gpurun_out/pmc_write_r3" -- python "$R/bench.py" --steps 1 --warmup 0 --cpu-sample 0 --no-steady > "$R/gpurun_out/pmc_write_r3.log" 2>&1
cd "$R"
python scripts/pmc_summary.py gpurun_out/pmc_fetch_r3 gpurun_out/pmc_write_r3 gpurun_out/r3 | cut -c1-150 | tail -4
cp gpurun_out/r3_pmc_hbm_traffic.json gpurun_out/r3_pmc_hbm_traffic.txt profiles/
timeout 300 python bench.py 2> gpurun_out/bench_r3.err | tail -1 > gpurun_out/bench_r3.log
python - <<'PY'
import json
d = json.loads(open('gpurun_out/bench_r3.log').read().strip().splitlines()[-1])
print('bench', d['value'], d['ms_per_step'], d['roofline'])
PY
timeout 200 python bench.py --workload branch --size 256 --steps ${BRANCH_STEPS:-12} 2> gpurun_out/r3g_branch256.err | tail -1 > gpurun_out/r3g_branch_256.json
python - <<'PY'
import json
d = json.load(open('gpurun_out/r3g_branch_256.json'))
print('branch 256 s/step %.3f' % (d['ms_per_step'] / 1e3), [(round(p['seconds'], 2), p['eig_solves'], p['eig_inner_iterations']) for p in d['per_step']][:6])
PY
(cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats -d "$R/gpurun_out/prof_r3" -- python "$R/bench.py" --steps 3 --warmup 1 --cpu-sample 0 --no-steady > "$R/gpurun_out/prof_r3.log" 2>&1)
python scripts/prof_summary.py gpurun_out/prof_r3 300 > gpurun_out/prof_r3_summary.txt 2>&1
head -8 gpurun_out/prof_r3_summary.txt | cut -c1-170
