#!/bin/bash
# round 3 evidence on the final sources: GPU suite, smoke, bench (default + driver arguments), rocprofv3 stats + PMC passes,
# cost-model inputs, the driver's 8-rank geometry from a bare shell over the host-staged communicator, RCCL bootstrap with 1 rank
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export PYTHONPATH="$PWD"
PART="${PART:-AB}"
if [[ "$PART" == *A* ]]; then
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -30 > gpurun_out/r3f_pytest.log
tail -6 gpurun_out/r3f_pytest.log | cut -c1-300
SKIP_TESTS=1 bash scripts/gpu_profile.sh r3 > gpurun_out/r3f_profile.out 2>&1
tail -3 gpurun_out/smoke.log
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 > gpurun_out/r3f_bench_driver_args.json
timeout 300 python bench.py --size 256 --steps 10 --warmup 3 2>/dev/null | tail -1 > gpurun_out/r3f_bench_256.json
bash scripts/gpu_cost_model_inputs.sh > gpurun_out/r3f_cost_inputs.out 2>&1
tail -10 gpurun_out/r3f_cost_inputs.out | cut -c1-250
BK_FORCE_DIST=1 timeout 300 python bench.py --steps 3 --warmup 1 --cpu-sample 0 --no-steady 2> gpurun_out/r3f_forcedist.err | tail -1 > gpurun_out/r3f_forcedist.json
OUT=gpurun_out/r3f_hostcomm8.jsonl
: > $OUT
BK_BENCH_HOSTCOMM=1 timeout 600 python bench.py --gpus 8 --size 256 --steps 1 --warmup 1 --cpu-sample 0 --no-steady 2> gpurun_out/r3f_hostcomm8_256.err | tail -1 >> $OUT
[ "${HOSTCOMM512:-0}" = 1 ] && BK_BENCH_HOSTCOMM=1 timeout 900 python bench.py --gpus 8 --size 512 --steps 1 --warmup 1 --cpu-sample 0 --no-steady 2> gpurun_out/r3f_hostcomm8_512.err | tail -1 >> $OUT
python - <<'PY'
import json
def show(path):
    for l in open(path):
        try:
            d = json.loads(l); c = d['config']
            print(path.split('/')[-1], c['grid'], 'ranks', d['n_gpus'], 'ms %.2f' % d['ms_per_step'], 'steps/s %.3f' % d['value'], 'itlin', c['itlinear_per_step'], 'p', c['full_corrector']['p'],
                  'roof', d['roofline'] and (d['roofline']['kernel'], round(d['roofline']['frac'], 3), d['roofline']['traffic']), 'comm', (d.get('comm') or {}).get('backend'), (d.get('comm') or {}).get('ranks_in_communicator'))
        except Exception as e:
            print('unparsed', path, e, l[:200])
for p in ('gpurun_out/bench_r3.log', 'gpurun_out/r3f_bench_driver_args.json', 'gpurun_out/r3f_bench_256.json', 'gpurun_out/r3f_forcedist.json', 'gpurun_out/r3f_hostcomm8.jsonl'):
    show(p)
PY
# config 3 eigensolve under rocprofv3: the kernel list must not contain a rocBLAS kernel any more
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$OLDPWD/gpurun_out/prof_c3" -- python "$OLDPWD/scripts/c3_fullsize.py" 1024 eig block > "$OLDPWD/gpurun_out/r3f_c3_prof.log" 2>&1)
python scripts/prof_summary.py gpurun_out/prof_c3 5 2>/dev/null | head -14 | cut -c1-200 > gpurun_out/r3f_c3_eigensolve_kernel_stats.txt
cat gpurun_out/r3f_c3_eigensolve_kernel_stats.txt | cut -c1-160
fi
if [[ "$PART" == *B* ]]; then
# config 5: one MINRES solve at 512^3 with the fused passes off / on (kernel families), the whole branch at 256^3 (bench.py
# --workload branch --steps 50 ends at p_min), three steps with the reference's own eigensolver tolerance, then 512^3
timeout 300 python scripts/micro/minres_fused_diag.py 512 2>&1 | grep -v "^$" | grep -v history > gpurun_out/r3f_minres_fused_512.txt
cat gpurun_out/r3f_minres_fused_512.txt | cut -c1-330
timeout 900 python bench.py --workload branch --size 256 --steps 50 2> gpurun_out/r3f_branch256.err | tail -1 > gpurun_out/r3f_branch_256_50steps.json
timeout 600 python bench.py --workload branch --size 256 --steps 3 --eig-tol 1e-12 2>> gpurun_out/r3f_branch256.err | tail -1 > gpurun_out/r3f_branch_256_3steps_eigtol1e-12.json
python - <<'PY'
import json
for f in ('gpurun_out/r3f_branch_256_50steps.json', 'gpurun_out/r3f_branch_256_3steps_eigtol1e-12.json'):
    try:
        d = json.load(open(f))
        ps = d['per_step']
        print(f.split('/')[-1], 'steps', len(ps), 's/step %.3f' % (d['ms_per_step'] / 1e3), 'init', d['config']['initialisation'], 'first/last', [(round(p['seconds'], 2), p['eig_solves'], p['eig_inner_iterations'], p['itlinear'], p['eig_converged']) for p in (ps[0], ps[-1])], 'p_end', d['param'][-1])
    except Exception as e:
        print(f, 'failed', e)
PY
[ "${SKIP_BRANCH512:-0}" = 1 ] || timeout 1500 python bench.py --workload branch --size 512 --steps 2 2> gpurun_out/r3f_branch512.err | tail -1 > gpurun_out/r3f_branch_512_2steps.json
python - <<'PY'
import json
try:
    d = json.load(open('gpurun_out/r3f_branch_512_2steps.json'))
    print('branch 512: s/step %.1f' % (d['ms_per_step'] / 1e3), 'init', d['config']['initialisation'], [(p['seconds'], p['eig_solves'], p['eig_inner_iterations'], p['itlinear']) for p in d['per_step']])
except Exception as e:
    print('branch 512 failed', e)
PY
fi
