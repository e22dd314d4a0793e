#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export PYTHONPATH="$PWD"
cd /tmp && export TMPDIR=/tmp
R="$OLDPWD"
rm -rf "$R/gpurun_out/kt_dct" "$R/gpurun_out/pmc_sq_dct"
BK_SWEEP_FAST=1 timeout 600 rocprofv3 --kernel-trace -d "$R/gpurun_out/kt_dct" -- python "$R/scripts/kernel_sweep.py" 512 precond > "$R/gpurun_out/kt_dct.log" 2>&1
BK_SWEEP_FAST=1 timeout 600 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES --kernel-trace -d "$R/gpurun_out/pmc_sq_dct" -- python "$R/scripts/kernel_sweep.py" 512 precond > "$R/gpurun_out/pmc_sq_dct.log" 2>&1
cd "$R"
python scripts/prof_summary.py gpurun_out/kt_dct 100 2>&1 | cut -c1-190 | tee gpurun_out/kt_dct_summary.txt
python - <<'PY' | tee gpurun_out/pmc_sq_dct_summary.txt
import sqlite3, glob, collections
f = glob.glob('gpurun_out/pmc_sq_dct/**/*.db', recursive=True)[0]
cur = sqlite3.connect(f).cursor()
rows = cur.execute("select kernel_name, counter_name, value, (end-start)/1e3 from counters_collection where kernel_name like '%dct_f%' and (end-start) > 100000").fetchall()
agg = collections.defaultdict(list)
for k, c, v, us in rows: agg[(k[-45:], c)].append((v, us))
for k in sorted(agg):
    vals = [x[0] for x in agg[k]]; print(k, len(vals), 'mean %.4g' % (sum(vals)/len(vals)), 'mean_us %.1f' % (sum(x[1] for x in agg[k])/len(vals)))
PY
