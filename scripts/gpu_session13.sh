#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export PYTHONPATH="$PWD"
cd /tmp && export TMPDIR=/tmp
R="$OLDPWD"
timeout 600 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY --kernel-trace -d "$R/gpurun_out/pmc_sq_dct" -- python "$R/scripts/kernel_sweep.py" 512 precond > "$R/gpurun_out/pmc_sq_dct.log" 2>&1
cd "$R"
python - <<'PY'
import sqlite3, glob, collections
f = glob.glob('gpurun_out/pmc_sq_dct/**/*.db', recursive=True)[0]
cur = sqlite3.connect(f).cursor()
rows = cur.execute("select kernel_name, counter_name, value, (end-start)/1e3, lds_block_size, workgroup_size from counters_collection where kernel_name like '%dct_fft%' and (end-start) > 300000").fetchall()
agg = collections.defaultdict(list)
for k, c, v, us, lds, wg in rows: agg[(wg, c)].append((v, us))
for k in sorted(agg): 
    vals = [x[0] for x in agg[k]]; print(k, len(vals), 'mean', sum(vals)/len(vals), 'mean_us', sum(x[1] for x in agg[k])/len(vals))
PY
