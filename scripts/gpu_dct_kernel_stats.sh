#!/bin/bash
# per-kernel durations of the DCT pass kernels (rocprofv3 kernel trace of a short bench) for a list of option settings:
#   VARIANTS="dct_xcd_map=0 dct_xcd_map=3 dct_xcd_map=7" bash scripts/gpu_dct_kernel_stats.sh
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export PYTHONPATH="$PWD"
R="$PWD"
[ -n "${SKIPTEST:-}" ] || timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "dct or precond" 2>&1 | tail -2
cd /tmp && export TMPDIR=/tmp
for v in ${VARIANTS:-dct_xcd_map=0 dct_xcd_map=3 dct_xcd_map=7}; do
g=$(echo $v | tr '=,' '__')
rm -rf "$R/gpurun_out/prof_o_$g"
timeout 600 rocprofv3 --kernel-trace --stats -d "$R/gpurun_out/prof_o_$g" -- python "$R/bench.py" --steps 2 --warmup 1 --cpu-sample 0 --no-steady $(echo $v | tr ',' '\n' | sed 's/^/--opt /') > "$R/gpurun_out/prof_o_$g.log" 2>&1
python "$R/scripts/prof_summary.py" "$R/gpurun_out/prof_o_$g" 300 > "$R/gpurun_out/prof_o_${g}_summary.txt" 2>&1
echo "variant $v"; grep dct_fused "$R/gpurun_out/prof_o_${g}_summary.txt" | sed 's/void bk::(anonymous namespace):://' | awk '{print $1,$2,$3,$4, $(NF-5), $(NF-4), $(NF-3), $(NF-2)}'
done
