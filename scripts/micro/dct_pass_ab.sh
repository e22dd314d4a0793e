#!/bin/bash
# per-pass kernel durations of the spectral preconditioner for the two LDS layouts (rocprofv3 --kernel-trace --stats)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export PYTHONPATH="$PWD"; mkdir -p gpurun_out; R="$PWD"
cd /tmp && export TMPDIR=/tmp
for L in ${LIBS:-layout1 layout0}; do
  LIB="$R/bifurcationkit.jl_amd/lib/libbkhip.so"; [ $L != layout1 ] && LIB="$R/bifurcationkit.jl_amd/lib/libbkhip_$L.so"
  BKHIP_LIB=$LIB timeout 300 rocprofv3 --kernel-trace --stats -d "$R/gpurun_out/prof_dct_$L" -- python "$R/scripts/micro/precond_loop.py" 512 > "$R/gpurun_out/prof_dct_$L.log" 2>&1
  echo "== $L"; python "$R/scripts/prof_summary.py" "$R/gpurun_out/prof_dct_$L" 300 2>/dev/null | grep -E "dct_fused|kernel " | cut -c1-60,100-170
done > "$R/gpurun_out/r4_dct_pass_layout_ab.txt"
cat "$R/gpurun_out/r4_dct_pass_layout_ab.txt"
rm -rf "$R"/gpurun_out/prof_dct_*
