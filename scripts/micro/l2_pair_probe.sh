#!/bin/bash
# VERDICT r5 Next 6 (exploratory): x -> y pass pair, plane by plane on one XCD, second pass from L2?  Copy-kernel upper bound.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; R="$PWD"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -Wno-unused-value scripts/micro/l2_pair_probe.hip -o gpurun_out/l2_pair_probe || exit 1
timeout 120 gpurun_out/l2_pair_probe 512 8 | tee gpurun_out/l2_pair_timing.jsonl
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d "$R/gpurun_out/l2_pair_fetch" -- "$R/gpurun_out/l2_pair_probe" 512 3 > "$R/gpurun_out/l2_pair_fetch.log" 2>&1
timeout 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d "$R/gpurun_out/l2_pair_write" -- "$R/gpurun_out/l2_pair_probe" 512 3 > "$R/gpurun_out/l2_pair_write.log" 2>&1
cd "$R"
python scripts/pmc_summary.py gpurun_out/l2_pair_fetch gpurun_out/l2_pair_write gpurun_out/l2_pair 2>&1 | cut -c1-200 | tee gpurun_out/l2_pair_pmc.txt
