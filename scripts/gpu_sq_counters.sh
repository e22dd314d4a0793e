#!/bin/bash
# One rocprofv3 SQ-counter pass (8 SQ slots, --kernel-trace only) over one bench step: stall breakdown per kernel.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
TAG=${1:-r2}
mkdir -p gpurun_out
export PYTHONPATH="$PWD"
R="$PWD"
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT \
    --kernel-trace -d "$R/gpurun_out/sq_${TAG}" -- python "$R/bench.py" --steps 1 --warmup 0 --cpu-sample 0 --no-steady > "$R/gpurun_out/sq_${TAG}.log" 2>&1
cd "$R"
tail -3 gpurun_out/sq_${TAG}.log | cut -c1-300
python scripts/sq_summary.py gpurun_out/sq_${TAG} gpurun_out/${TAG}_sq_stall_breakdown.txt | cut -c1-170
rm -rf gpurun_out/sq_${TAG}      # the raw counter database (> 64 MiB) stays on the box
