#!/bin/bash
# round 6: the whole GPU suite + smoke on the candidate final sources, the L2 pair probe (Next 6), SQ counters of the z round trip at 256 / 512 lanes (Next 3)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export PYTHONPATH="$PWD"; mkdir -p gpurun_out; R="$PWD"
rm -f gpurun_out/stencil_free_probe.jsonl gpurun_out/fullsize_phases.jsonl
( time timeout 1700 python -m pytest tests -m gpu -q 2>&1 | tail -40 ) > gpurun_out/pytest_gpu.log 2>&1
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -3 > gpurun_out/smoke.log
bash scripts/micro/l2_pair_probe.sh > gpurun_out/l2_pair.log 2>&1
cd /tmp && export TMPDIR=/tmp
for v in 256 512; do
  timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT \
      --kernel-trace -d "$R/gpurun_out/sq_z$v" -- python "$R/scripts/micro/precond_loop.py" 512 dct_rt_lanes=$v > "$R/gpurun_out/sq_z$v.log" 2>&1
done
cd "$R"
for v in 256 512; do python scripts/sq_summary.py gpurun_out/sq_z$v gpurun_out/sq_z${v}_summary.txt > /dev/null 2>&1; rm -rf gpurun_out/sq_z$v; done
tail -12 gpurun_out/pytest_gpu.log | cut -c1-300; cat gpurun_out/smoke.log; tail -12 gpurun_out/l2_pair.log | cut -c1-220
grep dct_fused gpurun_out/sq_z256_summary.txt gpurun_out/sq_z512_summary.txt | cut -c1-200
