cd "${GRAFT_REPO_ROOT:-/root/repo}"; export PYTHONPATH="$PWD"; mkdir -p gpurun_out
: > gpurun_out/r4_c2_block_ab.jsonl
for o in "gmres_sstep=0" "gmres_sstep=4" "gmres_sstep=4,two_lanes=0" "gmres_sstep=0,two_lanes=0" "gmres_sstep=3" ; do
  echo "{\"opts\": \"$o\"}" >> gpurun_out/r4_c2_block_ab.jsonl
  BK_ONLY=c2 BK_OPTS="$o" timeout 200 python scripts/bench_configs.py 2>/dev/null | tail -1 >> gpurun_out/r4_c2_block_ab.jsonl
done
cut -c1-330 gpurun_out/r4_c2_block_ab.jsonl
