#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export PYTHONPATH="$PWD"
timeout 900 python -m pytest tests/test_distributed.py -m gpu -q 2>&1 | tail -15 | cut -c1-400
timeout 300 python scripts/bench_configs.py 2>/dev/null | tee gpurun_out/r3m_configs_c2_c3.jsonl | cut -c1-600
timeout 600 python bench.py --workload branch --size 256 --steps 6 2>/dev/null | tail -1 > gpurun_out/r3m_branch_256.json
python - <<'PY'
import json
d = json.load(open('gpurun_out/r3m_branch_256.json'))
print('branch 256: s/step', d['ms_per_step'] / 1e3, 'init', d['config']['initialisation'])
for p in d['per_step']:
    print({k: (round(v, 4) if isinstance(v, float) else v) for k, v in p.items() if k != 'rightmost'})
PY
