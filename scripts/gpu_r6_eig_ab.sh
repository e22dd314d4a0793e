#!/bin/bash
# round 6, item 5(b): inner solver of the shift-invert eigensolver -- MINRES (fused Lanczos step) against GMRES on the stencil-free operator
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export PYTHONPATH="$PWD"; mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_stencil_free.py -m gpu -q 2>&1 | tail -6 > gpurun_out/e_pytest_stencil_free.log
timeout 900 python -m pytest tests/test_gpu_fullsize.py -m gpu -q -k "generic_state and hex" 2>&1 | tail -8 > gpurun_out/e_pytest_fullsize.log
B="python bench.py --workload branch --cpu-sample 0"
for v in "minres" "gmres --eig-inner-dim 30" "gmres --eig-inner-dim 40"; do
  tag=$(echo $v | tr -d ' -')
  timeout 400 $B --size 256 --steps 3 --eig-inner $v > gpurun_out/e_branch_256_${tag}.json 2> gpurun_out/e_branch_256_${tag}.err
done
for v in "minres" "gmres --eig-inner-dim 40"; do
  tag=$(echo $v | tr -d ' -')
  timeout 900 $B --size 512 --steps 2 --eig-inner $v > gpurun_out/e_branch_512_${tag}.json 2> gpurun_out/e_branch_512_${tag}.err
done
tail -3 gpurun_out/e_pytest_stencil_free.log gpurun_out/e_pytest_fullsize.log
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/e_branch_*.json")):
    try:
        o = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], "s/step %.2f" % (o["ms_per_step"] / 1e3), "init", o["config"]["initialisation"],
              [(p["seconds"], p["eig_solves"], p["eig_inner_iterations"], p["eig_converged"], p["n_unstable"], ["%.6f" % v for v in p["rightmost"][:3]]) for p in o["per_step"]])
    except Exception as e:
        print(f, "failed", repr(e), open(f.replace(".json", ".err")).read()[-300:])
PY
