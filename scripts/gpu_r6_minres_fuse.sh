#!/bin/bash
# round 6: MINRES recurrence axpy riding in the preconditioner's x-forward pass (option minres_fuse_axpy) -- parity, then A/B on the branch workload
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export PYTHONPATH="$PWD"; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -q -k "minres or krylovls or symmetric or eig or shift_invert or dct or precond" 2>&1 | tail -6 > gpurun_out/f_pytest.log
B="python bench.py --workload branch --cpu-sample 0"
for v in 1 0; do
  timeout 300 $B --size 256 --steps 4 --opt minres_fuse_axpy=$v > gpurun_out/f_branch_256_fuse$v.json 2> gpurun_out/f_branch_256_fuse$v.err
  timeout 600 $B --size 512 --steps 2 --opt minres_fuse_axpy=$v > gpurun_out/f_branch_512_fuse$v.json 2> gpurun_out/f_branch_512_fuse$v.err
done
tail -4 gpurun_out/f_pytest.log
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/f_branch_*.json")):
    try:
        o = json.loads(open(f).read().strip().splitlines()[-1])
        it = sum(p["eig_inner_iterations"] for p in o["per_step"]); sec = sum(p["seconds"] for p in o["per_step"])
        print(f.split("/")[-1], "s/step %.2f" % (o["ms_per_step"] / 1e3), "ms per inner iteration %.3f" % (sec / it * 1e3),
              [(round(p["seconds"], 2), p["eig_solves"], p["eig_inner_iterations"], p["eig_converged"], "%.8f" % p["rightmost"][0], "%.10f" % p["p"]) for p in o["per_step"]])
    except Exception as e:
        print(f, "failed", repr(e), open(f.replace(".json", ".err")).read()[-300:])
PY
