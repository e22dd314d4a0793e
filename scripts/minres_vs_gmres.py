"""Inner solve of the shift-invert eigensolver, (J - sigma) x = v on the tiled hexagon state: GMRES(30) (the reference's
GMRESKrylovKit) against KrylovLS(:minres), same preconditioner and tolerance.  Usage: python scripts/minres_vs_gmres.py [n=256]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from bk_amd import hip  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
ctx = hip.Context(0)
tiles = bench.tiles_for(n)
cprob, cls_, c0, c1 = bench.cell_branch_points(ctx, hip, 1.0, -0.001)
prob = hip.SwiftHohenberg(ctx, (n, n, n), tuple(l * t for l, t in zip(bench.CELL_L, tiles)), l=0.1, nu=1.2)
P = hip.DCTPreconditioner(prob, 1.0)
x0 = hip.HipVec(ctx, bench.tile_cell(c0["u"].t, tiles, prob.slab, ctx.torch_device), prob.nglobal)
J = prob.jacobian(x0, 0.0)          # J(u; l) - sigma = J(u; l - sigma): sigma = 0.1 folded into the parameter, as SH3dEig folds it into the operator
g = torch.Generator(device="cuda").manual_seed(3)
v = hip.HipVec(ctx, torch.rand(prob.nlocal, dtype=torch.float64, device="cuda", generator=g) - 0.5)
for name, ls in (("GMRESKrylovKit(30)", hip.GMRESKrylovKit(dim=30, rtol=1e-9, atol=1e-12, maxiter=150, Pl=P)),
                 ("KrylovLS(:minres)", hip.KrylovLSSymmetric("minres", rtol=1e-9, atol=1e-12, itmax=4000, Pl=P))):
    ls(J, v)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    x, ok, it = ls(J, v)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    r = J(x).add_(v, -1.0)
    print(json.dumps(dict(solver=name, n=n, converged=ok, iterations=it, seconds=dt, ms_per_iteration=dt / max(it, 1) * 1e3,
                          true_residual_rel=r.norm() / v.norm())))
