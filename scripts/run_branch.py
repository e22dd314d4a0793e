"""BASELINE config 5 in miniature: a PALC branch of the tiled hexagon state with the reference example's settings
(examples/SH3d.jl:160-166: Bordered tangent, BorderingBLS(check_precision = false), ds = -0.001, dsmax = 0.005,
newton tol 1e-9, normC = norminf, eigensolve every step with sigma = 0.1, nev = 15, Krylov dimension 45), driven by
the restated continuation engine through the plugin surface.  Prints one JSON line.
Usage: python scripts/run_branch.py [size=256] [steps=3] [nev=15] [native=0] [gmres|minres: eigensolver's inner solver]   (native=1: every step is one bk_cont_step call)"""
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from bk_amd import continuation as Cn  # noqa: E402
from bk_amd import hip  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
nev = int(sys.argv[3]) if len(sys.argv) > 3 else 15
native = len(sys.argv) > 4 and sys.argv[4] == "1"
eig_minres = len(sys.argv) > 5 and sys.argv[5] == "minres"      # inner solver of the eigensolver: KrylovLS(:minres)
ctx = hip.Context(0)
tiles = bench.tiles_for(n)
cprob, cls_, c0, c1 = bench.cell_branch_points(ctx, hip, 1.0, -0.001)
prob = hip.SwiftHohenberg(ctx, (n, n, n), tuple(l * t for l, t in zip(bench.CELL_L, tiles)), l=0.1, nu=1.2)
P = hip.DCTPreconditioner(prob, 1.0)
ls = hip.GMRESKrylovKit(dim=30, rtol=1e-9, atol=1e-12, maxiter=150, Pl=P)
els = hip.KrylovLSSymmetric("minres", rtol=1e-9, atol=1e-12, itmax=4000, Pl=P) if eig_minres else ls
eig = hip.ShiftInvert(0.1, els, tol=1e-8, maxiter=20, hermitian=True, save_vectors=False)
x0 = hip.HipVec(ctx, bench.tile_cell(c0["u"].t, tiles, prob.slab, ctx.torch_device), prob.nglobal)
nopt = Cn.NewtonPar(tol=1e-9, max_iterations=15, linsolver=ls, eigsolver=eig)
cp = Cn.ContinuationPar(ds=-0.001, dsmin=1e-4, dsmax=0.005, p_min=-0.1, p_max=0.15, max_steps=steps, nev=nev,
                        detect_bifurcation=3, newton_options=nopt)
alg = Cn.PALC(tangent="bordered", theta=0.5, bls=hip.BorderingBLS(None, check_precision=False))
torch.cuda.synchronize()
t0 = time.perf_counter()
br = (Cn.continuation_native if native else Cn.continuation)(prob, x0, 0.1, alg, cp, normC=Cn.norminf, verbosity=1)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(json.dumps(dict(driver="bk_cont_step" if native else "python mirror", eig_inner_solver="minres" if eig_minres else "gmres(30)", size=n, steps=len(br.param) - 1, seconds=dt, seconds_per_continuation_step=dt / max(1, len(br.param) - 1),
                      param=br.param, itnewton=br.itnewton, itlinear=br.itlinear, ds=br.ds, n_unstable=br.n_unstable,
                      rightmost=[[float(v.real) for v in e[:4]] if e is not None else None for e in br.eig])))
