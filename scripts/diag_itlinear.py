"""Why does the PALC corrector need 4-8x more GMRES iterations on a running branch than from the tiled predictor?
(Weak 2).  Runs a short native branch of the tiled hexagon state with the solver trace on and reports, for
the corrector of every step: iteration counts of the two solves of the bordered system, where along the residual history
the iterations are spent, and the tile-symmetry defect of the state / tangent / right-hand side (fraction of the 2-norm
outside the subspace of even-reflection tilings of a cell field -- the slow phase modes of the big domain live entirely
outside it).  Then it re-solves J x = F(z_pred) of the last step with alternative solver settings.
Usage: python scripts/diag_itlinear.py [size=256] [steps=3]"""
import json
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from bk_amd import continuation as Cn  # noqa: E402
from bk_amd import hip  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
ctx = hip.Context(0)
tiles = bench.tiles_for(n)
cprob, cls_, c0, c1 = bench.cell_branch_points(ctx, hip, 1.0, -0.001)
prob = hip.SwiftHohenberg(ctx, (n, n, n), tuple(l * t for l, t in zip(bench.CELL_L, tiles)), l=0.1, nu=1.2)
P = hip.DCTPreconditioner(prob, 1.0)
cx, cy, cz = bench.CELL


def symmetrize(t):
    """Projection of a grid field on the even-reflection tilings of a cell field."""
    a = t.reshape(tiles[2], cz, tiles[1], cy, tiles[0], cx).clone()
    a[1::2] = a[1::2].flip(1)
    a[:, :, 1::2] = a[:, :, 1::2].flip(3)
    a[:, :, :, :, 1::2] = a[:, :, :, :, 1::2].flip(5)
    m = a.mean(dim=(0, 2, 4), keepdim=True).expand_as(a).clone()
    m[1::2] = m[1::2].flip(1)
    m[:, :, 1::2] = m[:, :, 1::2].flip(3)
    m[:, :, :, :, 1::2] = m[:, :, :, :, 1::2].flip(5)
    return m.reshape(-1)


def defect(v):
    s = symmetrize(v.t)
    return float((v.t - s).norm() / v.t.norm())


def milestones(h):
    r0 = h[0]
    out = {}
    for lvl in (1e-3, 1e-6, 1e-8, 3e-9, 1e-9):
        k = next((i for i, v in enumerate(h) if v <= lvl * r0), None)
        out[f"{lvl:g}"] = k
    return out


def run(tangent, ls, label):
    nopt = Cn.NewtonPar(tol=1e-9, max_iterations=15, linsolver=ls, eigsolver=None)
    cp = Cn.ContinuationPar(ds=-0.001, dsmin=1e-4, dsmax=0.005, p_min=-0.1, p_max=0.15, max_steps=steps, nev=15,
                            detect_bifurcation=0, newton_options=nopt)
    alg = Cn.PALC(tangent=tangent, theta=0.5, bls=hip.BorderingBLS(None, check_precision=False))
    x0 = hip.HipVec(ctx, bench.tile_cell(c0["u"].t, tiles, prob.slab, ctx.torch_device), prob.nglobal)
    ctx.set_option("solver_trace", 1)
    ctx.solver_history()
    rec, last = [], {}

    def fin(get, r):
        st = get()
        hs = ctx.solver_history()
        rec.append(dict(step=r.step, p=r.p, itnewton=r.itnewton, itlinear=r.itlinear,
                        solves=[dict(its=len(h) - 1, r0=h[0], rel_end=h[-1] / h[0], reach=milestones(h)) for h in hs],
                        defect_u=defect(st["z"].u), defect_tau=defect(st["tau"].u)))
        last.update(st)
        return True

    br = Cn.continuation_native(prob, x0, 0.1, alg, cp, normC=Cn.norminf, finalise_solution=fin)
    ctx.set_option("solver_trace", 0)
    print(json.dumps(dict(run=label, size=n, itlinear=br.itlinear, itnewton=br.itnewton, steps=rec)), flush=True)
    return last


gm = hip.GMRESKrylovKit(dim=30, rtol=1e-9, atol=1e-12, maxiter=150, Pl=P)
st = run("bordered", gm, "bordered tangent, GMRES(30) rtol 1e-9 atol 1e-12 (examples/SH3d.jl:93,160-163)")
run("secant", gm, "secant tangent, same solver")

# ---- the corrector's first right-hand side of the next step, re-solved with other settings
z, tau, ds = st["z"], st["tau"], st["ds"]
pred = z.copy().add_(tau, ds)
R = prob.residual(pred.u, pred.p)
J = prob.jacobian(pred.u, pred.p)
Rs = hip.HipVec(ctx, symmetrize(R.t), prob.nglobal)
noise = hip.HipVec(ctx, (R.t - Rs.t).contiguous(), prob.nglobal)
info = dict(R_inf=R.norminf(), R_2=R.norm(), defect_R=defect(R), noise_inf=noise.norminf(),
            PlR_2=P.ldiv(R).norm(), Pl_noise_2=P.ldiv(noise).norm(), sqrtN=math.sqrt(prob.nglobal))
print(json.dumps(dict(rhs=info)), flush=True)
ctx.set_option("solver_trace", 1)
floor = 1e-12 * math.sqrt(prob.nglobal)
variants = [
    ("GMRES(30) rtol 1e-9", hip.GMRESKrylovKit(dim=30, rtol=1e-9, atol=1e-12, maxiter=150, Pl=P), R),
    ("GMRES(30) rtol 1e-9, symmetrized rhs", hip.GMRESKrylovKit(dim=30, rtol=1e-9, atol=1e-12, maxiter=150, Pl=P), Rs),
    ("GMRES(60) rtol 1e-9", hip.GMRESKrylovKit(dim=60, rtol=1e-9, atol=1e-12, maxiter=150, Pl=P), R),
    ("MINRES rtol 1e-9", hip.KrylovLSSymmetric("minres", rtol=1e-9, atol=1e-12, itmax=4000, Pl=P), R),
    (f"GMRES(30) rtol 1e-9 atol 1e-12*sqrt(N) = {floor:.2e}", hip.GMRESKrylovKit(dim=30, rtol=1e-9, atol=floor, maxiter=150, Pl=P), R),
    (f"GMRES(30) rtol 1e-9 atol 1e-13*sqrt(N) = {0.1 * floor:.2e}", hip.GMRESKrylovKit(dim=30, rtol=1e-9, atol=0.1 * floor, maxiter=150, Pl=P), R),
    ("GMRES(30) rtol 1e-7", hip.GMRESKrylovKit(dim=30, rtol=1e-7, atol=1e-12, maxiter=150, Pl=P), R),
]
for label, ls, rhs in variants:
    ctx.solver_history()
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    x, ok, it = ls(J, rhs)
    ev1.record()
    torch.cuda.synchronize()
    h = ctx.solver_history()[0]
    # true residual of the unpreconditioned system
    r = J(x).add_(rhs, 1.0, -1.0)
    print(json.dumps(dict(solve=label, ok=ok, numops=it, ms=ev0.elapsed_time(ev1), its=len(h) - 1, r0=h[0], reach=milestones(h),
                          true_res_inf=r.norminf(), true_res_rel2=r.norm() / rhs.norm(),
                          hist=[float(f"{v / h[0]:.3e}") for v in h[::max(1, len(h) // 24)]])), flush=True)
ctx.close()
