"""BASELINE config 3 at full size: cGL2d 1024 x 1024 (h of examples/cGL2d.jl:82-87 kept: lx = pi*1024/41, ly = (pi/2)*1024/21),
trivial branch u = 0.  (i) ShiftInvert(sigma = 1, nev = 9) (EigArpack(1.0, :LM), cGL2d.jl:96,100) against the closed-form
spectrum r + lam_Lap(i, j) +- i nu; (ii) native PALC continuation in r across the first Hopf point r* = -lam_Lap(1, 1) with
detection and bisection (detect_bifurcation = 3).  Prints one JSON line per part.
Preconditioner: "block" = the 2x2-block spectral preconditioner (bk_precond_cgl_create: exact on the trivial branch at the
reference value of r, the role of the reference's sparse LU) or "laplace" = (Lap - I)^-1 per field (round 1).
Usage: python scripts/c3_fullsize.py [n=1024] [part: eig,hopf] [precond: block|laplace]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from bk_amd import continuation as Cn  # noqa: E402
from bk_amd import hip  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
parts = sys.argv[2].split(",") if len(sys.argv) > 2 else ["eig", "hopf"]
dims, ls_ = (n, n), (np.pi * n / 41, (np.pi / 2) * n / 21)
ctx = hip.Context(0)
if os.environ.get("BK_GMRES_CHUNK"):
    ctx.set_option("gmres_chunk", float(os.environ["BK_GMRES_CHUNK"]))      # 1: host-driven Arnoldi steps; >= 2: device-resident chunks
prob = hip.CGL2d(ctx, dims, ls_, r=0.5)
n2 = 2 * n * n
lam = []
for n_, l_ in zip(dims, ls_):
    h = 2 * l_ / n_
    lam.append(-(4 / h ** 2) * np.sin(np.pi * np.arange(1, n_ + 1) / (2 * (n_ + 1))) ** 2)
lap = np.sort((lam[0][:, None] + lam[1][None, :]).ravel())[::-1]              # Laplacian eigenvalues, descending
kind = sys.argv[3] if len(sys.argv) > 3 else "block"
NU, SIGMA = 1.0, 1.0


def solvers(r0):
    if kind == "laplace":
        ls_ = hip.GMRESIterativeSolvers(reltol=1e-10, restart=60, maxiter=600, Pl=hip.LaplacePreconditioner(prob, 1.0))
        return ls_, ls_
    return (hip.GMRESIterativeSolvers(reltol=1e-10, restart=60, maxiter=600, Pl=hip.CGLBlockPreconditioner(prob, r0, NU)),
            hip.GMRESIterativeSolvers(reltol=1e-10, restart=60, maxiter=600, Pl=hip.CGLBlockPreconditioner(prob, r0 - SIGMA, NU)))


zero = prob.vec(np.zeros(n2))


def timed(fn):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    r = fn()
    torch.cuda.synchronize()
    return time.perf_counter() - t0, r


if "eig" in parts:
    r0 = 0.5
    J = prob.jacobian(zero, r0)
    ls, lse = solvers(r0)
    eig = hip.ShiftInvert(SIGMA, lse, tol=1e-8, maxiter=300, hermitian=False, save_vectors=False)
    dt, (vals, _, ok, nops) = timed(lambda: eig(J, 9))
    exact = np.array([complex(r0 + l, s) for l in lap[:5] for s in (1.0, -1.0)])
    good = vals[~np.isnan(vals.real)]
    err = max(np.abs(exact - v).min() for v in good) if len(good) else None
    # every returned value is an eigenvalue AND the returned set is the rightmost one (no pair skipped)
    miss = max(np.abs(good - e).min() for e in exact[:len(good) - len(good) % 2]) if len(good) else None
    print(json.dumps(dict(part="C3 shift-invert eigensolve", precond=kind, n=n, seconds=dt, converged=bool(ok), inner_solves=nops,
                          inner_iterations=int(ctx.get_option("eig_last_inner_ops")), nvals=len(vals), n_converged=len(good),
                          max_error_vs_closed_form=err, max_missing=miss,
                          vals=[[float(v.real), float(v.imag)] for v in vals], exact_real=[float(r0 + l) for l in lap[:5]])),
          flush=True)

if "hopf" in parts:
    rstar = -lap[:3]                                   # r* of the first Hopf points (lam + r = 0)
    ls, lse = solvers(float(rstar[0]))
    eig = hip.ShiftInvert(SIGMA, lse, tol=1e-8, maxiter=300, hermitian=False, save_vectors=False)
    nopt = Cn.NewtonPar(tol=1e-9, max_iterations=20, linsolver=ls, eigsolver=eig)
    width = float(rstar[1] - rstar[0])
    p_start = float(rstar[0] - 1.6 * width)
    cp = Cn.ContinuationPar(ds=0.5 * width, dsmin=1e-3 * width, dsmax=0.6 * width, p_min=p_start - width, p_max=float(rstar[2]),
                            max_steps=6, nev=9, newton_options=nopt, n_inversion=4, max_bisection_steps=10,
                            dsmin_bisection=1e-4 * width, tol_stability=1e-10)
    alg = Cn.PALC(tangent="secant", theta=0.5, bls=hip.BorderingBLS(None, check_precision=False))
    ctx.set_option("eig_thick_start", 1)
    steps = []
    dt, br = timed(lambda: Cn.continuation_native(prob, zero, p_start, alg, cp, normC=Cn.norminf, bisection=True,
                                                  finalise_solution=lambda get, r: steps.append((r.p, r.n_unstable, r.eig_numops)) or True))
    print(json.dumps(dict(part="C3 Hopf detection + bisection", precond=kind, n=n, seconds=dt, rstar=[float(x) for x in rstar], param=br.param,
                          n_unstable=br.n_unstable, n_imag=br.n_imag, steps=steps,
                          specialpoint=[{k: (list(v) if isinstance(v, tuple) else v) for k, v in sp.items()} for sp in br.specialpoint])),
          flush=True)
ctx.close()
