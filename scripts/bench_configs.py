"""Timings of BASELINE.json's cache-resident configs (not the headline bench): C2 = SH2d 512 x 512 PALC corrector from the
tiled hexagon branch point (the setup of tests/test_gpu_configs.py), C3 = cGL2d 1024 x 1024 JVP / bordered solve.
Vectors of 2 / 16 MiB sit in the 256 MB Infinity Cache: these configs are launch- and synchronisation-bound, not
HBM-bound.  Prints one JSON line per config."""
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from bk_amd import hip  # noqa: E402

ctx = hip.Context(0)


def apply_opts():
    """Library options of an experiment, BK_OPTS="key=value,key=value".  Applied AFTER the cell solves that define the input of the
    timed step (their last digits move with solver options, and the secant of two close solutions amplifies them), so that every
    option set times the same step."""
    for kv in os.environ.get("BK_OPTS", "").split(","):
        if "=" in kv:
            ctx.set_option(kv.split("=")[0], float(kv.split("=")[1]))


ONLY = os.environ.get("BK_ONLY", "")                                       # "c2": stop after config 2
if os.environ.get("BK_GMRES_CHUNK"):
    ctx.set_option("gmres_chunk", float(os.environ["BK_GMRES_CHUNK"]))      # 1: host-driven Arnoldi steps; >= 2: device-resident chunks


def timeit(fn, reps=5, warm=1):
    for _ in range(warm):
        r = fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        r = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps, r


# ---- C2
cell, cell_l, tiles = (64, 128), (2 * np.pi, 2 * np.pi / np.sqrt(3)), (8, 4)
cprob = hip.SwiftHohenberg(ctx, cell, cell_l, l=-0.1, nu=1.3)
cls_ = hip.GMRESKrylovKit(dim=30, rtol=1e-9, atol=1e-12, maxiter=150, Pl=hip.DCTPreconditioner(cprob, 1.0))
X, Y = np.meshgrid(*[-l + 2 * l / n * np.arange(n) for n, l in zip(cell, cell_l)], indexing="ij")
g = np.cos(X) + np.cos(X / 2) * np.cos(np.sqrt(3) * Y / 2)
g = ((g - g.min()) / (g - g.min()).max() - 0.25) * 1.7
u0c = cprob.vec(np.ascontiguousarray(g.reshape(-1, order="F")))
ds, theta = -0.001, 0.5
c0 = hip.newton_native(cprob, u0c, -0.1, cls_, tol=5e-9, max_iterations=40, norm_inf=True)
c1 = hip.newton_native(cprob, c0["u"], -0.1 + ds / 150, cls_, tol=5e-9, max_iterations=20, norm_inf=True)
assert c0["converged"] and c1["converged"]
apply_opts()
idx = [np.concatenate([np.arange(nc) if c % 2 == 0 else np.arange(nc)[::-1] for c in range(T)]) for nc, T in zip(cell, tiles)]
tile = lambda a: np.ascontiguousarray(a.reshape(cell[1], cell[0])[np.ix_(idx[1], idx[0])]).reshape(-1)
dims, ls_ = (512, 512), (16 * np.pi, 8 * np.pi / np.sqrt(3))
prob = hip.SwiftHohenberg(ctx, dims, ls_, l=-0.1, nu=1.3)
ls = hip.GMRESKrylovKit(dim=30, rtol=1e-9, atol=1e-12, maxiter=150, Pl=hip.DCTPreconditioner(prob, 1.0))
B = hip.BorderedArray
z0, z1 = B(prob.vec(tile(c0["u"].numpy())), -0.1), B(prob.vec(tile(c1["u"].numpy())), -0.1 + ds / 150)
tau = z1.copy().add_(z0, -1.0)
nrm = math.sqrt(tau.u.inner(tau.u) / prob.nglobal * theta + tau.p * tau.p * (1 - theta))
tau.scale_(math.copysign(1.0, ds) / nrm)
zp = z0.copy().add_(tau, ds)
step = lambda: hip.newton_palc_native(prob, z0, tau, zp, ds, theta, hip.BorderingBLS(ls, check_precision=False), tol=0.0,
                                      max_iterations=1, norm_inf=True)
dt, r = timeit(step, reps=10, warm=2)
J = prob.jacobian(z0.u, -0.1)
v = prob.vec(np.random.default_rng(0).standard_normal(prob.nglobal))
tj, _ = timeit(lambda: J(v), reps=200, warm=10)
print(json.dumps(dict(config="C2 SH2d 512x512 PALC corrector iteration", ms_per_step=dt * 1e3, steps_per_s=1 / dt,
                      itlinear=r["itlineartot"], ms_per_operator_application=dt * 1e3 / max(1, r["itlineartot"]),
                      jvp_us=tj * 1e6, jvp_gbs=24.0 * prob.nglobal / tj / 1e9, residuals=r["residuals"])))

if ONLY == "c2":
    sys.exit(0)
# ---- C3
dims3, ls3 = (1024, 1024), (np.pi * 1024 / 41, (np.pi / 2) * 1024 / 21)
p3 = hip.CGL2d(ctx, dims3, ls3, r=0.5)
rng = np.random.default_rng(3)
n2 = 2 * 1024 * 1024
u3 = p3.vec(0.5 * rng.standard_normal(n2))
J3 = p3.jacobian(u3, 0.7)
v3 = p3.vec(rng.standard_normal(n2))
tj3, _ = timeit(lambda: J3(v3), reps=200, warm=10)
P3 = hip.LaplacePreconditioner(p3, 1.0)
tp3, _ = timeit(lambda: P3.ldiv(v3), reps=20, warm=2)
ls3_ = hip.GMRESIterativeSolvers(reltol=1e-10, restart=60, maxiter=900, Pl=P3)
dR, dzu, R = (p3.vec(rng.standard_normal(n2)) for _ in range(3))
tb, rb = timeit(lambda: hip.BorderingBLS(ls3_, check_precision=False)(J3, dR, dzu, 0.4, R, 0.3, 0.5, 0.5, dotscale=1.0 / n2),
                reps=3, warm=1)
print(json.dumps(dict(config="C3 cGL2d 1024x1024", jvp_us=tj3 * 1e6, jvp_gbs=24.0 * n2 / tj3 / 1e9, precond_ms=tp3 * 1e3,
                      bordered_solve_ms=tb * 1e3, bordered_itlinear=list(rb[3]), bordered_converged=bool(rb[2]))))
# the workload SURVEY section 8(d) names for C3 is the TRIVIAL branch (u = 0, continuation in r from 0.5): there the 2x2-block
# spectral preconditioner (bk_precond_cgl_create) is the exact inverse -- the role of the reference's sparse LU
z3 = p3.vec(np.zeros(n2))
J30 = p3.jacobian(z3, 0.5)
for name, Pk in (("laplace", P3), ("block", hip.CGLBlockPreconditioner(p3, 0.5, 1.0))):
    lsk = hip.GMRESIterativeSolvers(reltol=1e-10, restart=60, maxiter=900, Pl=Pk)
    tk, rk = timeit(lambda: hip.BorderingBLS(lsk, check_precision=False)(J30, dR, dzu, 0.4, R, 0.3, 0.5, 0.5, dotscale=1.0 / n2),
                    reps=3, warm=1)
    tpk, _ = timeit(lambda: Pk.ldiv(v3), reps=20, warm=2)
    print(json.dumps(dict(config="C3 cGL2d 1024x1024, trivial branch r = 0.5", precond=name, precond_ms=tpk * 1e3,
                          bordered_solve_ms=tk * 1e3, bordered_itlinear=list(rk[3]), bordered_converged=bool(rk[2]))))

