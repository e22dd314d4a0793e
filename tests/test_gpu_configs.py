"""GPU tests of BASELINE.json's configs 2 and 3 at their FULL sizes (config 5 = test_gpu_fullsize.py, config 4 =
the same file under BK_FULLSIZE=256, config 1 = tests/test_oracle.py on the CPU path):

* C2  SH2d 512 x 512 (examples/SH2d-fronts.jl operator, lx = 16 pi, ly = 8 pi / sqrt 3, l = -0.1, nu = 1.3, hexagon
      state of :47-55, Pl = lu(L1 + I) :121): residual / JVP against the oracle's assembled sparse operator, one
      preconditioned GMRES solve and the PALC corrector against the oracle run at the SAME full size on the same inputs.
* C3  cGL2d 1024 x 1024 (examples/cGL2d.jl, lx = pi 1024/41, ly = (pi/2) 1024/21, parameters :90): residual / JVP
      against the oracle's sparse Jacobian, exactness of the DST preconditioner on (Lap - c) v, and the PALC bordered
      solve through its defining equations (size-independent residual identities, test_linear.jl:172-244).
"""
import math
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import bordered, krylov, operators, palc  # noqa: E402  (checker only)


def _hip():
    from bk_amd import hip
    return hip


def test_c2_sh2d_512_residual_jvp_gmres_corrector_match_oracle(ctx):
    hip = _hip()
    dims, ls_ = (512, 512), (16 * np.pi, 8 * np.pi / np.sqrt(3))
    sh = operators.SwiftHohenberg(dims, ls_)
    prob = hip.SwiftHohenberg(ctx, dims, ls_, l=-0.1, nu=1.3)
    rng = np.random.default_rng(2)
    mk = lambda Pl: (lambda J_, r, a0=0.0, a1=1.0: krylov.gmres_krylovkit(J_, r, a0, a1, krylovdim=30, maxiter=150,
                                                                          rtol=1e-9, atol=1e-12, Pl=Pl)[:3])
    # --- the hexagon state of the example (sol_hexa): Newton on ONE periodic cell (64 x 128 points, the same h), tiled
    # by even reflections to 8 x 4 cells = the full 512 x 512 domain, where it is an exact discrete solution.
    # eps * |L1| * |u| ~ 4e-10 at h_y = 0.057, hence Newton tolerances of 5e-9 in this config.
    cell, cell_l, tiles = (64, 128), (2 * np.pi, 2 * np.pi / np.sqrt(3)), (8, 4)
    shc = operators.SwiftHohenberg(cell, cell_l)
    pc = palc.Problem(lambda x, p: shc.F(x, p, 1.3), lambda x, p: (lambda dx: shc.dF(x, p, 1.3, dx)))
    lsc = mk(operators.dct_preconditioner(cell, cell_l, 1.0))
    ds, theta = -0.001, 0.5
    c0 = palc.newton(pc, shc.guess(), -0.1, lsc, tol=5e-9, max_iterations=40, normN=palc.norminf)
    c1 = palc.newton(pc, c0["u"], -0.1 + ds / 150.0, lsc, tol=5e-9, max_iterations=20, normN=palc.norminf)
    assert c0["converged"] and c1["converged"] and np.abs(c0["u"]).max() > 1.0
    idx = [np.concatenate([np.arange(nc) if c % 2 == 0 else np.arange(nc)[::-1] for c in range(T)])
           for nc, T in zip(cell, tiles)]
    tile = lambda a: np.ascontiguousarray(a.reshape(cell[1], cell[0])[np.ix_(idx[1], idx[0])]).reshape(-1)
    u, u1 = tile(c0["u"]), tile(c1["u"])
    v = rng.standard_normal(sh.N)
    # --- residual and JVP at full size: rounding of a 13-term stencil sum with cancellation, |L1| ~ 16/h^4
    scale = 64 * np.finfo(float).eps * (16.0 / min(2 * l / n for l, n in zip(ls_, dims)) ** 4)
    F = prob.residual(prob.vec(u), -0.1).numpy()
    Fo = sh.F(u, -0.1, 1.3)
    assert np.abs(Fo).max() <= 2e-8 and np.abs(F - Fo).max() <= scale * np.abs(u).max()
    J = prob.jacobian(prob.vec(u), -0.1)
    Jv = J(prob.vec(v), 0.3, 0.7).numpy()
    Jvo = 0.3 * v + 0.7 * sh.dF(u, -0.1, 1.3, v)
    assert np.abs(Jv - Jvo).max() <= scale * np.abs(v).max()
    # --- one preconditioned GMRES solve (GMRESKrylovKit, Pl = (L1 + I)^-1, random right-hand side) on both sides
    Plo = operators.dct_preconditioner(dims, ls_, 1.0)
    ols = mk(Plo)
    rhs = rng.standard_normal(sh.N)
    xo, oko, ito = ols(lambda dx: sh.dF(u, -0.1, 1.3, dx), rhs)
    P = hip.DCTPreconditioner(prob, 1.0)
    ls = hip.GMRESKrylovKit(dim=30, rtol=1e-9, atol=1e-12, maxiter=150, Pl=P)
    x, ok, it = ls(J, prob.vec(rhs))
    assert ok and oko and abs(it - ito) <= max(2, ito // 15), (it, ito)
    xn = x.numpy()
    assert np.abs(xn - xo).max() <= 1e-6 * np.abs(xo).max()
    # the preconditioned residual the solver controls: |Pl^-1 (J x - rhs)| <= rtol |Pl^-1 rhs|
    pres = Plo(sh.dF(u, -0.1, 1.3, xn) - rhs)
    assert np.linalg.norm(pres) <= 2e-9 * np.linalg.norm(Plo(rhs)) + 1e-12
    # --- the PALC corrector from the secant predictor, identical inputs on both sides (Palc.jl:187-305)
    z0, z1 = (u, -0.1), (u1, -0.1 + ds / 150.0)
    tau = palc.secant_tangent(z1, z0, ds, theta)
    zp = palc.add_tangent(z0, tau, ds)
    # dF/dp on both sides in the cancellation-free form ((p + eps) - p) / eps * u (dF/dl = u; bk_residual_dparam on the
    # device, Problem.dparam_factor in the oracle): both sides then solve THE SAME right-hand sides and the operator-
    # application counts can be compared two-sidedly.  (The literal two-residual quotient of Palc.jl:239-240 puts 4e-5
    # relative white noise -- eps_mach |L1 u| / eps -- on dF/dp, far above rtol = 1e-9: oracle 44 + 13 applications with a
    # restart, another noise realisation on the device; that comparison was one-sided in round 2.)
    oprob = palc.Problem(lambda x_, p: sh.F(x_, p, 1.3), lambda x_, p: (lambda dx: sh.dF(x_, p, 1.3, dx)),
                         dparam_factor=lambda x_, p: x_)
    obls = lambda *a, **k: bordered.bordering_bls(ols, *a, check_precision=False, **k)
    so = palc.newton_palc(oprob, z0, tau, zp, ds, theta, obls, tol=5e-9, max_iterations=15, normN=palc.norminf)
    B = hip.BorderedArray
    sg = hip.newton_palc_native(prob, B(prob.vec(z0[0]), z0[1]), B(prob.vec(tau[0]), tau[1]), B(prob.vec(zp[0]), zp[1]),
                                ds, theta, hip.BorderingBLS(ls, check_precision=False), tol=5e-9, max_iterations=15,
                                norm_inf=True)
    assert so["converged"] and sg["converged"] and sg["itnewton"] == so["itnewton"] >= 1
    r0 = so["residuals"][0]
    # same predictor => same residual up to the rounding of one stencil evaluation, eps * |L1| * |u| ~ 4e-10 at h_y = 0.057
    # (the 1e-10-relative figure of the 3-D configs, h = 0.196, is below that floor here); the later entries of the
    # history additionally carry the linear-solve error rtol * |rhs| of either side
    floor = scale * np.abs(u).max()
    assert abs(sg["residuals"][0] - r0) <= floor, (sg["residuals"], so["residuals"])
    for rg, ro in zip(sg["residuals"][1:], so["residuals"][1:]):
        assert abs(rg - ro) <= 2 * floor + 1e-7 * r0, (sg["residuals"], so["residuals"])
    assert abs(sg["u"].p - so["p"]) <= 1e-9 and np.abs(sg["u"].u.numpy() - so["u"]).max() <= 1e-6
    # Counts inside the corrector: the right-hand side of the R solve is the predictor residual itself, |F| ~ 1e-5 carrying
    # the stencil's rounding noise ~4e-10 absolute = 4e-5 relative, far above rtol = 1e-9 -- and the noise realisation of the
    # oracle's assembled sparse L1 differs from the matrix-free kernel's, so that solve chases different noise on the two
    # sides (oracle 44 with a restart + 13, HIP 32 in total, with either form of dF/dp).  Inside the corrector only the upper
    # bound is meaningful ...
    assert sg["itlineartot"] <= so["itlineartot"] + max(4, so["itlineartot"] // 3), (sg["itlineartot"], so["itlineartot"])
    # ... and the lower bound comes from handing BOTH sides byte-identical inputs: the oracle's residual and dF/dp at the
    # predictor, uploaded, through one bordered solve (solve_bls_palc, src/LinearBorderSolver.jl:16-36).  Same right-hand
    # sides, same algorithm => the same operator-application count per solve (+-2 where the estimate crosses the tolerance
    # within rounding) and the same solution; a HIP solve that stopped early would show up in both.
    from bk_amd import continuation as Cn
    Rf = sh.F(zp[0], zp[1], 1.3)
    dR = palc.dF_dparam(oprob, zp[0], zp[1])
    n_res = palc.arc_length_eq(zp[0], z0[0], zp[1] - z0[1], tau[0], tau[1], theta, ds)
    Jo = lambda dx: sh.dF(zp[0], zp[1], 1.3, dx)
    xo2, plo, oko2, ito2 = bordered.solve_bls_palc(obls, theta, tau[0], tau[1], Jo, dR, Rf, n_res)
    Jg = prob.jacobian(prob.vec(zp[0]), zp[1])
    xg2, plg, okg2, itg2 = Cn.solve_bls_palc(hip.BorderingBLS(ls, check_precision=False), theta,
                                             B(prob.vec(tau[0]), tau[1]), Jg, prob.vec(dR), prob.vec(Rf), n_res)
    assert oko2 and okg2
    assert all(abs(a - b) <= 2 for a, b in zip(itg2, ito2)), (itg2, ito2)
    assert abs(plg - plo) <= 1e-8 * max(abs(plo), 1e-3) + 1e-12, (plg, plo)
    xg2n = xg2.numpy()
    assert np.abs(xg2n - xo2).max() <= 1e-5 * np.abs(xo2).max() + 1e-12
    # the bordered system itself, through the oracle's operator: |J dX + dl dR - R| small in the preconditioned norm
    rb = Plo(sh.dF(zp[0], zp[1], 1.3, xg2n) + plg * dR - Rf)
    assert np.linalg.norm(rb) <= 5e-9 * (np.linalg.norm(Plo(Rf)) + abs(plg) * np.linalg.norm(Plo(dR))) + 1e-12
    # and the literal quotient still shows the round-2 pathology on the oracle side (the reference's own formula): at least
    # as many applications as the cancellation-free form
    oprob_lit = palc.Problem(lambda x_, p: sh.F(x_, p, 1.3), lambda x_, p: (lambda dx: sh.dF(x_, p, 1.3, dx)))
    sl = palc.newton_palc(oprob_lit, z0, tau, zp, ds, theta, obls, tol=5e-9, max_iterations=15, normN=palc.norminf)
    assert sl["converged"] and sl["itlineartot"] >= so["itlineartot"], (sl["itlineartot"], so["itlineartot"])


def test_c3_cgl2d_1024_jvp_preconditioner_bordered_solve(ctx):
    hip = _hip()
    dims, ls_ = (1024, 1024), (np.pi * 1024 / 41, (np.pi / 2) * 1024 / 21)
    c = operators.CGL2d(dims, ls_)
    prob = hip.CGL2d(ctx, dims, ls_, r=0.5)
    rng = np.random.default_rng(3)
    n2 = 2 * c.n
    u, v, w = 0.5 * rng.standard_normal(n2), rng.standard_normal(n2), rng.standard_normal(n2)
    pars = c.default_params()
    pars["r"] = 0.7
    hmin = min(2 * l / n for l, n in zip(ls_, dims))
    scale = 64 * np.finfo(float).eps * (8.0 / hmin ** 2 + 10.0)
    F = prob.residual(prob.vec(u), 0.7).numpy()
    assert np.abs(F - c.F(u, **pars)).max() <= scale * max(1.0, np.abs(u).max() ** 5)
    J = prob.jacobian(prob.vec(u), 0.7)
    V, W = prob.vec(v), prob.vec(w)
    Jv = J(V, -0.2, 0.9)
    assert np.abs(Jv.numpy() - (-0.2 * v + 0.9 * (c.J(u, **pars) @ v))).max() <= scale * np.abs(v).max() * 10
    # linearity (the Jacobian is NOT symmetric: rotation nu, mu terms)
    lhs = J(V.copy().add_(W, -0.7, 2.0))
    rhs = J(V).copy().add_(J(W), -0.7, 2.0)
    assert lhs.add_(rhs, -1.0).norminf() <= 1e-9 * rhs.norminf()
    # DST preconditioner: (Lap - cI)^-1 (Lap - cI) v = v with the operator taken from the trivial-state Jacobian
    # (u = 0, nu = 0, r = -c: J = Lap - c on both fields)
    c0 = 1.0
    prob0 = hip.CGL2d(ctx, dims, ls_, r=-c0, nu=0.0)
    J0 = prob0.jacobian(prob0.vec(np.zeros(n2)), -c0)
    P = hip.LaplacePreconditioner(prob, c0)
    back = P.ldiv(J0(V))
    assert back.add_(V, -1.0).norminf() <= 1e-9 * V.norminf()
    # PALC bordered solve on the cGL Jacobian: the solution satisfies both block rows (residualBEC,
    # LinearBorderSolver.jl:146-166) to the solver tolerance
    ls = hip.GMRESIterativeSolvers(reltol=1e-10, restart=60, maxiter=900, Pl=P)
    dR, dzu, R = (prob.vec(rng.standard_normal(n2)) for _ in range(3))
    theta, dzp, nn = 0.5, 0.4, 0.3
    dX, dl, ok, it = hip.BorderingBLS(ls, check_precision=False)(J, dR, dzu, dzp, R, nn, theta, 1 - theta,
                                                                  dotscale=1.0 / n2)
    assert ok, it
    top = J(dX).add_(dR, dl).add_(R, -1.0)                       # J dX + dl dR - R
    assert top.norm() <= 1e-6 * R.norm(), (top.norm(), R.norm())
    bot = theta * dzu.inner(dX) / n2 + (1 - theta) * dzp * dl - nn
    assert abs(bot) <= 1e-9


def _cgl_full():
    n = 1024
    dims, ls_ = (n, n), (np.pi * n / 41, (np.pi / 2) * n / 21)
    lam = []
    for n_, l_ in zip(dims, ls_):
        h = 2 * l_ / n_
        lam.append(-(4 / h ** 2) * np.sin(np.pi * np.arange(1, n_ + 1) / (2 * (n_ + 1))) ** 2)
    lap = np.sort((lam[0][:, None] + lam[1][None, :]).ravel())[::-1]          # Dirichlet Laplacian eigenvalues, descending
    return dims, ls_, lap


def _cgl_solvers(hip, prob, kind, r0=0.5, sigma=1.0, nu=1.0):
    """(linear solver, eigensolver) with the (Lap - I)^-1 preconditioner of round 1 ("laplace") or the 2x2-block spectral
    preconditioner ("block": exact on the trivial branch at r0 -- the role of the reference's sparse LU)."""
    if kind == "laplace":
        P = hip.LaplacePreconditioner(prob, 1.0)
        ls = hip.GMRESIterativeSolvers(reltol=1e-10, restart=60, maxiter=600, Pl=P)
        return ls, hip.ShiftInvert(sigma, ls, tol=1e-8, maxiter=300, hermitian=False, save_vectors=False)
    ls = hip.GMRESIterativeSolvers(reltol=1e-10, restart=60, maxiter=600, Pl=hip.CGLBlockPreconditioner(prob, r0, nu))
    lse = hip.GMRESIterativeSolvers(reltol=1e-10, restart=60, maxiter=600, Pl=hip.CGLBlockPreconditioner(prob, r0 - sigma, nu))
    return ls, hip.ShiftInvert(sigma, lse, tol=1e-8, maxiter=300, hermitian=False, save_vectors=False)


@pytest.mark.parametrize("precond", ["laplace", "block"])
def test_c3_cgl_1024_shift_invert_eigenvalues_match_closed_form(ctx, precond):
    """examples/cGL2d.jl:96,100 at full size: ShiftInvert(sigma = 1, nev = 9) (EigArpack(1.0, :LM)) on the trivial state at
    r = 0.5.  The Jacobian at u = 0 is Lap (+) Lap + [[r, -nu], [nu, r]], so its eigenvalues are r + lam_Lap(i, j) +- i nu in
    closed form; on this domain (25 x the example's) consecutive ones are 1e-3 apart, 1.118 away from the shift."""
    hip = _hip()
    dims, ls_, lap = _cgl_full()
    prob = hip.CGL2d(ctx, dims, ls_, r=0.5)
    n2 = 2 * dims[0] * dims[1]
    ls, eig = _cgl_solvers(hip, prob, precond)
    vals, _, ok, nops = eig(prob.jacobian(prob.vec(np.zeros(n2)), 0.5), 9)
    if precond == "block":                      # Pl^-1 (J - sigma I) = I on the trivial state: one GMRES iteration per solve
        assert int(ctx.get_option("eig_last_inner_ops")) <= 2 * nops
    assert ok and len(vals) == 10 and not np.isnan(vals.real).any()            # 9 -> 10: the cut never splits a conjugate pair
    exact = np.array([complex(0.5 + l, s) for l in lap[:5] for s in (1.0, -1.0)])
    # the returned set IS the rightmost five pairs (nothing skipped, nothing spurious), to 1e-8
    assert max(np.abs(exact - v).min() for v in vals) <= 1e-8
    assert max(np.abs(vals - e).min() for e in exact) <= 1e-8
    assert np.all(np.diff(vals.real) <= 1e-12)                                  # sorted by decreasing real part


@pytest.mark.parametrize("precond", ["laplace", "block"])
def test_c3_cgl_1024_first_hopf_point_is_detected_and_bracketed(ctx, precond):
    """Native PALC continuation in r along the trivial branch at full size across the first Hopf point r* = -lam_Lap(1, 1)
    (closed form): n_unstable goes 0 -> 2 with a complex pair, the special point is classified `hopf`, and the bisection
    (locate_bifurcation!, src/Bifurcations.jl:159-349; a few steps only, each is a full eigensolve) brackets r*."""
    hip = _hip()
    from bk_amd import continuation as Cn
    dims, ls_, lap = _cgl_full()
    prob = hip.CGL2d(ctx, dims, ls_, r=0.5)
    n2 = 2 * dims[0] * dims[1]
    rstar = -lap[:2]
    ls, eig = _cgl_solvers(hip, prob, precond, r0=float(rstar[0]))
    width = float(rstar[1] - rstar[0])
    nopt = Cn.NewtonPar(tol=1e-9, max_iterations=20, linsolver=ls, eigsolver=eig)
    cp = Cn.ContinuationPar(ds=0.5 * width, dsmin=1e-3 * width, dsmax=0.6 * width, p_min=float(rstar[0] - 2 * width),
                            p_max=float(rstar[1]), max_steps=2, nev=9, newton_options=nopt, n_inversion=2,
                            max_bisection_steps=3, dsmin_bisection=1e-4 * width)
    alg = Cn.PALC(tangent="secant", theta=0.5, bls=hip.BorderingBLS(None, check_precision=False))
    ctx.set_option("eig_thick_start", 1)
    try:
        br = Cn.continuation_native(prob, prob.vec(np.zeros(n2)), float(rstar[0] - 0.7 * width), alg, cp, normC=Cn.norminf,
                                    bisection=True)
    finally:
        ctx.set_option("eig_thick_start", 0)
    assert br.n_unstable[0] == 0 and br.n_unstable[-1] == 2 and br.n_imag[-1] == 2
    assert len(br.specialpoint) == 1
    sp = br.specialpoint[0]
    lo, hi = sp["interval"]
    assert sp["type"] == "hopf" and lo - 1e-9 <= rstar[0] <= hi + 1e-9 and hi - lo <= 0.6 * width, (sp, rstar)
