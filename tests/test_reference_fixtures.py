"""Reference-pinned parity: consumes tests/golden/julia_fixtures.json, the dump of the REAL BifurcationKit.jl (+ KrylovKit,
Arpack) produced by julia/gen_fixtures.jl on the three problems of the hot path.  There is no Julia in this repository's
build image, so the fixture file does not exist yet: every test here SKIPS until someone runs the generator (DESIGN.md
section 2: "parity unpinned").  With the file present, the CPU tests pin the oracle to the reference and the `-m gpu`
tests pin the HIP path to it, through the same summaries (norms, a probe inner product, entries at fixed positions).

Reference call sites: examples/SH3d.jl:88-166, examples/SH2d-fronts.jl:8-127, examples/cGL2d.jl:6-100,
test/newton/test_newton.jl:23-52, test/linear_solvers/test_linear.jl:106-244."""
import json
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
FIXTURES = os.environ.get("BK_FIXTURES", os.path.join(HERE, "golden", "julia_fixtures.json"))


@pytest.fixture(scope="module")
def fx():
    if not os.path.exists(FIXTURES):
        pytest.skip("tests/golden/julia_fixtures.json missing: run `julia julia/gen_fixtures.jl` on a machine with "
                    "Julia + BifurcationKit.jl (the reference cannot run in this image)")
    return json.load(open(FIXTURES))


def probe(k, N):
    i = np.arange(1, N + 1, dtype=np.float64)
    return np.sin((0.37 + 0.11 * k) * i) + 0.5 * np.cos((1.3 + 0.07 * k) * i + 0.1)


def positions(N):
    return [0, 1, 16, N // 3 - 1, N // 2, N - 1]            # gen_fixtures.jl positions(N), 0-based


def summary(v):
    v = np.asarray(v, dtype=np.float64)
    return dict(norm2=float(np.linalg.norm(v)), norminf=float(np.abs(v).max()), dot_probe9=float(v @ probe(9, v.size)),
                entries=[float(v[i]) for i in positions(v.size)])


def close(mine, ref, rtol, what=""):
    s = summary(mine)
    scale = max(ref["norminf"], 1e-300)
    assert abs(s["norm2"] - ref["norm2"]) <= rtol * ref["norm2"] + 1e-300, (what, s["norm2"], ref["norm2"])
    assert abs(s["norminf"] - ref["norminf"]) <= rtol * scale, (what, s["norminf"], ref["norminf"])
    assert abs(s["dot_probe9"] - ref["dot_probe9"]) <= rtol * ref["norm2"] * np.sqrt(len(np.asarray(mine))), what
    assert np.allclose(s["entries"], ref["entries"], rtol=0, atol=rtol * scale), (what, s["entries"], ref["entries"])


def sh_guess(dims, ls):
    ax = [-l + 2.0 * l / n * np.arange(n) for n, l in zip(dims, ls)]
    if len(dims) == 3:
        X, Y, Z = np.meshgrid(*ax, indexing="ij")
        s = np.cos(X) * np.cos(Y) + 0 * Z
        s = (s - s.min()); s = s / s.max() * 1.2
    else:
        X, Y = np.meshgrid(*ax, indexing="ij")
        s = np.cos(X) + np.cos(X / 2) * np.cos(np.sqrt(3) * Y / 2)
        s = s - s.min(); s = s / s.max(); s = (s - 0.25) * 1.7
    return np.ascontiguousarray(s.reshape(-1, order="F"))


# (sh3d_64 / sh2d_128x64, round 6: grids on which the library's default Arnoldi step is the stencil-free one, with the reference
# example's own Pl = cholesky(L1); a case the fixture file does not hold is skipped)
SH_CASES = ["sh3d_22", "sh2d_151x100", "sh3d_64", "sh2d_128x64"]


# ------------------------------------------------------------------------------------------------ oracle vs reference (CPU)
@pytest.mark.parametrize("name", SH_CASES)
def test_oracle_swift_hohenberg_matches_reference(fx, name):
    import scipy.sparse.linalg as spla
    from oracle import bordered, krylov, operators, palc
    if name not in fx:
        pytest.skip(f"{name}: not in this fixture file (julia/gen_fixtures.jl emits it from round 6 on)")
    c = fx[name]
    dims, ls, l, nu = tuple(c["dims"]), tuple(c["ls"]), c["l"], c["nu"]
    sh = operators.SwiftHohenberg(dims, ls)
    u0 = sh_guess(dims, ls)
    N = sh.N
    close(sh.F(u0, l, nu), c["F_u0"], 1e-11, "F")
    close(sh.dF(u0, l, nu, probe(1, N)), c["dF_u0_probe1"], 1e-11, "dF")
    lu = spla.splu(sh.L1.tocsc())
    ols = lambda J, r, a0=0.0, a1=1.0: krylov.gmres_krylovkit(J, r, a0, a1, krylovdim=30, maxiter=150, rtol=1e-9, atol=1e-12,
                                                               Pl=lu.solve)[:3]
    oprob = palc.Problem(lambda x, p: sh.F(x, p, nu), lambda x, p: (lambda dx: sh.dF(x, p, nu, dx)))
    so = palc.newton(oprob, u0, l, ols, tol=1e-8, max_iterations=20, normN=palc.norminf)
    rn = c["newton"]
    assert so["converged"] == rn["converged"] and so["itnewton"] == rn["itnewton"]
    assert abs(so["residuals"][0] - rn["residuals"][0]) <= 1e-10 * rn["residuals"][0]      # north-star figure
    assert abs(so["itlineartot"] - rn["itlineartot"]) <= 2 * rn["itnewton"]
    close(so["u"], rn["u"], 1e-7, "newton u")
    us = so["u"]
    J = lambda dx: sh.dF(us, l, nu, dx)
    r1, r2, r3 = probe(1, N), probe(2, N), probe(3, N)
    x, ok, it = ols(J, r1)
    assert ok == c["gmres"]["converged"] and abs(it - c["gmres"]["numops"]) <= 2
    close(x, c["gmres"]["x"], 1e-7, "gmres")
    x, ok, it = ols(J, r1, 0.3, 0.9)
    assert ok == c["gmres_shift"]["converged"] and abs(it - c["gmres_shift"]["numops"]) <= 2
    close(x, c["gmres_shift"]["x"], 1e-7, "gmres shift")
    for key, fn in (("minres", krylov.minres_krylovjl), ("cg", krylov.cg_krylovjl)):      # Krylov.jl's symmetric solvers
        if key in c:
            x, ok, it = fn(J, r1, c[key]["a0"], c[key]["a1"], atol=1e-13, rtol=1e-10, M=lu.solve)
            assert ok == c[key]["converged"] and abs(it - c[key]["niter"]) <= max(2, c[key]["niter"] // 10), (key, it)
            close(x, c[key]["x"], 1e-7, key)
    if "gmres_is_pr" in c:                                     # GMRESIterativeSolvers with Pl and Pr (src/LinearSolver.jl:178,198-201)
        import scipy.sparse as sp
        g = c["gmres_is_pr"]
        lu1 = spla.splu((sh.L1 + g["pl_shift"] * sp.identity(N)).tocsc())
        lu3 = spla.splu((sh.L1 + g["pr_shift"] * sp.identity(N)).tocsc())
        x, ok, it = krylov.gmres_iterativesolvers(J, r1, g["a0"], g["a1"], restart=g["restart"], maxiter=4000, reltol=g["reltol"],
                                                  Pl=lu1.solve, Pr=lu3.solve)
        assert ok == g["converged"] and abs(it - g["niter"]) <= max(2, g["niter"] // 20), (it, g["niter"])
        close(x, g["x"], 1e-7, "gmres Pl + Pr")
    dX, dl, ok, its = bordered.bordering_bls(ols, J, r2, r3, 0.4, r1, 0.3, 0.5, 0.5, check_precision=False,
                                             dotp=lambda a, b: float(a @ b) / N)
    assert abs(dl - c["bordering"]["dl"]) <= 1e-7 * max(1.0, abs(c["bordering"]["dl"]))
    close(dX, c["bordering"]["dX"], 1e-6, "bordering")
    assert all(abs(a - b) <= 2 for a, b in zip(its, c["bordering"]["itlinear"]))
    obls = lambda *a, **k: bordered.bordering_bls(ols, *a, check_precision=False, **k)
    br = palc.continuation(oprob, us, l, ls=ols, bls=obls, tangent="bordered", normC=palc.norminf, ds=-0.001, dsmin=1e-4,
                           dsmax=0.005, p_min=-0.1, p_max=0.15, max_steps=len(c["branch"]["param"]) - 1, tol=1e-9,
                           max_iterations=15)
    assert np.allclose(br.param, c["branch"]["param"], rtol=0, atol=1e-8)
    assert all(abs(a - b) <= 1 for a, b in zip(br.itnewton, c["branch"]["itnewton"]))


def test_oracle_cgl_matches_reference(fx):
    from oracle import operators
    c = fx["cgl_41x21"]
    cg = operators.CGL2d(tuple(c["dims"]), tuple(c["ls"]))
    n2 = 2 * cg.n
    pars = dict(cg.default_params(), r=1.2)
    u, du = 0.4 * probe(5, n2), probe(6, n2)
    close(cg.F(u, **pars), c["F_probe"], 1e-11, "cgl F")
    close(cg.J(u, **pars) @ du, c["J_probe_du"], 1e-11, "cgl J")
    dense = np.linalg.eigvals(cg.J(np.zeros(n2), **cg.default_params()).toarray())
    for re, im in c["eig_trivial_r0.5"]["vals"]:
        assert np.abs(dense - complex(re, im)).min() <= 1e-8


# ------------------------------------------------------------------------------------------------ HIP vs reference (GPU)
@pytest.mark.gpu
@pytest.mark.parametrize("name", SH_CASES)
def test_hip_swift_hohenberg_matches_reference(fx, ctx, name):
    from bk_amd import continuation as Cn
    from bk_amd import hip
    if name not in fx:
        pytest.skip(f"{name}: not in this fixture file (julia/gen_fixtures.jl emits it from round 6 on)")
    c = fx[name]
    dims, ls, l, nu = tuple(c["dims"]), tuple(c["ls"]), c["l"], c["nu"]
    prob = hip.SwiftHohenberg(ctx, dims, ls, l=l, nu=nu)
    N = prob.nglobal
    u0 = prob.vec(sh_guess(dims, ls))
    close(prob.residual(u0, l).numpy(), c["F_u0"], 1e-10, "F")
    close(prob.jacobian(u0, l)(prob.vec(probe(1, N))).numpy(), c["dF_u0_probe1"], 1e-10, "dF")
    P = hip.DCTPreconditioner(prob, 0.0)                                            # = cholesky(L1) \ .
    ls_ = hip.GMRESKrylovKit(dim=30, rtol=1e-9, atol=1e-12, maxiter=150, Pl=P)
    sn = hip.newton_native(prob, u0, l, ls_, tol=1e-8, max_iterations=20, norm_inf=True)
    rn = c["newton"]
    assert sn["converged"] == rn["converged"] and sn["itnewton"] == rn["itnewton"]
    assert abs(sn["residuals"][0] - rn["residuals"][0]) <= 1e-10 * rn["residuals"][0]
    close(sn["u"].numpy(), rn["u"], 1e-7, "newton u")
    J = prob.jacobian(sn["u"], l)
    r1, r2, r3 = (prob.vec(probe(k, N)) for k in (1, 2, 3))
    # KrylovKit's numops, whatever the block size of the library's Arnoldi process (single steps, blocks of 4) and whichever form of
    # the preconditioned operator it iterates on (stencil-free forced / the literal chain): the counter is the reference's, not ours
    for sstep, sfree in ((-1, 1), (0, 1), (4, 2), (4, 0)):
        ctx.set_option("gmres_sstep", sstep)
        ctx.set_option("gmres_stencil_free", sfree)
        try:
            x, ok, it = ls_(J, r1)
            assert ok == c["gmres"]["converged"] and abs(it - c["gmres"]["numops"]) <= 2, (sstep, sfree, it, c["gmres"]["numops"])
            close(x.numpy(), c["gmres"]["x"], 1e-7, "gmres")
            x, ok, it = ls_(J, r1, 0.3, 0.9)
            assert abs(it - c["gmres_shift"]["numops"]) <= 2, (sstep, sfree, it, c["gmres_shift"]["numops"])
            close(x.numpy(), c["gmres_shift"]["x"], 1e-7, "gmres shift")
        finally:
            ctx.set_option("gmres_sstep", -1)
            ctx.set_option("gmres_stencil_free", 1)
    if "gmres_is_pr" in c:                            # GMRESIterativeSolvers with Pl and Pr: bk_gmres_opts.pr
        g = c["gmres_is_pr"]
        lis = hip.GMRESIterativeSolvers(reltol=g["reltol"], restart=g["restart"], maxiter=4000,
                                        Pl=hip.DCTPreconditioner(prob, g["pl_shift"]), Pr=hip.DCTPreconditioner(prob, g["pr_shift"]))
        x, ok, it = lis(J, r1, g["a0"], g["a1"])
        assert ok == g["converged"] and abs(it - g["niter"]) <= max(2, g["niter"] // 20), (it, g["niter"])
        close(x.numpy(), g["x"], 1e-7, "gmres Pl + Pr")
    for key in ("minres", "cg"):                      # the fused-pass path (3-D) / the separate passes (2-D stencil)
        if key in c:
            ks = hip.KrylovLSSymmetric(key, atol=1e-13, rtol=1e-10, Pl=P)
            x, ok, it = ks(J, r1, c[key]["a0"], c[key]["a1"])
            assert ok == c[key]["converged"] and abs(it - c[key]["niter"]) <= max(2, c[key]["niter"] // 10), (key, it)
            close(x.numpy(), c[key]["x"], 1e-7, key)
    dX, dl, ok, its = hip.BorderingBLS(ls_, check_precision=False)(J, r2, r3, 0.4, r1, 0.3, 0.5, 0.5, dotscale=1.0 / N)
    assert abs(dl - c["bordering"]["dl"]) <= 1e-7 * max(1.0, abs(c["bordering"]["dl"]))
    close(dX.numpy(), c["bordering"]["dX"], 1e-6, "bordering")
    mf = hip.MatrixFreeBLS(hip.GMRESKrylovKit(dim=30, rtol=1e-9, atol=1e-12, maxiter=150))
    dX, dl, ok, it = mf(J, r2, r3, 0.4, r1, 0.3, 0.5, 0.5, dotscale=1.0 / N)
    assert ok == c["matrixfree"]["converged"]
    if ok:
        assert abs(dl - c["matrixfree"]["dl"]) <= 1e-6 * max(1.0, abs(c["matrixfree"]["dl"]))
        close(dX.numpy(), c["matrixfree"]["dX"], 1e-5, "matrixfree")
    eig = hip.ShiftInvert(0.1, ls_, tol=1e-10, maxiter=40, hermitian=True, save_vectors=False, krylovdim=36)
    vals, _, cv, _ = eig(J, 6)
    assert np.allclose(np.sort(vals.real)[::-1][:len(c["shift_invert"]["vals"])], c["shift_invert"]["vals"], rtol=0, atol=1e-7)
    nopt = Cn.NewtonPar(tol=1e-9, max_iterations=15, linsolver=ls_)
    cp = Cn.ContinuationPar(ds=-0.001, dsmin=1e-4, dsmax=0.005, p_min=-0.1, p_max=0.15,
                            max_steps=len(c["branch"]["param"]) - 1, detect_bifurcation=0, newton_options=nopt)
    alg = Cn.PALC(tangent="bordered", theta=0.5, bls=hip.BorderingBLS(None, check_precision=False))
    br = Cn.continuation_native(prob, sn["u"], l, alg, cp, normC=Cn.norminf)
    assert np.allclose(br.param, c["branch"]["param"], rtol=0, atol=1e-8)
    assert all(abs(a - b) <= 1 for a, b in zip(br.itnewton, c["branch"]["itnewton"]))
    if "branch_matrixfree" in c:                      # the corrector with MatrixFreeBLS: bk_bordering_opts.kind = 1, one call per corrector
        bm = c["branch_matrixfree"]
        mfc = hip.MatrixFreeBLS(hip.GMRESKrylovKit(dim=60, rtol=1e-10, atol=1e-13, maxiter=300))
        cpm = Cn.ContinuationPar(ds=-0.001, dsmin=1e-4, dsmax=0.005, p_min=-0.1, p_max=0.15, max_steps=len(bm["param"]) - 1,
                                 detect_bifurcation=0, newton_options=Cn.NewtonPar(tol=1e-9, max_iterations=15, linsolver=ls_))
        brm = Cn.continuation(prob, sn["u"], l, Cn.PALC(tangent="bordered", theta=0.5, bls=mfc), cpm, normC=Cn.norminf)
        assert np.allclose(brm.param, bm["param"], rtol=0, atol=1e-8)
        assert all(abs(a - b) <= 1 for a, b in zip(brm.itnewton, bm["itnewton"]))


@pytest.mark.gpu
def test_hip_cgl_matches_reference(fx, ctx):
    from bk_amd import continuation as Cn
    from bk_amd import hip
    c = fx["cgl_41x21"]
    dims, ls = tuple(c["dims"]), tuple(c["ls"])
    n2 = 2 * dims[0] * dims[1]
    prob = hip.CGL2d(ctx, dims, ls, r=0.5)
    u, du = prob.vec(0.4 * probe(5, n2)), prob.vec(probe(6, n2))
    close(prob.residual(u, 1.2).numpy(), c["F_probe"], 1e-10, "cgl F")
    close(prob.jacobian(u, 1.2)(du).numpy(), c["J_probe_du"], 1e-10, "cgl J")
    P = hip.LaplacePreconditioner(prob, 1.0)
    ls_ = hip.GMRESIterativeSolvers(reltol=1e-11, restart=60, maxiter=600, Pl=P)
    eig = hip.ShiftInvert(1.0, ls_, tol=1e-9, maxiter=60, hermitian=False, save_vectors=False)
    zero = prob.vec(np.zeros(n2))
    vals, _, cv, _ = eig(prob.jacobian(zero, 0.5), 9)
    ref = np.array([complex(a, b) for a, b in c["eig_trivial_r0.5"]["vals"]])
    for v in vals[~np.isnan(vals.real)]:
        assert np.abs(ref - v).min() <= 1e-6 or v.real < ref.real.min() + 1e-6
    # the reference's branch: same Hopf points (type, parameter interval) from the native continuation with bisection
    rb = c["branch"]
    nopt = Cn.NewtonPar(tol=1e-9, max_iterations=20, linsolver=ls_, eigsolver=eig)
    cp = Cn.ContinuationPar(ds=0.001, dsmin=0.001, dsmax=0.15, p_min=-1.0, p_max=2.5, max_steps=len(rb["param"]) - 1, nev=9,
                            newton_options=nopt, n_inversion=6, detect_bifurcation=3)
    alg = Cn.PALC(tangent="secant", theta=0.5, bls=hip.BorderingBLS(None, check_precision=False))
    bn = Cn.continuation_native(prob, zero, 0.5, alg, cp, normC=Cn.norminf, bisection=True)
    hopf_ref = [sp for sp in rb["specialpoint"] if sp["type"] == "hopf"]
    hopf_mine = [sp for sp in bn.specialpoint if sp.get("type") == "hopf"]
    assert len(hopf_mine) >= min(len(hopf_ref), 1)
    for a, b in zip(hopf_mine, hopf_ref):
        assert abs(a["param"] - b["param"]) <= 1e-3
