"""Writes a file with the SCHEMA of tests/golden/julia_fixtures.json (julia/gen_fixtures.jl) from the NumPy oracle -- NOT
reference data: it pins nothing (the consumer then compares the oracle with itself).  Its only purpose is to keep the
consumer tests/test_reference_fixtures.py executable end to end until the real file exists (tests/test_fixture_plumbing.py
runs the CPU consumers against it), so that the day someone runs the Julia script the plumbing is known to work.  The file
carries "generator": "oracle" and is never written into tests/golden/.
Usage: python tests/golden/make_schema_fixture.py OUT.json"""
import json
import os
import sys

import numpy as np
import scipy.sparse.linalg as spla

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import bordered, krylov, operators, palc  # noqa: E402
from test_reference_fixtures import probe, sh_guess, summary  # noqa: E402


def sh_case(dims, ls, l, nu, branch_steps):
    sh = operators.SwiftHohenberg(dims, ls)
    N = sh.N
    u0 = sh_guess(dims, ls)
    c = dict(dims=list(dims), ls=list(ls), l=l, nu=nu)
    c["F_u0"] = summary(sh.F(u0, l, nu))
    c["dF_u0_probe1"] = summary(sh.dF(u0, l, nu, probe(1, N)))
    lu = spla.splu(sh.L1.tocsc())
    ols = lambda J, r, a0=0.0, a1=1.0: krylov.gmres_krylovkit(J, r, a0, a1, krylovdim=30, maxiter=150, rtol=1e-9, atol=1e-12,
                                                               Pl=lu.solve)[:3]
    oprob = palc.Problem(lambda x, p: sh.F(x, p, nu), lambda x, p: (lambda dx: sh.dF(x, p, nu, dx)))
    so = palc.newton(oprob, u0, l, ols, tol=1e-8, max_iterations=20, normN=palc.norminf)
    c["newton"] = dict(converged=bool(so["converged"]), residuals=[float(x) for x in so["residuals"]], itnewton=int(so["itnewton"]),
                       itlineartot=int(so["itlineartot"]), u=summary(so["u"]))
    us = so["u"]
    J = lambda dx: sh.dF(us, l, nu, dx)
    r1, r2, r3 = probe(1, N), probe(2, N), probe(3, N)
    x, ok, it = ols(J, r1)
    c["gmres"] = dict(converged=bool(ok), numops=int(it), x=summary(x))
    x, ok, it = ols(J, r1, 0.3, 0.9)
    c["gmres_shift"] = dict(a0=0.3, a1=0.9, converged=bool(ok), numops=int(it), x=summary(x))
    x, ok, it = krylov.minres_krylovjl(J, r1, -0.1, 1.0, atol=1e-13, rtol=1e-10, M=lu.solve)
    c["minres"] = dict(a0=-0.1, a1=1.0, converged=bool(ok), niter=int(it), x=summary(x))
    x, ok, it = krylov.cg_krylovjl(J, r1, 2.0, -1.0, atol=1e-13, rtol=1e-10, M=lu.solve)
    c["cg"] = dict(a0=2.0, a1=-1.0, converged=bool(ok), niter=int(it), x=summary(x))
    dotp = lambda a, b: float(a @ b) / N
    dX, dl, ok, its = bordered.bordering_bls(ols, J, r2, r3, 0.4, r1, 0.3, 0.5, 0.5, check_precision=False, dotp=dotp)
    c["bordering"] = dict(converged=bool(ok), itlinear=[int(i) for i in its], dl=float(dl), dX=summary(dX))
    # the two entries only the GPU consumers read are filled from direct solves (cheap; this file pins nothing anyway)
    import scipy.sparse as sp
    Jm = sh.J(us, l, nu).tocsc()
    A = sp.bmat([[Jm, sp.csc_matrix(r2[:, None])], [sp.csc_matrix(0.5 * r3[None, :] / N), sp.csc_matrix([[0.5 * 0.4]])]]).tocsc()
    sol = spla.spsolve(A, np.concatenate([r1, [0.3]]))
    c["matrixfree"] = dict(converged=True, itlinear=0, dl=float(sol[-1]), dX=summary(sol[:-1]))
    lu1, lu3 = spla.splu((sh.L1 + sp.identity(N)).tocsc()), spla.splu((sh.L1 + 1e5 * sp.identity(N)).tocsc())
    x, ok, it = krylov.gmres_iterativesolvers(J, r1, 0.4, -1.0, restart=63, maxiter=4000, reltol=1e-10, Pl=lu1.solve, Pr=lu3.solve)
    c["gmres_is_pr"] = dict(a0=0.4, a1=-1.0, reltol=1e-10, restart=63, pl_shift=1.0, pr_shift=1e5, converged=bool(ok), niter=int(it),
                            x=summary(x))
    sigma = 0.1
    near = spla.eigsh(Jm, k=6, sigma=sigma, which="LM", tol=1e-12, return_eigenvectors=False)
    c["shift_invert"] = dict(sigma=sigma, converged=6, numops=0, vals=[float(v) for v in np.sort(near)[::-1]])
    obls = lambda *a, **k: bordered.bordering_bls(ols, *a, check_precision=False, **k)
    br = palc.continuation(oprob, us, l, ls=ols, bls=obls, tangent="bordered", normC=palc.norminf, ds=-0.001, dsmin=1e-4,
                           dsmax=0.005, p_min=-0.1, p_max=0.15, max_steps=branch_steps, tol=1e-9, max_iterations=15)
    c["branch"] = dict(param=[float(p) for p in br.param], itnewton=[int(i) for i in br.itnewton],
                       itlinear=[int(i) for i in br.itlinear], ds=[float(d) for d in br.ds])
    # (the MatrixFreeBLS branch is the same curve: only the GPU consumer reads it, against the native corrector)
    c["branch_matrixfree"] = dict(param=c["branch"]["param"][:3], itnewton=c["branch"]["itnewton"][:3], itlinear=[0, 0, 0])
    return c


def cgl_case():
    dims, ls = (41, 21), (np.pi, np.pi / 2)
    cg = operators.CGL2d(dims, ls)
    n2 = 2 * cg.n
    pars = dict(cg.default_params(), r=1.2)
    u, du = 0.4 * probe(5, n2), probe(6, n2)
    c = dict(dims=list(dims), ls=list(ls))
    c["F_probe"] = summary(cg.F(u, **pars))
    c["J_probe_du"] = summary(cg.J(u, **pars) @ du)
    dense = np.linalg.eigvals(cg.J(np.zeros(n2), **cg.default_params()).toarray())
    near = sorted(dense, key=lambda z: abs(z - 1.0))[:9]
    near = sorted(near, key=lambda z: (-z.real, -z.imag))
    c["eig_trivial_r0.5"] = dict(converged=True, vals=[[float(z.real), float(z.imag)] for z in near])
    c["branch"] = dict(param=[0.5], n_unstable=[0], n_imag=[0], specialpoint=[])
    return c


def generate(path):
    out = dict(generator="oracle", versions=dict(note="oracle-generated schema exercise, NOT reference output"))
    out["sh3d_22"] = sh_case((22, 22, 22), (np.pi,) * 3, 0.1, 1.2, 4)
    out["sh2d_151x100"] = sh_case((151, 100), (8 * np.pi, 4 * np.pi / np.sqrt(3)), -0.1, 1.3, 3)
    # (round 6: a grid on which the library's default Arnoldi step is the stencil-free one; the 64^3 case of julia/gen_fixtures.jl needs a
    # sparse factorisation of a 262 144-unknown 25-point operator and is left to the real generator -- its consumers skip)
    out["sh2d_128x64"] = sh_case((128, 64), (12.5, 6.0), -0.1, 1.3, 2)
    out["cgl_41x21"] = cgl_case()
    json.dump(out, open(path, "w"), indent=1)


if __name__ == "__main__":
    generate(sys.argv[1])
