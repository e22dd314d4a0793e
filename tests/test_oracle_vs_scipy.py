"""Second, independent anchor of the oracle's Krylov loops (oracle/krylov.py restates KrylovKit / IterativeSolvers / Krylov.jl
"from package knowledge"): SciPy ships its own implementations of the same published algorithms --
restarted GMRES with modified Gram-Schmidt + Givens (Saad & Schultz), Paige-Saunders MINRES, Hestenes-Stiefel CG -- and wraps
the same Fortran ARPACK that `Arpack.jl` (`EigArpack`, src/EigSolver.jl:67-102) wraps.  On the PDE operators of the hot path
the iterates of two correct implementations of one algorithm coincide: residual histories to rounding, iteration counts
exactly (+-1 where a stopping test sits on its threshold), ARPACK's eigenvalues to its tolerance.  This does not pin the Julia
packages themselves (no Julia here: parity stays "unpinned"); it pins the restatements to the published
algorithms those packages implement."""
import numpy as np
import pytest
import scipy.sparse as sp
import scipy.sparse.linalg as spla

from oracle import krylov, operators


def _sh2d(dims=(24, 18), ls=(2 * np.pi, 2 * np.pi / np.sqrt(3)), seed=7):
    sh = operators.SwiftHohenberg(dims, ls)
    rng = np.random.default_rng(seed)
    u = sh.guess() + 0.05 * rng.standard_normal(sh.N)
    J = sh.J(u, -0.1, 1.3)                                   # assembled sparse Jacobian (symmetric)
    Pl = operators.dct_preconditioner(dims, ls, 1.0)
    return sh, J, Pl, rng


def _scipy_gmres_history(A, b, restart, rtol, maxiter, M=None):
    """SciPy's GMRES with the preconditioned-residual callback: ||M r_k|| / ||b|| after every inner iteration."""
    hist = []
    x, info = spla.gmres(A, b, rtol=rtol, atol=0.0, restart=restart, maxiter=maxiter, M=M, callback=hist.append,
                         callback_type="pr_norm")
    return x, info, np.array(hist)


def test_gmres_iterates_match_scipy_on_the_preconditioned_sh_jacobian():
    """GMRES residual history of the oracle (KrylovKit restatement: Arnoldi on Pl^-1 J with MGS2, Givens) against
    scipy.sparse.linalg.gmres with the same left preconditioner: the minimal-residual iterates of one Krylov space are unique,
    so the preconditioned residual norms agree iteration by iteration (SciPy stops on its own, unpreconditioned criterion, so
    the histories are compared, not the counts)."""
    sh, J, Pl, rng = _sh2d()
    n = sh.N
    b = rng.standard_normal(n)
    M = spla.LinearOperator((n, n), matvec=Pl)
    A = spla.LinearOperator((n, n), matvec=lambda v: J @ v)
    xs, info, hist = _scipy_gmres_history(A, b, restart=n, rtol=1e-10, maxiter=200, M=M)
    assert info == 0 and len(hist) > 10
    ours = []
    xk, okk, numops, _ = krylov.gmres_krylovkit(J, b, krylovdim=200, maxiter=5, rtol=1e-13, atol=0.0, Pl=Pl, history=ours)
    assert okk
    ours = np.array(ours) / np.linalg.norm(b)              # SciPy reports ||M r_k|| / ||b|| after inner iteration k = 1, 2, ...
    m = min(len(hist), len(ours))
    rel = np.abs(ours[:m] - hist[:m]) / hist[:m]
    assert rel[hist[:m] > 1e-9].max() <= 1e-6, rel             # identical iterates down to where rounding takes over
    assert np.abs(xk - xs).max() <= 1e-8 * np.abs(xs).max()
    # the IterativeSolvers and Krylov.jl restatements walk the same iterates: they stop at the first iteration whose
    # preconditioned residual is below their tolerance -- read off SciPy's history
    scale = np.linalg.norm(b) / np.linalg.norm(Pl(b))       # SciPy's history relative to ||M b||, the solvers' reference norm
    first = lambda tol: int(np.argmax(hist * scale <= tol)) + 1
    xo, ok, it = krylov.gmres_iterativesolvers(J, b, reltol=1e-8, restart=200, maxiter=200, Pl=Pl)
    assert ok and it == first(1e-8), (it, first(1e-8))
    xj, okj, itj = krylov.gmres_krylovjl(J, b, memory=200, rtol=1e-8, atol=0.0, M=Pl)
    assert okj and itj == first(1e-8), (itj, first(1e-8))
    assert np.abs(xo - xs).max() <= 1e-6 * np.abs(xs).max() and np.abs(xj - xs).max() <= 1e-6 * np.abs(xs).max()
    # KrylovKit's count adds its two bookkeeping applications (A * x0 and the explicit final residual) to the iterations
    h2 = []
    _, ok2, numops2, _ = krylov.gmres_krylovkit(J, b, krylovdim=200, maxiter=5, rtol=1e-8, atol=0.0, Pl=Pl, history=h2)
    assert ok2 and numops2 == first(1e-8) + 2, (numops2, first(1e-8))


def test_restarted_gmres_matches_scipy_cycle_by_cycle():
    """GMRES(m) with restarts on the shifted operator a0 I + a1 J (the Hopf / shift-invert call shape): the restart
    structure -- m inner steps, explicit residual, new cycle -- gives the same total iteration count as SciPy's."""
    sh, J, Pl, rng = _sh2d(seed=11)
    n = sh.N
    a0, a1 = -0.6, 1.0
    S = (a0 * sp.identity(n) + a1 * J).tocsr()
    b = rng.standard_normal(n)
    M = spla.LinearOperator((n, n), matvec=Pl)
    xs, info, hist = _scipy_gmres_history(S, b, restart=8, rtol=1e-9, maxiter=400, M=M)
    assert info == 0
    xo, ok, it = krylov.gmres_iterativesolvers(J, b, a0, a1, reltol=1e-9, restart=8, maxiter=3000, Pl=Pl)
    assert ok
    assert abs(it - len(hist)) <= 2, (it, len(hist))
    assert np.abs(xo - xs).max() <= 1e-7 * np.abs(xs).max()


def test_minres_and_cg_iterates_match_scipy():
    """Krylov.jl-flavoured MINRES / CG of the oracle against SciPy's (both ports of the same Paige-Saunders / Hestenes-Stiefel
    recurrences) on the symmetric SH Jacobian with the SPD spectral preconditioner."""
    sh, J, Pl, rng = _sh2d(seed=3)
    n = sh.N
    b = rng.standard_normal(n)
    M = spla.LinearOperator((n, n), matvec=Pl)
    S = (J - 0.4 * sp.identity(n)).tocsr()                   # J - sigma I, sigma right of the spectrum: negative definite
    its = []
    xs, info = spla.minres(S, b, rtol=1e-10, M=M, maxiter=4 * n, callback=lambda xk: its.append(1))
    assert info == 0
    xo, ok, it = krylov.minres_krylovjl(J, b, -0.4, 1.0, atol=0.0, rtol=1e-10, itmax=4 * n, M=Pl)
    assert ok and np.abs(xo - xs).max() <= 1e-7 * np.abs(xs).max()
    assert abs(it - len(its)) <= 2, (it, len(its))           # the two stop on differently scaled estimates of the same residual
    itc = []
    xs2, info = spla.cg(-S, -b, rtol=1e-10, atol=0.0, M=M, maxiter=4 * n, callback=lambda xk: itc.append(1))
    assert info == 0
    xc, okc, itco = krylov.cg_krylovjl(lambda v: -(S @ v), -b, atol=0.0, rtol=1e-10, itmax=4 * n, M=Pl)
    assert okc and np.abs(xc - xs2).max() <= 1e-7 * np.abs(xs2).max()
    assert abs(itco - len(itc)) <= 2, (itco, len(itc))


def test_shift_invert_krylovschur_matches_arpack_on_the_sh_jacobian():
    """`EigArpack(sigma, :LM)` (src/EigSolver.jl:85-102) is ARPACK's shift-invert mode; SciPy drives the same Fortran.  The
    oracle's ShiftInvert + Krylov-Schur (src/EigSolver.jl:246-266, examples/SH3d.jl:96-113) must return the same rightmost
    eigenvalues of the SH Jacobian."""
    sh, J, Pl, rng = _sh2d(dims=(20, 16), seed=5)
    n = sh.N
    sigma, nev = 0.4, 8
    ref = spla.eigsh(J.tocsc(), k=nev, sigma=sigma, which="LM", tol=1e-12, return_eigenvectors=False)
    ref = np.sort(ref)[::-1]
    # the shift is folded into the operator BEFORE the preconditioned solve, as SH3dEig does (examples/SH3d.jl:106-107):
    # GMRESKrylovKit with Pl applies a0 after the preconditioner (src/LinearSolver.jl:268-277)
    ls = lambda A, r, a0=0.0, a1=1.0: krylov.gmres_krylovkit(lambda v: a0 * v + a1 * (A @ v), r, krylovdim=60, maxiter=20,
                                                              rtol=1e-12, atol=1e-14, Pl=Pl)[:3]
    eig = lambda Amap, k: krylov.eigsolve_krylovschur(Amap, rng.random(n), k, "LM", tol=1e-10, krylovdim=40, maxiter=50,
                                                      hermitian=True)
    vals, vecs, cv, _ = krylov.shift_invert(J, nev, sigma, ls, eig)
    got = np.sort(np.real(vals[:nev]))[::-1]
    assert np.abs(got - ref).max() <= 1e-8 * max(1.0, np.abs(ref).max()), (got, ref)


def test_shift_invert_matches_arpack_on_the_cgl_jacobian_complex_pairs():
    """Non-symmetric case: the cGL Jacobian of the trivial state has the complex pairs r + lam_Lap +- i nu
    (examples/cGL2d.jl:96-100: EigArpack(1.0, :LM), nev = 9); ARPACK and the oracle's Krylov-Schur agree on them."""
    dims, ls_ = (12, 9), (np.pi, np.pi / 2)
    c = operators.CGL2d(dims, ls_)
    p = c.default_params()
    p["r"] = 0.5
    Jm = c.J(np.zeros((2 * c.n)), **p).tocsc()
    sigma, nev = 1.0, 6
    rng = np.random.default_rng(2)
    ref = spla.eigs(Jm, k=nev, sigma=sigma, which="LM", tol=1e-12, return_eigenvectors=False, v0=rng.random((2 * c.n)))
    lu = spla.splu((Jm - sigma * sp.identity((2 * c.n))).tocsc())
    ls = lambda A, r, a0=0.0, a1=1.0: (lu.solve(r), True, 1)            # DefaultLS on J - sigma I (the reference's sparse LU)
    eig = lambda Amap, k: krylov.eigsolve_krylovschur(Amap, rng.random((2 * c.n)), k, "LM", tol=1e-11, krylovdim=30, maxiter=100)
    vals, vecs, cv, _ = krylov.shift_invert(Jm, nev, sigma, ls, eig)
    key = lambda z: (-round(z.real, 9), -round(z.imag, 9))
    got = np.array(sorted(vals[:nev], key=key))
    want = np.array(sorted(ref, key=key))
    assert np.abs(got - want).max() <= 1e-8, (got, want)
    assert np.abs(np.abs(got.imag) - p["nu"]).max() <= 1e-8               # the +- i nu pairs of the closed form
