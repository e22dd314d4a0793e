"""ORACLE (test infrastructure) -- bordered linear solvers of src/LinearBorderSolver.jl on flat
NumPy vectors.

  bordering_bls    BorderingBLS call + BEC + residualBEC      src/LinearBorderSolver.jl:88-166
  matrixfree_bls   MatrixFreeBLSmap (flat ``vcat`` path) + MatrixFreeBLS   :308-322, :424-437
  matrix_bls       MatrixBLS: explicit (N+1) matrix + dense ``\\``            :231-264
  solve_bls_palc   the PALC adapter (xi_u = theta, xi_p = 1-theta, dotp = dot/N)   :16-36

Linear solvers are callables ``ls(J, rhs, a0=0.0, a1=1.0) -> (x, ok, it)``; the two-RHS default
(src/LinearSolver.jl:15-19) is ``solve2``.
"""
from __future__ import annotations

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla

from .krylov import apply


def default_ls(J, rhs, a0=0.0, a1=1.0):
    """DefaultLS, src/LinearSolver.jl:101-103: ``(a0 I + a1 J) \\ rhs`` -> (x, true, 1)."""
    n = rhs.shape[0]
    if sp.issparse(J):
        M = (a0 * sp.identity(n, format="csc") + a1 * J.tocsc()).tocsc()
        return spla.spsolve(M, rhs), True, 1
    return np.linalg.solve(a0 * np.eye(n) + a1 * np.asarray(J), rhs), True, 1


def solve2(ls, J, rhs1, rhs2, **kw):
    """src/LinearSolver.jl:15-19: two sequential solves, AND the flags, tuple of iterations."""
    x1, f1, it1 = ls(J, rhs1, **kw)
    x2, f2, it2 = ls(J, rhs2, **kw)
    return x1, x2, bool(f1 and f2), (it1, it2)


def bec(ls, J, dR, dzu, dzp, R, n, xiu, xip, shift, dotp):
    """BEC, src/LinearBorderSolver.jl:125-144."""
    if shift is None:
        x1, dx, ok, it = solve2(ls, J, R, dR)
    else:
        x1, dx, ok, it = solve2(ls, J, R, dR, a0=shift)
    dl = (n - dotp(dzu, x1) * xiu) / (dzp * xip - dotp(dzu, dx) * xiu)
    x1 = x1 - dl * dx
    return x1, dl, ok, it


def residual_bec(J, dR, dzu, dzp, R, n, dX, dl, xiu, xip, shift, dotp):
    """residualBEC, src/LinearBorderSolver.jl:146-166."""
    dXr = apply(J, dX)
    if shift is not None:
        dXr = dXr + shift * dX
    dXr = dXr + dl * dR
    dXr = R - dXr
    dlr = n - xip * dzp * dl - xiu * dotp(dzu, dX)
    return dXr, dlr


def bordering_bls(ls, J, dR, dzu, dzp, R, n, xiu=1.0, xip=1.0, *, shift=None, dotp=np.dot,
                  tol=1e-12, check_precision=True, k=1):
    """(lbs::BorderingBLS)(...), src/LinearBorderSolver.jl:88-123.  Returns (dX, dl, cv, itlinear)."""
    dX, dl, cv, itl = bec(ls, J, dR, dzu, dzp, R, n, xiu, xip, shift, dotp)
    kk = 0
    fail = True
    while check_precision and kk < k and fail:
        dXr, dlr = residual_bec(J, dR, dzu, dzp, R, n, dX, dl, xiu, xip, shift, dotp)
        fail = np.linalg.norm(dXr) > tol or abs(dlr) > tol
        if fail:
            dX1, dl1, cv, itl = bec(ls, J, dR, dzu, dzp, dXr, dlr, xiu, xip, shift, dotp)
            dX = dX + dX1
            dl = dl + dl1
            kk += 1
    return dX, dl, cv, itl


def bordering_bls_block(ls, J, b, c, d, rhst, rhsb):
    """solve_bls_block(::BorderingBLS, J, b::NTuple, c::NTuple, d::Matrix, rhst, rhsb),
    src/LinearBorderSolver.jl:173-206 -> (u1, u2, cv, its).  The flag of the J x1 = rhst solve is overwritten by
    `cv = true` (:186-189), exactly as written there."""
    d = np.atleast_2d(np.asarray(d, dtype=float))
    m = d.shape[0]
    if not (len(b) == len(c) == m):
        raise ValueError("Linear bordered solver, wrong sizes!")
    x1, cv, it = ls(J, rhst)
    x2s, its, cv = [], [], True
    for bi in b:
        x2, flag, it = ls(J, bi)
        x2s.append(x2); its.append(it)
        cv = cv and flag
    S = np.array([[d[i, j] - np.dot(c[i], x2s[j]) for j in range(m)] for i in range(m)])
    h = np.array([rhsb[i] - np.dot(c[i], x1) for i in range(m)])
    u2 = np.linalg.solve(S, h)
    u1 = x1.copy()
    for i in range(m):
        u1 -= u2[i] * x2s[i]
    return u1, u2, cv, tuple(its)


def matrixfree_bls_block(ls, J, a, b, c, rhst, rhsb, *, shift=None, dotp=np.dot):
    """solve_bls_block(::MatrixFreeBLS, J, a, b, c, rhst, rhsb; shift, dotp), src/LinearBorderSolver.jl:440-450, with the
    m-column MatrixFreeBLSmap of :338-352 on the flat vector [u; p]: one linear solve on the (N + m) operator."""
    c = np.atleast_2d(np.asarray(c, dtype=float))
    m, n = c.shape[0], rhst.shape[0]

    def op(x):
        xu, xp = x[:n], x[n:]
        out = np.empty_like(x)
        out[:n] = apply(J, xu) + sum(xp[i] * a[i] for i in range(m))
        if shift is not None:
            out[:n] += shift * xu
        out[n:] = c @ xp + np.array([dotp(b[i], xu) for i in range(m)])
        return out

    sol, cv, it = ls(op, np.concatenate([rhst, np.asarray(rhsb, dtype=float)]))
    return sol[:n], sol[n:], cv, it


def matrixfree_blsmap(J, a, b, c, shift, dot):
    """MatrixFreeBLSmap on a flat vector, src/LinearBorderSolver.jl:308-322."""
    def op(x):
        xu, xp = x[:-1], x[-1]
        out = np.empty_like(x)
        out[:-1] = apply(J, xu) + xp * a
        if shift is not None:
            out[:-1] += shift * xu
        out[-1] = dot(b, xu) + c * xp
        return out
    return op


def matrixfree_bls(ls, J, dR, dzu, dzp, R, n, xiu=1.0, xip=1.0, *, shift=None, dotp=np.dot):
    """(lbs::MatrixFreeBLS)(...), src/LinearBorderSolver.jl:424-437 (``vcat`` path)."""
    op = matrixfree_blsmap(J, dR, xiu * dzu, dzp * xip, shift, dotp)
    rhs = np.concatenate([R, [n]])
    sol, cv, it = ls(op, rhs)
    return sol[:-1].copy(), float(sol[-1]), cv, it


def matrix_bls(J, dR, dzu, dzp, R, n, xiu=1.0, xip=1.0, *, shift=None, apply_xiu=None):
    """MatrixBLS, src/LinearBorderSolver.jl:231-264: assemble the (N+1)x(N+1) matrix, dense/sparse
    ``\\``.  In the PALC call the last row is ``xiu * applyxiu(dzu)`` with applyxiu = scale by 1/N."""
    N = R.shape[0]
    Jm = sp.csc_matrix(J) if not sp.issparse(J) else J.tocsc()
    if shift is not None:
        Jm = Jm + shift * sp.identity(N, format="csc")
    row = dzu.copy() if apply_xiu is None else apply_xiu(dzu.copy())
    A = sp.bmat([[Jm, sp.csc_matrix(dR.reshape(-1, 1))],
                 [sp.csc_matrix((xiu * row).reshape(1, -1)), sp.csc_matrix([[xip * dzp]])]], format="csc")
    sol = spla.spsolve(A, np.concatenate([R, [n]]))
    return sol[:-1], float(sol[-1]), True, 1


def solve_bls_palc(bls, theta, tau_u, tau_p, J, dR, R, n, *, shift=None):
    """solve_bls_palc, src/LinearBorderSolver.jl:16-36: xi_u = theta, xi_p = 1 - theta,
    ``dotp = getdot(iter).dot`` = NormalisedDot = dot/length (src/continuation/Palc.jl:1-6,41)."""
    N = R.shape[0]
    dotp = lambda x, y: float(np.dot(x, y)) / N
    return bls(J, dR, tau_u, tau_p, R, n, theta, 1.0 - theta, shift=shift, dotp=dotp)
