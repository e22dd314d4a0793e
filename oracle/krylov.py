"""ORACLE (test infrastructure) -- host Krylov loops the reference delegates to third-party
Julia packages, restated from their published algorithms.

The packages are NOT under /root/reference (Project.toml:53,56 compat ranges only, no
Manifest) and Julia is not installed here, so these restatements are "parity unpinned"
against real package output; they are anchored on the reference's call sites
(src/LinearSolver.jl:198,256-291; src/EigSolver.jl:157-160,257-266; examples/SH3d.jl:96-113)
and on its test identities (test/linear_solvers/test_linear.jl:106-169, :666-677).

  gmres_krylovkit      KrylovKit.linsolve (GMRES) ^0.7-0.10: restarted GMRES(krylovdim) on
                       (a0 + a1 A), Krylov space built on A itself, shift applied to the
                       Hessenberg; ModifiedGramSchmidt2 (KrylovDefaults.orth); maxiter = number
                       of restart cycles; stop on |y[k+1]| <= max(atol, rtol*||b||) then verify
                       with an explicit residual; returns numops (operator applications), which
                       BK reports as "iterations" (src/LinearSolver.jl:291).
  gmres_iterativesolvers  IterativeSolvers.gmres 0.8.4-0.9: restarted GMRES, single-pass MGS,
                       progressive Givens, maxiter = total inner iterations, stop on
                       ||r|| <= max(reltol*||r0||, abstol)  (src/LinearSolver.jl:186-206).
  eigsolve_krylovschur KrylovKit.eigsolve: Arnoldi/Lanczos + Krylov-Schur thick restart
                       (src/EigSolver.jl:149-160, examples/SH3d.jl:109).
  shift_invert         src/EigSolver.jl:257-266.
"""
from __future__ import annotations

import numpy as np
import scipy.linalg as sla


def apply(J, x):
    """src/Utils.jl:191-195: ``A*x`` for matrices, ``f(x)`` for anything callable."""
    if callable(J):
        return J(x)
    return J @ x


def axpy_op(J, v, a0, a1):
    """_axpy_op, src/LinearSolver.jl:46-64: (a0 I + a1 J) v."""
    return a0 * v + a1 * apply(J, v)


def _mgs2(w, V, k):
    """ModifiedGramSchmidt2: one MGS sweep then one re-orthogonalisation sweep against V[0..k-1]."""
    h = np.zeros(k)
    for i in range(k):
        h[i] = V[i] @ w
        w = w - h[i] * V[i]
    for i in range(k):
        s = V[i] @ w
        w = w - s * V[i]
        h[i] += s
    return w, h


class GramCGS:
    """The library's Arnoldi orthogonalisation since round 3, restated (csrc/solver.hip: arnoldi_step, gram branch;
    csrc/vecops.hip: multidot_c_kernel<GRAM>): ONE classical Gram-Schmidt pass whose projection coefficients are corrected
    with the MEASURED Gram matrix of the basis.  The pass that computes a = V'w also returns g = V'v_{k-1}, the Gram column
    of the newest vector; with G known, c = G^-1 a (c = a - E a + E E a, E = G - I) are the coefficients of the orthogonal
    projection of w onto span(V), so w - V c is orthogonal to every v_i up to the rounding of this one pass whatever defect
    the earlier vectors carry, and beta^2 = w'w - c'a.  Not a reference algorithm (KrylovKit: ModifiedGramSchmidt2, _mgs2
    above): the tests show that it reproduces MGS2's Hessenberg / residual history / iteration counts."""

    def __init__(self, m):
        self.G = np.zeros((m + 1, m + 1))
        self.n = 0

    def reset(self):
        self.n = 0

    def __call__(self, w, V, k):
        assert self.n == k - 1, "one call per Arnoldi step, in order"
        a = V[:k] @ w
        g = V[:k] @ V[k - 1]
        self.G[:k, k - 1] = g
        self.G[k - 1, :k] = g
        self.n = k
        E = self.G[:k, :k] - np.eye(k)
        e1 = E @ a
        c = a - e1 + E @ e1
        b2 = float(w @ w) - float(c @ a)
        wn = w - V[:k].T @ c
        # the library normalises with the Pythagorean beta; hand the remainder back scaled so that ||.|| == sqrt(b2)
        nrm = np.linalg.norm(wn)
        if b2 > 1e-8 * float(w @ w) and nrm > 0.0:
            wn = wn * (np.sqrt(b2) / nrm)
        return wn, c


def _cgs1(w, V, k):
    """ONE classical Gram-Schmidt pass without any correction (control experiment of the tests: loses orthogonality)."""
    h = V[:k] @ w
    return w - V[:k].T @ h, h


def _givens(f, g):
    """Real Givens (c, s, r) with [c s; -s c] [f; g] = [r; 0]."""
    if g == 0.0:
        return 1.0, 0.0, f
    if f == 0.0:
        return 0.0, 1.0, g
    r = np.hypot(f, g)
    return f / r, g / r, r


def gmres_krylovkit(A, b, a0=0.0, a1=1.0, *, krylovdim=30, maxiter=100, atol=1e-12, rtol=1e-12,
                    Pl=None, history=None, orth="mgs2", basis_out=None):
    """Solve (a0 + a1 A) x = b from x0 = 0.  Returns (x, converged, numops, normres).

    ``Pl`` (callable applying Pl^-1) selects the preconditioned branch of GMRESKrylovKit,
    src/LinearSolver.jl:268-288: the operator becomes ``dx -> a0 dx + a1 Pl^-1 (A dx)`` (shift applied
    AFTER the preconditioner -- the reference's quirk, kept) and the right-hand side ``Pl^-1 b``;
    tolerances then act on the preconditioned residual."""
    if Pl is not None:
        A_, a0_, a1_ = A, a0, a1
        lin = lambda dx: a1_ * Pl(apply(A_, dx)) + a0_ * dx
        return gmres_krylovkit(lin, Pl(np.asarray(b, dtype=float)), 0.0, 1.0, krylovdim=krylovdim,
                               maxiter=maxiter, atol=atol, rtol=rtol, history=history, orth=orth, basis_out=basis_out)
    b = np.asarray(b, dtype=float)
    n = b.shape[0]
    x = np.zeros(n)
    r = b.copy()                       # y0 = A*0 = 0 (the package still applies A once)
    numops = 1
    beta = np.linalg.norm(r)
    tol = max(atol, rtol * np.linalg.norm(b))
    if beta < tol:
        return x, True, numops, beta
    m = krylovdim
    V = np.zeros((m + 1, n))
    R = np.zeros((m, m))
    y = np.zeros(m + 1)
    cs = np.zeros(m)
    sn = np.zeros(m)
    numiter = 0
    # orth: "mgs2" = KrylovKit's ModifiedGramSchmidt2 (the reference); "cgs_gram" = the library's Gram-corrected single pass;
    # "cgs1" = one uncorrected classical pass (control)
    gram = GramCGS(m) if orth == "cgs_gram" else None
    ortho = {"mgs2": _mgs2, "cgs1": _cgs1}.get(orth, gram)

    def start(r, beta):
        # ArnoldiIterator initialize: v1 = r/||r||, w = A v1, h11, residual
        V[0] = r / beta
        if gram is not None:
            gram.reset()
        w = apply(A, V[0])
        w, h = ortho(w, V, 1)
        return w, h, np.linalg.norm(w)

    w, h, nrm = start(r, beta)
    numops += 1
    while numiter < maxiter:
        numiter += 1
        y[:] = 0.0
        y[0] = beta
        k = 1
        R[0, 0] = a0 + a1 * h[0]
        cs[0], sn[0], R[0, 0] = _givens(R[0, 0], a1 * nrm)
        y[1] = -sn[0] * y[0]
        y[0] = cs[0] * y[0]
        beta = abs(y[1])
        if history is not None:
            history.append(beta)
        while beta > tol and k < m:
            V[k] = w / nrm
            w = apply(A, V[k])
            numops += 1
            w, h = ortho(w, V, k + 1)
            nrm = np.linalg.norm(w)
            k += 1
            col = a1 * h
            col[k - 1] += a0
            for i in range(k - 1):
                t = cs[i] * col[i] + sn[i] * col[i + 1]
                col[i + 1] = -sn[i] * col[i] + cs[i] * col[i + 1]
                col[i] = t
            cs[k - 1], sn[k - 1], col[k - 1] = _givens(col[k - 1], a1 * nrm)
            R[:k, k - 1] = col
            y[k] = -sn[k - 1] * y[k - 1]
            y[k - 1] = cs[k - 1] * y[k - 1]
            beta = abs(y[k])
            if history is not None:
                history.append(beta)
        yk = sla.solve_triangular(R[:k, :k], y[:k])
        x = x + V[:k].T @ yk
        if basis_out is not None:
            basis_out.append(V[:k].copy())
        if beta > tol:
            # residual from the Krylov data: r = y[k+1] * (V_{k+1} G_1' ... G_k') e_{k+1}
            V[k] = w / nrm
            z = np.zeros(k + 1)
            z[k] = 1.0
            for i in range(k - 1, -1, -1):
                t = cs[i] * z[i] - sn[i] * z[i + 1]
                z[i + 1] = sn[i] * z[i] + cs[i] * z[i + 1]
                z[i] = t
            r = y[k] * (V[: k + 1].T @ z)
        else:
            r = b - a0 * x - a1 * apply(A, x)
            numops += 1
            beta = np.linalg.norm(r)
            if beta < tol:
                return x, True, numops, beta
        if numiter < maxiter:
            beta = np.linalg.norm(r)
            w, h, nrm = start(r, beta)
            numops += 1
    return x, False, numops, beta


def leja_shifts(H, kk, count=4, origin=0.0):
    """Up to ``count`` real parts of the eigenvalues of H[:kk, :kk] in Leja order (the point farthest from ``origin`` first, then the
    point that maximises the product of distances to those already chosen): the Newton shifts of the library's block Arnoldi step.
    ``origin``: 0 for an ordinary operator; an operator iterated in the rearranged form A = W + theta0 I passes theta0, the origin
    of W (csrc/solver.hip: ritz_shifts, bk_op::monomial_shift), so that the order is the one W itself would get."""
    pts = list(np.linalg.eigvals(H[:kk, :kk]).real)
    out = []
    while pts and len(out) < count:
        score = [abs(p - origin) if not out else float(np.prod([abs(p - q) for q in out])) for p in pts]
        out.append(pts.pop(int(np.argmax(score))))
    return out


def block_arnoldi_coefficients(G, H, k, u, s, Aq, Gp, pivot_tol=1e-8, theta=None):
    """Host algebra of the library's BLOCK Arnoldi step restated (csrc/sstep.h: block_coefficients).  ``G``: measured Gram
    matrix (valid for all k basis vectors on entry), ``H``: raw Hessenberg (columns 0..k-2 valid), ``Aq`` = Q'P (k x s),
    ``Gp`` = P'P.  The block is truncated at the first Cholesky pivot below ``pivot_tol`` of its column's squared norm.
    Returns (C, R, s_eff, ratio of the last accepted pivot) with P[:s_eff] = Q C + Q_new R, and writes columns
    k-1 .. k+s_eff-2 of H; None if not even the first column is acceptable."""
    j = k - 1
    C = np.linalg.solve(G[:k, :k], Aq)
    S = Gp - 0.5 * (C.T @ Aq + Aq.T @ C)
    R = np.zeros((s, s))
    s_eff, ratio = s, 1.0
    for a in range(s):
        v = S[a, a] - R[:a, a] @ R[:a, a]
        if not v > pivot_tol * Gp[a, a]:
            s_eff = a
            break
        R[a, a] = np.sqrt(v)
        ratio = v / Gp[a, a]
        R[a, a + 1:] = (S[a, a + 1:] - R[:a, a] @ R[:a, a + 1:]) / R[a, a]
    if s_eff == 0:
        return None
    s = s_eff
    C, R = C[:, :s], R[:s, :s]
    Pc = np.vstack([C, R])
    Bc = np.zeros((k + s, s))
    Bc[j, 0] = 1.0
    Bc[:, 1:] = Pc[:, :s - 1]
    rhs = Pc.copy()
    if theta is not None:
        rhs += Bc * np.asarray(theta[:s])[None, :]       # A B = P + B diag(theta)
    rhs[:k] -= H[:k, :j] @ Bc[:j]
    U = np.vstack([Bc[j:j + 1], Bc[k:k + s - 1]])
    H[:k + s, j:j + s] = sla.solve_triangular(U, rhs.T, trans="T", lower=False).T
    for q in range(s):
        H[j + q + 2:k + s, j + q] = 0.0
    return C, R, s_eff, ratio


def gmres_block(A, b, a0=0.0, a1=1.0, *, krylovdim=30, maxiter=100, atol=1e-12, rtol=1e-12, Pl=None, block=4,
                history=None, basis_out=None, stats=None, newton=True, shifts=None, keep_shifts=False, defer=True,
                mono_shift=0.0, predict_margin=1.0, leja_origin=None):
    """The library's GMRES for vectors that stream from HBM since round 4, restated (csrc/solver.hip: gmres_core with
    arnoldi_block): KrylovKit's restarted GMRES -- same stopping rules, restart and numops bookkeeping as gmres_krylovkit
    above -- whose Arnoldi steps are taken in BLOCKS of up to ``block``: p_1 = A q_j, .., p_s = A p_{s-1}, then ONE pass of
    projections (Q'P, P'P and the Gram columns of the previous block's vectors) and ONE update pass for the s steps.  The
    projection uses the MEASURED Gram matrix (C = G^-1 Q'P: the block form of GramCGS above), the new vectors come from the
    Cholesky factor of the projected block's Gram matrix, the s Hessenberg columns from the change of basis.  The block size
    is capped by the number of steps the residual estimate predicts to need, so no operator application is wasted in the
    cases at hand (``stats['wasted']``); numops counts consumed steps, as the library does.  ``defer``: the update pass of a
    block (Q_new = R^-T (P - C'Q)) waits until the new vectors are needed explicitly -- the next block, a restart's residual
    -- and a solve that ends inside the block folds it into the solution update, x += Q (y_old - C R^-1 y_new) + P (R^-1
    y_new) (csrc/solver.hip: PendingBlock; ``stats['folded']`` counts those).  ``mono_shift`` (round 5): the operator is iterated in
    the rearranged form A = W + mono_shift I (the stencil-free preconditioned operator T = Pl^-1 J + I); blocks without Ritz values then
    run on powers of W, p_{i+1} = (A - mono_shift) p_i, and the Leja order of the Newton shifts starts from the point farthest from
    W's origin -- the blocks are W's blocks.  ``leja_origin`` (round 6; default: mono_shift): the origin of the Leja order on its own --
    the library's default since round 6 is the first block on powers of T itself (mono_shift = 0) with the Leja order still taken from
    W's origin (leja_origin = theta0; csrc/ops.h: bk_op::rearranged_origin), the better-conditioned first block with every later block
    unchanged.  ``predict_margin``: the speculated block length aims at margin x tolerance (library
    option gmres_predict_margin, 1 since round 5).  Not a reference algorithm: the tests show it reproduces the reference
    restatement's counts, residual history and solution."""
    if Pl is not None:
        A_, a0_, a1_ = A, a0, a1
        lin = lambda dx: a1_ * Pl(apply(A_, dx)) + a0_ * dx
        return gmres_block(lin, Pl(np.asarray(b, dtype=float)), 0.0, 1.0, krylovdim=krylovdim, maxiter=maxiter, atol=atol,
                           rtol=rtol, block=block, history=history, basis_out=basis_out, stats=stats, newton=newton,
                           shifts=shifts, keep_shifts=keep_shifts, defer=defer, mono_shift=mono_shift, predict_margin=predict_margin,
                           leja_origin=leja_origin)
    if leja_origin is None:
        leja_origin = mono_shift
    b = np.asarray(b, dtype=float)
    n = b.shape[0]
    x = np.zeros(n)
    r = b.copy()
    numops = 1
    beta = np.linalg.norm(r)
    tol = max(atol, rtol * np.linalg.norm(b))
    if stats is None:
        stats = {}
    stats.update(wasted=0, refused=0, void=0, blocks=[], ratios=[], shifts=None, folded=0)
    if beta < tol:
        return x, True, numops, beta
    m = krylovdim
    blk_cur = block
    # Newton shifts p_{i+1} = (A - theta_i) p_i: Leja-ordered Ritz values, from this solve's Hessenberg as soon as a block
    # exists; ``shifts`` = a set for the FIRST block (none in the library by default: monomial, at most 3 long), replaced by
    # Ritz values after it unless ``keep_shifts`` (then kept until it truncates a block: the library's option
    # gmres_newton_carry, which starts from the previous solve's set)
    shifts = list(shifts) if (shifts and newton) else []
    carried = bool(shifts) and keep_shifts
    for numiter in range(1, maxiter + 1):
        Q = np.zeros((m + 1, n))
        H = np.zeros((m + 2, m))
        G = np.eye(m + 1)
        Q[0] = r / beta
        cs, sn, y, Rm = np.zeros(m), np.zeros(m), np.zeros(m + 1), np.zeros((m, m))
        y[0] = beta
        j, gram_n, k_done = 0, 0, 0
        res, res_prev = beta, None
        classic = False
        pend = None                                      # (k, got, C, R, P) of a block whose update pass has not run

        def flush():
            nonlocal pend
            if pend is not None:
                k0, g, C0, R0, P0 = pend
                Q[k0:k0 + g] = sla.solve_triangular(R0, P0[:g] - C0.T @ Q[:k0], trans="T", lower=False)
                pend = None

        while j < m and res > tol:
            k = j + 1
            flush()
            sb = min(blk_cur, block if shifts else min(block, 3), m - j)     # monomial blocks: at most 3 (csrc/solver.hip: kMonomialMax)
            capped = sb < blk_cur
            predicted = False
            if res_prev is not None and sb > 1:
                rho = res / res_prev if res < res_prev else 1.0
                need, bb = 1, res * rho
                while bb > predict_margin * tol and need < sb:
                    bb *= rho
                    need += 1
                predicted = need < sb
                sb = need
            done = False
            if not classic:
                P = np.zeros((sb, n))
                p = Q[j]
                theta = [shifts[i % len(shifts)] for i in range(sb)] if shifts else ([mono_shift] * sb if mono_shift != 0.0 else None)
                for i in range(sb):
                    p = apply(A, p) - (theta[i] * p if theta else 0.0)
                    P[i] = p
                u = k - gram_n
                Gn = Q[:k] @ Q[k - u:k].T
                G[:k, k - u:k] = Gn
                G[k - u:k, :k] = Gn.T
                out = block_arnoldi_coefficients(G, H, k, u, sb, Q[:k] @ P.T, P @ P.T, theta=theta)
                if out is None:
                    classic = True                       # refused: the rest of the cycle runs step by step (MGS2 here)
                    gram_n = j
                    stats["refused"] += 1
                    stats["void"] += sb
                else:
                    C, R, got, ratio = out
                    gram_n = k
                    pend = (k, got, C, R, P)
                    if not defer:
                        flush()
                    stats["void"] += sb - got            # operator applications of a truncated block's tail
                    if got < sb and carried:
                        shifts, carried = [], False
                    if newton and not carried and (len(shifts) < 4 or j + got <= 12):
                        shifts = leja_shifts(H, j + got, origin=leja_origin)
                    if got < sb:
                        blk_cur = got
                    elif not capped and not predicted and blk_cur < block and ratio >= 1e-4:
                        blk_cur += 1
                    sb = got
                    numops += sb
                    stats["blocks"].append(sb)
                    stats["ratios"].append(ratio)        # last pivot ratio of the block (conditioning margin, csrc/sstep.h)
                    done = True
            if not done:
                sb = 1
                w, h = _mgs2(apply(A, Q[j]), Q, k)
                numops += 1
                H[:k, j] = h
                H[k, j] = np.linalg.norm(w)
                Q[k] = w / H[k, j]
            for jj in range(j, j + sb):
                col = a1 * H[:jj + 2, jj]
                col[jj] += a0
                for i in range(jj):
                    t = cs[i] * col[i] + sn[i] * col[i + 1]
                    col[i + 1] = -sn[i] * col[i] + cs[i] * col[i + 1]
                    col[i] = t
                cs[jj], sn[jj], col[jj] = _givens(col[jj], col[jj + 1])
                Rm[:jj + 1, jj] = col[:jj + 1]
                y[jj + 1] = -sn[jj] * y[jj]
                y[jj] = cs[jj] * y[jj]
                res_prev, res = res, abs(y[jj + 1])
                k_done = jj + 1
                if history is not None:
                    history.append(res)
                if not res > tol:
                    stats["wasted"] += j + sb - k_done
                    numops -= j + sb - k_done
                    break
            j += sb
        k = k_done
        yk = sla.solve_triangular(Rm[:k, :k], y[:k])
        if pend is not None and k > pend[0]:
            k0, g, C0, R0, P0 = pend
            c = k - k0
            tnew = sla.solve_triangular(R0[:c, :c], yk[k0:k], lower=False)        # R^-1 y_new (leading c x c block)
            x = x + Q[:k0].T @ (yk[:k0] - C0[:, :c] @ tnew) + P0[:c].T @ tnew
            stats["folded"] += 1
        else:
            x = x + Q[:k].T @ yk
        flush()
        if basis_out is not None:
            basis_out.append(Q[:k].copy())
        beta = res
        if beta > tol:
            z = np.zeros(k + 1)
            z[k] = 1.0
            for i in range(k - 1, -1, -1):
                t = cs[i] * z[i] - sn[i] * z[i + 1]
                z[i + 1] = sn[i] * z[i] + cs[i] * z[i + 1]
                z[i] = t
            r = y[k] * (Q[:k + 1].T @ z)
        else:
            r = b - a0 * x - a1 * apply(A, x)
            numops += 1
            beta = np.linalg.norm(r)
            if beta < tol:
                stats["shifts"] = list(shifts)
                return x, True, numops, beta
        beta = np.linalg.norm(r)
    stats["shifts"] = list(shifts)
    return x, False, numops, beta


def minres_krylovjl(A, b, a0=0.0, a1=1.0, *, atol=None, rtol=None, itmax=0, M=None):
    """Krylov.jl `minres` (KrylovLS(KrylovAlg = :minres), src/LinearSolver.jl:339-341: symmetric operator v -> a0 v + a1 A v,
    "centered" SPD preconditioner M = Pl passed as a callable applying Pl^-1): Paige-Saunders MINRES, x0 = 0, stop when
    the estimated residual norm (in the M^-1 inner product) <= atol + rtol * beta1.  Returns (x, solved, niter)."""
    b = np.asarray(b, dtype=float)
    n = b.shape[0]
    eps = np.finfo(float).eps
    atol = np.sqrt(eps) if atol is None else atol
    rtol = np.sqrt(eps) if rtol is None else rtol
    itmax = 2 * n if itmax == 0 else itmax
    op = lambda v: axpy_op(A, v, a0, a1)
    prec = (lambda v: v) if M is None else M
    x = np.zeros(n)
    r1 = b.copy()
    y = prec(r1)
    beta1 = float(np.dot(r1, y))
    if beta1 < 0:
        raise ValueError("minres: the preconditioner is not positive definite")
    if beta1 == 0.0:
        return x, True, 0
    beta1 = np.sqrt(beta1)
    tol = atol + rtol * beta1
    oldb, beta, dbar, epsln, phibar = 0.0, beta1, 0.0, 0.0, beta1
    cs, sn = -1.0, 0.0
    w = np.zeros(n); w2 = np.zeros(n)
    r2 = r1.copy()
    it = 0
    solved = phibar <= tol
    while not solved and it < itmax:
        it += 1
        v = y / beta
        y = op(v)
        if it >= 2:
            y = y - (beta / oldb) * r1
        alfa = float(np.dot(v, y))
        y = y - (alfa / beta) * r2
        r1, r2 = r2, y
        y = prec(r2)
        oldb = beta
        b2 = float(np.dot(r2, y))
        if b2 < 0:
            raise ValueError("minres: the preconditioner is not positive definite")
        beta = np.sqrt(b2)
        oldeps = epsln
        delta = cs * dbar + sn * alfa
        gbar = sn * dbar - cs * alfa
        epsln = sn * beta
        dbar = -cs * beta
        gamma = max(np.hypot(gbar, beta), eps)
        cs, sn = gbar / gamma, beta / gamma
        phi = cs * phibar
        phibar = sn * phibar
        w1, w2 = w2, w
        w = (v - oldeps * w1 - delta * w2) / gamma
        x = x + phi * w
        solved = phibar <= tol
    return x, bool(solved), it


def cg_krylovjl(A, b, a0=0.0, a1=1.0, *, atol=None, rtol=None, itmax=0, M=None):
    """Krylov.jl `cg` (KrylovLS(KrylovAlg = :cg)): preconditioned conjugate gradients for an SPD operator, x0 = 0,
    residual norm sqrt(r' M^-1 r) <= atol + rtol * its initial value.  Returns (x, solved, niter)."""
    b = np.asarray(b, dtype=float)
    n = b.shape[0]
    eps = np.finfo(float).eps
    atol = np.sqrt(eps) if atol is None else atol
    rtol = np.sqrt(eps) if rtol is None else rtol
    itmax = 2 * n if itmax == 0 else itmax
    op = lambda v: axpy_op(A, v, a0, a1)
    prec = (lambda v: v) if M is None else M
    x = np.zeros(n)
    r = b.copy()
    z = prec(r)
    p = z.copy()
    gamma = float(np.dot(r, z))
    rnorm = np.sqrt(max(gamma, 0.0))
    if gamma == 0.0:
        return x, True, 0
    tol = atol + rtol * rnorm
    it = 0
    solved = rnorm <= tol
    while not solved and it < itmax:
        Ap = op(p)
        pAp = float(np.dot(p, Ap))
        if pAp <= 0.0:                      # not positive definite along p: Krylov.jl stops (zero / negative curvature)
            break
        alpha = gamma / pAp
        x = x + alpha * p
        r = r - alpha * Ap
        z = prec(r)
        gnext = float(np.dot(r, z))
        rnorm = np.sqrt(max(gnext, 0.0))
        beta = gnext / gamma
        gamma = gnext
        p = z + beta * p
        it += 1
        solved = rnorm <= tol
    return x, bool(solved), it


def gmres_iterativesolvers(A, b, a0=0.0, a1=1.0, *, restart=200, maxiter=100, reltol=1e-8, abstol=0.0,
                           Pl=None, Pr=None):
    """IterativeSolvers.gmres on v -> a0 v + a1 A v (src/LinearSolver.jl:195-201), x0 = 0.
    ``Pl`` / ``Pr`` (optional) are callables applying Pl^-1 / Pr^-1: the iteration runs on Pl^-1 A Pr^-1 y = Pl^-1 b and the
    solution is x = Pr^-1 y (the package's expand! / update_solution!).  Returns (x, isconverged, iters)."""
    if Pr is not None:
        A_, a0_, a1_ = A, a0, a1
        y, ok, it = gmres_iterativesolvers(lambda v: axpy_op(A_, Pr(v), a0_, a1_), b, 0.0, 1.0, restart=restart, maxiter=maxiter,
                                           reltol=reltol, abstol=abstol, Pl=Pl)
        return Pr(y), ok, it
    b = np.asarray(b, dtype=float)
    n = b.shape[0]
    op = lambda v: axpy_op(A, v, a0, a1)
    prec = (lambda v: v) if Pl is None else Pl
    x = np.zeros(n)
    restart = min(restart, n)
    r = prec(b.copy())
    beta = np.linalg.norm(r)
    tol = max(reltol * beta, abstol)
    iters = 0
    if beta <= tol:
        return x, True, 0
    V = np.zeros((restart + 1, n))
    while iters < maxiter:
        H = np.zeros((restart + 1, restart))
        cs = np.zeros(restart)
        sn = np.zeros(restart)
        g = np.zeros(restart + 1)
        g[0] = beta
        V[0] = r / beta
        k = 0
        while k < restart and iters < maxiter:
            w = prec(op(V[k]))
            for i in range(k + 1):
                H[i, k] = V[i] @ w
                w = w - H[i, k] * V[i]
            H[k + 1, k] = np.linalg.norm(w)
            if H[k + 1, k] != 0.0:
                V[k + 1] = w / H[k + 1, k]
            for i in range(k):
                t = cs[i] * H[i, k] + sn[i] * H[i + 1, k]
                H[i + 1, k] = -sn[i] * H[i, k] + cs[i] * H[i + 1, k]
                H[i, k] = t
            cs[k], sn[k], H[k, k] = _givens(H[k, k], H[k + 1, k])
            H[k + 1, k] = 0.0
            g[k + 1] = -sn[k] * g[k]
            g[k] = cs[k] * g[k]
            k += 1
            iters += 1
            beta = abs(g[k])
            if beta <= tol:
                break
        yk = sla.solve_triangular(H[:k, :k], g[:k])
        x = x + V[:k].T @ yk
        if beta <= tol:
            return x, True, iters
        r = prec(b - op(x))
        beta = np.linalg.norm(r)
    return x, beta <= tol, iters


def gmres_krylovjl(A, b, a0=0.0, a1=1.0, *, memory=20, restart=False, itmax=0, atol=np.sqrt(np.finfo(float).eps),
                   rtol=np.sqrt(np.finfo(float).eps), M=None, N=None):
    """Krylov.jl `gmres` as BifurcationKit's KrylovLS calls it (src/LinearSolver.jl:336-345: `krylov_solve(Val(:gmres), Jmap,
    rhs; kwargs..., M = Pl, N = Pr)`) on v -> a0 v + a1 A v, x0 = 0: left preconditioner M, modified Gram-Schmidt, Givens QR
    of the Hessenberg matrix, stop on ||M r|| <= atol + rtol ||M r0||.  `memory` is the initial basis size; with
    `restart = false` (the package default) the basis simply keeps growing, with `restart = true` the method restarts
    every `memory` steps.  ``N``: right preconditioner (the method solves M A N y = M b, x = N y).  Package knowledge
    (SURVEY Appendix B), not verifiable here.  Returns (x, solved, niter)."""
    if N is not None:
        A_, a0_, a1_ = A, a0, a1
        y, ok, it = gmres_krylovjl(lambda v: axpy_op(A_, N(v), a0_, a1_), b, 0.0, 1.0, memory=memory, restart=restart, itmax=itmax,
                                   atol=atol, rtol=rtol, M=M)
        return N(y), ok, it
    b = np.asarray(b, dtype=float)
    n = b.shape[0]
    op = lambda v: axpy_op(A, v, a0, a1)
    prec = (lambda v: v) if M is None else M
    itmax = itmax if itmax > 0 else 2 * n
    x = np.zeros(n)
    r = prec(b.copy())
    beta = np.linalg.norm(r)
    tol = atol + rtol * beta
    it = 0
    if beta <= tol:
        return x, True, 0
    cap = memory if restart else itmax
    while it < itmax:
        V = [r / beta]
        Hc, cs, sn, g = [], [], [], [beta]
        k = 0
        while k < cap and it < itmax:
            w = prec(op(V[k]))
            h = np.zeros(k + 2)
            for i in range(k + 1):
                h[i] = V[i] @ w
                w = w - h[i] * V[i]
            h[k + 1] = np.linalg.norm(w)
            if h[k + 1] != 0.0:
                V.append(w / h[k + 1])
            for i in range(k):
                t = cs[i] * h[i] + sn[i] * h[i + 1]
                h[i + 1] = -sn[i] * h[i] + cs[i] * h[i + 1]
                h[i] = t
            c_, s_, rr = _givens(h[k], h[k + 1])
            cs.append(c_); sn.append(s_)
            h[k], h[k + 1] = rr, 0.0
            Hc.append(h[:k + 1].copy())
            g.append(-s_ * g[k])
            g[k] = c_ * g[k]
            k += 1
            it += 1
            beta = abs(g[k])
            if beta <= tol or len(V) <= k:
                break
        R = np.zeros((k, k))
        for j, col in enumerate(Hc):
            R[:j + 1, j] = col
        yk = sla.solve_triangular(R, np.asarray(g[:k]))
        x = x + np.asarray(V[:k]).T @ yk
        if beta <= tol:
            return x, True, it
        r = prec(b - op(x))
        beta = np.linalg.norm(r)
    return x, beta <= tol, it


def _which_key(which):
    if which == "LM":
        return lambda lam: -np.abs(lam)
    if which == "LR":
        return lambda lam: -np.real(lam)
    if which == "SR":
        return lambda lam: np.real(lam)
    raise ValueError(which)


def eigsolve_krylovschur(A, x0, howmany, which="LM", *, tol=1e-12, krylovdim=30, maxiter=100,
                         hermitian=False):
    """Arnoldi (Lanczos when ``hermitian``) with Krylov-Schur thick restart.
    Returns (vals[complex], vecs[list of ndarray], nconverged, numops)."""
    x0 = np.asarray(x0, dtype=float)
    n = x0.shape[0]
    m = min(krylovdim, n)
    key = _which_key(which)
    V = np.zeros((m + 1, n))
    H = np.zeros((m + 1, m))
    V[0] = x0 / np.linalg.norm(x0)
    k = 0
    numops = 0
    numiter = 0
    while True:
        numiter += 1
        while k < m:
            w = apply(A, V[k])
            numops += 1
            w, h = _mgs2(w, V, k + 1)
            beta = np.linalg.norm(w)
            H[: k + 1, k] = h
            H[k + 1, k] = beta
            V[k + 1] = w / beta if beta > 0 else 0.0
            k += 1
        B = H[:m, :m].copy()
        bvec = H[m, :m].copy()
        if hermitian:
            B = 0.5 * (B + B.T)
            lam, Y = np.linalg.eigh(B)
            lam = lam.astype(complex)
            Y = Y.astype(complex)
        else:
            lam, Y = np.linalg.eig(B)
            Y = Y / np.linalg.norm(Y, axis=0)
        order = np.argsort([key(l) for l in lam], kind="stable")
        lam, Y = lam[order], Y[:, order]
        res = np.abs(bvec @ Y)
        nconv = 0
        while nconv < m and res[nconv] < tol:
            nconv += 1
        if nconv >= howmany or numiter >= maxiter or m == n:
            break
        keep = (3 * m + 2 * nconv) // 5
        # never split a complex-conjugate pair
        if (not hermitian) and keep < m and abs(lam[keep - 1].imag) > 0 and \
                np.isclose(lam[keep - 1], np.conj(lam[keep])):
            keep += 1
        # real orthonormal basis Q of the invariant subspace spanned by the kept Ritz vectors
        Z = np.concatenate([Y[:, :keep].real, Y[:, :keep].imag], axis=1)
        Uz, sv, _ = np.linalg.svd(Z, full_matrices=False)
        Q = Uz[:, :keep]
        Vnew = Q.T @ V[:m]
        Bk = Q.T @ B @ Q
        bk = bvec @ Q
        V[:keep] = Vnew
        V[keep] = V[m]
        H[:] = 0.0
        H[:keep, :keep] = Bk
        H[keep, :keep] = bk
        k = keep
    nout = max(howmany, min(nconv, m)) if nconv > howmany else howmany
    nout = min(nout, m)
    vals = lam[:nout]
    vecs = []
    for i in range(nout):
        v = (Y[:, i] @ V[:m].astype(complex)) if np.iscomplexobj(Y) else Y[:, i] @ V[:m]
        if np.abs(v.imag).max() < 1e-14 * max(1.0, np.abs(v.real).max()):
            v = v.real
        vecs.append(v)
    return vals, vecs, min(nconv, nout), numops


def shift_invert(J, nev, sigma, ls, eig):
    """ShiftInvert, src/EigSolver.jl:257-266.  ``ls(J, rhs, a0, a1) -> (x, ok, it)``;
    ``eig(Jmap, nev) -> (vals, vecs, cv, n)``.  Back-transform 1/mu + sigma, sort by real part desc."""
    Jmap = lambda rhs: ls(J, rhs, -sigma, 1.0)[0]
    vals, vecs, cv, nops = eig(Jmap, nev)
    lam = 1.0 / np.asarray(vals) + sigma
    ind = sorted(range(len(lam)), key=lambda i: -lam[i].real)      # __sort_spectrum :16-19
    return np.asarray([complex(lam[i]) for i in ind]), [vecs[i] for i in ind], cv, nops


def default_eig(J, nev):
    """DefaultEig, src/EigSolver.jl:42-49: ``LA.eigen(Array(J); sortby = real)`` then the last ``nev``
    entries in reverse order (decreasing real part).  Eigenvectors as LAPACK *geev returns them (unit
    2-norm, largest component real) -- NumPy calls the same routine."""
    A = J.toarray() if hasattr(J, "toarray") else np.asarray(J, dtype=float)
    w, V = np.linalg.eig(A)
    order = np.argsort(w.real, kind="stable")
    w, V = w[order], V[:, order]
    nev2 = min(nev, len(w))
    sel = list(range(len(w) - 1, len(w) - 1 - nev2, -1))
    return w[sel].astype(complex), V[:, sel].astype(complex), True, 1
