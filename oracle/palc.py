"""ORACLE (test infrastructure) -- Newton, the PALC corrector, tangent predictors, step-size
control and a minimal continuation loop, restated on flat NumPy vectors.

  newton                 _newton                     src/Newton.jl:66-114
  dot_theta / arc_length_eq / newton_palc            src/continuation/Palc.jl:23-56, 187-305
  secant_tangent / bordered_tangent / add_tangent    src/continuation/Tangents.jl:8-15, 28-54, 71-104
  step_size_control      _step_size_control!         src/continuation/Contbase.jl:77-102
  is_stable                                          src/Bifurcations.jl:5-19
  continuation           iterate / iterate_from_two_points / iterate(it,state)
                                                     src/Continuation.jl:349-456, 458-504
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np

from .bordered import solve_bls_palc

EPS_FD = float(np.sqrt(np.finfo(float).eps))       # getdelta: src/Problems.jl:69-70, 452


@dataclass
class Problem:
    """Mirror of BifurcationProblem's (F, J) pair (src/Problems.jl:89-123, 133-149)."""
    F: callable           # F(x, p) -> residual
    J: callable           # J(x, p) -> matrix | callable (opaque to the engine, :98-101)
    delta: float = EPS_FD
    # optional: the pointwise factor phi_p(x) of dF/dp when the continuation parameter multiplies a pointwise term
    # (F = S x + p phi_p(x) + ...).  The reference's quotient (F(x, p + eps) - F(x, p)) / eps then equals
    # ((p + eps) - p) / eps * phi_p(x) EXACTLY in exact arithmetic, the stencil part cancelling identically; evaluating it
    # in this form removes the ~eps_mach |L1 x| / eps rounding noise of the two-residual form.  None
    # (default) = the literal form of Palc.jl:239-240 / Tangents.jl:77-82.
    dparam_factor: callable = None


def dF_dparam(prob, x, p, res_f=None):
    """dFdp = (F(x, p + eps) - F(x, p)) / eps (Palc.jl:239-240, Tangents.jl:77-82); cancellation-free when the problem
    supplies ``dparam_factor`` (what bk_residual_dparam evaluates on the device)."""
    eps = prob.delta
    if prob.dparam_factor is not None:
        return (((p + eps) - p) / eps) * prob.dparam_factor(x, p)
    f0 = res_f if res_f is not None else prob.F(x, p)
    return (prob.F(x, p + eps) - f0) * (1.0 / eps)


def norminf(x):
    return float(np.max(np.abs(x))) if x.size else 0.0


def norm2(x):
    return float(np.linalg.norm(x))


def cb_default(state, **kw):
    """cb_default, src/Newton.jl:151."""
    return True


def cb_max_norm(maxres):
    """cbMaxNorm, src/Newton.jl:156-159: veto iterates whose residual exceeds ``maxres``."""
    return lambda state, **kw: state["residual"] < maxres


def newton(prob, x0, p, ls, *, tol=1e-12, max_iterations=25, normN=norm2, callback=cb_default):
    """_newton, src/Newton.jl:66-114.  Returns dict(u, residuals, converged, itnewton, itlineartot).
    ``callback(state; fromNewton)`` is consulted before the loop, after every iteration and for the final flag (:88,111,114)."""
    x = x0.copy()
    fx = prob.F(x, p)
    res = normN(fx)
    residuals = [res]
    step = 0
    itlin = 0
    compute = callback(dict(x=x, fx=fx, residual=res, step=step, residuals=residuals), fromNewton=True)
    while step < max_iterations and res > tol and compute:
        J = prob.J(x, p)
        u, cv, it = ls(J, fx)
        itlin += int(np.sum(it))
        x = x - u
        fx = prob.F(x, p)
        res = normN(fx)
        residuals.append(res)
        step += 1
        compute = callback(dict(x=x, fx=fx, residual=res, step=step, itlinear=it, residuals=residuals), fromNewton=True)
    flag = (residuals[-1] < tol) and bool(callback(dict(x=x, fx=fx, residual=res, step=step, residuals=residuals), fromNewton=True))
    return dict(u=x, residuals=residuals, converged=flag, itnewton=step, itlineartot=itlin)


def dot_theta(u1, u2, p1, p2, theta):
    """DotTheta, Palc.jl:35 with NormalisedDot (:1-6): theta*<u1,u2>/N + (1-theta)*p1*p2."""
    return float(np.dot(u1, u2)) / u1.shape[0] * theta + p1 * p2 * (1.0 - theta)


def norm_theta(u, p, theta):
    return float(np.sqrt(dot_theta(u, u, p, p, theta)))


def arc_length_eq(u1, u2, p, du, dp, theta, ds):
    """Palc.jl:44-56 (two separate dots, exactly as written there)."""
    return (dot_theta(u1, du, p, dp, theta) - ds) - (dot_theta(u2, du, p, 0.0, theta) - 0.0)


def newton_palc(prob, z0, tau0, z_pred, ds, theta, bls, *, tol=1e-12, max_iterations=25,
                p_min=-np.inf, p_max=np.inf, normN=norm2, linesearch=False, alpha=1.0, alpha_min=1e-3,
                callback=cb_default):
    """newton_palc, Palc.jl:187-305, both the plain update (:282-285) and the line search (:254-281), with the
    callback veto (:235, 294-297).  z0, tau0, z_pred are (u, p) pairs.
    Returns dict(u, p, residuals, converged, itnewton, itlineartot)."""
    eps = prob.delta
    N = lambda u, p: arc_length_eq(u, z0[0], p - z0[1], tau0[0], tau0[1], theta, ds)
    x = z_pred[0].copy()
    p = float(z_pred[1])
    res_f = prob.F(x, p)
    res_n = N(x, p)
    res = max(normN(res_f), abs(res_n))
    residuals = [res]
    step = 0
    itlin = 0
    alpha0 = alpha
    line_step = True
    compute = callback(dict(x=x, res_f=res_f, residual=res, step=step, z0=z0, p=p, residuals=residuals), fromNewton=False)
    while step < max_iterations and res > tol and line_step and compute:
        dFdp = dF_dparam(prob, x, p, res_f)
        J = prob.J(x, p)
        u, up, flag, it = solve_bls_palc(bls, theta, tau0[0], tau0[1], J, dFdp, res_f, res_n)
        itlin += int(np.sum(it))
        if linesearch:
            line_step = False
            while not line_step and alpha > alpha_min:
                x_pred = x - alpha * u
                p_pred = p - alpha * up
                res_f = prob.F(x_pred, p_pred)
                res_n = N(x_pred, p_pred)
                res = max(normN(res_f), abs(res_n))
                if res < residuals[-1]:
                    if res < residuals[-1] / 4 and alpha < 1:
                        alpha *= 2
                    line_step = True
                    x = x_pred
                    p = float(np.clip(p_pred, p_min, p_max))
                else:
                    alpha /= 2
            alpha = alpha0
        else:
            x = x - u
            p = float(np.clip(p - up, p_min, p_max))
            res_f = prob.F(x, p)
            res_n = N(x, p)
            res = max(normN(res_f), abs(res_n))
        residuals.append(res)
        step += 1
        compute = callback(dict(x=x, res_f=res_f, residual=res, step=step, itlinear=it, z0=z0, p=p, residuals=residuals),
                           fromNewton=False)
    flag = (residuals[-1] < tol) and bool(callback(dict(x=x, res_f=res_f, residual=res, step=step, p=p, residuals=residuals),
                                                   fromNewton=False))
    return dict(u=x, p=p, residuals=residuals, converged=flag, itnewton=step, itlineartot=itlin)


def natural_corrector(prob, z_pred, ls, *, p_min, p_max, **newton_kw):
    """corrector!(state, it, ::Natural), src/continuation/Natural.jl:38-58: plain Newton from z_pred.u at the clamped
    parameter; the new point takes p = z_pred.p (already clamped by the caller, Palc.jl:157-160)."""
    pc = float(np.clip(z_pred[1], p_min, p_max))
    sol = newton(prob, z_pred[0], pc, ls, **newton_kw)
    sol["p"] = pc
    return sol


def secant_tangent(z1, z0, ds, theta):
    """_secant_tangent!, Tangents.jl:28-42: tau = (z1 - z0) * sign(ds)/||z1 - z0||_theta."""
    tu = z1[0] - z0[0]
    tp = z1[1] - z0[1]
    a = np.sign(ds) / norm_theta(tu, tp, theta)
    return tu * a, tp * a


def bordered_tangent(prob, z, tau, theta, bls):
    """gettangent!(::Bordered), Tangents.jl:71-104.  Returns (tau_u, tau_p, flag)."""
    eps = prob.delta
    dFdl = dF_dparam(prob, z[0], z[1])
    J = prob.J(z[0], z[1])
    tu, tp, flag, _ = solve_bls_palc(bls, theta, tau[0], tau[1], J, dFdl, np.zeros_like(z[0]), 1.0)
    a = 1.0 / np.sqrt(dot_theta(tu, tu, tp, tp, theta))
    a *= np.sign(dot_theta(tau[0], tu, tau[1], tp, theta))
    return tu * a, tp * a, flag


def add_tangent(z, tau, ds):
    """addtangent!, Tangents.jl:8-15: z_pred = z + ds * tau."""
    return z[0] + ds * tau[0], z[1] + ds * tau[1]


def step_size_control(ds, converged, itnewton, *, a=0.5, Nmax=25, dsmin=1e-4, dsmax=1e-1):
    """_step_size_control!, Contbase.jl:77-102.  Returns (dsnew, stop)."""
    if not converged:
        if abs(ds) <= dsmin:
            return ds, True
        dsnew = np.sign(ds) * max(abs(ds) / 2.0, dsmin)
    else:
        factor = (Nmax - itnewton) / Nmax
        dsnew = ds * (1.0 + a * factor**2)
    dsnew = np.sign(dsnew) * min(max(abs(dsnew), dsmin), dsmax)      # clamp_ds, ContParameters.jl:107
    return float(dsnew), False


def is_stable(eigvalues, tol_stability=1e-10):
    """is_stable, Bifurcations.jl:5-19 -> (n_unstable, n_imag)."""
    ev = np.asarray(eigvalues)
    n_unstable = int(np.sum(ev.real > tol_stability))
    n_imag = int(np.sum((np.abs(ev.imag) > tol_stability) & (ev.real > tol_stability)))
    return n_unstable, n_imag


@dataclass
class Branch:
    param: list = field(default_factory=list)
    itnewton: list = field(default_factory=list)
    itlinear: list = field(default_factory=list)
    ds: list = field(default_factory=list)
    n_unstable: list = field(default_factory=list)
    residuals: list = field(default_factory=list)
    sol: list = field(default_factory=list)
    eig: list = field(default_factory=list)


def continuation(prob, x0, p0, *, ls, bls, ds=1e-2, dsmin=1e-4, dsmax=1e-1, theta=0.5, a=0.5,
                 p_min=-1.0, p_max=1.0, max_steps=10, eta=150.0, tol=1e-12, max_iterations=25,
                 tangent="secant", normC=norm2, eig=None, nev=3, tol_stability=1e-10,
                 keep_solutions=False):
    """Minimal PALC branch: two Newton solves -> secant tangent -> predictor/corrector loop.
    Continuation.jl:349-456 (first two points), :458-504 (one step).  ``eig(J, nev) -> (vals, vecs, cv, it)``."""
    newton_kw = dict(tol=tol, max_iterations=max_iterations, normN=normC)
    s0 = newton(prob, x0, p0, ls, **newton_kw)
    assert s0["converged"], "Newton failed on the initial guess"
    p1 = p0 + ds / eta
    s1 = newton(prob, s0["u"], p1, ls, **newton_kw)
    assert s1["converged"], "Newton failed for the initial tangent"
    z0 = (s0["u"], p0)
    z1 = (s1["u"], p1)
    n_unst = -1
    br = Branch()
    if eig is not None:
        vals, _, _, _ = eig(prob.J(z0[0], z0[1]), nev)
        n_unst = is_stable(vals, tol_stability)[0]
        br.eig.append(np.asarray(vals))
    tau = secant_tangent(z1, z0, ds, theta)              # initialize!, Palc.jl:112-123
    z = (z0[0].copy(), z0[1])
    z_pred = add_tangent(z, tau, ds)
    br.param.append(z[1]); br.itnewton.append(s0["itnewton"]); br.itlinear.append(s0["itlineartot"])
    br.ds.append(ds); br.n_unstable.append(n_unst); br.residuals.append(s0["residuals"])
    if keep_solutions:
        br.sol.append(z[0].copy())
    step = 0
    z_old = (z[0].copy(), z[1])
    n_prev = -1                                                            # state.n_unstable[2]
    while step < max_steps and (p_min < z[1] < p_max or step == 0):        # done(it, state), Continuation.jl:254-257
        if z_pred[1] <= p_min or z_pred[1] >= p_max:                       # corrector!(::PALC), Palc.jl:157-160
            z_pred = (z_pred[0], float(np.clip(z_pred[1], p_min, p_max)))
            sol = natural_corrector(prob, z_pred, ls, p_min=p_min, p_max=p_max, **newton_kw)
        else:
            sol = newton_palc(prob, z, tau, z_pred, ds, theta, bls, tol=tol, max_iterations=max_iterations,
                              p_min=p_min, p_max=p_max, normN=normC)
        conv = sol["converged"]
        if conv:
            z_old = (z[0].copy(), z[1])
            z = (sol["u"], sol["p"])
            if eig is not None:
                nev_ = max(n_prev + 5, nev)                                # n = state.n_unstable[2], Utils.jl:78-79
                vals, _, _, _ = eig(prob.J(z[0], z[1]), nev_)
                n_prev = n_unst
                n_unst = is_stable(vals, tol_stability)[0]
                br.eig.append(np.asarray(vals))
            step += 1
            br.param.append(z[1]); br.itnewton.append(sol["itnewton"]); br.itlinear.append(sol["itlineartot"])
            br.ds.append(ds); br.n_unstable.append(n_unst); br.residuals.append(sol["residuals"])
            if keep_solutions:
                br.sol.append(z[0].copy())
        ds, stop = step_size_control(ds, conv, sol["itnewton"], a=a, Nmax=max_iterations, dsmin=dsmin, dsmax=dsmax)
        if stop:
            break
        if conv:                                                           # Palc.jl:140-143
            if tangent == "secant":
                tau = secant_tangent(z, z_old, ds, theta)
            else:
                tu, tp, _ = bordered_tangent(prob, z, tau, theta, bls)
                tau = (tu, tp)
        z_pred = add_tangent(z, tau, ds)
    return br
