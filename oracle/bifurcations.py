"""ORACLE (test infrastructure) -- state-based continuation with bifurcation location by bisection, on flat NumPy vectors.

  ContState / copy                                   ContState, src/Continuation.jl:150-250 (the fields the PALC path uses)
  iterate                                            Base.iterate(it, state), src/Continuation.jl:458-504
  detect_bifurcation                                 src/Bifurcations.jl:22-28
  locate_bifurcation                                 locate_bifurcation!, src/Bifurcations.jl:159-349
  bifurcation_type                                   _get_bifurcation_type, src/Bifurcations.jl:80-150 (codim-1 cases)
  continuation                                       continuation!, src/Continuation.jl:505-600 with detect_bifurcation = 3
"""
from __future__ import annotations

import copy as _copy
from dataclasses import dataclass, field

import numpy as np

from . import palc


@dataclass
class ContPar:
    """The ContinuationPar / NewtonPar fields read on this path (src/ContParameters.jl:44-100)."""
    ds: float = 1e-2
    dsmin: float = 1e-4
    dsmax: float = 1e-1
    a: float = 0.5
    theta: float = 0.5
    p_min: float = -1.0
    p_max: float = 1.0
    max_steps: int = 20
    eta: float = 150.0
    nev: int = 3
    tol_stability: float = 1e-10
    tol: float = 1e-12                     # NewtonPar.tol, src/Newton.jl:19
    max_iterations: int = 25
    tangent: str = "secant"
    dsmin_bisection: float = 1e-16
    n_inversion: int = 2
    max_bisection_steps: int = 25
    tol_bisection_eigenvalue: float = 1e-16


@dataclass
class ContState:
    z: tuple
    z_old: tuple
    tau: tuple
    z_pred: tuple
    ds: float
    step: int = 0
    converged: bool = True
    itnewton: int = 0
    itlinear: int = 0
    stepsizecontrol: bool = True
    stop: bool = False
    n_unstable: tuple = (-1, -1)          # (current, previous)
    n_imag: tuple = (-1, -1)
    eigvals: np.ndarray | None = None
    residuals: list = field(default_factory=list)

    def copy(self):
        c = _copy.copy(self)
        for name in ("z", "z_old", "tau", "z_pred"):
            u, p = getattr(self, name)
            setattr(c, name, (u.copy(), p))
        c.eigvals = None if self.eigvals is None else self.eigvals.copy()
        c.residuals = list(self.residuals)
        return c


def detect_bifurcation(st: ContState) -> bool:
    n1, n2 = st.n_unstable
    return not (n1 == -1 or n2 == -1) and n1 != n2


def _eigen(prob, st, eig, cp):
    nev_ = max(st.n_unstable[1] + 5, cp.nev)                              # n = state.n_unstable[2], src/Utils.jl:78-79
    vals = np.asarray(eig(prob.J(st.z[0], st.z[1]), nev_)[0])
    nu, ni = palc.is_stable(vals, cp.tol_stability)
    st.n_unstable = (nu, st.n_unstable[0])                                # update_stability!, Continuation.jl:274-278
    st.n_imag = (ni, st.n_imag[0])
    st.eigvals = vals


def iterate(prob, st: ContState, *, ls, bls, eig, cp: ContPar, normC=palc.norm2):
    """One pass of Base.iterate (Continuation.jl:458-504); returns the state, or None when `done` says stop (:254-257)."""
    if not (st.step <= cp.max_steps and (cp.p_min < st.z[1] < cp.p_max or st.step == 0) and not st.stop):
        return None
    if st.z_pred[1] <= cp.p_min or st.z_pred[1] >= cp.p_max:              # corrector!(::PALC) -> Natural, Palc.jl:157-160
        st.z_pred = (st.z_pred[0], float(np.clip(st.z_pred[1], cp.p_min, cp.p_max)))
        sol = palc.natural_corrector(prob, st.z_pred, ls, p_min=cp.p_min, p_max=cp.p_max, tol=cp.tol,
                                     max_iterations=cp.max_iterations, normN=normC)
    else:
        sol = palc.newton_palc(prob, st.z, st.tau, st.z_pred, st.ds, cp.theta, bls, tol=cp.tol,
                               max_iterations=cp.max_iterations, p_min=cp.p_min, p_max=cp.p_max, normN=normC)
    st.converged, st.itnewton, st.itlinear, st.residuals = sol["converged"], sol["itnewton"], sol["itlineartot"], sol["residuals"]
    if st.converged:
        st.z_old = (st.z[0].copy(), st.z[1])
        st.z = (sol["u"], sol["p"])
        if eig is not None:
            _eigen(prob, st, eig, cp)
        st.step += 1
    if not st.stop and st.stepsizecontrol:                                # step_size_control!, Contbase.jl:69-75
        st.ds, stop = palc.step_size_control(st.ds, st.converged, st.itnewton, a=cp.a, Nmax=cp.max_iterations,
                                             dsmin=cp.dsmin, dsmax=cp.dsmax)
        st.stop = st.stop or stop
    if st.converged:                                                      # getpredictor!, Palc.jl:133-146
        if cp.tangent == "secant":
            st.tau = palc.secant_tangent(st.z, st.z_old, st.ds, cp.theta)
        else:
            tu, tp, _ = palc.bordered_tangent(prob, st.z, st.tau, cp.theta, bls)
            st.tau = (tu, tp)
    st.z_pred = palc.add_tangent(st.z, st.tau, st.ds)
    return st


def _rightmost_abs_real(vals):
    v = np.asarray(vals)
    v = v[~np.isnan(v.real)]
    return float(np.min(np.abs(v.real))) if v.size else np.inf


def locate_bifurcation(prob, S: ContState, step_fn, cp: ContPar):
    """locate_bifurcation!(iter, state) (Bifurcations.jl:159-349): bisection with the continuation iterator itself.  `S` is
    the state right after the step that changed n_unstable; it is updated in place to sit right after (status "guess" /
    "converged") or right before ("guessL") the bifurcation point.  Returns (status, (p_lo, p_hi))."""
    n2, n1 = S.n_unstable
    if n1 == -1 or n2 == -1 or abs(S.ds) < cp.dsmin:
        return "none", (0.0, 0.0)
    after, state, before = S.copy(), S.copy(), S.copy()
    before.n_unstable = (before.n_unstable[1], before.n_unstable[0])
    before.n_imag = (before.n_imag[1], before.n_imag[0])
    before.z_old, before.z = (before.z_old[0], before.z[1]), (before.z[0], before.z_old[1])      # swap the p components
    state.ds *= -1
    state.step = 0
    state.stepsizecontrol = False
    nxt = state
    nunst = [n2]
    nimag = [state.n_imag[0]]
    interval = sorted((state.z[1], state.z_old[1]))
    ind = 0 if interval[0] == state.z[1] else 1
    n_inv = 0
    while True:
        if not state.converged:
            break
        if nxt is None:
            break
        state = nxt
        nunst.append(state.n_unstable[0])
        nimag.append(state.n_imag[0])
        if nunst[-1] == nunst[-2]:
            state.ds /= 2
        else:
            state.ds /= -2
            n_inv += 1
            ind = 1 - ind
        state.z_pred = palc.add_tangent(state.z, state.tau, state.ds)     # update_predictor!, Palc.jl:148-151
        if n_inv % 2 == 0:
            after = state.copy()
        else:
            before = state.copy()
        if state.step > 0:
            interval[ind] = state.z[1]
        located = _rightmost_abs_real(state.eigvals) < cp.tol_bisection_eigenvalue
        if not (abs(state.ds) >= cp.dsmin_bisection and state.step < cp.max_bisection_steps and n_inv < cp.n_inversion
                and not located):
            break
        nxt = step_fn(state)
    if n_inv % 2 == 0:
        status = "converged" if n_inv >= cp.n_inversion else "guess"
        src = state
        S.n_unstable = (state.n_unstable[0], before.n_unstable[0])
        S.n_imag = (state.n_imag[0], before.n_imag[0])
        interval = (state.z[1], before.z[1])
    else:
        status = "guessL"
        src = after
        S.n_unstable = (after.n_unstable[0], state.n_unstable[0])
        S.n_imag = (after.n_imag[0], state.n_imag[0])
        interval = (state.z[1], after.z[1])
    for name in ("z_old", "z_pred", "z", "tau"):
        u, p = getattr(src, name)
        setattr(S, name, (u.copy(), p))
    S.eigvals = None if src.eigvals is None else src.eigvals.copy()
    S.z_pred = palc.add_tangent(S.z, S.tau, S.ds)                         # update_predictor!(_state, iter)
    return status, tuple(sorted(interval))


def bifurcation_type(S: ContState):
    """_get_bifurcation_type (Bifurcations.jl:80-150), the codim-1 classification by eigenvalue counts."""
    dn = abs(S.n_unstable[0] - S.n_unstable[1])
    di = abs(S.n_imag[0] - S.n_imag[1])
    if dn == 1:
        return "bp" if di == 0 else ("hopf" if di == 1 else "nd")
    if dn == 2:
        return "hopf" if di == 2 else "nd"
    return "nd" if dn > 2 else "none"


def continuation(prob, x0, p0, *, ls, bls, eig, cp: ContPar, normC=palc.norm2, detect_bifurcation_level=3):
    """continuation! (Continuation.jl:349-456, 505-600) with eigenvalues every step and bisection of detected bifurcations.
    Returns dict(param, n_unstable, ds, itnewton, specialpoint=[dict(type, status, interval, param, step, n_unstable, x)]).
    Pinned to reference-held known answers (special-point parameters and bisection intervals asserted by the reference's own
    tests to its default isapprox tolerance) in tests/test_reference_known_answers.py."""
    kw = dict(tol=cp.tol, max_iterations=cp.max_iterations, normN=normC)
    s0 = palc.newton(prob, x0, p0, ls, **kw)
    assert s0["converged"]
    p1 = p0 + cp.ds / cp.eta
    s1 = palc.newton(prob, s0["u"], p1, ls, **kw)
    assert s1["converged"]
    z0, z1 = (s0["u"], p0), (s1["u"], p1)
    tau = palc.secant_tangent(z1, z0, cp.ds, cp.theta)
    st = ContState(z=(z0[0].copy(), p0), z_old=(z0[0].copy(), p0), tau=tau, z_pred=palc.add_tangent(z0, tau, cp.ds), ds=cp.ds)
    _eigen(prob, st, eig, cp)
    st.n_unstable = (st.n_unstable[0], -1)
    st.n_imag = (st.n_imag[0], -1)
    out = dict(param=[st.z[1]], n_unstable=[st.n_unstable[0]], ds=[st.ds], itnewton=[s0["itnewton"]], specialpoint=[])
    step_fn = lambda s: iterate(prob, s, ls=ls, bls=bls, eig=eig, cp=cp, normC=normC)
    nxt = step_fn(st)
    while nxt is not None:
        st = nxt
        if st.converged and st.step <= cp.max_steps and st.step > 0:
            if detect_bifurcation(st):
                status, interval = "guess", tuple(sorted((st.z_old[1], st.z[1])))
                on_boundary = st.z[1] in (cp.p_min, cp.p_max)
                if detect_bifurcation_level > 2 and not on_boundary:
                    status, interval = locate_bifurcation(prob, st, step_fn, cp)
                if detect_bifurcation(st):
                    tp = bifurcation_type(st)
                    if tp != "none":
                        out["specialpoint"].append(dict(type=tp, status=status, interval=interval, param=st.z[1], step=st.step,
                                                        n_unstable=st.n_unstable, x=st.z[0].copy()))
            out["param"].append(st.z[1]); out["n_unstable"].append(st.n_unstable[0]); out["ds"].append(st.ds)
            out["itnewton"].append(st.itnewton)
        nxt = step_fn(st)
    return out
