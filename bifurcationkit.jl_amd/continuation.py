"""Host-side continuation engine restated minimally around the plugin surface: the CALLER of the hot path.

In a Julia deployment this file is not needed -- BifurcationKit's own ``continuation`` / ``newton`` call the
HIP plugins through ``julia/BifurcationKitHIP.jl``.  It exists so that the same call sequence can be driven
(and parity-tested) from Python.  It is written against duck-typed vectors (``HipVec`` / ``BorderedArray``
or any type with copy/zerovector/scale_/add_/inner/norm/norminf/copyto_/__len__) and plugin callables:

  ls(J, rhs, a0=0, a1=1) -> (x, ok, it)                   AbstractLinearSolver   src/LinearSolver.jl:12
  bls(J, dR, dzu, dzp, R, n, xiu, xip, shift=, dotp=)      AbstractBorderedLinearSolver  src/LinearBorderSolver.jl:3-6
  eig(J, nev) -> (vals, vecs, ok, it)                      AbstractEigenSolver    src/EigSolver.jl:4-12

Restated pieces (reference file:line):
  newton                _newton                              src/Newton.jl:66-114
  DotTheta / arc_length_eq / newton_palc                     src/continuation/Palc.jl:23-56, 187-305
  solve_bls_palc                                             src/LinearBorderSolver.jl:16-36
  Secant / Bordered tangents, addtangent!                    src/continuation/Tangents.jl:8-104
  _step_size_control!                                        src/continuation/Contbase.jl:77-102
  compute_eigenvalues / is_stable / detect_bifurcation       src/Utils.jl:67-104, src/Bifurcations.jl:5-28
  iterate (first two points, one step)                       src/Continuation.jl:349-456, 458-504
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field, replace

import numpy as np

from .hip import BorderedArray


def norminf(x):
    return x.norminf()


def norm2(x):
    return x.norm()


@dataclass
class NewtonPar:
    """src/Newton.jl:17-33."""
    tol: float = 1e-12
    max_iterations: int = 25
    verbose: bool = False
    linsolver: object = None
    eigsolver: object = None
    linesearch: bool = False         # :27, used by newton_palc only (Palc.jl:254-281)
    alpha: float = 1.0               # :29
    alphamin: float = 1e-3           # :31


def cb_default(state, **kw):         # src/Newton.jl:151
    return True


@dataclass
class cbMaxNorm:
    """src/Newton.jl:156-159: reject iterates whose residual exceeds ``maxres``.  The native correctors evaluate it inside
    the library (bk_newton_opts.max_residual); any other callable becomes a host callback (bk_newton_callback)."""
    maxres: float

    def __call__(self, state, **kw):
        return state["residual"] < self.maxres


@dataclass
class ContinuationPar:
    """src/ContParameters.jl:44-100 (the fields the PALC path reads)."""
    dsmin: float = 1e-4
    dsmax: float = 1e-1
    ds: float = 1e-2
    a: float = 0.5
    p_min: float = -1.0
    p_max: float = 1.0
    max_steps: int = 400
    eta: float = 150.0
    nev: int = 3
    tol_stability: float = 1e-10
    detect_bifurcation: int = 3
    detect_fold: bool = True                 # src/ContParameters.jl:75 (acts only when detect_bifurcation < 2)
    dsmin_bisection: float = 1e-16           # bisection of detected bifurcations (detect_bifurcation = 3),
    n_inversion: int = 2                     # src/ContParameters.jl:78-81
    max_bisection_steps: int = 25
    tol_bisection_eigenvalue: float = 1e-16
    save_sol_every_step: int = 0             # src/ContParameters.jl:65 -- every how many steps br.sol keeps the solution
    save_to_file: bool = False               # :64 -- checkpoint every accepted step (save_to_file, ext/JLD2Ext/save.jl:8-30)
    newton_options: NewtonPar = field(default_factory=NewtonPar)

    def __post_init__(self):
        """The consistency checks of the reference's constructor (src/ContParameters.jl:89-99)."""
        if not self.tol_stability >= 0:
            raise ValueError("You must provide a positive tolerance for tol_stability")
        if not (self.dsmax >= abs(self.ds) >= self.dsmin >= 0):
            raise ValueError(f"You must provide a valid interval (ordered) for ds. You passed {self.dsmax} >= {abs(self.ds)} >= {self.dsmin}")
        if not (abs(self.ds) >= self.dsmin_bisection >= 0):
            raise ValueError("You must provide a valid interval for `ds` and `dsmin_bisection`")
        if not self.p_max >= self.p_min:
            raise ValueError("You must provide a valid interval [p_min, p_max]")
        if self.n_inversion % 2 != 0:
            raise ValueError("The option `n_inversion` number must be even")
        if not 0 <= self.detect_bifurcation <= 3:
            raise ValueError("The option `detect_bifurcation` must belong to {0,1,2,3}")
        if not self.tol_bisection_eigenvalue >= 0:
            raise ValueError("The option `tol_bisection_eigenvalue` must be positive")


@dataclass
class PALC:
    """src/continuation/Palc.jl:70-84."""
    tangent: str = "secant"          # Secant() | Bordered()
    theta: float = 0.5
    bls: object = None

    def update(self, contparams):    # Palc.jl:100-110: a bls without solver inherits the Newton linsolver
        if self.bls is not None and getattr(self.bls, "solver", None) is None:
            return replace(self, bls=self.bls.update_bls(contparams.newton_options.linsolver))
        return self


@dataclass
class NonLinearSolution:
    """src/Newton.jl:49-63."""
    u: object
    residuals: list
    converged: bool
    itnewton: int
    itlineartot: int


def newton(prob, x0, p, options: NewtonPar, normN=norm2, callback=cb_default) -> NonLinearSolution:
    """_newton, src/Newton.jl:66-114."""
    x = x0.copy()
    fx = prob.residual(x, p)
    res = normN(fx)
    residuals = [res]
    step, itlin = 0, 0
    compute = callback(dict(x=x, fx=fx, residual=res, step=step, residuals=residuals), fromNewton=True)
    while step < options.max_iterations and res > options.tol and compute:
        J = prob.jacobian(x, p)
        u, cv, it = options.linsolver(J, fx)
        itlin += int(np.sum(it))
        x.add_(u, -1.0)                       # minus!!(x, u)
        fx = prob.residual(x, p)
        res = normN(fx)
        residuals.append(res)
        step += 1
        if options.verbose:
            print(f"  newton {step:3d}  res = {res:.4e}  itlinear = {it}")
        compute = callback(dict(x=x, fx=fx, residual=res, step=step, itlinear=it, residuals=residuals), fromNewton=True)
    flag = residuals[-1] < options.tol and bool(callback(dict(x=x, fx=fx, residual=res, step=step, residuals=residuals),
                                                         fromNewton=True))
    return NonLinearSolution(x, residuals, flag, step, itlin)


def dot_theta(u1, u2, p1, p2, theta):
    """DotTheta with NormalisedDot, Palc.jl:1-6, 35."""
    return u1.inner(u2) / len(u1) * theta + p1 * p2 * (1.0 - theta)


def solve_bls_palc(bls, theta, tau, J, dR, R, n, shift=None):
    """src/LinearBorderSolver.jl:16-36: xi_u = theta, xi_p = 1 - theta, dotp = dot/length."""
    N = len(R)
    return bls(J, dR, tau.u, tau.p, R, n, theta, 1.0 - theta, shift=shift,
               dotp=lambda x, y: x.inner(y) / N, dotscale=1.0 / N)


def _dFdp(prob, x, p, eps, res_f=None):
    """dFdp = (F(x, p + eps) - F(x, p)) / eps (Palc.jl:239-240, Tangents.jl:77-82).  Device problems evaluate the
    quotient cancellation-free in one pass (bk_residual_dparam), like the native corrector."""
    if hasattr(prob, "residual_dparam"):
        return prob.residual_dparam(x, p, eps)
    dFdp = prob.residual(x, p + eps)
    dFdp.add_(res_f if res_f is not None else prob.residual(x, p), -1.0)
    return dFdp.scale_(1.0 / eps)


def newton_palc(prob, z0, tau0, z_pred, ds, theta, bls, options: NewtonPar, p_min=-math.inf, p_max=math.inf,
                normN=norm2, callback=cb_default) -> NonLinearSolution:
    """newton_palc, Palc.jl:187-305 (plain update :282-285, line search :254-281, callback :235,294-297)."""
    eps = prob.delta

    def Nfun(u, p):                            # arc_length_eq, Palc.jl:44-56 (two dots, as written)
        return (dot_theta(u, tau0.u, p - z0.p, tau0.p, theta) - ds) - (dot_theta(z0.u, tau0.u, p - z0.p, 0.0, theta))

    x = z_pred.u.copy()
    p = float(z_pred.p)
    res_f = prob.residual(x, p)
    res_n = Nfun(x, p)
    res = max(normN(res_f), abs(res_n))
    residuals = [res]
    step, itlin = 0, 0
    alpha, line_step = options.alpha, True
    compute = callback(dict(x=x, res_f=res_f, residual=res, step=step, z0=z0, p=p, residuals=residuals), fromNewton=False)
    while step < options.max_iterations and res > options.tol and line_step and compute:
        dFdp = _dFdp(prob, x, p, eps, res_f)
        J = prob.jacobian(x, p)
        u, up, flag, it = solve_bls_palc(bls, theta, tau0, J, dFdp, res_f, res_n)
        itlin += int(np.sum(it))
        if options.linesearch:
            line_step = False
            while not line_step and alpha > options.alphamin:
                x_pred = x.copy().add_(u, -alpha)
                p_pred = p - alpha * up
                res_f = prob.residual(x_pred, p_pred)
                res_n = Nfun(x_pred, p_pred)
                res = max(normN(res_f), abs(res_n))
                if res < residuals[-1]:
                    if res < residuals[-1] / 4 and alpha < 1:
                        alpha *= 2
                    line_step = True
                    x.copyto_(x_pred)
                    p = min(max(p_pred, p_min), p_max)
                else:
                    alpha /= 2
            alpha = options.alpha
        else:
            x.add_(u, -1.0)
            p = min(max(p - up, p_min), p_max)
            res_f = prob.residual(x, p)
            res_n = Nfun(x, p)
            res = max(normN(res_f), abs(res_n))
        residuals.append(res)
        step += 1
        if options.verbose:
            print(f"  newton_palc {step:3d}  res = {res:.4e}  itlinear = {it}")
        compute = callback(dict(x=x, res_f=res_f, residual=res, step=step, itlinear=it, z0=z0, p=p, residuals=residuals),
                           fromNewton=False)
    flag = residuals[-1] < options.tol and bool(callback(dict(x=x, res_f=res_f, residual=res, step=step, p=p,
                                                              residuals=residuals), fromNewton=False))
    return NonLinearSolution(BorderedArray(x, p), residuals, flag, step, itlin)


def secant_tangent(z1, z0, ds, theta):
    """_secant_tangent!, Tangents.jl:28-42."""
    tau = z1.copy()
    tau.add_(z0, -1.0)
    a = math.copysign(1.0, ds) / math.sqrt(dot_theta(tau.u, tau.u, tau.p, tau.p, theta))
    return tau.scale_(a)


def bordered_tangent(prob, z, tau, theta, bls):
    """gettangent!(::Bordered), Tangents.jl:71-104."""
    eps = prob.delta
    dFdl = _dFdp(prob, z.u, z.p, eps)
    J = prob.jacobian(z.u, z.p)
    tu, tp, flag, _ = solve_bls_palc(bls, theta, tau, J, dFdl, z.u.zerovector(), 1.0)
    a = 1.0 / math.sqrt(dot_theta(tu, tu, tp, tp, theta))
    a *= math.copysign(1.0, dot_theta(tau.u, tu, tau.p, tp, theta))
    return BorderedArray(tu, tp).scale_(a), flag


def step_size_control(ds, converged, itnewton, cp: ContinuationPar):
    """_step_size_control!, Contbase.jl:77-102 -> (dsnew, stop)."""
    if not converged:
        if abs(ds) <= cp.dsmin:
            return ds, True
        dsnew = math.copysign(max(abs(ds) / 2.0, cp.dsmin), ds)
    else:
        Nmax = cp.newton_options.max_iterations
        factor = (Nmax - itnewton) / Nmax
        dsnew = ds * (1.0 + cp.a * factor ** 2)
    dsnew = math.copysign(min(max(abs(dsnew), cp.dsmin), cp.dsmax), dsnew)     # clamp_ds
    return dsnew, False


def is_stable(eigvalues, tol_stability):
    """Bifurcations.jl:5-19 -> (n_unstable, n_imag)."""
    ev = np.asarray(eigvalues)
    return (int(np.sum(ev.real > tol_stability)),
            int(np.sum((np.abs(ev.imag) > tol_stability) & (ev.real > tol_stability))))


@dataclass
class ContResult:
    """The per-step record of src/Continuation.jl:259-272 plus detected stability changes."""
    param: list = field(default_factory=list)
    itnewton: list = field(default_factory=list)
    itlinear: list = field(default_factory=list)
    ds: list = field(default_factory=list)
    n_unstable: list = field(default_factory=list)
    n_imag: list = field(default_factory=list)
    eig: list = field(default_factory=list)
    residuals: list = field(default_factory=list)
    specialpoint: list = field(default_factory=list)
    sol: list = field(default_factory=list)


def detect_fold(p1, p2, p3):
    """src/Bifurcations.jl:32."""
    return (p3 - p2) * (p2 - p1) < 0


def locate_fold(br, cp, p_state):
    """locate_fold!(contres, iter, state), src/Bifurcations.jl:35-69: a fold is flagged by the loss of monotony of the
    parameter along the last three points ALREADY in the branch record -- the reference calls it before save!
    (src/Continuation.jl:524 vs :579), so the point just computed is not among them and only supplies ``param =
    getp(state)``; only when bifurcations are not detected through eigenvalues (detect_bifurcation < 2, "to avoid
    duplicates").  Call it before the new point is recorded."""
    n = len(br.param)
    if cp.detect_fold and cp.detect_bifurcation < 2 and n > 2 and detect_fold(br.param[-3], br.param[-2], br.param[-1]):
        # Julia's 1-based branch[n_br - 1] is the MIDDLE point of the three: 0-based index n - 2; `step` = n_br - 1 is the
        # (0-based) step number of the last recorded point
        br.specialpoint.append(dict(type="fold", step=n - 1, idx=n - 2, param=float(p_state), status="guess",
                                    interval=(br.param[-2], br.param[-2])))
        return True
    return False


def mod_counter(step, every):
    """src/Utils.jl:183-188."""
    if step == 0 or every == 0:
        return False
    return True if every == 1 else step % every == 0


def _to_host(x):
    """Download a state vector (HipVec -> NumPy; the checkpoint is the one place where the state leaves the device)."""
    if hasattr(x, "numpy"):
        return np.asarray(x.numpy())
    return np.asarray(getattr(x, "a", x), dtype=float)


def _rank_suffix(x):
    ctx = getattr(x, "ctx", None)
    nr = getattr(ctx, "nranks", 1) if ctx is not None else 1
    return f"-rank{ctx.rank}of{nr}" if nr > 1 else ""


def save_to_file(filename, ds, sol, p, i, br):
    """save_to_file(iter, sol, p, i, br), ext/JLD2Ext/save.jl:8-30: the solution of step ``i`` goes to the group
    ``sol-fw-i`` / ``sol-bw-i`` (forward / backward branch by the sign of ds) with its parameter, and the branch record is
    rewritten next to it.  JLD2 groups become one ``.npz`` per step, ``<filename>-sol-<fw|bw>-<i>.npz`` (a rank suffix when
    the state is a z-slab of a distributed run), the branch ``<filename>-branch.json``."""
    import json
    fd = "fw" if ds >= 0 else "bw"
    np.savez(f"{filename}-sol-{fd}-{i}{_rank_suffix(sol)}.npz", sol=_to_host(sol), param=float(p), step=int(i))
    rec = dict(param=[float(x) for x in br.param], itnewton=list(br.itnewton), itlinear=list(br.itlinear),
               ds=[float(x) for x in br.ds], n_unstable=list(br.n_unstable), n_imag=list(br.n_imag),
               eig=[None if v is None else [[float(z.real), float(z.imag)] for z in np.atleast_1d(v)] for v in br.eig],
               specialpoint=[{k: (list(v) if isinstance(v, tuple) else v) for k, v in sp.items()} for sp in br.specialpoint])
    # the branch record is identical on every rank of a distributed run: rank 0 writes it.  The reference opens the file
    # with "a+" and keeps the other direction's branch (ext/JLD2Ext/save.jl:22-28): merge with what is there, and replace
    # the file atomically so that a reader (or a crash) never sees a truncated record
    ctx = getattr(sol, "ctx", None)
    if ctx is not None and getattr(ctx, "nranks", 1) > 1 and ctx.rank != 0:
        return
    import os
    path = f"{filename}-branch.json"
    doc = {}
    if os.path.exists(path):
        try:
            with open(path) as f:
                doc = json.load(f)
        except (OSError, ValueError):
            doc = {}
    doc["branch" + fd] = rec
    tmp = f"{path}.tmp{os.getpid()}"
    with open(tmp, "w") as f:
        json.dump(doc, f)
    os.replace(tmp, path)


def load_solution(filename, i, fd="fw", rank_suffix=""):
    """Read a checkpoint written by :func:`save_to_file`: (solution as a NumPy array, parameter)."""
    with np.load(f"{filename}-sol-{fd}-{i}{rank_suffix}.npz") as z:
        return np.array(z["sol"]), float(z["param"])


def load_branch(filename, fd=None):
    """The branch record written by :func:`save_to_file`; ``fd`` = "fw" / "bw" picks a direction (default: the only / first one)."""
    import json
    with open(f"{filename}-branch.json") as f:
        d = json.load(f)
    return d["branch" + fd] if fd else next(iter(d.values()))


def continuation(prob, x0, p0, alg: PALC, cp: ContinuationPar, normC=norm2, verbosity=0, save_sol=False,
                 corrector=newton_palc, callback_newton=cb_default, filename=None) -> ContResult:
    """PALC branch.  Continuation.jl:349-456 (two Newton solves + secant tangent), :458-504 (one step:
    corrector!, compute_eigenvalues!, step_size_control!, getpredictor!).  ``br.sol`` keeps (x, p, step) every
    ``cp.save_sol_every_step`` steps and at the last point (save!, :280-292); ``cp.save_to_file`` writes a checkpoint after
    every accepted step (:579) under ``filename``."""
    alg = alg.update(cp)
    if cp.save_to_file and filename is None:
        import datetime
        filename = "branch-" + datetime.datetime.now().isoformat()      # ContIterable default, Continuation.jl:46
    nopt = cp.newton_options
    eig = nopt.eigsolver if cp.detect_bifurcation > 0 else None
    sol0 = newton(prob, x0, p0, nopt, normC, callback_newton)
    if not sol0.converged:
        raise RuntimeError("Newton failed to converge for the initial guess on the branch")
    p1 = p0 + cp.ds / cp.eta
    sol1 = newton(prob, sol0.u, p1, nopt, normC, callback_newton)
    if not sol1.converged:
        raise RuntimeError("Newton failed to converge. Required for the computation of the initial tangent")
    z0 = BorderedArray(sol0.u, p0)
    z1 = BorderedArray(sol1.u, p1)
    br = ContResult()
    n_unst, n_imag = -1, -1

    def eigen(z, n_prev):
        nev_ = max(n_prev + 5, cp.nev)                                  # n = state.n_unstable[2], Utils.jl:78-79
        vals, _, cv, it = eig(prob.jacobian(z.u, z.p), nev_)
        nu, ni = is_stable(vals, cp.tol_stability)
        return vals, nu, ni

    def record(z, sol, ds, vals):
        br.param.append(z.p); br.itnewton.append(sol.itnewton); br.itlinear.append(sol.itlineartot)
        br.ds.append(ds); br.n_unstable.append(n_unst); br.n_imag.append(n_imag)
        br.residuals.append(list(sol.residuals)); br.eig.append(vals)
        if save_sol:
            br.sol.append(z.u.copy())
        elif cp.save_sol_every_step > 0:
            finished = not (step < cp.max_steps and (cp.p_min < z.p < cp.p_max or step == 0))
            if mod_counter(step, cp.save_sol_every_step) or finished:
                br.sol.append(dict(x=z.u.copy(), p=z.p, step=step))

    vals = None
    n_unst_prev = -1
    if eig is not None:
        vals, n_unst, n_imag = eigen(z0, -1)
    ds = cp.ds
    tau = secant_tangent(z1, z0, ds, alg.theta)                         # initialize!, Palc.jl:112-123
    z = z0.copy()
    z_old = z0.copy()
    z_pred = z.copy().add_(tau, ds)                                     # addtangent!
    step = 0
    record(z, sol0, ds, vals)
    while step < cp.max_steps and (cp.p_min < z.p < cp.p_max or step == 0):        # done, Continuation.jl:254-257
        if z_pred.p <= cp.p_min or z_pred.p >= cp.p_max:
            # corrector!(::PALC) hands over to Natural at the clamped parameter (Palc.jl:157-160, Natural.jl:38-58)
            z_pred.p = min(max(z_pred.p, cp.p_min), cp.p_max)
            sn = newton(prob, z_pred.u, z_pred.p, nopt, normC, callback_newton)
            sol = NonLinearSolution(BorderedArray(sn.u, z_pred.p), sn.residuals, sn.converged, sn.itnewton, sn.itlineartot)
        elif corrector is newton_palc:
            sol = corrector(prob, z, tau, z_pred, ds, alg.theta, alg.bls, nopt, cp.p_min, cp.p_max, normC, callback_newton)
        else:
            sol = corrector(prob, z, tau, z_pred, ds, alg.theta, alg.bls, nopt, cp.p_min, cp.p_max, normC)
        conv = sol.converged
        if verbosity:
            print(f"step {step:3d} ds={ds:+.3e} p={z.p:+.6f} -> {sol.u.p:+.6f} conv={conv} "
                  f"itnewton={sol.itnewton} itlinear={sol.itlineartot}")
        if conv:
            z_old.copyto_(z)
            z.copyto_(sol.u)
            prev_unst = n_unst
            if eig is not None:
                vals, n_unst, n_imag = eigen(z, n_unst_prev)
                n_unst_prev = prev_unst
                if cp.detect_bifurcation > 1 and prev_unst != -1 and n_unst != prev_unst:   # Continuation.jl:530, Bifurcations.jl:22-28
                    br.specialpoint.append(dict(step=step + 1, param=z.p, n_unstable=(prev_unst, n_unst)))
            step += 1
            locate_fold(br, cp, z.p)
            record(z, sol, ds, vals)
            if cp.save_to_file:
                save_to_file(filename, cp.ds, z.u, z.p, step, br)
        ds, stop = step_size_control(ds, conv, sol.itnewton, cp)
        if stop:
            break
        if conv:                                                         # Palc.jl:140-143
            if alg.tangent == "secant":
                tau = secant_tangent(z, z_old, ds, alg.theta)
            else:
                tau, _ = bordered_tangent(prob, z, tau, alg.theta, alg.bls)
        z_pred = z.copy().add_(tau, ds)
    return br


def continuation_native(prob, x0, p0, alg: PALC, cp: ContinuationPar, normC=norm2, verbosity=0, save_sol=False,
                        bisection=False, finalise_solution=None, callback_newton=None, on_init=None, filename=None) -> ContResult:
    """The same branch with every step issued as ONE library call (``bk_cont_step``: corrector, eigenvalues, step-size
    control, tangent and predictor -- the body of ``iterate``, src/Continuation.jl:458-504) and the two initial Newton
    solves as ``bk_newton``.  Needs the native solver types (GMRES* + BorderingBLS + ShiftInvert); ``normC`` must be
    ``norm2`` or ``norminf``.  Returns the same record as :func:`continuation`.  ``bisection=True`` (the reference's
    ``detect_bifurcation = 3``): every detected change of stability is located by ``bk_cont_locate_bifurcation``
    (locate_bifurcation!, src/Bifurcations.jl:159-349) and the special point carries its interval, status and type.
    ``finalise_solution(state, r) -> bool`` (the reference's hook of the same name, src/Continuation.jl:296-310) is called
    after every accepted step with a ``get()`` accessor of the device state; returning False stops the run.
    ``cp.save_sol_every_step`` / ``cp.save_to_file`` / ``filename`` as in :func:`continuation`."""
    import ctypes as C

    from . import _lib as L
    from . import hip

    alg = alg.update(cp)
    nopt = cp.newton_options
    ls, bls = nopt.linsolver, alg.bls
    if normC not in (norm2, norminf):
        raise TypeError("continuation_native: normC must be norm2 or norminf")
    inf = normC is norminf
    ctx = prob.ctx
    eig = nopt.eigsolver if cp.detect_bifurcation > 0 else None
    s0 = hip.newton_native(prob, x0, p0, ls, nopt.tol, nopt.max_iterations, inf, callback=callback_newton)
    if not s0["converged"]:
        raise RuntimeError("Newton failed to converge for the initial guess on the branch")
    p1 = p0 + cp.ds / cp.eta
    s1 = hip.newton_native(prob, s0["u"], p1, ls, nopt.tol, nopt.max_iterations, inf, callback=callback_newton)
    if not s1["converged"]:
        raise RuntimeError("Newton failed to converge. Required for the computation of the initial tangent")
    big = 1.7e308
    co = L.ContOpts(cp.ds, cp.dsmin, cp.dsmax, cp.a, alg.theta, max(cp.p_min, -big), min(cp.p_max, big),
                    0 if alg.tangent == "secant" else 1, 1 if eig is not None else 0, cp.nev, cp.tol_stability)
    no = hip.newton_opts(nopt.tol, nopt.max_iterations, inf, nopt.linesearch, nopt.alpha, nopt.alphamin, callback_newton)
    bo = L.BorderingOpts(bls.tol, 1 if bls.check_precision else 0, bls.k)
    lo = bls.solver._opts()
    eo = elo = None
    epl = None
    if eig is not None:
        kd = eig.krylovdim if eig.krylovdim is not None else 0
        eo = L.EigOpts(float(eig.sigma), int(min(kd, 63)), int(eig.maxiter), float(eig.tol), 1 if eig.hermitian else 0,
                       int(eig.seed))
        elo, epl = eig.ls._opts(), eig.ls._pl()
    pv = prob._pvec(p0)
    arr = (C.c_double * len(pv))(*pv)
    h = C.c_void_p()
    r = L.ContStepResult()
    ctx.check(ctx.lib.bk_cont_create(
        ctx.h, prob.h, arr, len(pv), prob.ipar, hip._ptr(s0["u"].t), float(p0), hip._ptr(s1["u"].t), float(p1),
        C.byref(co), C.byref(no), C.byref(bo), C.byref(lo), bls.solver._pl(),
        C.byref(eo) if eo is not None else None, C.byref(elo) if elo is not None else None, epl, C.byref(r),
        C.byref(h)), "bk_cont_create")
    br = ContResult()
    if cp.save_to_file and filename is None:
        import datetime
        filename = "branch-" + datetime.datetime.now().isoformat()

    def state_vec():
        u = s0["u"].similar()
        ctx.check(ctx.lib.bk_cont_get(h, hip._ptr(u.t), None, None, None, None), "bk_cont_get")
        return u

    def vals_of(r):
        return np.array([complex(r.vals_re[i], r.vals_im[i]) for i in range(r.nvals)]) if eig is not None else None

    def record(r, itnewton, itlinear, residuals):
        br.param.append(r.p); br.itnewton.append(itnewton); br.itlinear.append(itlinear)
        br.ds.append(r.ds_used); br.n_unstable.append(r.n_unstable); br.n_imag.append(r.n_imag)
        br.residuals.append(residuals); br.eig.append(vals_of(r))
        if save_sol:
            br.sol.append(state_vec())
        elif cp.save_sol_every_step > 0:
            finished = not (step < cp.max_steps and (cp.p_min < r.p < cp.p_max or step == 0)) or bool(r.stop)
            if mod_counter(step, cp.save_sol_every_step) or finished:
                br.sol.append(dict(x=state_vec(), p=r.p, step=step))

    step = 0
    try:
        record(r, s0["itnewton"], s0["itlineartot"], list(s0["residuals"]))
        if on_init is not None:
            on_init(r)
        while step < cp.max_steps:
            prev_unst = r.n_unstable
            ctx.check(ctx.lib.bk_cont_step(h, C.byref(r)), "bk_cont_step")
            if r.stop == 2:
                break
            if verbosity:
                print(f"step {step:3d} ds={r.ds_used:+.3e} -> p={r.p:+.6f} conv={bool(r.converged)} "
                      f"itnewton={r.itnewton} itlinear={r.itlinear}")
            if r.converged:
                if r.bifurcation and cp.detect_bifurcation > 1:                # Continuation.jl:530
                    sp = dict(step=step + 1, param=r.p, n_unstable=(prev_unst, r.n_unstable))
                    on_boundary = r.p in (cp.p_min, cp.p_max)
                    keep = True
                    if bisection and cp.detect_bifurcation > 2 and not on_boundary:   # Continuation.jl:537-541
                        bo = L.BisectionOpts(cp.dsmin_bisection, cp.n_inversion, cp.max_bisection_steps,
                                             cp.tol_bisection_eigenvalue, cp.max_steps)
                        res = L.BisectionResult()
                        ctx.check(ctx.lib.bk_cont_locate_bifurcation(h, C.byref(bo), C.byref(res)), "bk_cont_locate_bifurcation")
                        if res.status != 0:                                    # status none: the state was left untouched
                            sp.update(status=("none", "guess", "converged", "guessL")[res.status],
                                      type=("none", "bp", "hopf", "nd")[res.type], interval=(res.interval[0], res.interval[1]),
                                      param=res.p, n_unstable=(res.n_unstable[1], res.n_unstable[0]), bisection_steps=res.steps)
                            # the record takes the bisected state: p, counts AND its eigenvalues (_state.eigvals = state.eigvals)
                            r.p, r.n_unstable, r.n_imag = res.p, res.n_unstable[0], res.n_imag[0]
                            r.nvals = res.nvals
                            for i in range(res.nvals):
                                r.vals_re[i], r.vals_im[i] = res.vals_re[i], res.vals_im[i]
                            # "double-check that the bisection did not remove the bifurcation point" (Continuation.jl:543-546)
                            keep = res.n_unstable[0] != res.n_unstable[1] and res.type != 0
                    if keep:
                        br.specialpoint.append(sp)
                step += 1
                locate_fold(br, cp, r.p)
                record(r, r.itnewton, r.itlinear, [r.residuals[i] for i in range(r.itnewton + 1)])
                if cp.save_to_file:
                    save_to_file(filename, cp.ds, state_vec(), r.p, step, br)
                if finalise_solution is not None:
                    def get():
                        u, tu = s0["u"].similar(), s0["u"].similar()
                        pp, tp, dd = C.c_double(), C.c_double(), C.c_double()
                        ctx.check(ctx.lib.bk_cont_get(h, hip._ptr(u.t), C.byref(pp), hip._ptr(tu.t), C.byref(tp), C.byref(dd)),
                                  "bk_cont_get")
                        return dict(z=BorderedArray(u, pp.value), tau=BorderedArray(tu, tp.value), ds=dd.value)
                    if finalise_solution(get, r) is False:
                        break
            if r.stop:
                break
    finally:
        ctx.lib.bk_cont_destroy(h)
    return br
