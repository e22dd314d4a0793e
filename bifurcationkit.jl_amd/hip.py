"""Host-side mirror of BifurcationKit's plugin surface, backed by ``libbkhip.so``.

Same names, argument meaning and error behaviour as the reference types they stand in for
(paths relative to the BifurcationKit.jl checkout):

  HipVec              state vector with the VectorInterface subset the PALC path uses
                      (src/BorderedArrays.jl:86-217; checklist in SURVEY.md section 8b)
  GMRESKrylovKit      src/LinearSolver.jl:223-291      ``ls(J, rhs; a0, a1) -> (x, success, numops)``
  GMRESIterativeSolvers  src/LinearSolver.jl:149-206
  BorderingBLS        src/LinearBorderSolver.jl:59-166  ``bls(J, dR, dzu, dzp, R, n, xi_u, xi_p; shift, dotp)``
  MatrixFreeBLS       src/LinearBorderSolver.jl:404-437
  ShiftInvert         src/EigSolver.jl:246-266 (+ the SH3dEig of examples/SH3d.jl:96-113)
  SwiftHohenberg / SwiftHohenberg1D / CGL2d   the operator definitions of examples/SH3d.jl,
                      SH2d-fronts.jl, SHpde_snaking.jl, cGL2d.jl as ``BifurcationProblem``-like objects
                      whose ``jacobian`` returns an opaque device handle (src/Problems.jl:98-101)

PyTorch is used for device memory and the HIP stream only; all arithmetic runs in the HIP library.
Non-convergence is never an exception (flags, like the reference); misuse and device errors raise.
"""
from __future__ import annotations

import ctypes as C
import math
from dataclasses import dataclass, field

import numpy as np
import torch

from . import _lib as L


def _ptr(t):
    return C.c_void_p(t.data_ptr())


class Context:
    """``bk_ctx``: one per process / GPU.  ``comm`` = None | ("rccl", rank, nranks, id_bytes) |
    ("host", rank, nranks, allreduce, sendrecv[, lane_user]) -- ``lane_user`` (optional): the `user` value under which the SAME
    callbacks serve the second lane's communicator (bk_ctx_set_lane_comm; hostcomm.py: a second gloo group)."""

    def __init__(self, device: int = 0, comm=None):
        if not torch.cuda.is_available():
            raise RuntimeError("bifurcationkit.jl_amd needs a visible MI355X (no CPU fallback)")
        self.lib = L.load()
        self.device = device
        torch.cuda.set_device(device)
        self.torch_device = torch.device("cuda", device)
        stream = torch.cuda.current_stream(device).cuda_stream
        h = C.c_void_p()
        self.rank, self.nranks = 0, 1
        self._keep = []
        if comm is None:
            st = self.lib.bk_ctx_create(C.byref(h), device, C.c_void_p(stream))
        elif comm[0] == "rccl":
            _, rank, nranks, idb = comm
            buf = C.create_string_buffer(bytes(idb), L.BK_UNIQUE_ID_BYTES)
            st = self.lib.bk_ctx_create_dist(C.byref(h), device, C.c_void_p(stream), rank, nranks, buf)
            self.rank, self.nranks = rank, nranks
        elif comm[0] == "host":
            _, rank, nranks, allreduce, sendrecv = comm[:5]
            ar = L.ALLREDUCE_FN(allreduce)
            sr = L.SENDRECV_FN(sendrecv)
            self._keep += [ar, sr]
            st = self.lib.bk_ctx_create_hostcomm(C.byref(h), device, C.c_void_p(stream), rank, nranks, ar, sr, None)
            if st == 0 and len(comm) > 5 and comm[5] is not None:
                st = self.lib.bk_ctx_set_lane_comm(h, ar, sr, C.c_void_p(int(comm[5])))
            self.rank, self.nranks = rank, nranks
        else:
            raise ValueError(comm)
        if st != 0 or not h.value:
            raise L.BkHipError(f"bk_ctx_create failed ({st})")
        self.h = h

    @staticmethod
    def unique_id() -> bytes:
        lib = L.load()
        buf = C.create_string_buffer(L.BK_UNIQUE_ID_BYTES)
        if lib.bk_comm_unique_id(buf) != 0:
            raise L.BkHipError("bk_comm_unique_id failed")
        return buf.raw

    def check(self, st, what=""):
        L.check(self.h, st, what)

    def set_option(self, key: str, value: float):
        self.check(self.lib.bk_ctx_set_option(self.h, key.encode(), float(value)), "bk_ctx_set_option")

    def get_option(self, key: str) -> float:
        v = C.c_double()
        self.check(self.lib.bk_ctx_get_option(self.h, key.encode(), C.byref(v)), "bk_ctx_get_option")
        return v.value

    def sync(self):
        self.check(self.lib.bk_ctx_sync(self.h), "bk_ctx_sync")

    def comm_info(self):
        """(kind, rank, nranks) as the communicator itself reports them (RCCL: ncclCommUserRank / ncclCommCount)."""
        k, r, n = C.c_int(), C.c_int(), C.c_int()
        self.check(self.lib.bk_comm_info(self.h, C.byref(k), C.byref(r), C.byref(n)), "bk_comm_info")
        return {0: "none", 1: "rccl", 2: "host"}[k.value], r.value, n.value

    def comm_probe(self, what: str, count: int, reps: int = 20) -> float:
        """Microseconds per call of the hot path's collectives on this communicator (collective: all ranks call it):
        what = "allreduce" (count doubles, in-stream) | "halo" (count doubles to / from each z-neighbour)."""
        us = C.c_double()
        self.check(self.lib.bk_comm_probe(self.h, {"allreduce": 0, "halo": 1}[what], int(count), int(reps), C.byref(us)),
                   "bk_comm_probe")
        return us.value

    def prof_enable(self, on=True):
        self.check(self.lib.bk_prof_enable(self.h, 1 if on else 0))

    def prof_reset(self):
        self.check(self.lib.bk_prof_reset(self.h))

    def prof_get(self, name: str):
        ms, calls, nbytes = C.c_double(), C.c_longlong(), C.c_double()
        self.check(self.lib.bk_prof_get(self.h, name.encode(), C.byref(ms), C.byref(calls), C.byref(nbytes)))
        return dict(ms=ms.value, calls=calls.value, bytes=nbytes.value)

    BLOCK_LOG_FIELDS = ("solve", "j", "steps", "got", "last_pivot_ratio", "theta0", "theta1", "theta2", "theta3", "beta", "tol")

    def solver_block_log(self, reset=True):
        """Block log of the GMRES solves since the last reset (needs ``set_option("gmres_block_log", 1)``): one dict per Arnoldi
        block -- solve number, first column ``j``, steps issued / accepted (``got``), last pivot ratio, the Newton shifts
        (NaN: unused), residual estimate ``beta`` and tolerance when the block was issued (include/bkhip.h: bk_solver_block_log)."""
        n = C.c_size_t()
        self.check(self.lib.bk_solver_block_log(self.h, None, 0, C.byref(n), 0))
        buf = (C.c_double * max(n.value, 1))()
        self.check(self.lib.bk_solver_block_log(self.h, buf, n.value, C.byref(n), 1 if reset else 0))
        k = len(self.BLOCK_LOG_FIELDS)
        recs = []
        for i in range(0, n.value - n.value % k, k):
            r = dict(zip(self.BLOCK_LOG_FIELDS, buf[i:i + k]))
            for f in ("solve", "j", "steps", "got"):
                r[f] = int(r[f])
            recs.append(r)
        return recs

    def solver_history(self, reset=True):
        """Residual histories of the linear solves since the last reset (needs ``set_option("solver_trace", 1)``):
        a list with one list per solve, [initial residual, estimate after iteration 1, ...]."""
        n = C.c_size_t()
        self.check(self.lib.bk_solver_history(self.h, None, 0, C.byref(n), 0))
        buf = (C.c_double * max(n.value, 1))()
        self.check(self.lib.bk_solver_history(self.h, buf, n.value, C.byref(n), 1 if reset else 0))
        out = []
        for v in buf[:n.value]:
            if v < 0:
                out.append([])
            else:
                out[-1].append(v)
        return out

    def empty(self, n: int) -> torch.Tensor:
        return torch.empty(int(n), dtype=torch.float64, device=self.torch_device)

    def close(self):
        if getattr(self, "h", None) is not None and self.h.value:
            self.lib.bk_ctx_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class HipVec:
    """Device state vector (local slab) with the VectorInterface subset of SURVEY.md section 8b.
    ``nglobal`` is what Julia's ``length(x)`` would return for the undistributed vector."""

    __slots__ = ("ctx", "t", "nglobal")

    def __init__(self, ctx: Context, t: torch.Tensor, nglobal: int | None = None):
        assert t.dtype == torch.float64 and t.is_cuda and t.is_contiguous()
        self.ctx, self.t = ctx, t
        self.nglobal = int(nglobal if nglobal is not None else t.numel())

    # --- construction / transfer
    @classmethod
    def from_numpy(cls, ctx, a, nglobal=None):
        t = torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).to(ctx.torch_device)
        return cls(ctx, t, nglobal)

    def numpy(self):
        self.ctx.sync()
        return self.t.cpu().numpy()

    @property
    def n(self):
        return self.t.numel()

    def __len__(self):           # Base.length: used by NormalisedDot, src/continuation/Palc.jl:4
        return self.nglobal

    def similar(self):
        return HipVec(self.ctx, self.ctx.empty(self.n), self.nglobal)

    # --- VectorInterface
    def zerovector(self):        # VI.zerovector
        z = self.similar()
        self.ctx.check(self.ctx.lib.bk_vec_zero(self.ctx.h, self.n, _ptr(z.t)))
        return z

    def copy(self):              # _copy, src/BorderedArrays.jl:30
        z = self.similar()
        self.ctx.check(self.ctx.lib.bk_vec_copy(self.ctx.h, self.n, _ptr(self.t), _ptr(z.t)))
        return z

    def copyto_(self, src):      # _copyto!, src/BorderedArrays.jl:34
        self.ctx.check(self.ctx.lib.bk_vec_copy(self.ctx.h, self.n, _ptr(src.t), _ptr(self.t)))
        return self

    def scale_(self, a):         # VI.scale!
        self.ctx.check(self.ctx.lib.bk_vec_scale(self.ctx.h, self.n, float(a), _ptr(self.t)))
        return self

    def add_(self, x, a=1.0, b=1.0):   # VI.add!(y, x, a, b): y = b*y + a*x
        self.ctx.check(self.ctx.lib.bk_vec_axpby(self.ctx.h, self.n, float(a), _ptr(x.t), float(b), _ptr(self.t)))
        return self

    def inner(self, y):          # VI.inner
        out = C.c_double()
        self.ctx.check(self.ctx.lib.bk_vec_dot(self.ctx.h, self.n, _ptr(self.t), _ptr(y.t), C.byref(out)))
        return out.value

    def norm(self):              # VI.norm / LinearAlgebra.norm
        out = C.c_double()
        self.ctx.check(self.ctx.lib.bk_vec_nrm2(self.ctx.h, self.n, _ptr(self.t), C.byref(out)))
        return out.value

    def norminf(self):           # norm(x, Inf) = norminf, src/LinearSolver.jl:4
        out = C.c_double()
        self.ctx.check(self.ctx.lib.bk_vec_nrminf(self.ctx.h, self.n, _ptr(self.t), C.byref(out)))
        return out.value


# ------------------------------------------------------------------------------------------ problems
class HipJacobian:
    """What ``jacobian(prob, x, p)`` returns: an opaque operator handle (``bk_op``).  Keeps ``x`` alive, like
    the Julia closure ``dx -> dF_sh(x, p, dx)`` (examples/SH3d.jl:119) captures it."""

    def __init__(self, prob, x: HipVec, params, adjoint=False):
        self.prob, self.ctx, self.x = prob, prob.ctx, x
        arr = (C.c_double * len(params))(*params)
        h = C.c_void_p()
        fn = self.ctx.lib.bk_jacobian_adjoint if adjoint else self.ctx.lib.bk_jacobian
        self.ctx.check(fn(prob.h, _ptr(x.t), arr, len(params), C.byref(h)), "bk_jacobian")
        self.h = h

    def __call__(self, dx: HipVec, a0=0.0, a1=1.0) -> HipVec:      # apply(J, dx), src/Utils.jl:191-195
        out = dx.similar()
        self.ctx.check(self.ctx.lib.bk_op_apply(self.h, _ptr(dx.t), float(a0), float(a1), _ptr(out.t)), "bk_op_apply")
        return out

    def __del__(self):
        try:
            if self.h.value:
                self.ctx.lib.bk_op_destroy(self.h)
        except Exception:
            pass


class _PdeProblem:
    """BifurcationProblem-like: ``residual(x, p)``, ``jacobian(x, p)``, parameter 'lens' = index into params
    (src/Problems.jl:89-149)."""
    pde = None
    param_names = ()
    nfields = 1

    def __init__(self, ctx: Context, dims, ls, params: dict, lens: str):
        self.ctx = ctx
        self.dims = tuple(int(d) for d in dims)
        self.ls = tuple(float(x) for x in ls)
        self.params = dict(params)
        self.lens = lens
        self.ipar = self.param_names.index(lens)
        self.delta = math.sqrt(np.finfo(float).eps)         # getdelta: src/Problems.jl:69-70
        d = L.ProblemDesc()
        d.pde = self.pde
        d.ndim = len(self.dims)
        for i in range(3):
            d.n[i] = self.dims[i] if i < len(self.dims) else 1
            d.l[i] = self.ls[i] if i < len(self.ls) else 1.0
        h = C.c_void_p()
        ctx.check(ctx.lib.bk_problem_create(ctx.h, C.byref(d), C.byref(h)), "bk_problem_create")
        self.h = h
        nl, lo, hi = C.c_size_t(), C.c_int(), C.c_int()
        ctx.check(ctx.lib.bk_problem_nlocal(h, C.byref(nl), C.byref(lo), C.byref(hi)))
        self.nlocal, self.slab = nl.value, (lo.value, hi.value)
        self.nglobal = int(np.prod(self.dims)) * self.nfields

    def _pvec(self, p):
        vals = dict(self.params)
        vals[self.lens] = float(p)
        return [float(vals[k]) for k in self.param_names]

    def residual(self, x: HipVec, p: float) -> HipVec:
        out = x.similar()
        pv = self._pvec(p)
        arr = (C.c_double * len(pv))(*pv)
        self.ctx.check(self.ctx.lib.bk_residual(self.h, _ptr(x.t), arr, len(pv), _ptr(out.t)), "bk_residual")
        return out

    def residual_dparam(self, x: HipVec, p: float, eps: float | None = None) -> HipVec:
        """(F(x, p + eps) - F(x, p)) / eps for the continuation parameter, cancellation-free (bk_residual_dparam)."""
        out = x.similar()
        pv = self._pvec(p)
        arr = (C.c_double * len(pv))(*pv)
        self.ctx.check(self.ctx.lib.bk_residual_dparam(self.h, _ptr(x.t), arr, len(pv), self.ipar,
                                                       float(self.delta if eps is None else eps), _ptr(out.t)),
                       "bk_residual_dparam")
        return out

    def jacobian(self, x: HipVec, p: float) -> HipJacobian:
        return HipJacobian(self, x, self._pvec(p))

    def jacobian_adjoint(self, x: HipVec, p: float) -> HipJacobian:
        """J(x, p)' (the JAd of src/codim2/MinAugHopf.jl:66-80)."""
        return HipJacobian(self, x, self._pvec(p), adjoint=True)

    def vec(self, a_global: np.ndarray) -> HipVec:
        """Scatter a global NumPy state (x fastest) to this rank's slab."""
        a = np.asarray(a_global, dtype=np.float64).reshape(-1)
        if self.ctx.nranks == 1:
            return HipVec.from_numpy(self.ctx, a, self.nglobal)
        plane = int(np.prod(self.dims[:-1]))
        lo, hi = self.slab
        return HipVec.from_numpy(self.ctx, a[lo * plane:hi * plane], self.nglobal)

    def __del__(self):
        try:
            if self.h.value:
                self.ctx.lib.bk_problem_destroy(self.h)
        except Exception:
            pass


class SwiftHohenberg(_PdeProblem):
    """2-D/3-D quadratic-cubic SH, Neumann-ghost: examples/SH3d.jl:16-53, examples/SH2d-fronts.jl:13-34."""
    pde = L.BK_PDE_SH
    param_names = ("l", "nu")

    def __init__(self, ctx, dims, ls, l=0.1, nu=1.2, lens="l"):
        super().__init__(ctx, dims, ls, dict(l=l, nu=nu), lens)


class SwiftHohenberg1D(_PdeProblem):
    """1-D cubic-quintic SH, Dirichlet: examples/SHpde_snaking.jl:8-31."""
    pde = L.BK_PDE_SH1D
    param_names = ("lam", "nu")

    def __init__(self, ctx, N, l, lam=-0.1, nu=2.0, lens="lam"):
        super().__init__(ctx, (N,), (l,), dict(lam=lam, nu=nu), lens)


class CGL2d(_PdeProblem):
    """2-D cubic-quintic complex Ginzburg-Landau, SoA [u1; u2], Dirichlet: examples/cGL2d.jl:6-91."""
    pde = L.BK_PDE_CGL2D
    param_names = ("r", "mu", "nu", "c3", "c5", "gamma")
    nfields = 2

    def __init__(self, ctx, dims, ls, r=0.5, mu=0.1, nu=1.0, c3=-1.0, c5=1.0, gamma=0.0, lens="r"):
        super().__init__(ctx, dims, ls, dict(r=r, mu=mu, nu=nu, c3=c3, c5=c5, gamma=gamma), lens)


class DCTPreconditioner:
    """``Pl``: exact ((I+Lap)^2 + shift)^-1 -- ``cholesky(Symmetric(L1))`` of examples/SH3d.jl:88 (shift=0),
    ``lu(L1 + I)`` of examples/SH2d-fronts.jl:121 (shift=1)."""

    def __init__(self, prob: SwiftHohenberg, shift: float = 0.0):
        self.ctx, self.prob = prob.ctx, prob
        h = C.c_void_p()
        self.ctx.check(self.ctx.lib.bk_precond_sh_create(prob.h, float(shift), C.byref(h)), "bk_precond_sh_create")
        self.h = h

    def ldiv(self, v: HipVec) -> HipVec:        # Pl \ v
        out = v.similar()
        self.ctx.check(self.ctx.lib.bk_precond_apply(self.h, _ptr(v.t), _ptr(out.t)), "bk_precond_apply")
        return out

    def linmap(self, J: "HipJacobian", v: HipVec, a0=0.0, a1=1.0):
        """``a0 v + a1 Pl \\ (J v)``: the ``_linmap`` closure of GMRESKrylovKit with a left preconditioner
        (src/LinearSolver.jl:270-277) -- what every Arnoldi step of a preconditioned solve applies.  Returns (vector,
        stencil_free): whether the library evaluated it without the stencil (context option ``gmres_stencil_free``)."""
        out = v.similar()
        sf = C.c_int(0)
        self.ctx.check(self.ctx.lib.bk_precond_op_apply(self.ctx.h, self.h, J.h, _ptr(v.t), float(a0), float(a1), _ptr(out.t),
                                                       C.byref(sf)), "bk_precond_op_apply")
        return out, bool(sf.value)

    def __del__(self):
        try:
            if self.h.value:
                self.ctx.lib.bk_precond_destroy(self.h)
        except Exception:
            pass


class LaplacePreconditioner(DCTPreconditioner):
    """``Pl`` = (Lap - c I)^-1 on both cGL fields (DST-I; Dirichlet Laplacian of examples/cGL2d.jl:6-22).  Matrix-free
    stand-in for the sparse LU the reference uses on cGL2d (DefaultLS / ARPACK shift-invert, cGL2d.jl:96)."""

    def __init__(self, prob: "CGL2d", c: float = 1.0):
        self.ctx, self.prob = prob.ctx, prob
        h = C.c_void_p()
        self.ctx.check(self.ctx.lib.bk_precond_lap_create(prob.h, float(c), C.byref(h)), "bk_precond_lap_create")
        self.h = h


class CGLBlockPreconditioner(DCTPreconditioner):
    """``Pl`` = (Lap (x) I_2 + [[a, -b], [b, a]])^-1 on the stacked cGL fields: with a = r, b = nu the exact inverse of the
    Jacobian of the trivial state (Jcgl, examples/cGL2d.jl:57-79), with a = r - sigma that of the shift-inverted operator
    of ``EigArpack(sigma, :LM)`` (cGL2d.jl:96) -- what the reference's sparse LU provides there."""

    def __init__(self, prob: "CGL2d", a: float, b: float):
        self.ctx, self.prob = prob.ctx, prob
        h = C.c_void_p()
        self.ctx.check(self.ctx.lib.bk_precond_cgl_create(prob.h, float(a), float(b), C.byref(h)), "bk_precond_cgl_create")
        self.h = h


# ------------------------------------------------------------------------------------------ linear solvers
class _GMRES:
    flavor = L.BK_GMRES_KRYLOVKIT

    def _opts(self):
        o = L.GmresOpts()
        o.flavor, o.dim, o.maxiter, o.atol, o.rtol = self.flavor, self.dim, self.maxiter, self.atol, self.rtol
        pr = getattr(self, "Pr", None)                 # right preconditioner (GMRESIterativeSolvers.Pr, KrylovLS.Pr)
        o.pr = pr.h if pr is not None else None
        return o

    def _pl(self):
        return self.Pl.h if self.Pl is not None else None

    def __call__(self, J: HipJacobian, rhs: HipVec, a0=0.0, a1=1.0):
        """(ls)(J, rhs; a0, a1) -> (x, success, niter): src/LinearSolver.jl:12."""
        ctx = rhs.ctx
        x = rhs.similar()
        cv, it, rn = C.c_int(), C.c_int(), C.c_double()
        o = self._opts()
        ctx.check(ctx.lib.bk_gmres(ctx.h, J.h, _ptr(rhs.t), _ptr(x.t), float(a0), float(a1), C.byref(o), self._pl(),
                                   C.byref(cv), C.byref(it), C.byref(rn)), "bk_gmres")
        self.last_resnorm = rn.value
        return x, bool(cv.value), it.value

    def solve_complex(self, J: HipJacobian, rhs, a0: complex, a1: float = 1.0):
        """(ls)(J, rhs; a0::Complex, a1) on a complex right-hand side given as ``(re, im)`` HipVecs (``im`` may be
        None): src/NormalForms.jl:1053.  Returns ((x_re, x_im), success, niter)."""
        rr, ri = rhs
        ctx = rr.ctx
        xr, xi = rr.similar(), rr.similar()
        cv, it, rn = C.c_int(), C.c_int(), C.c_double()
        lo = self._opts()
        a0 = complex(a0)
        ctx.check(ctx.lib.bk_gmres_cshift(ctx.h, J.h, _ptr(rr.t), _ptr(ri.t) if ri is not None else None, _ptr(xr.t),
                                          _ptr(xi.t), a0.real, a0.imag, float(a1), C.byref(lo), self._pl(),
                                          C.byref(cv), C.byref(it), C.byref(rn)), "bk_gmres_cshift")
        return (xr, xi), bool(cv.value), it.value

    def solve2(self, J, rhs1, rhs2, a0=0.0, a1=1.0):
        """(ls)(J, rhs1, rhs2): src/LinearSolver.jl:15-19."""
        x1, f1, it1 = self(J, rhs1, a0, a1)
        x2, f2, it2 = self(J, rhs2, a0, a1)
        return x1, x2, f1 and f2, (it1, it2)


@dataclass
class GMRESKrylovKit(_GMRES):
    """Fields of src/LinearSolver.jl:223-250 (defaults = KrylovDefaults, :225-234).  ``issymmetric`` / ``ishermitian`` /
    ``isposdef`` are the switches the reference forwards to ``KrylovKit.linsolve`` (:256-267): KrylovKit keeps GMRES unless
    the map is declared symmetric AND positive definite, where it switches to CG -- here that combination selects the
    library's preconditioned CG (BK_KRYLOV_CG, the ``KrylovLS(:cg)`` core, M = Pl), same stopping rule
    max(atol, rtol*||b||) up to the M-norm; the symmetric-indefinite case stays GMRES exactly as in KrylovKit."""
    dim: int = 30
    atol: float = 1e-12
    rtol: float = 1e-12
    maxiter: int = 100
    Pl: DCTPreconditioner | None = None
    issymmetric: bool = False
    ishermitian: bool = False
    isposdef: bool = False

    @property
    def flavor(self):
        return L.BK_KRYLOV_CG if (self.isposdef and (self.issymmetric or self.ishermitian)) else L.BK_GMRES_KRYLOVKIT


@dataclass
class GMRESIterativeSolvers(_GMRES):
    """Fields of src/LinearSolver.jl:149-180: reltol, abstol, restart, maxiter, Pl, Pr (the solver iterates on
    Pl^-1 (a0 I + a1 J) Pr^-1 y = Pl^-1 rhs and returns x = Pr^-1 y, :198-201)."""
    reltol: float = 1e-8
    abstol: float = 0.0
    restart: int = 63          # the reference default (200) allocates a 201-vector basis; capped
    maxiter: int = 100
    Pl: DCTPreconditioner | None = None
    Pr: DCTPreconditioner | None = None
    flavor = L.BK_GMRES_ITERATIVESOLVERS

    @property
    def dim(self):
        return self.restart

    @property
    def atol(self):
        return self.abstol

    @property
    def rtol(self):
        return self.reltol


@dataclass
class KrylovLS(_GMRES):
    """KrylovLS / KrylovLSInplace with KrylovAlg = :gmres (src/LinearSolver.jl:316-414): Krylov.jl's stopping rule
    ||r|| <= atol + rtol ||r0||, `itmax` iterations, left preconditioner M = Pl applied to the shifted operator.  Krylov.jl
    restarts every `memory` steps only with ``restart = true``; by default the basis keeps growing past `memory` -- here it
    grows up to the library's 63 vectors (then restarts), which is the package's behaviour for every solve of <= 63
    iterations."""
    atol: float = 1.4901161193847656e-08
    rtol: float = 1.4901161193847656e-08
    memory: int = 20
    itmax: int = 2000
    restart: bool = False
    Pl: DCTPreconditioner | None = None
    Pr: DCTPreconditioner | None = None      # Krylov.jl's right preconditioner N (src/LinearSolver.jl:343)
    flavor = L.BK_GMRES_KRYLOVJL

    @property
    def dim(self):
        return self.memory if self.restart else 63

    @property
    def maxiter(self):
        return self.itmax


@dataclass
class KrylovLSSymmetric(_GMRES):
    """KrylovLS(KrylovAlg = :minres | :cg) (src/LinearSolver.jl:336-341): Krylov.jl's symmetric solvers on the shifted
    operator with the "centered" SPD preconditioner ``M = Pl``; stopping rule atol + rtol * (initial M^-1-norm of the
    residual); ``itmax = 0`` means 2n.  Short recurrences: 8 (MINRES) / 4 (CG) work vectors instead of a Krylov basis."""
    KrylovAlg: str = "minres"
    atol: float = 1.4901161193847656e-08
    rtol: float = 1.4901161193847656e-08
    itmax: int = 0
    Pl: DCTPreconditioner | None = None
    dim = 0

    @property
    def flavor(self):
        if self.KrylovAlg not in ("minres", "cg"):
            raise ValueError("KrylovLSSymmetric: KrylovAlg must be 'minres' or 'cg'")
        return L.BK_KRYLOV_MINRES if self.KrylovAlg == "minres" else L.BK_KRYLOV_CG

    @property
    def maxiter(self):
        return self.itmax


# ------------------------------------------------------------------------------------------ bordered solvers
@dataclass
class BorderedArray:
    """(u, p) pair, src/BorderedArrays.jl:23-26; VectorInterface methods :86-217."""
    u: object
    p: float

    def copy(self):
        return BorderedArray(self.u.copy(), self.p)

    def zerovector(self):
        return BorderedArray(self.u.zerovector(), 0.0)

    def copyto_(self, src):
        self.u.copyto_(src.u)
        self.p = src.p
        return self

    def scale_(self, a):
        self.u.scale_(a)
        self.p *= a
        return self

    def add_(self, x, a=1.0, b=1.0):
        self.u.add_(x.u, a, b)
        self.p = b * self.p + a * x.p
        return self

    def inner(self, y):
        return self.u.inner(y.u) + self.p * y.p

    def norm(self):
        return math.sqrt(self.u.norm() ** 2 + self.p ** 2)

    def __len__(self):
        return len(self.u) + 1


@dataclass
class BorderingBLS:
    """src/LinearBorderSolver.jl:59-79.  With a native GMRES ``solver`` and HipVec arguments the whole bordered
    solve is ONE library call (bk_bls_bordering); any other solver / vector type takes the generic path, which
    is the reference's algorithm line by line (:88-166)."""
    solver: object = None
    tol: float = 1e-12
    check_precision: bool = True
    k: int = 1

    def __post_init__(self):
        assert self.k > 0, "Number of recursions must be positive"

    def update_bls(self, ls):                      # update_bls, :490-493
        return BorderingBLS(ls, self.tol, self.check_precision, self.k)

    def __call__(self, J, dR, dzu, dzp, R, n, xiu=1.0, xip=1.0, *, shift=None, dotp=None, dotscale=None):
        native = isinstance(self.solver, _GMRES) and isinstance(R, HipVec) and (dotp is None or dotscale is not None)
        if native:
            ctx = R.ctx
            dX = R.similar()
            dl, cv = C.c_double(), C.c_int()
            it = (C.c_int * 2)()
            bo = L.BorderingOpts(self.tol, 1 if self.check_precision else 0, self.k)
            lo = self.solver._opts()
            ctx.check(ctx.lib.bk_bls_bordering(
                ctx.h, J.h, _ptr(dR.t), _ptr(dzu.t), float(dzp), _ptr(R.t), float(n), float(xiu), float(xip),
                0 if shift is None else 1, 0.0 if shift is None else float(shift),
                1.0 if dotscale is None else float(dotscale), C.byref(bo), C.byref(lo), self.solver._pl(),
                _ptr(dX.t), C.byref(dl), C.byref(cv), it), "bk_bls_bordering")
            return dX, dl.value, bool(cv.value), (it[0], it[1])
        return self._generic(J, dR, dzu, dzp, R, n, xiu, xip, shift, dotp or (lambda x, y: x.inner(y)))

    def solve_complex(self, J, dR, dzu, dzp, R, n, xiu=1.0, xip=1.0, *, shift: complex, dotscale=1.0):
        """(lbs::BorderingBLS)(J, dR, dzu, dzp, R, n; shift::Complex) on complex data ((re, im) pairs of HipVecs; ``im``
        may be None), one BEC pass: src/codim2/MinAugHopf.jl:17, 72-76.  Returns ((dX_re, dX_im), dl::complex, cv, its)."""
        ctx = R[0].ctx
        xr, xi = R[0].similar(), R[0].similar()
        p = lambda v: _ptr(v.t) if v is not None else None
        dl = (C.c_double * 2)()
        it = (C.c_int * 2)()
        cv = C.c_int()
        lo = self.solver._opts()
        dzp, n, shift = complex(dzp), complex(n), complex(shift)
        ctx.check(ctx.lib.bk_bls_bordering_cshift(
            ctx.h, J.h, p(dR[0]), p(dR[1]), p(dzu[0]), p(dzu[1]), dzp.real, dzp.imag, p(R[0]), p(R[1]), n.real, n.imag,
            float(xiu), float(xip), shift.real, shift.imag, float(dotscale), C.byref(lo), self.solver._pl(), p(xr), p(xi),
            dl, C.byref(cv), it), "bk_bls_bordering_cshift")
        return (xr, xi), complex(dl[0], dl[1]), bool(cv.value), (it[0], it[1])

    def solve_block(self, J, b, c, d, rhst, rhsb):
        """solve_bls_block(lbs::BorderingBLS, J, b::NTuple, c::NTuple, d, rhst, rhsb) -> (u1, u2, cv, its),
        src/LinearBorderSolver.jl:173-206 (m-column border: normal forms / Bogdanov-Takens)."""
        if not isinstance(self.solver, _GMRES):
            raise TypeError("solve_block (HIP) needs a native GMRES solver")
        d = np.atleast_2d(np.asarray(d, dtype=np.float64))
        m = d.shape[0]
        if not (len(b) == len(c) == m == d.shape[1]):
            raise ValueError("Linear bordered solver, wrong sizes!")
        ctx = rhst.ctx
        u1 = rhst.similar()
        bp = (C.c_void_p * m)(*[x.t.data_ptr() for x in b])
        cp = (C.c_void_p * m)(*[x.t.data_ptr() for x in c])
        dd = (C.c_double * (m * m))(*d.ravel().tolist())
        rb = (C.c_double * m)(*[float(v) for v in rhsb])
        u2 = (C.c_double * m)()
        its = (C.c_int * m)()
        cv = C.c_int()
        lo = self.solver._opts()
        ctx.check(ctx.lib.bk_bls_block_bordering(ctx.h, J.h, m, bp, cp, dd, _ptr(rhst.t), rb, C.byref(lo),
                                                 self.solver._pl(), _ptr(u1.t), u2, C.byref(cv), its),
                  "bk_bls_block_bordering")
        return u1, np.array(list(u2)), bool(cv.value), tuple(its)

    # generic path: BEC / residualBEC exactly as written in the reference
    def _bec(self, J, dR, dzu, dzp, R, n, xiu, xip, shift, dotp):
        kw = {} if shift is None else dict(a0=shift)
        if hasattr(self.solver, "solve2"):
            x1, dx, ok, it = self.solver.solve2(J, R, dR, **kw)
        else:
            x1, f1, i1 = self.solver(J, R, **kw)
            dx, f2, i2 = self.solver(J, dR, **kw)
            ok, it = f1 and f2, (i1, i2)
        dl = (n - dotp(dzu, x1) * xiu) / (dzp * xip - dotp(dzu, dx) * xiu)
        x1.add_(dx, -dl)
        return x1, dl, ok, it

    def _generic(self, J, dR, dzu, dzp, R, n, xiu, xip, shift, dotp):
        dX, dl, cv, it = self._bec(J, dR, dzu, dzp, R, n, xiu, xip, shift, dotp)
        k, fail = 0, True
        while self.check_precision and k < self.k and fail:
            dXr = J(dX)
            if shift is not None:
                dXr.add_(dX, shift)
            dXr.add_(dR, dl)
            dXr.add_(R, 1.0, -1.0)
            dlr = n - xip * dzp * dl - xiu * dotp(dzu, dX)
            fail = dXr.norm() > self.tol or abs(dlr) > self.tol
            if fail:
                dX1, dl1, cv, it = self._bec(J, dR, dzu, dzp, dXr, dlr, xiu, xip, shift, dotp)
                dX.add_(dX1, 1.0)
                dl += dl1
                k += 1
        return dX, dl, cv, it


@dataclass
class MatrixFreeBLS:
    """src/LinearBorderSolver.jl:404-437 on BorderedArray(u, p): one GMRES on the (N+1) operator
    MatrixFreeBLSmap (:326-335), device-resident with the scalar border on the host."""
    solver: _GMRES = None

    def update_bls(self, ls):
        return MatrixFreeBLS(ls)

    def __call__(self, J, dR, dzu, dzp, R, n, xiu=1.0, xip=1.0, *, shift=None, dotp=None, dotscale=None):
        if not isinstance(self.solver, _GMRES) or not isinstance(R, HipVec):
            raise TypeError("MatrixFreeBLS (HIP) needs a native GMRES solver and HipVec arguments")
        if dotp is not None and dotscale is None:
            raise TypeError("pass dotscale (dotp(x,y) = dotscale*<x,y>) for the native MatrixFreeBLS")
        ctx = R.ctx
        dX = R.similar()
        dl, cv, it = C.c_double(), C.c_int(), C.c_int()
        lo = self.solver._opts()
        ctx.check(ctx.lib.bk_bls_matrixfree(
            ctx.h, J.h, _ptr(dR.t), _ptr(dzu.t), float(dzp), _ptr(R.t), float(n), float(xiu), float(xip),
            0 if shift is None else 1, 0.0 if shift is None else float(shift),
            1.0 if dotscale is None else float(dotscale), C.byref(lo), _ptr(dX.t), C.byref(dl), C.byref(cv),
            C.byref(it)), "bk_bls_matrixfree")
        return dX, dl.value, bool(cv.value), it.value


    def solve_block(self, J, a, b, c, rhst, rhsb, *, shift=None, dotscale=1.0):
        """solve_bls_block(lbs::MatrixFreeBLS, J, a, b, c, rhst, rhsb; shift, dotp) -> (u1, u2, cv, it),
        src/LinearBorderSolver.jl:440-450: one GMRES on the (N + m) operator."""
        c = np.atleast_2d(np.asarray(c, dtype=np.float64))
        m = c.shape[0]
        if not (len(a) == len(b) == m == c.shape[1]):
            raise ValueError("Linear bordered solver, wrong sizes!")
        ctx = rhst.ctx
        u1 = rhst.similar()
        ap = (C.c_void_p * m)(*[x.t.data_ptr() for x in a])
        bp = (C.c_void_p * m)(*[x.t.data_ptr() for x in b])
        cc = (C.c_double * (m * m))(*c.ravel().tolist())
        rb = (C.c_double * m)(*[float(v) for v in rhsb])
        u2 = (C.c_double * m)()
        cv, it = C.c_int(), C.c_int()
        lo = self.solver._opts()
        ctx.check(ctx.lib.bk_bls_block_matrixfree(ctx.h, J.h, m, ap, bp, cc, _ptr(rhst.t), rb, 0 if shift is None else 1,
                                                  0.0 if shift is None else float(shift), float(dotscale), C.byref(lo),
                                                  _ptr(u1.t), u2, C.byref(cv), C.byref(it)), "bk_bls_block_matrixfree")
        return u1, np.array(list(u2)), bool(cv.value), it.value


# ------------------------------------------------------------------------------------------ eigensolver
@dataclass
class ShiftInvert:
    """ShiftInvert(sigma, ls, eig) of src/EigSolver.jl:246-266 with a KrylovKit.eigsolve-style outer
    iteration -- the SH3dEig of examples/SH3d.jl:96-113: ``(eig)(J, nev) -> (vals, vecs, converged, numops)``,
    eigenvalues Complex, sorted by decreasing real part; ``geteigenvector(eig, vecs, n) = vecs[n]``."""
    sigma: float
    ls: _GMRES
    tol: float = 1e-12
    maxiter: int = 20
    krylovdim: int | None = None         # None -> max(30, nev + 30), examples/SH3d.jl:109
    hermitian: bool = False
    seed: int = 1234
    save_vectors: bool = True
    x0: object = None                    # start vector (HipVec); None -> rand(N), examples/SH3d.jl:109

    def __call__(self, J: HipJacobian, nev: int, **kwargs):
        ctx = J.ctx
        if self.x0 is not None:
            ctx.check(ctx.lib.bk_eig_set_start_vector(ctx.h, _ptr(self.x0.t)), "bk_eig_set_start_vector")
        kd = self.krylovdim if self.krylovdim is not None else max(30, nev + 30)
        kd = min(kd, 63, J.prob.nglobal - 1)
        nev = min(nev, kd)
        eo = L.EigOpts(float(self.sigma), int(kd), int(self.maxiter), float(self.tol), 1 if self.hermitian else 0,
                       int(self.seed))
        lo = self.ls._opts()
        re = (C.c_double * (nev + 1))()
        im = (C.c_double * (nev + 1))()
        n = J.prob.nlocal
        ld = (n + 31) // 32 * 32
        vr = ctx.empty(ld * (nev + 1)) if self.save_vectors else None
        vi = ctx.empty(ld * (nev + 1)) if (self.save_vectors and not self.hermitian) else None
        nvals, nconv, nops = C.c_int(), C.c_int(), C.c_int()
        ctx.check(ctx.lib.bk_eig_shiftinvert(
            ctx.h, J.h, nev, C.byref(eo), C.byref(lo), self.ls._pl(), re, im,
            _ptr(vr) if vr is not None else None, _ptr(vi) if vi is not None else None, ld,
            C.byref(nvals), C.byref(nconv), C.byref(nops)), "bk_eig_shiftinvert")
        m = nvals.value                                   # nev, or nev + 1 to keep a complex pair together
        vals = np.array([complex(re[i], im[i]) for i in range(m)])        # NaN = not converged (ignored by is_stable)
        vecs = None
        if vr is not None:
            vecs = [(HipVec(ctx, vr[i * ld:i * ld + n], J.prob.nglobal),
                     HipVec(ctx, vi[i * ld:i * ld + n], J.prob.nglobal) if vi is not None else None)
                    for i in range(m)]
        return vals, vecs, nconv.value >= nev, nops.value

    @staticmethod
    def geteigenvector(vecs, n):
        return vecs[n]


@dataclass
class EigKrylovKit:
    """EigKrylovKit (src/EigSolver.jl:117-166) with ``which = :LR``: KrylovKit.eigsolve on the Jacobian itself, no inner
    solves.  ``(eig)(J, nev) -> (vals, vecs, converged, numops)`` sorted by decreasing real part."""
    tol: float = 1e-6
    maxiter: int = 100
    krylovdim: int = 30
    hermitian: bool = False
    seed: int = 1234
    save_vectors: bool = False
    x0: object = None                    # EigKrylovKit.x0, src/EigSolver.jl:143

    def __call__(self, J: HipJacobian, nev: int, **kwargs):
        ctx = J.ctx
        if self.x0 is not None:
            ctx.check(ctx.lib.bk_eig_set_start_vector(ctx.h, _ptr(self.x0.t)), "bk_eig_set_start_vector")
        kd = min(self.krylovdim, 63, J.prob.nglobal - 1)
        nev = min(nev, kd)
        eo = L.EigOpts(0.0, int(kd), int(self.maxiter), float(self.tol), 1 if self.hermitian else 0, int(self.seed))
        re = (C.c_double * (nev + 1))()
        im = (C.c_double * (nev + 1))()
        n = J.prob.nlocal
        ld = (n + 31) // 32 * 32
        vr = ctx.empty(ld * (nev + 1)) if self.save_vectors else None
        vi = ctx.empty(ld * (nev + 1)) if (self.save_vectors and not self.hermitian) else None
        nvals, nconv, nops = C.c_int(), C.c_int(), C.c_int()
        ctx.check(ctx.lib.bk_eig_krylovkit(ctx.h, J.h, nev, C.byref(eo), re, im, _ptr(vr) if vr is not None else None,
                                           _ptr(vi) if vi is not None else None, ld, C.byref(nvals), C.byref(nconv),
                                           C.byref(nops)), "bk_eig_krylovkit")
        m = nvals.value
        vals = np.array([complex(re[i], im[i]) for i in range(m)])
        vecs = None
        if vr is not None:
            vecs = [(HipVec(ctx, vr[i * ld:i * ld + n], J.prob.nglobal),
                     HipVec(ctx, vi[i * ld:i * ld + n], J.prob.nglobal) if vi is not None else None) for i in range(m)]
        return vals, vecs, nconv.value >= nev, nops.value

    @staticmethod
    def geteigenvector(vecs, n):
        return vecs[n]


@dataclass
class EigArpack:
    """EigArpack(sigma = nothing, which = :LR; tol, maxiter, ncv, v0) (src/EigSolver.jl:67-102): the same
    ``(eig)(J, nev) -> (vals, vecs, true, 1)`` contract on the library's Krylov-Schur cores.  With ``sigma`` ARPACK
    factorises ``J - sigma I`` (matrix only); the matrix-free replacement solves with the iterative ``ls`` (required then),
    ``which`` applying to 1/(lambda - sigma) as in ARPACK.  Without ``sigma`` only ``which = "LR"`` is offered.  ARPACK
    defaults: ``ncv = max(20, 2 nev + 1)``, ``maxiter = 300``, ``tol = 0`` (-> machine precision; clamped to what the inner
    solves can deliver, 10 x their rtol)."""
    sigma: float | None = None
    which: str = "LR"
    ls: _GMRES | None = None
    tol: float = 0.0
    maxiter: int = 300
    ncv: int | None = None
    v0: object = None
    hermitian: bool = False
    save_vectors: bool = True

    def __call__(self, J: HipJacobian, nev: int, **kwargs):
        ncv = self.ncv if self.ncv is not None else max(20, 2 * nev + 1)
        if self.sigma is None:
            if self.which != "LR":
                raise NotImplementedError("EigArpack (HIP): without sigma only which = :LR is available")
            core = EigKrylovKit(tol=max(self.tol, 1e-10), maxiter=self.maxiter, krylovdim=ncv, hermitian=self.hermitian,
                                save_vectors=self.save_vectors, x0=self.v0)
        else:
            if self.ls is None:
                raise TypeError("EigArpack (HIP): the matrix-free shift-invert needs a linear solver `ls` for J - sigma I")
            if self.which != "LM":
                raise NotImplementedError("EigArpack (HIP): with sigma use which = :LM (eigenvalues closest to sigma)")
            core = ShiftInvert(self.sigma, self.ls, tol=max(self.tol, 10.0 * self.ls.rtol), maxiter=self.maxiter,
                               krylovdim=ncv, hermitian=self.hermitian, save_vectors=self.save_vectors, x0=self.v0)
        vals, vecs, cv, it = core(J, nev)
        return vals, vecs, True, 1                         # __sort_arpack returns (.., true, 1), src/EigSolver.jl:98-102

    @staticmethod
    def geteigenvector(vecs, n):
        return vecs[n]


@dataclass
class EigArnoldiMethod:
    """EigArnoldiMethod(; sigma = nothing, which = LR(), x0, tol, mindim, maxdim, restarts) (src/EigSolver.jl:182-235):
    implicitly restarted Arnoldi (Krylov-Schur) for the rightmost eigenvalues of the operator; like the reference, the
    shift-invert strategy is not available for maps (:218-220 warns and ignores ``sigma``).  ArnoldiMethod defaults:
    ``tol = sqrt(eps)``, ``mindim = max(10, nev)``, ``maxdim = max(20, 2 nev)``, ``restarts = 200``."""
    sigma: float | None = None
    which: str = "LR"
    x0: object = None
    tol: float = 1.4901161193847656e-08
    maxdim: int | None = None
    restarts: int = 200
    hermitian: bool = False
    save_vectors: bool = True

    def __call__(self, J: HipJacobian, nev: int, **kwargs):
        if self.sigma is not None:
            import warnings
            warnings.warn("Shift-Invert strategy not implemented for maps")      # the reference's own message
        if self.which != "LR":
            raise NotImplementedError("EigArnoldiMethod (HIP): which = LR() only")
        maxdim = self.maxdim if self.maxdim is not None else max(20, 2 * nev)
        core = EigKrylovKit(tol=self.tol, maxiter=self.restarts, krylovdim=maxdim, hermitian=self.hermitian,
                            save_vectors=self.save_vectors, x0=self.x0)
        vals, vecs, cv, it = core(J, nev)
        return vals, vecs, cv, 1

    @staticmethod
    def geteigenvector(vecs, n):
        return vecs[n]


# ------------------------------------------------------------------------------------------ native correctors
def newton_opts(tol, max_iterations, norm_inf, linesearch=False, alpha=1.0, alphamin=1e-3, callback=None):
    """``bk_newton_opts`` from the NewtonPar fields (src/Newton.jl:17-33) and a reference-style callback
    ``callback(state; fromNewton) -> Bool`` (src/Newton.jl:88): a ``cbMaxNorm``-like object (attribute ``maxres``) is
    evaluated inside the library; any other callable becomes a host callback receiving
    ``dict(x, fx | res_f, residual, step, itlinear, p, z0)`` where ``x``, ``fx`` / ``res_f`` and ``z0[0]`` are raw device
    addresses (ints; None for a plain Newton's ``z0``) of vectors of the problem's local length."""
    no = L.NewtonOpts()
    no.tol, no.max_iterations, no.norm_inf = float(tol), int(max_iterations), 1 if norm_inf else 0
    no.linesearch, no.alpha, no.alpha_min = 1 if linesearch else 0, float(alpha), float(alphamin)
    if callback is not None and hasattr(callback, "maxres"):
        no.max_residual = float(callback.maxres)
    elif callback is not None:
        def tramp(user, x, fx, residual, step, itlinear, p, z0u, z0p, from_newton):
            st = dict(x=x, residual=residual, step=step, itlinear=itlinear, p=p, z0=(z0u, z0p))
            st["fx" if from_newton else "res_f"] = fx
            try:
                return 1 if callback(st, fromNewton=bool(from_newton)) else 0
            except Exception:      # an exception must not unwind through the C frames
                import traceback
                traceback.print_exc()
                return 0
        no.callback = L.NEWTON_CALLBACK(tramp)
        no._keepalive = (tramp, no.callback)
    return no


def newton_native(prob: _PdeProblem, x0: HipVec, p: float, ls: _GMRES, tol=1e-12, max_iterations=25, norm_inf=False,
                  callback=None):
    """_newton (src/Newton.jl:66-114) as one library call."""
    ctx = prob.ctx
    x = x0.copy()
    pv = prob._pvec(p)
    arr = (C.c_double * len(pv))(*pv)
    no = newton_opts(tol, max_iterations, norm_inf, callback=callback)
    lo = ls._opts()
    res = L.NewtonResult()
    ctx.check(ctx.lib.bk_newton(ctx.h, prob.h, _ptr(x.t), arr, len(pv), C.byref(no), C.byref(lo), ls._pl(),
                                C.byref(res)), "bk_newton")
    return dict(u=x, converged=bool(res.converged), itnewton=res.itnewton, itlineartot=res.itlinear,
                residuals=[res.residuals[i] for i in range(res.itnewton + 1)])


def newton_palc_native(prob: _PdeProblem, z0: BorderedArray, tau: BorderedArray, z_pred: BorderedArray, ds, theta,
                       bls: BorderingBLS, tol=1e-12, max_iterations=25, p_min=-math.inf, p_max=math.inf,
                       norm_inf=False, linesearch=False, alpha=1.0, alphamin=1e-3, callback=None):
    """newton_palc (src/continuation/Palc.jl:187-305) as one library call, with BorderingBLS or MatrixFreeBLS."""
    ctx = prob.ctx
    x = z_pred.u.copy()
    p = C.c_double(z_pred.p)
    pv = prob._pvec(z_pred.p)
    arr = (C.c_double * len(pv))(*pv)
    no = newton_opts(tol, max_iterations, norm_inf, linesearch, alpha, alphamin, callback)
    if isinstance(bls, MatrixFreeBLS):                     # bk_bordering_opts.kind = 1: one GMRES on the (N + 1) operator per iteration
        bo = L.BorderingOpts(0.0, 0, 1, 1)
    else:
        bo = L.BorderingOpts(bls.tol, 1 if bls.check_precision else 0, bls.k, 0)
    lo = bls.solver._opts()
    res = L.NewtonResult()
    big = 1.7e308
    ctx.check(ctx.lib.bk_newton_palc(
        ctx.h, prob.h, _ptr(x.t), C.byref(p), _ptr(z0.u.t), float(z0.p), _ptr(tau.u.t), float(tau.p), float(ds),
        float(theta), arr, len(pv), prob.ipar, max(float(p_min), -big), min(float(p_max), big), C.byref(no),
        C.byref(bo), C.byref(lo), bls.solver._pl(), C.byref(res)), "bk_newton_palc")
    return dict(u=BorderedArray(x, p.value), converged=bool(res.converged), itnewton=res.itnewton,
                itlineartot=res.itlinear, residuals=[res.residuals[i] for i in range(res.itnewton + 1)])


# ------------------------------------------------------------------------------------------ deflation
@dataclass
class DeflationOperator:
    """DeflationOperator(power, dot = VI.inner, alpha, roots; accumulator) of src/DeflationOperator.jl:61-141 on device
    vectors: ``M(u) = prod_i(<u - root_i, u - root_i>^-power + alpha)`` (or the mean), ``dM`` by finite differences
    (autodiff = false, delta = 1e-8, :160-169)."""
    power: float
    alpha: float
    roots: list = field(default_factory=list)
    accumulator: str = "prod"
    delta: float = 1e-8

    def push(self, v):
        self.roots.append(v)

    def __len__(self):
        return len(self.roots)

    def __call__(self, u: HipVec) -> float:
        if not self.roots:
            return 1.0
        out = None
        for r in self.roots:
            d = u.copy().add_(r, -1.0)
            m = 1.0 / d.inner(d) ** self.power + self.alpha
            out = m if out is None else (out * m if self.accumulator == "prod" else out + m)
        return out / len(self.roots) if self.accumulator == "mean" else out

    def dM(self, u: HipVec, du: HipVec) -> float:
        if not self.roots:
            return 0.0
        return (self(u.copy().add_(du, self.delta)) - self(u)) / self.delta


def deflated_custom_ls(ls, prob, defop: DeflationOperator, u: HipVec, p, rhs: HipVec):
    """(dfl::DeflatedProblemCustomLS)(J = (u, p, defPb), rhs), src/DeflationOperator.jl:264-312, line by line on the
    plugin surface: two solves with the plain Jacobian, then h = (h1 - z h2)/M(u)."""
    Fu = prob.residual(u, p)
    Mu = defop(u)
    Ju = prob.jacobian(u, p)
    if len(defop) == 0:
        h1, _, it1 = ls(Ju, rhs)
        return h1, True, (it1, 0)
    h1, h2, _, (it1, it2) = ls.solve2(Ju, rhs, Fu)
    z1 = defop.dM(u, h1)
    z2 = defop.dM(u, h2)
    z = z1 / (Mu + z2)
    return h1.add_(h2, -z).scale_(1.0 / Mu), True, (it1, it2)


def newton_deflated(prob, defop: DeflationOperator, x0: HipVec, p, ls, tol=1e-12, max_iterations=25, norm_inf=False):
    """solve(prob, defOp, options, DeflatedProblemCustomLS()) (:340-355) driven call by call through the plugin surface."""
    nrm = (lambda v: v.norminf()) if norm_inf else (lambda v: v.norm())
    x = x0.copy()
    fx = prob.residual(x, p).scale_(defop(x))
    res = [nrm(fx)]
    step, itlin = 0, 0
    while step < max_iterations and res[-1] > tol:
        h, _, it = deflated_custom_ls(ls, prob, defop, x, p, fx)
        itlin += int(np.sum(it))
        x.add_(h, -1.0)
        fx = prob.residual(x, p).scale_(defop(x))
        res.append(nrm(fx))
        step += 1
    return dict(u=x, converged=res[-1] < tol, itnewton=step, itlineartot=itlin, residuals=res)


def newton_deflated_native(prob: _PdeProblem, defop: DeflationOperator, x0: HipVec, p: float, ls: _GMRES, tol=1e-12,
                           max_iterations=25, norm_inf=False):
    """The same as one library call (bk_newton_deflated)."""
    ctx = prob.ctx
    x = x0.copy()
    pv = prob._pvec(p)
    arr = (C.c_double * len(pv))(*pv)
    no = newton_opts(tol, max_iterations, norm_inf)
    lo = ls._opts()
    res = L.NewtonResult()
    m = len(defop)
    rp = (C.c_void_p * max(m, 1))(*[r.t.data_ptr() for r in defop.roots])
    ctx.check(ctx.lib.bk_newton_deflated(ctx.h, prob.h, _ptr(x.t), arr, len(pv), rp, m, float(defop.power),
                                         float(defop.alpha), 1 if defop.accumulator == "mean" else 0,
                                         float(defop.delta), C.byref(no), C.byref(lo), ls._pl(), C.byref(res)),
              "bk_newton_deflated")
    return dict(u=x, converged=bool(res.converged), itnewton=res.itnewton, itlineartot=res.itlinear,
                residuals=[res.residuals[i] for i in range(res.itnewton + 1)])
