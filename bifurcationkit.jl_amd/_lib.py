"""ctypes binding of ``lib/libbkhip.so`` (C ABI: ``include/bkhip.h``).

The binding is the Python twin of the Julia ``ccall`` shim (``julia/BifurcationKitHIP.jl``): same entry
points, same argument order.  There is NO fallback: if the shared library is missing the import of any
product module raises, and every compute call needs a visible MI355X.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# BKHIP_LIB: an explicit library file (A/B builds of the same sources, e.g. `make layout0`); default = the in-tree build
LIB_PATH = os.environ.get("BKHIP_LIB") or os.path.join(_HERE, "lib", "libbkhip.so")

BK_ABI_VERSION = 6          # include/bkhip.h: layout version of the option structs mirrored below
BK_UNIQUE_ID_BYTES = 128
BK_MAX_PARAMS = 8
BK_MAX_NEWTON_ITER = 64

BK_PDE_SH, BK_PDE_SH1D, BK_PDE_CGL2D = 1, 2, 3
BK_GMRES_KRYLOVKIT, BK_GMRES_ITERATIVESOLVERS, BK_GMRES_KRYLOVJL = 0, 1, 2
BK_KRYLOV_MINRES, BK_KRYLOV_CG = 3, 4

c_double_p = C.POINTER(C.c_double)
c_int_p = C.POINTER(C.c_int)


class ProblemDesc(C.Structure):
    _fields_ = [("pde", C.c_int), ("ndim", C.c_int), ("n", C.c_int * 3), ("l", C.c_double * 3)]


class GmresOpts(C.Structure):
    _fields_ = [("flavor", C.c_int), ("dim", C.c_int), ("maxiter", C.c_int), ("atol", C.c_double),
                ("rtol", C.c_double), ("pr", C.c_void_p)]


class BorderingOpts(C.Structure):
    _fields_ = [("tol", C.c_double), ("check_precision", C.c_int), ("k", C.c_int), ("kind", C.c_int)]


class EigOpts(C.Structure):
    _fields_ = [("sigma", C.c_double), ("krylovdim", C.c_int), ("maxiter", C.c_int), ("tol", C.c_double),
                ("hermitian", C.c_int), ("seed", C.c_ulonglong)]


NEWTON_CALLBACK = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_int, C.c_int, C.c_double,
                              C.c_void_p, C.c_double, C.c_int)


class NewtonOpts(C.Structure):
    _fields_ = [("tol", C.c_double), ("max_iterations", C.c_int), ("norm_inf", C.c_int), ("linesearch", C.c_int),
                ("alpha", C.c_double), ("alpha_min", C.c_double), ("max_residual", C.c_double),
                ("callback", NEWTON_CALLBACK), ("callback_user", C.c_void_p)]


class NewtonResult(C.Structure):
    _fields_ = [("converged", C.c_int), ("itnewton", C.c_int), ("itlinear", C.c_int),
                ("residuals", C.c_double * (BK_MAX_NEWTON_ITER + 1))]


BK_MAX_NEV = 62


class ContOpts(C.Structure):
    _fields_ = [("ds", C.c_double), ("dsmin", C.c_double), ("dsmax", C.c_double), ("a", C.c_double),
                ("theta", C.c_double), ("p_min", C.c_double), ("p_max", C.c_double), ("tangent", C.c_int),
                ("detect", C.c_int), ("nev", C.c_int), ("tol_stability", C.c_double)]


class ContStepResult(C.Structure):
    _fields_ = [("converged", C.c_int), ("itnewton", C.c_int), ("itlinear", C.c_int),
                ("residuals", C.c_double * (BK_MAX_NEWTON_ITER + 1)), ("p", C.c_double), ("ds_used", C.c_double),
                ("ds_next", C.c_double), ("step", C.c_int), ("stop", C.c_int), ("n_unstable", C.c_int),
                ("n_imag", C.c_int), ("bifurcation", C.c_int), ("nvals", C.c_int), ("eig_converged", C.c_int),
                ("eig_numops", C.c_int), ("vals_re", C.c_double * (BK_MAX_NEV + 1)),
                ("vals_im", C.c_double * (BK_MAX_NEV + 1)), ("tangent_converged", C.c_int), ("natural", C.c_int)]


class BisectionOpts(C.Structure):
    _fields_ = [("dsmin_bisection", C.c_double), ("n_inversion", C.c_int), ("max_bisection_steps", C.c_int),
                ("tol_bisection_eigenvalue", C.c_double), ("max_steps", C.c_int)]


class BisectionResult(C.Structure):
    _fields_ = [("status", C.c_int), ("type", C.c_int), ("interval", C.c_double * 2), ("p", C.c_double),
                ("n_unstable", C.c_int * 2), ("n_imag", C.c_int * 2), ("steps", C.c_int), ("nvals", C.c_int),
                ("vals_re", C.c_double * (BK_MAX_NEV + 1)), ("vals_im", C.c_double * (BK_MAX_NEV + 1))]


ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, c_double_p, C.c_int, C.c_int)
SENDRECV_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, c_double_p, C.c_size_t, C.c_int, c_double_p, C.c_size_t, C.c_int)

VP = C.c_void_p     # opaque handles and device pointers travel as void*
D = C.c_double
I = C.c_int
SZ = C.c_size_t

# name -> (restype, argtypes): every symbol include/bkhip.h declares
SIGNATURES = {
    "bk_version": (I, []),
    "bk_abi_version": (I, []),
    "bk_ctx_create": (I, [C.POINTER(VP), I, VP]),
    "bk_comm_unique_id": (I, [VP]),
    "bk_ctx_create_dist": (I, [C.POINTER(VP), I, VP, I, I, VP]),
    "bk_ctx_create_hostcomm": (I, [C.POINTER(VP), I, VP, I, I, ALLREDUCE_FN, SENDRECV_FN, VP]),
    "bk_ctx_set_lane_comm": (I, [VP, ALLREDUCE_FN, SENDRECV_FN, VP]),
    "bk_comm_info": (I, [VP, c_int_p, c_int_p, c_int_p]),
    "bk_comm_probe": (I, [VP, I, SZ, I, c_double_p]),
    "bk_ctx_destroy": (I, [VP]),
    "bk_last_error": (C.c_char_p, [VP]),
    "bk_ctx_sync": (I, [VP]),
    "bk_ctx_set_option": (I, [VP, C.c_char_p, D]),
    "bk_ctx_get_option": (I, [VP, C.c_char_p, c_double_p]),
    "bk_prof_enable": (I, [VP, I]),
    "bk_prof_reset": (I, [VP]),
    "bk_prof_get": (I, [VP, C.c_char_p, c_double_p, C.POINTER(C.c_longlong), c_double_p]),
    "bk_solver_history": (I, [VP, c_double_p, SZ, C.POINTER(SZ), I]),
    "bk_solver_block_log": (I, [VP, c_double_p, SZ, C.POINTER(SZ), I]),
    "bk_malloc": (I, [VP, SZ, C.POINTER(VP)]),
    "bk_free": (I, [VP, VP]),
    "bk_upload": (I, [VP, VP, c_double_p, SZ]),
    "bk_download": (I, [VP, c_double_p, VP, SZ]),
    "bk_vec_copy": (I, [VP, SZ, VP, VP]),
    "bk_vec_zero": (I, [VP, SZ, VP]),
    "bk_vec_scale": (I, [VP, SZ, D, VP]),
    "bk_vec_axpby": (I, [VP, SZ, D, VP, D, VP]),
    "bk_vec_dot": (I, [VP, SZ, VP, VP, c_double_p]),
    "bk_vec_nrm2": (I, [VP, SZ, VP, c_double_p]),
    "bk_vec_nrminf": (I, [VP, SZ, VP, c_double_p]),
    "bk_krylov_multidot": (I, [VP, SZ, VP, SZ, I, VP, c_double_p]),
    "bk_krylov_multiaxpy": (I, [VP, SZ, VP, SZ, I, c_double_p, VP, D, VP, c_double_p]),
    "bk_problem_create": (I, [VP, C.POINTER(ProblemDesc), C.POINTER(VP)]),
    "bk_problem_destroy": (I, [VP]),
    "bk_problem_nlocal": (I, [VP, C.POINTER(SZ), c_int_p, c_int_p]),
    "bk_residual": (I, [VP, VP, c_double_p, I, VP]),
    "bk_residual_dparam": (I, [VP, VP, c_double_p, I, I, D, VP]),
    "bk_jacobian": (I, [VP, VP, c_double_p, I, C.POINTER(VP)]),
    "bk_op_destroy": (I, [VP]),
    "bk_jacobian_adjoint": (I, [VP, VP, c_double_p, I, C.POINTER(VP)]),
    "bk_op_apply": (I, [VP, VP, D, D, VP]),
    "bk_precond_sh_create": (I, [VP, D, C.POINTER(VP)]),
    "bk_precond_lap_create": (I, [VP, D, C.POINTER(VP)]),
    "bk_precond_cgl_create": (I, [VP, D, D, C.POINTER(VP)]),
    "bk_precond_destroy": (I, [VP]),
    "bk_precond_apply": (I, [VP, VP, VP]),
    "bk_precond_op_apply": (I, [VP, VP, VP, VP, D, D, VP, c_int_p]),
    "bk_gmres_default_opts": (None, [C.POINTER(GmresOpts), I]),
    "bk_gmres": (I, [VP, VP, VP, VP, D, D, C.POINTER(GmresOpts), VP, c_int_p, c_int_p, c_double_p]),
    "bk_gmres2": (I, [VP, VP, VP, VP, VP, VP, D, D, C.POINTER(GmresOpts), VP, c_int_p, c_int_p]),
    "bk_bls_bordering": (I, [VP, VP, VP, VP, D, VP, D, D, D, I, D, D, C.POINTER(BorderingOpts),
                             C.POINTER(GmresOpts), VP, VP, c_double_p, c_int_p, c_int_p]),
    "bk_bls_matrixfree": (I, [VP, VP, VP, VP, D, VP, D, D, D, I, D, D, C.POINTER(GmresOpts), VP, c_double_p,
                              c_int_p, c_int_p]),
    "bk_bls_block_bordering": (I, [VP, VP, I, C.POINTER(VP), C.POINTER(VP), c_double_p, VP, c_double_p,
                                   C.POINTER(GmresOpts), VP, VP, c_double_p, c_int_p, c_int_p]),
    "bk_bls_block_matrixfree": (I, [VP, VP, I, C.POINTER(VP), C.POINTER(VP), c_double_p, VP, c_double_p, I, D, D,
                                    C.POINTER(GmresOpts), VP, c_double_p, c_int_p, c_int_p]),
    "bk_gmres_cshift": (I, [VP, VP, VP, VP, VP, VP, D, D, D, C.POINTER(GmresOpts), VP, c_int_p, c_int_p, c_double_p]),
    "bk_bls_bordering_cshift": (I, [VP, VP, VP, VP, VP, VP, D, D, VP, VP, D, D, D, D, D, D, D, C.POINTER(GmresOpts),
                                    VP, VP, VP, c_double_p, c_int_p, c_int_p]),
    "bk_eig_shiftinvert": (I, [VP, VP, I, C.POINTER(EigOpts), C.POINTER(GmresOpts), VP, c_double_p, c_double_p,
                               VP, VP, SZ, c_int_p, c_int_p, c_int_p]),
    "bk_eig_set_start_vector": (I, [VP, VP]),
    "bk_eig_krylovkit": (I, [VP, VP, I, C.POINTER(EigOpts), c_double_p, c_double_p, VP, VP, SZ, c_int_p, c_int_p, c_int_p]),
    "bk_newton": (I, [VP, VP, VP, c_double_p, I, C.POINTER(NewtonOpts), C.POINTER(GmresOpts), VP,
                      C.POINTER(NewtonResult)]),
    "bk_cont_create": (I, [VP, VP, c_double_p, I, I, VP, D, VP, D, C.POINTER(ContOpts), C.POINTER(NewtonOpts),
                           C.POINTER(BorderingOpts), C.POINTER(GmresOpts), VP, C.POINTER(EigOpts),
                           C.POINTER(GmresOpts), VP, C.POINTER(ContStepResult), C.POINTER(VP)]),
    "bk_cont_step": (I, [VP, C.POINTER(ContStepResult)]),
    "bk_cont_get": (I, [VP, VP, c_double_p, VP, c_double_p, c_double_p]),
    "bk_cont_destroy": (I, [VP]),
    "bk_cont_clone": (I, [VP, C.POINTER(VP)]),
    "bk_cont_locate_bifurcation": (I, [VP, C.POINTER(BisectionOpts), C.POINTER(BisectionResult)]),
    "bk_newton_deflated": (I, [VP, VP, VP, c_double_p, I, C.POINTER(VP), I, D, D, I, D, C.POINTER(NewtonOpts),
                               C.POINTER(GmresOpts), VP, C.POINTER(NewtonResult)]),
    "bk_newton_palc": (I, [VP, VP, VP, c_double_p, VP, D, VP, D, D, D, c_double_p, I, I, D, D,
                           C.POINTER(NewtonOpts), C.POINTER(BorderingOpts), C.POINTER(GmresOpts), VP,
                           C.POINTER(NewtonResult)]),
}

_lib = None


class BkHipError(RuntimeError):
    pass


def load():
    """Load libbkhip.so (once) and attach prototypes.  Raises if the library is missing: the product has
    no CPU fallback by design."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(make -C bifurcationkit.jl_amd/csrc).  There is no CPU fallback.")
    # one HIP runtime per process: if torch is in use it must be imported first so that its bundled
    # libamdhip64.so.7 / librccl.so.1 are the ones this library binds to (same SONAMEs as /opt/rocm's).
    try:
        import torch  # noqa: F401
    except Exception:  # pragma: no cover - torch is optional for the C ABI itself
        pass
    lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError here = header/library mismatch
        fn.restype = res
        fn.argtypes = args
    if lib.bk_abi_version() != BK_ABI_VERSION:
        raise ImportError(f"{LIB_PATH}: option-struct layout version {lib.bk_abi_version()}, this binding was written for "
                          f"{BK_ABI_VERSION} (include/bkhip.h: BK_ABI_VERSION)")
    _lib = lib
    return lib


def check(ctx_handle, status, what=""):
    if status != 0:
        msg = ""
        if _lib is not None and ctx_handle:
            raw = _lib.bk_last_error(ctx_handle)
            msg = raw.decode() if raw else ""
        raise BkHipError(f"{what} failed with status {status}: {msg}")
