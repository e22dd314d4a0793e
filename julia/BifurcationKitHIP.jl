# BifurcationKitHIP.jl -- the reference-side binding of libbkhip.so (C ABI: include/bkhip.h).
#
# Thin `ccall` layer that makes the MI355X-native corrector a drop-in behind BifurcationKit's own plugin
# surface, so `continuation` / `newton` call into it unchanged:
#
#   HipVec                  state vector (device pointer + context) with the VectorInterface methods the PALC
#                           path calls (checklist: SURVEY.md section 8b; src/BorderedArrays.jl:86-217)
#   HipJacobian             what `prob.VF.J(x, p)` returns: an opaque operator handle (src/Problems.jl:98-101)
#   HipGMRES                <: BK.AbstractIterativeLinearSolver   (src/LinearSolver.jl:8-12, 223-291)
#   HipBorderingBLS         <: BK.AbstractBorderedLinearSolver    (src/LinearBorderSolver.jl:1-6, 59-166)
#   HipMatrixFreeBLS        <: BK.AbstractBorderedLinearSolver    (src/LinearBorderSolver.jl:404-437)
#   HipShiftInvert          <: BK.AbstractEigenSolver             (src/EigSolver.jl:4-12, 246-266)
#
# NOTE: there is no Julia in the build container of this repository, so this file has not been executed; the
# identical call sequence is exercised through Python ctypes (bifurcationkit.jl_amd/_lib.py, tests/).  The
# argument order of every ccall below is the order of the prototypes in include/bkhip.h.
module BifurcationKitHIP

using BifurcationKit
import BifurcationKit: AbstractIterativeLinearSolver, AbstractBorderedLinearSolver, AbstractEigenSolver
import LinearAlgebra
import KrylovKit: VectorInterface
const BK = BifurcationKit
const VI = VectorInterface

const libbkhip = Ref{String}(joinpath(@__DIR__, "..", "bifurcationkit.jl_amd", "lib", "libbkhip.so"))

# ------------------------------------------------------------------------------------------------ C structs
struct ProblemDesc
    pde::Cint
    ndim::Cint
    n::NTuple{3, Cint}
    l::NTuple{3, Cdouble}
end
struct GmresOpts
    flavor::Cint; dim::Cint; maxiter::Cint; atol::Cdouble; rtol::Cdouble
    pr::Ptr{Cvoid}             # right preconditioner handle or C_NULL (bk_gmres_opts.pr)
end
struct BorderingOpts
    tol::Cdouble; check_precision::Cint; k::Cint
    kind::Cint                 # correctors: 0 = BorderingBLS, 1 = MatrixFreeBLS (bk_bordering_opts.kind)
end
struct EigOpts
    sigma::Cdouble; krylovdim::Cint; maxiter::Cint; tol::Cdouble; hermitian::Cint; seed::Culonglong
end

const BK_ABI_VERSION = Cint(6)     # include/bkhip.h
const BK_MAX_NEWTON_ITER = 64
struct NewtonOpts            # bk_newton_opts
    tol::Cdouble; max_iterations::Cint; norm_inf::Cint; linesearch::Cint; alpha::Cdouble; alpha_min::Cdouble
    max_residual::Cdouble; callback::Ptr{Cvoid}; callback_user::Ptr{Cvoid}
end
struct NewtonResult          # bk_newton_result
    converged::Cint; itnewton::Cint; itlinear::Cint; residuals::NTuple{BK_MAX_NEWTON_ITER + 1, Cdouble}
end
NewtonResult() = NewtonResult(0, 0, 0, ntuple(_ -> 0.0, BK_MAX_NEWTON_ITER + 1))

const BK_PDE_SH, BK_PDE_SH1D, BK_PDE_CGL2D = Cint(1), Cint(2), Cint(3)

# ------------------------------------------------------------------------------------------------ context
mutable struct HipContext
    h::Ptr{Cvoid}
    function HipContext(device::Integer = 0)
        # the option structs above mirror include/bkhip.h at BK_ABI_VERSION; refuse a library with another layout
        abi = ccall((:bk_abi_version, libbkhip[]), Cint, ())
        abi == BK_ABI_VERSION || error("libbkhip: option-struct layout version $abi, this binding was written for $BK_ABI_VERSION")
        r = Ref{Ptr{Cvoid}}(C_NULL)
        st = ccall((:bk_ctx_create, libbkhip[]), Cint, (Ref{Ptr{Cvoid}}, Cint, Ptr{Cvoid}), r, device, C_NULL)
        st == 0 || error("bk_ctx_create failed ($st)")
        ctx = new(r[])
        finalizer(c -> ccall((:bk_ctx_destroy, libbkhip[]), Cint, (Ptr{Cvoid},), c.h), ctx)
    end
end
"(kind, rank, nranks) of the context's communicator as RCCL itself reports them (bk_comm_info)"
function comm_info(ctx::HipContext)
    k, r, n = Ref{Cint}(0), Ref{Cint}(0), Ref{Cint}(1)
    ccall((:bk_comm_info, libbkhip[]), Cint, (Ptr{Cvoid}, Ref{Cint}, Ref{Cint}, Ref{Cint}), ctx.h, k, r, n)
    return (:none, :rccl, :host)[k[] + 1], Int(r[]), Int(n[])
end
"microseconds per call of the hot path's collectives on this communicator (bk_comm_probe; what = 0 all-reduce, 1 halo)"
function comm_probe(ctx::HipContext, what::Integer, count::Integer, reps::Integer = 20)
    us = Ref{Cdouble}(0)
    check(ctx, ccall((:bk_comm_probe, libbkhip[]), Cint, (Ptr{Cvoid}, Cint, Csize_t, Cint, Ref{Cdouble}), ctx.h, what, count, reps, us), "bk_comm_probe")
    return us[]
end
function check(ctx::HipContext, st::Cint, what = "")
    st == 0 && return nothing
    msg = unsafe_string(ccall((:bk_last_error, libbkhip[]), Cstring, (Ptr{Cvoid},), ctx.h))
    error("$what failed with status $st: $msg")
end

# ------------------------------------------------------------------------------------------------ HipVec
mutable struct HipVec
    ctx::HipContext
    p::Ptr{Cdouble}      # device pointer
    n::Int               # local length (== global length on one GPU)
    function HipVec(ctx::HipContext, n::Integer)
        r = Ref{Ptr{Cdouble}}(C_NULL)
        check(ctx, ccall((:bk_malloc, libbkhip[]), Cint, (Ptr{Cvoid}, Csize_t, Ref{Ptr{Cdouble}}), ctx.h, n, r), "bk_malloc")
        v = new(ctx, r[], n)
        finalizer(x -> ccall((:bk_free, libbkhip[]), Cint, (Ptr{Cvoid}, Ptr{Cdouble}), x.ctx.h, x.p), v)
    end
    # non-owning view of device memory the library hands to a callback (no finalizer)
    HipVec(ctx::HipContext, n::Integer, p::Ptr{Cdouble}, owns::Bool) = (@assert !owns; new(ctx, p, n))
end
function HipVec(ctx::HipContext, a::Vector{Float64})
    v = HipVec(ctx, length(a))
    check(ctx, ccall((:bk_upload, libbkhip[]), Cint, (Ptr{Cvoid}, Ptr{Cdouble}, Ptr{Cdouble}, Csize_t), ctx.h, v.p, a, length(a)), "bk_upload")
    v
end
function Base.Array(v::HipVec)
    a = Vector{Float64}(undef, v.n)
    check(v.ctx, ccall((:bk_download, libbkhip[]), Cint, (Ptr{Cvoid}, Ptr{Cdouble}, Ptr{Cdouble}, Csize_t), v.ctx.h, a, v.p, v.n), "bk_download")
    a
end
# Checkpoints (`save_to_file`, ext/JLD2Ext/save.jl:8-30; `save_solution` of the problem, src/Continuation.jl:289): the state
# leaves the device as a plain Vector{Float64}.  With JLD2 loaded, `JLD2.writeas(::Type{HipVec}) = Vector{Float64}` together
# with `Base.convert(::Type{Vector{Float64}}, v::HipVec) = Array(v)` makes the extension serialise branches unchanged.
Base.convert(::Type{Vector{Float64}}, v::HipVec) = Array(v)
Base.length(v::HipVec) = v.n
Base.eltype(::HipVec) = Float64
Base.similar(v::HipVec) = HipVec(v.ctx, v.n)

_copy!(dst::HipVec, src::HipVec) = (check(dst.ctx, ccall((:bk_vec_copy, libbkhip[]), Cint, (Ptr{Cvoid}, Csize_t, Ptr{Cdouble}, Ptr{Cdouble}), dst.ctx.h, dst.n, src.p, dst.p)); dst)
_axpby!(y::HipVec, x::HipVec, a, b) = (check(y.ctx, ccall((:bk_vec_axpby, libbkhip[]), Cint, (Ptr{Cvoid}, Csize_t, Cdouble, Ptr{Cdouble}, Cdouble, Ptr{Cdouble}), y.ctx.h, y.n, a, x.p, b, y.p)); y)

# VectorInterface (src/BorderedArrays.jl:86-217 is the template)
VI.scalartype(::Type{HipVec}) = Float64
VI.scalartype(::HipVec) = Float64
function VI.zerovector(v::HipVec, ::Type{S} = Float64) where {S}
    z = similar(v)
    check(v.ctx, ccall((:bk_vec_zero, libbkhip[]), Cint, (Ptr{Cvoid}, Csize_t, Ptr{Cdouble}), v.ctx.h, z.n, z.p))
    z
end
VI.zerovector!(v::HipVec) = (check(v.ctx, ccall((:bk_vec_zero, libbkhip[]), Cint, (Ptr{Cvoid}, Csize_t, Ptr{Cdouble}), v.ctx.h, v.n, v.p)); v)
VI.zerovector!!(v::HipVec) = VI.zerovector!(v)
VI.scale!(v::HipVec, a::Number) = (check(v.ctx, ccall((:bk_vec_scale, libbkhip[]), Cint, (Ptr{Cvoid}, Csize_t, Cdouble, Ptr{Cdouble}), v.ctx.h, v.n, a, v.p)); v)
VI.scale!!(v::HipVec, a::Number) = VI.scale!(v, a)
VI.scale(v::HipVec, a::Number) = VI.scale!(_copy!(similar(v), v), a)
VI.scale!(y::HipVec, x::HipVec, a::Number) = VI.scale!(_copy!(y, x), a)
VI.scale!!(y::HipVec, x::HipVec, a::Number) = VI.scale!(y, x, a)
VI.add!(y::HipVec, x::HipVec, a::Number = 1, b::Number = 1) = _axpby!(y, x, a, b)      # y = b*y + a*x
VI.add!!(y::HipVec, x::HipVec, a::Number = 1, b::Number = 1) = VI.add!(y, x, a, b)
VI.add(y::HipVec, x::HipVec, a::Number = 1, b::Number = 1) = VI.add!(_copy!(similar(y), y), x, a, b)
function VI.inner(x::HipVec, y::HipVec)
    r = Ref{Cdouble}(0)
    check(x.ctx, ccall((:bk_vec_dot, libbkhip[]), Cint, (Ptr{Cvoid}, Csize_t, Ptr{Cdouble}, Ptr{Cdouble}, Ref{Cdouble}), x.ctx.h, x.n, x.p, y.p, r))
    r[]
end
function VI.norm(x::HipVec)
    r = Ref{Cdouble}(0)
    check(x.ctx, ccall((:bk_vec_nrm2, libbkhip[]), Cint, (Ptr{Cvoid}, Csize_t, Ptr{Cdouble}, Ref{Cdouble}), x.ctx.h, x.n, x.p, r))
    r[]
end
function LinearAlgebra.norm(x::HipVec, p::Real = 2)
    p == 2 && return VI.norm(x)
    p == Inf || error("HipVec: only norm(x), norm(x, 2), norm(x, Inf)")
    r = Ref{Cdouble}(0)
    check(x.ctx, ccall((:bk_vec_nrminf, libbkhip[]), Cint, (Ptr{Cvoid}, Csize_t, Ptr{Cdouble}, Ref{Cdouble}), x.ctx.h, x.n, x.p, r))
    r[]
end
LinearAlgebra.dot(x::HipVec, y::HipVec) = VI.inner(x, y)
# the two internal helpers the engine calls on state vectors (src/BorderedArrays.jl:30-47)
BK._copy(v::HipVec) = _copy!(similar(v), v)
BK._copyto!(dst::HipVec, src::HipVec) = _copy!(dst, src)

# ------------------------------------------------------------------------------------------------ problems
mutable struct HipProblem
    ctx::HipContext
    h::Ptr{Cvoid}
    nparams::Int
end
function HipProblem(ctx::HipContext, pde::Cint, dims::NTuple{N, Int}, ls::NTuple{N, Float64}) where {N}
    n = ntuple(i -> Cint(i <= N ? dims[i] : 1), 3)
    l = ntuple(i -> Cdouble(i <= N ? ls[i] : 1.0), 3)
    r = Ref{Ptr{Cvoid}}(C_NULL)
    d = Ref(ProblemDesc(pde, Cint(N), n, l))
    check(ctx, ccall((:bk_problem_create, libbkhip[]), Cint, (Ptr{Cvoid}, Ref{ProblemDesc}, Ref{Ptr{Cvoid}}), ctx.h, d, r), "bk_problem_create")
    p = HipProblem(ctx, r[], pde == BK_PDE_CGL2D ? 6 : 2)
    finalizer(x -> ccall((:bk_problem_destroy, libbkhip[]), Cint, (Ptr{Cvoid},), x.h), p)
end
"Swift-Hohenberg 2-D/3-D (examples/SH3d.jl:16-53): params = (l, nu)"
SwiftHohenberg(ctx, dims, ls) = HipProblem(ctx, BK_PDE_SH, dims, ls)

# F(u, p) with p a NamedTuple whose fields are in the kernel's parameter order, e.g. (l = 0.1, ν = 1.2)
function residual(prob::HipProblem, u::HipVec, par)
    out = similar(u)
    pv = Cdouble[Float64(x) for x in Tuple(par)][1:prob.nparams]
    check(prob.ctx, ccall((:bk_residual, libbkhip[]), Cint, (Ptr{Cvoid}, Ptr{Cdouble}, Ptr{Cdouble}, Cint, Ptr{Cdouble}), prob.h, u.p, pv, length(pv), out.p), "bk_residual")
    out
end

# (F(u, p + eps) - F(u, p)) / eps for the parameter number `ipar` (0-based kernel order), evaluated without the ~1e-8 white
# rounding noise of the two-residual quotient (every parameter multiplies a pointwise term; bk_residual_dparam).  The
# reference's newton_palc forms the quotient itself from two `residual` calls (src/continuation/Palc.jl:239-240); the
# methods of section "dF/dp" below route the engine to this function by dispatch on the device problem type.
function residual_dparam(prob::HipProblem, u::HipVec, par, ipar::Integer; eps = sqrt(Base.eps(Float64)))
    out = similar(u)
    pv = Cdouble[Float64(x) for x in Tuple(par)][1:prob.nparams]
    check(prob.ctx, ccall((:bk_residual_dparam, libbkhip[]), Cint, (Ptr{Cvoid}, Ptr{Cdouble}, Ptr{Cdouble}, Cint, Cint, Cdouble, Ptr{Cdouble}),
                          prob.h, u.p, pv, length(pv), ipar, eps, out.p), "bk_residual_dparam")
    out
end

mutable struct HipJacobian
    prob::HipProblem
    h::Ptr{Cvoid}
    x::HipVec            # kept alive: the handle references it, like `dx -> dF_sh(x, p, dx)` captures x
end
function jacobian(prob::HipProblem, u::HipVec, par)
    pv = Cdouble[Float64(x) for x in Tuple(par)][1:prob.nparams]
    r = Ref{Ptr{Cvoid}}(C_NULL)
    check(prob.ctx, ccall((:bk_jacobian, libbkhip[]), Cint, (Ptr{Cvoid}, Ptr{Cdouble}, Ptr{Cdouble}, Cint, Ref{Ptr{Cvoid}}), prob.h, u.p, pv, length(pv), r), "bk_jacobian")
    J = HipJacobian(prob, r[], u)
    finalizer(j -> ccall((:bk_op_destroy, libbkhip[]), Cint, (Ptr{Cvoid},), j.h), J)
end
# apply(J, dx): src/Utils.jl:191-195 dispatches to J(dx) for callables
function (J::HipJacobian)(dx::HipVec)
    out = similar(dx)
    check(J.prob.ctx, ccall((:bk_op_apply, libbkhip[]), Cint, (Ptr{Cvoid}, Ptr{Cdouble}, Cdouble, Cdouble, Ptr{Cdouble}), J.h, dx.p, 0.0, 1.0, out.p), "bk_op_apply")
    out
end

# F as a callable STRUCT (not an anonymous closure): its type marks the BifurcationProblem as a device problem, so that the
# two places where the engine forms dF/dp by a two-residual quotient can be routed by dispatch (section "dF/dp" below).
struct HipResidualFn
    prob::HipProblem
end
(f::HipResidualFn)(u::HipVec, p) = residual(f.prob, u, p)
struct HipJacobianFn
    prob::HipProblem
end
(f::HipJacobianFn)(u::HipVec, p) = jacobian(f.prob, u, p)

"BifurcationProblem whose F and J live on the device.  `par` must list the kernel parameters first."
bifurcation_problem(prob::HipProblem, u0::HipVec, par, lens; kwargs...) =
    BK.BifurcationProblem(HipResidualFn(prob), u0, par, lens; J = HipJacobianFn(prob), kwargs...)

# the problem / iterator types the routed methods dispatch on (src/Problems.jl:89,344: BifurcationProblem{Tvf, ...} with
# Tvf = BifFunction{Tf, ...}; src/Continuation.jl:27: ContIterable{Tkind, Tprob, ...})
const HipBifProblem = BK.BifurcationProblem{<:BK.BifFunction{<:HipResidualFn}}
const HipContIterable = BK.ContIterable{<:BK.AbstractContinuationKind, <:HipBifProblem}
hipproblem(bp::HipBifProblem) = bp.VF.F.prob
# 0-based kernel index of the continuation parameter: the field of `par` the lens writes to
function _ipar(par, lens)
    a, b = Tuple(BK.set(par, lens, 1.0)), Tuple(BK.set(par, lens, 2.0))
    i = findfirst(k -> a[k] != b[k], 1:length(a))
    isnothing(i) && error("the continuation lens does not address a kernel parameter")
    return i - 1
end

mutable struct HipDCTPreconditioner     # Pl = (L1 + shift I)^-1: cholesky(L1) of SH3d.jl:88 / lu(L1 + I) of SH2d-fronts.jl:121
    prob::HipProblem
    h::Ptr{Cvoid}
end
function HipDCTPreconditioner(prob::HipProblem, shift::Real = 0.0)
    r = Ref{Ptr{Cvoid}}(C_NULL)
    check(prob.ctx, ccall((:bk_precond_sh_create, libbkhip[]), Cint, (Ptr{Cvoid}, Cdouble, Ref{Ptr{Cvoid}}), prob.h, shift, r), "bk_precond_sh_create")
    P = HipDCTPreconditioner(prob, r[])
    finalizer(x -> ccall((:bk_precond_destroy, libbkhip[]), Cint, (Ptr{Cvoid},), x.h), P)
end
"Pl = (Lap - c I)^-1 on both cGL fields (DST-I of the Dirichlet Laplacian, examples/cGL2d.jl:6-22)."
mutable struct HipLaplacePreconditioner
    prob::HipProblem
    h::Ptr{Cvoid}
end
function HipLaplacePreconditioner(prob::HipProblem, c::Real = 1.0)
    r = Ref{Ptr{Cvoid}}(C_NULL)
    check(prob.ctx, ccall((:bk_precond_lap_create, libbkhip[]), Cint, (Ptr{Cvoid}, Cdouble, Ref{Ptr{Cvoid}}), prob.h, c, r), "bk_precond_lap_create")
    P = HipLaplacePreconditioner(prob, r[])
    finalizer(x -> ccall((:bk_precond_destroy, libbkhip[]), Cint, (Ptr{Cvoid},), x.h), P)
end
"""
Pl = (Lap (x) I_2 + [[a, -b], [b, a]])^-1 on the stacked cGL fields: with `a = r, b = nu` the exact inverse of the Jacobian of
the trivial state (Jcgl, examples/cGL2d.jl:57-79) -- what `DefaultLS` computes there by sparse LU --, with `a = r - sigma` that
of the shift-inverted operator of `EigArpack(sigma, :LM)` (cGL2d.jl:96).
"""
mutable struct HipCGLBlockPreconditioner
    prob::HipProblem
    h::Ptr{Cvoid}
end
function HipCGLBlockPreconditioner(prob::HipProblem, a::Real, b::Real)
    r = Ref{Ptr{Cvoid}}(C_NULL)
    check(prob.ctx, ccall((:bk_precond_cgl_create, libbkhip[]), Cint, (Ptr{Cvoid}, Cdouble, Cdouble, Ref{Ptr{Cvoid}}), prob.h, a, b, r), "bk_precond_cgl_create")
    P = HipCGLBlockPreconditioner(prob, r[])
    finalizer(x -> ccall((:bk_precond_destroy, libbkhip[]), Cint, (Ptr{Cvoid},), x.h), P)
end
_plh(::Nothing) = C_NULL
_plh(P::Union{HipDCTPreconditioner, HipLaplacePreconditioner, HipCGLBlockPreconditioner}) = P.h

# ------------------------------------------------------------------------------------------------ linear solver
"""
    HipGMRES(; dim = 30, atol = 1e-12, rtol = 1e-12, maxiter = 100, Pl = nothing)

Device-resident restarted GMRES with the fields and the semantics of `GMRESKrylovKit`
(src/LinearSolver.jl:223-291): `ls(J, rhs; a₀, a₁) -> (x, success, numops)`.
"""
Base.@kwdef mutable struct HipGMRES{Tl} <: AbstractIterativeLinearSolver
    dim::Int = 30
    atol::Float64 = 1e-12
    rtol::Float64 = 1e-12
    maxiter::Int = 100
    Pl::Tl = nothing
end
_opts(l::HipGMRES) = GmresOpts(Cint(0), l.dim, l.maxiter, l.atol, l.rtol, C_NULL)      # GMRESKrylovKit has no Pr
_num(a, default) = a isa Number ? Float64(a) : default          # VI.Zero() / VI.One() defaults of the engine

function (l::HipGMRES)(J::HipJacobian, rhs::HipVec; a₀ = 0.0, a₁ = 1.0, kwargs...)
    ctx = rhs.ctx
    x = similar(rhs)
    cv, it, rn = Ref{Cint}(0), Ref{Cint}(0), Ref{Cdouble}(0)
    check(ctx, ccall((:bk_gmres, libbkhip[]), Cint,
        (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cdouble}, Ptr{Cdouble}, Cdouble, Cdouble, Ref{GmresOpts}, Ptr{Cvoid}, Ref{Cint}, Ref{Cint}, Ref{Cdouble}),
        ctx.h, J.h, rhs.p, x.p, _num(a₀, 0.0), _num(a₁, 1.0), Ref(_opts(l)), _plh(l.Pl), cv, it, rn), "bk_gmres")
    return x, cv[] == 1, Int(it[])
end
# (ls)(J, rhs1, rhs2) keeps the default of src/LinearSolver.jl:15-19 (two calls of the method above).

"""
    HipKrylovLS(KrylovAlg = :gmres | :minres | :cg; atol, rtol, memory, itmax, restart, Pl)

`KrylovLS` (src/LinearSolver.jl:316-345): Krylov.jl semantics; `:minres` / `:cg` are the symmetric solvers with the
centered preconditioner `M = Pl` (8 / 4 work vectors instead of a Krylov basis).
"""
Base.@kwdef mutable struct HipKrylovLS{Tl} <: AbstractIterativeLinearSolver
    KrylovAlg::Symbol = :gmres
    atol::Float64 = sqrt(eps())
    rtol::Float64 = sqrt(eps())
    memory::Int = 20
    itmax::Int = 0
    restart::Bool = false      # Krylov.jl: restart every `memory` steps only when true; else the basis grows (here: up to 63)
    Pl::Tl = nothing
    Pr = nothing               # right preconditioner N = Pr of the non-symmetric methods (src/LinearSolver.jl:343)
end
_flavor(l::HipKrylovLS) = l.KrylovAlg == :gmres ? Cint(2) : l.KrylovAlg == :minres ? Cint(3) : l.KrylovAlg == :cg ? Cint(4) :
                          error("HipKrylovLS: KrylovAlg must be :gmres, :minres or :cg")
_opts(l::HipKrylovLS) = GmresOpts(_flavor(l), l.restart ? l.memory : 63, l.KrylovAlg == :gmres && l.itmax == 0 ? 2000 : l.itmax, l.atol, l.rtol,
                                  _plh(l.Pr))
function (l::HipKrylovLS)(J::HipJacobian, rhs::HipVec; a₀ = 0.0, a₁ = 1.0, kwargs...)
    ctx = rhs.ctx
    x = similar(rhs)
    cv, it, rn = Ref{Cint}(0), Ref{Cint}(0), Ref{Cdouble}(0)
    check(ctx, ccall((:bk_gmres, libbkhip[]), Cint,
        (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cdouble}, Ptr{Cdouble}, Cdouble, Cdouble, Ref{GmresOpts}, Ptr{Cvoid}, Ref{Cint}, Ref{Cint}, Ref{Cdouble}),
        ctx.h, J.h, rhs.p, x.p, _num(a₀, 0.0), _num(a₁, 1.0), Ref(_opts(l)), _plh(l.Pl), cv, it, rn), "bk_gmres")
    return x, cv[] == 1, Int(it[])
end

# Complex device vectors of the Hopf machinery: (re, im) pairs of HipVec.
struct HipCVec
    re::HipVec
    im::HipVec
end
# ls(L, rhs; a₀ = Complex(0, 2ω), a₁ = -1), src/NormalForms.jl:1053
function (l::HipGMRES)(J::HipJacobian, rhs::HipCVec; a₀ = 0.0, a₁ = 1.0, kwargs...)
    ctx = rhs.re.ctx
    x = HipCVec(similar(rhs.re), similar(rhs.re))
    cv, it, rn = Ref{Cint}(0), Ref{Cint}(0), Ref{Cdouble}(0)
    a0 = a₀ isa Number ? ComplexF64(a₀) : ComplexF64(0)
    check(ctx, ccall((:bk_gmres_cshift, libbkhip[]), Cint,
        (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Cdouble, Cdouble, Cdouble, Ref{GmresOpts},
         Ptr{Cvoid}, Ref{Cint}, Ref{Cint}, Ref{Cdouble}),
        ctx.h, J.h, rhs.re.p, rhs.im.p, x.re.p, x.im.p, real(a0), imag(a0), _num(a₁, 1.0), Ref(_opts(l)), _plh(l.Pl), cv, it, rn),
        "bk_gmres_cshift")
    return x, cv[] == 1, Int(it[])
end
# the adjoint Jacobian JAd of src/codim2/MinAugHopf.jl:66-80 (transposed pointwise block for cGL, J itself for SH)
function jacobian_adjoint(prob::HipProblem, u::HipVec, par)
    pv = Cdouble[Float64(x) for x in Tuple(par)][1:prob.nparams]
    r = Ref{Ptr{Cvoid}}(C_NULL)
    check(prob.ctx, ccall((:bk_jacobian_adjoint, libbkhip[]), Cint, (Ptr{Cvoid}, Ptr{Cdouble}, Ptr{Cdouble}, Cint, Ref{Ptr{Cvoid}}),
                          prob.h, u.p, pv, length(pv), r), "bk_jacobian_adjoint")
    J = HipJacobian(prob, r[], u)
    finalizer(j -> ccall((:bk_op_destroy, libbkhip[]), Cint, (Ptr{Cvoid},), j.h), J)
end

# ------------------------------------------------------------------------------------------------ bordered solvers
"""
    HipBorderingBLS(solver; tol = 1e-12, check_precision = true, k = 1)

`BorderingBLS` (src/LinearBorderSolver.jl:59-166) as ONE library call per bordered solve.
"""
Base.@kwdef struct HipBorderingBLS{S <: Union{HipGMRES, Nothing}} <: AbstractBorderedLinearSolver
    solver::S = nothing
    tol::Float64 = 1e-12
    check_precision::Bool = true
    k::Int = 1
end
HipBorderingBLS(ls::HipGMRES) = HipBorderingBLS(solver = ls)
BK.update_bls(lbs::HipBorderingBLS, ls) = HipBorderingBLS(ls, lbs.tol, lbs.check_precision, lbs.k)   # :490-493

# dotp of the PALC call is NormalisedDot = dot/length (src/continuation/Palc.jl:1-6): pass it as `dotscale`.
_dotscale(dotp, x) = dotp isa BK.NormalisedDot ? 1.0 / length(x) : 1.0

function (lbs::HipBorderingBLS)(J::HipJacobian, dR::HipVec, dzu::HipVec, dzp::T, R::HipVec, n::T,
                                ξu::Tξ = 1.0, ξp::Tξ = 1.0; shift = nothing, dotp = nothing, applyξu! = nothing) where {T, Tξ}
    ctx = R.ctx
    dX = similar(R)
    dl, cv, it = Ref{Cdouble}(0), Ref{Cint}(0), zeros(Cint, 2)
    bo = BorderingOpts(lbs.tol, lbs.check_precision, lbs.k, 0)
    check(ctx, ccall((:bk_bls_bordering, libbkhip[]), Cint,
        (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cdouble}, Ptr{Cdouble}, Cdouble, Ptr{Cdouble}, Cdouble, Cdouble, Cdouble, Cint, Cdouble, Cdouble,
         Ref{BorderingOpts}, Ref{GmresOpts}, Ptr{Cvoid}, Ptr{Cdouble}, Ref{Cdouble}, Ref{Cint}, Ptr{Cint}),
        ctx.h, J.h, dR.p, dzu.p, dzp, R.p, n, ξu, ξp, isnothing(shift) ? 0 : 1, isnothing(shift) ? 0.0 : shift, _dotscale(dotp, R),
        Ref(bo), Ref(_opts(lbs.solver)), _plh(lbs.solver.Pl), dX.p, dl, cv, it), "bk_bls_bordering")
    return dX, dl[], cv[] == 1, (Int(it[1]), Int(it[2]))
end

# bls(J, a, b, 0, 0, 1; shift = Complex(0, -ω)) of src/codim2/MinAugHopf.jl:17, 72-76: complex border, one BEC pass
function (lbs::HipBorderingBLS)(J::HipJacobian, dR::HipCVec, dzu::HipCVec, dzp, R::HipCVec, n, ξu = 1.0, ξp = 1.0;
                                shift = nothing, dotp = nothing, applyξu! = nothing)
    ctx = R.re.ctx
    dX = HipCVec(similar(R.re), similar(R.re))
    dl, cv, it = zeros(Cdouble, 2), Ref{Cint}(0), zeros(Cint, 2)
    sh = isnothing(shift) ? ComplexF64(0) : ComplexF64(shift)
    zp, nn = ComplexF64(dzp), ComplexF64(n)
    check(ctx, ccall((:bk_bls_bordering_cshift, libbkhip[]), Cint,
        (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Cdouble, Cdouble, Ptr{Cdouble}, Ptr{Cdouble},
         Cdouble, Cdouble, Cdouble, Cdouble, Cdouble, Cdouble, Cdouble, Ref{GmresOpts}, Ptr{Cvoid}, Ptr{Cdouble}, Ptr{Cdouble},
         Ptr{Cdouble}, Ref{Cint}, Ptr{Cint}),
        ctx.h, J.h, dR.re.p, dR.im.p, dzu.re.p, dzu.im.p, real(zp), imag(zp), R.re.p, R.im.p, real(nn), imag(nn), ξu, ξp,
        real(sh), imag(sh), _dotscale(dotp, R.re), Ref(_opts(lbs.solver)), _plh(lbs.solver.Pl), dX.re.p, dX.im.p, dl, cv, it),
        "bk_bls_bordering_cshift")
    return dX, complex(dl[1], dl[2]), cv[] == 1, (Int(it[1]), Int(it[2]))
end

# m-column border (normal forms / Bogdanov-Takens), src/LinearBorderSolver.jl:173-206
function BK.solve_bls_block(lbs::HipBorderingBLS, J::HipJacobian, b::NTuple{M, HipVec}, c::NTuple{M, HipVec},
                            d::AbstractMatrix, rhst::HipVec, rhsb) where {M}
    m = size(d, 1)
    (length(b) == length(c) == m == M) || error("Linear bordered solver, wrong sizes!")
    ctx = rhst.ctx
    u1 = similar(rhst)
    bp = [x.p for x in b]; cp = [x.p for x in c]
    dd = collect(Cdouble, permutedims(d))                 # row-major d[i*m + j]
    u2 = zeros(Cdouble, m); its = zeros(Cint, m); cv = Ref{Cint}(0)
    check(ctx, ccall((:bk_bls_block_bordering, libbkhip[]), Cint,
        (Ptr{Cvoid}, Ptr{Cvoid}, Cint, Ptr{Ptr{Cdouble}}, Ptr{Ptr{Cdouble}}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble},
         Ref{GmresOpts}, Ptr{Cvoid}, Ptr{Cdouble}, Ptr{Cdouble}, Ref{Cint}, Ptr{Cint}),
        ctx.h, J.h, m, bp, cp, dd, rhst.p, collect(Cdouble, rhsb), Ref(_opts(lbs.solver)), _plh(lbs.solver.Pl),
        u1.p, u2, cv, its), "bk_bls_block_bordering")
    return u1, u2, cv[] == 1, Tuple(Int.(its))
end

"`MatrixFreeBLS` (src/LinearBorderSolver.jl:404-437): one GMRES on the (N+1) operator, border scalar on the host."
struct HipMatrixFreeBLS{S <: Union{HipGMRES, Nothing}} <: AbstractBorderedLinearSolver
    solver::S
end
HipMatrixFreeBLS() = HipMatrixFreeBLS(nothing)
BK.update_bls(::HipMatrixFreeBLS, ls) = HipMatrixFreeBLS(ls)
function (lbs::HipMatrixFreeBLS)(J::HipJacobian, dR::HipVec, dzu::HipVec, dzp::T, R::HipVec, n::T,
                                 ξu::Tξ = 1.0, ξp::Tξ = 1.0; shift = nothing, dotp = nothing, applyξu! = nothing) where {T, Tξ}
    ctx = R.ctx
    dX = similar(R)
    dl, cv, it = Ref{Cdouble}(0), Ref{Cint}(0), Ref{Cint}(0)
    check(ctx, ccall((:bk_bls_matrixfree, libbkhip[]), Cint,
        (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cdouble}, Ptr{Cdouble}, Cdouble, Ptr{Cdouble}, Cdouble, Cdouble, Cdouble, Cint, Cdouble, Cdouble,
         Ref{GmresOpts}, Ptr{Cdouble}, Ref{Cdouble}, Ref{Cint}, Ref{Cint}),
        ctx.h, J.h, dR.p, dzu.p, dzp, R.p, n, ξu, ξp, isnothing(shift) ? 0 : 1, isnothing(shift) ? 0.0 : shift, _dotscale(dotp, R),
        Ref(_opts(lbs.solver)), dX.p, dl, cv, it), "bk_bls_matrixfree")
    return dX, dl[], cv[] == 1, Int(it[])
end

# ------------------------------------------------------------------------------------------------ dF/dp
# The engine forms dF/dp = (F(x, p + ϵ) - F(x, p)) / ϵ from two residuals in exactly two places on the PALC path:
# newton_palc (src/continuation/Palc.jl:223-226, 239-240) and gettangent!(::Bordered) (src/continuation/Tangents.jl:77-82).
# With ϵ = 1.5e-8 the quotient of two stencil evaluations is white noise at 1e-8 relative -- ten times the GMRES
# tolerance of the examples; on a 256^3 / 512^3 grid that noise costs 117-228 operator applications per step instead of
# 20-25 (DESIGN.md section 7).  Every parameter of these PDEs multiplies a pointwise term, so the stencil part cancels
# identically and bk_residual_dparam evaluates the same quotient without the noise.  The two methods below are the
# engine's own methods specialised on the device problem type (nothing in BifurcationKit is edited): `continuation`
# runs unchanged and picks them up by dispatch.

# gettangent!(state, iter, ::Bordered, dotθ): src/continuation/Tangents.jl:71-104 with the routed quotient
function BK.gettangent!(state::BK.AbstractContinuationState, iter::HipContIterable, ::BK.Bordered, dotθ)
    (iter.verbosity > 0) && println("Predictor: Bordered (device dF/dp)")
    ϵ = BK.getdelta(iter.prob)
    τ = state.τ
    θ = BK.getθ(iter)
    T = eltype(iter)
    par = BK.setparam(iter, state.z.p)
    dFdl = residual_dparam(hipproblem(iter.prob), state.z.u, par, _ipar(BK.getparams(iter.prob), BK.getlens(iter)); eps = ϵ)
    J = BK.jacobian(iter.prob, state.z.u, par)
    τu, τp, flag, iterl = BK.solve_bls_palc(BK.get_bordered_linsolver(iter), iter, state, J, dFdl,
                                           VI.zerovector(state.z.u), one(T))
    ~flag && @warn "Linear solver failed to converge in tangent computation with type ::Bordered"
    α = one(T) / sqrt(dotθ(τu, τu, τp, τp, θ))
    α *= sign(dotθ(τ.u, τu, τ.p, τp, θ))
    BK._copyto!(τ.u, τu)
    τ.p = τp
    VI.scale!(τ, α)
end

# C-side view of the engine's Newton callback (bk_newton_callback, include/bkhip.h): the device pointers are wrapped as
# non-owning HipVec views and the engine's callback is called with the fields it documents (src/Newton.jl:151-159).
mutable struct _CbBox
    ctx::HipContext
    n::Int
    cb::Any
    z0::Any
    contparams::Any
    residuals::Vector{Float64}
    linsolver::Any
    kwargs::Any
end
_view(ctx, p::Ptr{Cdouble}, n) = HipVec(ctx, n, p, false)          # (ctx, n, pointer, owns = false)
function _newton_cb(user::Ptr{Cvoid}, x::Ptr{Cdouble}, fx::Ptr{Cdouble}, res::Cdouble, step::Cint, itlinear::Cint, p::Cdouble,
                    z0u::Ptr{Cdouble}, z0p::Cdouble, from_newton::Cint)::Cint
    b = unsafe_pointer_to_objref(user)::_CbBox
    step > 0 && push!(b.residuals, res)
    ok = b.cb((; x = _view(b.ctx, x, b.n), res_f = _view(b.ctx, fx, b.n), residual = res, step = Int(step),
                itlinear = Int(itlinear), contparams = b.contparams, z0 = b.z0, p = p, residuals = b.residuals,
                options = (; linsolver = b.linsolver)); fromNewton = from_newton != 0, b.kwargs...)
    return ok ? Cint(1) : Cint(0)
end

# newton_palc(iter, state, dotθ; normN, callback, kwargs...): src/continuation/Palc.jl:187-305 as ONE library call
# (bk_newton_palc: cancellation-free dF/dp, Jacobian handle, the bordered solve, update, clamping, line search :254-281,
# callback veto :235,294-297) for every combination the library covers: HipBorderingBLS or HipMatrixFreeBLS (bk_bordering_opts.
# kind) around any device linear solver, the standard DotTheta, norm or norminf.  Anything else (a custom dotθ or norm) is
# handed back to the engine's own method with `invoke` -- nothing of Palc.jl is restated here -- at the price of the engine's
# two-residual dF/dp quotient (DESIGN.md section 7).
function BK.newton_palc(iter::HipContIterable, state::BK.AbstractContinuationState, dotθ = BK.getdot(iter);
                        normN = LinearAlgebra.norm, callback = BK.cb_default, kwargs...)
    prob = iter.prob
    hp = hipproblem(prob)
    ctx = hp.ctx
    par = BK.getparams(prob)
    lens = BK.getlens(iter)
    contparams = BK.getcontparams(iter)
    θ = BK.getθ(iter)
    z0 = BK.getsolution(state)
    τ0 = state.τ
    (; z_pred, ds) = state
    (; tol, max_iterations, verbose, α, αmin, linesearch) = contparams.newton_options
    (; p_min, p_max) = contparams
    lbs = BK.get_bordered_linsolver(iter)
    native = (lbs isa HipBorderingBLS || lbs isa HipMatrixFreeBLS) && !isnothing(lbs.solver) && dotθ isa BK.DotTheta &&
             dotθ.dot isa BK.NormalisedDot && (normN === LinearAlgebra.norm || normN === BK.norminf)
    if !native
        @warn "BifurcationKitHIP: newton_palc falls back to the engine's generic loop (two-residual dF/dp)" maxlog = 1
        return invoke(BK.newton_palc, Tuple{BK.ContIterable, BK.AbstractContinuationState, Any}, iter, state, dotθ;
                      normN, callback, kwargs...)
    end
    x = BK._copy(z_pred.u)
    p = Ref{Cdouble}(z_pred.p)
    pv = Cdouble[Float64(v) for v in Tuple(BK.set(par, lens, z_pred.p))][1:hp.nparams]
    box = _CbBox(ctx, x.n, callback, z0, contparams, Float64[], lbs, kwargs)
    cfun = callback === BK.cb_default ? C_NULL :
           @cfunction(_newton_cb, Cint, (Ptr{Cvoid}, Ptr{Cdouble}, Ptr{Cdouble}, Cdouble, Cint, Cint, Cdouble, Ptr{Cdouble}, Cdouble, Cint))
    no = NewtonOpts(tol, max_iterations, normN === BK.norminf ? 1 : 0, linesearch ? 1 : 0, α, αmin, 0.0, cfun,
                    callback === BK.cb_default ? C_NULL : pointer_from_objref(box))
    bo = lbs isa HipMatrixFreeBLS ? BorderingOpts(0.0, 0, 1, 1) : BorderingOpts(lbs.tol, lbs.check_precision, lbs.k, 0)
    res = Ref(NewtonResult())
    GC.@preserve box begin
        check(ctx, ccall((:bk_newton_palc, libbkhip[]), Cint,
            (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cdouble}, Ref{Cdouble}, Ptr{Cdouble}, Cdouble, Ptr{Cdouble}, Cdouble, Cdouble, Cdouble,
             Ptr{Cdouble}, Cint, Cint, Cdouble, Cdouble, Ref{NewtonOpts}, Ref{BorderingOpts}, Ref{GmresOpts}, Ptr{Cvoid}, Ref{NewtonResult}),
            ctx.h, hp.h, x.p, p, z0.u.p, z0.p, τ0.u.p, τ0.p, ds, θ, pv, length(pv), _ipar(par, lens),
            max(p_min, -1.7e308), min(p_max, 1.7e308), Ref(no), Ref(bo), Ref(_opts(lbs.solver)), _plh(lbs.solver.Pl), res),
            "bk_newton_palc")
    end
    r = res[]
    residuals = Float64[r.residuals[i] for i in 1:(r.itnewton + 1)]
    verbose && foreach(i -> BK.print_nonlinear_step(i - 1, residuals[i]), eachindex(residuals))
    return BK.NonLinearSolution(BK.BorderedArray(x, p[]), prob, residuals, r.converged == 1, Int(r.itnewton), Int(r.itlinear))
end

# ------------------------------------------------------------------------------------------------ eigensolver
"""
    HipShiftInvert(σ, ls; tol = 1e-12, maxiter = 20, hermitian = false)

`ShiftInvert` (src/EigSolver.jl:246-266) with a Krylov-Schur outer iteration; the `SH3dEig` of
examples/SH3d.jl:96-113.  Returns `(vals::Vector{ComplexF64}, vecs::Vector{HipVec}, converged, numops)`,
eigenvalues sorted by decreasing real part.
"""
Base.@kwdef struct HipShiftInvert <: AbstractEigenSolver
    σ::Float64
    ls::HipGMRES
    tol::Float64 = 1e-12
    maxiter::Int = 20
    hermitian::Bool = false
    seed::UInt64 = 1234
    x₀::Union{Nothing, HipVec} = nothing      # start vector (EigKrylovKit.x₀, src/EigSolver.jl:143); nothing -> rand(N), SH3d.jl:109
end
BK.geteigenvector(::HipShiftInvert, vecs, n::Union{Int, AbstractVector{Int64}}) = vecs[n]     # like SH3dEig, SH3d.jl:101

function (e::HipShiftInvert)(J::HipJacobian, nev::Int; kwargs...)
    ctx = J.prob.ctx
    kd = min(max(30, nev + 30), 63)                                   # SH3d.jl:109
    re, im = zeros(Cdouble, nev + 1), zeros(Cdouble, nev + 1)      # nev + 1: a complex pair is never split
    n = J.x.n
    ld = cld(n, 32) * 32
    buf = HipVec(ctx, ld * (nev + 1))
    nvals, nconv, nops = Ref{Cint}(0), Ref{Cint}(0), Ref{Cint}(0)
    eo = EigOpts(e.σ, kd, e.maxiter, e.tol, e.hermitian, e.seed)
    isnothing(e.x₀) || check(ctx, ccall((:bk_eig_set_start_vector, libbkhip[]), Cint, (Ptr{Cvoid}, Ptr{Cdouble}), ctx.h, e.x₀.p), "bk_eig_set_start_vector")
    check(ctx, ccall((:bk_eig_shiftinvert, libbkhip[]), Cint,
        (Ptr{Cvoid}, Ptr{Cvoid}, Cint, Ref{EigOpts}, Ref{GmresOpts}, Ptr{Cvoid}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble},
         Csize_t, Ref{Cint}, Ref{Cint}, Ref{Cint}),
        ctx.h, J.h, nev, Ref(eo), Ref(_opts(e.ls)), _plh(e.ls.Pl), re, im, buf.p, C_NULL, ld, nvals, nconv, nops), "bk_eig_shiftinvert")
    m = Int(nvals[])
    vecs = [(v = HipVec(ctx, n); ccall((:bk_vec_copy, libbkhip[]), Cint, (Ptr{Cvoid}, Csize_t, Ptr{Cdouble}, Ptr{Cdouble}),
                                       ctx.h, n, buf.p + (i - 1) * ld * sizeof(Cdouble), v.p); v) for i in 1:m]
    return Complex.(re[1:m], im[1:m]), vecs, nconv[] >= nev, Int(nops[])
end

# ------------------------------------------------------------------------------------------------ usage
# ctx  = HipContext(0)
# prob = SwiftHohenberg(ctx, (512, 512, 512), (16π, 16π, 16π))
# Pl   = HipDCTPreconditioner(prob, 1.0)
# ls   = HipGMRES(rtol = 1e-9, maxiter = 150, Pl = Pl)                         # examples/SH3d.jl:93
# bp   = bifurcation_problem(prob, HipVec(ctx, vec(sol0)), (l = 0.1, ν = 1.2), (@optic _.l); issymmetric = true)
# optn = NewtonPar(tol = 1e-8, max_iterations = 20, linsolver = ls, eigsolver = HipShiftInvert(σ = 0.1, ls = ls, hermitian = true))
# br   = continuation(bp, PALC(tangent = Bordered(), bls = HipBorderingBLS(solver = ls, check_precision = false)),
#                     ContinuationPar(dsmax = 0.005, ds = -0.001, newton_options = optn, nev = 15); normC = norminf)

end # module
