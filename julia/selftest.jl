# One-command self-test of the binding for a maintainer with Julia + BifurcationKit + an MI355X:
#     julia --project --check-bounds=yes julia/selftest.jl [path/to/libbkhip.so]
# It checks what cannot be checked in the build container of this repository (no Julia there): that the file parses and loads,
# that the struct layouts match the C ABI, and -- the part that silently degrades if it breaks -- that the DISPATCH this binding
# relies on still resolves to its own methods:
#   * bifurcation_problem(...) yields a HipBifProblem (the reference constructor keeps the callable struct HipResidualFn as
#     the type parameter of BifFunction: src/Problems.jl:463-468),
#   * BK.newton_palc / BK.gettangent!(::Bordered) on a ContIterable of that problem resolve to the methods of this file (else
#     `continuation` silently takes the engine's two-residual dF/dp quotient again, DESIGN.md section 7),
#   * one corrector through `continuation` runs as ONE bk_newton_palc call and agrees with the engine's generic loop.
include(joinpath(@__DIR__, "BifurcationKitHIP.jl"))
using .BifurcationKitHIP, BifurcationKit, Test
const BK = BifurcationKit
const H = BifurcationKitHIP

length(ARGS) >= 1 && (H.libbkhip[] = ARGS[1])

@testset "struct layouts match include/bkhip.h" begin
    @test ccall((:bk_abi_version, H.libbkhip[]), Cint, ()) == H.BK_ABI_VERSION        # the library's layout version (round 5)
    @test sizeof(H.GmresOpts) == 40 && fieldoffset(H.GmresOpts, 6) == 32            # flavor dim maxiter | atol rtol | pr
    @test sizeof(H.BorderingOpts) == 24 && fieldoffset(H.BorderingOpts, 4) == 16      # tol | check_precision k kind (+pad)
    @test sizeof(H.ProblemDesc) == 48
end

ctx = H.HipContext(0)
dims, ls = (32, 32, 32), (pi, pi, pi)
prob = H.SwiftHohenberg(ctx, dims, Float64.(ls))
u0 = H.HipVec(ctx, Float64[0.5 * (cos(x) + cos(x / 2) * cos(sqrt(3) * y / 2)) for x in range(-pi, pi; length = 33)[1:32],
                                                                                y in range(-pi, pi; length = 33)[1:32], z in 1:32][:])
par = (l = 0.1, ν = 1.2)
bp = H.bifurcation_problem(prob, u0, par, (BK.@optic _.l))

@testset "dispatch" begin
    @test bp isa H.HipBifProblem
    P = H.HipDCTPreconditioner(prob, 1.0)
    ls_ = H.HipGMRES(dim = 30, rtol = 1e-9, atol = 1e-12, maxiter = 150, Pl = P)
    opts = BK.ContinuationPar(ds = -0.001, dsmin = 1e-4, dsmax = 0.005, p_min = -0.1, p_max = 0.15, max_steps = 2, nev = 3,
                              detect_bifurcation = 0, newton_options = BK.NewtonPar(tol = 1e-9, max_iterations = 15, linsolver = ls_))
    alg = BK.PALC(tangent = BK.Bordered(), bls = H.HipBorderingBLS(solver = ls_, check_precision = false))
    it = BK.ContIterable(bp, alg, opts)
    @test it isa H.HipContIterable
    st = iterate(it)[1]
    m1 = which(BK.newton_palc, Tuple{typeof(it), typeof(st), typeof(BK.getdot(it))})
    m2 = which(BK.gettangent!, Tuple{typeof(st), typeof(it), BK.Bordered, typeof(BK.getdot(it))})
    @test m1.module === H && m2.module === H
    # the same corrector through the override (ONE bk_newton_palc call) and through the engine's own loop
    a = BK.newton_palc(it, st)
    b = invoke(BK.newton_palc, Tuple{BK.ContIterable, BK.AbstractContinuationState, Any}, it, st, BK.getdot(it))
    @test a.converged && b.converged && abs(a.u.p - b.u.p) <= 1e-9
    # MatrixFreeBLS takes the one-call path too (bk_bordering_opts.kind = 1)
    it2 = BK.ContIterable(bp, BK.PALC(tangent = BK.Bordered(), bls = H.HipMatrixFreeBLS(ls_)), opts)
    c = BK.newton_palc(it2, iterate(it2)[1])
    @test c.converged && abs(c.u.p - a.u.p) <= 1e-8
    br = BK.continuation(bp, alg, opts)
    @test length(br) >= 2
end
println("BifurcationKitHIP self-test passed")
