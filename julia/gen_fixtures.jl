# gen_fixtures.jl -- reference-pinned fixtures for the parity tests (VERDICT r1, item 5).
#
# Runs the REAL BifurcationKit.jl (+ KrylovKit / Arpack, whatever versions the environment resolves; they are written
# into the fixture) on the three problems of this repository's hot path and dumps what the reference computes at the
# plugin boundary: residuals / JVPs, GMRESKrylovKit solves (incl. the Pl + shift branch), GMRESIterativeSolvers with Pl and Pr,
# KrylovLS(:minres / :cg), BorderingBLS and MatrixFreeBLS solves, Newton and newton_palc histories along a short PALC branch (with
# BorderingBLS and with MatrixFreeBLS as the corrector's bordered solver), shift-invert eigenvalues.  The `numops` / `niter` counters
# are what the consumers hold the HIP solvers to for EVERY block size of its Arnoldi process (gmres_sstep = 0 and 4).
#
#     julia --project=<env with BifurcationKit> julia/gen_fixtures.jl        # writes tests/golden/julia_fixtures.json
#
# There is no Julia in the build container of this repository: this script has NOT been executed there.  Until someone
# runs it, tests/test_reference_fixtures.py skips and the oracle stays "parity unpinned" (DESIGN.md section 1).
#
# Inputs are closed-form (no RNG, no input files): probe(k, N)[i] = sin(a_k i) + 0.5 cos(b_k i + 0.1), i = 1..N -- the
# Python side (tests/test_reference_fixtures.py: probe) builds the same vectors with i = index0 + 1.  Vectors are not
# dumped whole: each one is summarised by (norm2, norminf, <v, probe(9)>, 6 entries at fixed 1-based positions).
using BifurcationKit, LinearAlgebra, SparseArrays, Printf
import KrylovKit
const BK = BifurcationKit

# ------------------------------------------------------------------------------------------------ helpers
probe(k, N) = [sin((0.37 + 0.11k) * i) + 0.5cos((1.3 + 0.07k) * i + 0.1) for i in 1:N]
positions(N) = [1, 2, 17, N ÷ 3, N ÷ 2 + 1, N]
summary_of(v) = Dict("norm2" => norm(v), "norminf" => norm(v, Inf), "dot_probe9" => dot(v, probe(9, length(v))),
                     "entries" => [v[i] for i in positions(length(v))])

# second difference on N points of [-l, l) (right end excluded, h = 2l/N); Neumann-ghost (corner entries -1/h^2) or
# Dirichlet (missing neighbour = 0) -- the two boundary closures of the examples
function second_difference(N, l; neumann)
    h = 2l / N
    D = spdiagm(0 => fill(-2.0, N), 1 => ones(N - 1), -1 => ones(N - 1)) / h^2
    if neumann
        D[1, 1] = -1 / h^2
        D[N, N] = -1 / h^2
    end
    return sparse(D)
end
eye(n) = sparse(1.0I, n, n)
# flat index = i + Nx (j + Ny k), x fastest
function laplacian(dims, ls; neumann)
    Ds = [second_difference(n, l; neumann) for (n, l) in zip(dims, ls)]
    if length(dims) == 2
        return kron(eye(dims[2]), Ds[1]) + kron(Ds[2], eye(dims[1]))
    end
    return kron(eye(dims[3]), kron(eye(dims[2]), Ds[1]) + kron(Ds[2], eye(dims[1]))) + kron(Ds[3], eye(dims[1] * dims[2]))
end
axes_of(dims, ls) = [[-l + 2l / n * (i - 1) for i in 1:n] for (n, l) in zip(dims, ls)]

# tiny JSON writer (no package dependency)
json(x::AbstractString) = "\"" * escape_string(x) * "\""
json(x::Bool) = x ? "true" : "false"
json(x::Integer) = string(x)
json(x::Real) = isfinite(x) ? @sprintf("%.17g", x) : "null"
json(x::Complex) = json([real(x), imag(x)])
json(x::Union{AbstractVector, Tuple}) = "[" * join((json(v) for v in x), ", ") * "]"
json(x::AbstractDict) = "{" * join((json(string(k)) * ": " * json(v) for (k, v) in sort(collect(x), by = first)), ",\n ") * "}"

# GMRESKrylovKit applies Pl through ldiv!(out, Pl, rhs) (src/LinearSolver.jl:278); CHOLMOD factors need this method
# (the reference example adds the same one, examples/SH3d.jl:89-90)
LinearAlgebra.ldiv!(o::Vector{Float64}, P::SparseArrays.CHOLMOD.Factor, v::Vector{Float64}) = (o .= P \ v)

out = Dict{String, Any}()
out["versions"] = Dict("julia" => string(VERSION),
                       "BifurcationKit" => string(pkgversion(BK)), "KrylovKit" => string(pkgversion(KrylovKit)))

# ------------------------------------------------------------------------------------------------ Swift-Hohenberg
F_sh(u, p) = -(p.L1 * u) .+ (p.l .* u .+ p.ν .* u .^ 2 .- u .^ 3)
dF_sh(u, p, du) = -(p.L1 * du) .+ (p.l .+ 2 .* p.ν .* u .- 3 .* u .^ 2) .* du

function sh_case(name, dims, ls, l, ν, guess; newton_tol = 1e-8, branch_steps = 4)
    N = prod(dims)
    L1 = (eye(N) + laplacian(dims, ls; neumann = true))^2
    par = (l = l, ν = ν, L1 = L1)
    u0 = vec(guess)
    c = Dict{String, Any}("dims" => collect(dims), "ls" => collect(ls), "l" => l, "nu" => ν)
    du = probe(1, N)
    c["F_u0"] = summary_of(F_sh(u0, par))
    c["dF_u0_probe1"] = summary_of(dF_sh(u0, par, du))
    # GMRESKrylovKit with the exact left preconditioner of the example (sparse Cholesky of L1)
    Pl = cholesky(Symmetric(L1))
    ls_ = GMRESKrylovKit(verbose = 0, rtol = 1e-9, maxiter = 150, ishermitian = true, Pl = Pl)
    prob = BifurcationProblem(F_sh, u0, par, (@optic _.l); J = (x, p) -> (dx -> dF_sh(x, p, dx)), issymmetric = true)
    optn = NewtonPar(verbose = false, tol = newton_tol, max_iterations = 20, linsolver = ls_)
    sol = BK.solve(prob, Newton(), optn; normN = x -> norm(x, Inf))
    c["newton"] = Dict("converged" => BK.converged(sol), "residuals" => sol.residuals, "itnewton" => sol.itnewton,
                       "itlineartot" => sol.itlineartot, "u" => summary_of(sol.u))
    us = sol.u
    J = dx -> dF_sh(us, par, dx)
    r1, r2, r3 = probe(1, N), probe(2, N), probe(3, N)
    x, ok, it = ls_(J, r1)
    c["gmres"] = Dict("converged" => ok, "numops" => it, "x" => summary_of(x))
    x, ok, it = ls_(J, r1; a₀ = 0.3, a₁ = 0.9)                   # the Pl + shift branch, src/LinearSolver.jl:268-288
    c["gmres_shift"] = Dict("a0" => 0.3, "a1" => 0.9, "converged" => ok, "numops" => it, "x" => summary_of(x))
    # KrylovLS(:minres) / (:cg): Krylov.jl's symmetric solvers with the same factor as the "centered" preconditioner M
    # (src/LinearSolver.jl:336-341; `ldiv = true` as in the commented line of examples/SH3d.jl:94) -- the inner solver of
    # config 5 on the HIP side, with two of its passes fused into the stencil / transform kernels
    try
        kmr = KrylovLS(KrylovAlg = :minres, Pl = Pl, rtol = 1e-10, atol = 1e-13, ldiv = true)
        x, ok, it = kmr(J, r1; a₀ = -0.1, a₁ = 1.0)
        c["minres"] = Dict("a0" => -0.1, "a1" => 1.0, "converged" => ok, "niter" => it, "x" => summary_of(x))
        kcg = KrylovLS(KrylovAlg = :cg, Pl = Pl, rtol = 1e-10, atol = 1e-13, ldiv = true)
        x, ok, it = kcg(J, r1; a₀ = 2.0, a₁ = -1.0)           # -(J - 2I) is positive definite
        c["cg"] = Dict("a0" => 2.0, "a1" => -1.0, "converged" => ok, "niter" => it, "x" => summary_of(x))
    catch err
        c["minres_error"] = sprint(showerror, err)
    end
    dotp = (a, b) -> dot(a, b) / N
    bls = BorderingBLS(solver = ls_, check_precision = false)
    dX, dl, ok, its = bls(J, r2, r3, 0.4, r1, 0.3, 0.5, 0.5; dotp = dotp)
    c["bordering"] = Dict("converged" => ok, "itlinear" => collect(its), "dl" => dl, "dX" => summary_of(dX))
    mf = MatrixFreeBLS(GMRESKrylovKit(verbose = 0, rtol = 1e-9, maxiter = 150))
    dX, dl, ok, its = mf(J, r2, r3, 0.4, r1, 0.3, 0.5, 0.5; dotp = dotp)
    c["matrixfree"] = Dict("converged" => ok, "itlinear" => sum(its), "dl" => dl, "dX" => summary_of(dX))
    # GMRESIterativeSolvers with a left AND a right preconditioner (src/LinearSolver.jl:178,198-201): the iteration runs on
    # Pl^-1 (a0 + a1 J) Pr^-1 y = Pl^-1 rhs, x = Pr^-1 y.  Pl = lu(L1 + I), Pr = lu(L1 + 1e5 I) (bk_gmres_opts.pr on the HIP side; the
    # large shift keeps Pl^-1 A Pr^-1 well conditioned -- 20-30 iterations -- while Pr still is a genuine, non-scalar operator)
    try
        lsis = GMRESIterativeSolvers(reltol = 1e-10, restart = 63, maxiter = 4000, N = N, Pl = lu(L1 + I), Pr = lu(L1 + 1e5I))
        x, ok, it = lsis(J, r1; a₀ = 0.4, a₁ = -1.0)
        c["gmres_is_pr"] = Dict("a0" => 0.4, "a1" => -1.0, "reltol" => 1e-10, "restart" => 63, "pl_shift" => 1.0, "pr_shift" => 1e5,
                                "converged" => ok, "niter" => it, "x" => summary_of(x))
    catch err
        c["gmres_is_pr_error"] = sprint(showerror, err)
    end
    # shift-invert eigenvalues the way the example's user-defined eigensolver does it (sigma = 0.1, KrylovKit.eigsolve)
    σ = 0.1
    A = dx -> ls_(v -> J(v) .- σ .* v, dx)[1]
    vals, _, info = KrylovKit.eigsolve(A, probe(4, N), 6, :LM; tol = 1e-10, maxiter = 40, ishermitian = true, krylovdim = 36)
    λ = sort(real.(1 ./ vals .+ σ), rev = true)
    c["shift_invert"] = Dict("sigma" => σ, "converged" => info.converged, "numops" => info.numops, "vals" => λ[1:min(6, end)])
    # a short PALC branch with the example's settings (Bordered tangent, BorderingBLS without the precision check)
    optc = ContinuationPar(dsmin = 1e-4, dsmax = 0.005, ds = -0.001, p_max = 0.15, p_min = -0.1,
                           newton_options = NewtonPar(optn; tol = 1e-9, max_iterations = 15), max_steps = branch_steps,
                           detect_bifurcation = 0, save_sol_every_step = 0)
    probb = BK.re_make(prob; u0 = us)
    br = continuation(probb, PALC(tangent = Bordered(), bls = BorderingBLS(solver = ls_, check_precision = false)), optc;
                      normC = x -> norm(x, Inf), verbosity = 0)
    c["branch"] = Dict("param" => br.param, "itnewton" => br.itnewton, "itlinear" => br.itlinear, "ds" => br.ds)
    # the same branch with MatrixFreeBLS as the corrector's bordered solver (src/LinearBorderSolver.jl:424-437: one unpreconditioned
    # GMRES on the (N + 1) operator per Newton iteration) -- bk_bordering_opts.kind = 1 on the HIP side, one library call per corrector
    try
        mfc = MatrixFreeBLS(GMRESKrylovKit(verbose = 0, dim = 60, rtol = 1e-10, atol = 1e-13, maxiter = 300))
        optm = ContinuationPar(optc; max_steps = 2)
        brm = continuation(probb, PALC(tangent = Bordered(), bls = mfc), optm; normC = x -> norm(x, Inf), verbosity = 0)
        c["branch_matrixfree"] = Dict("param" => brm.param, "itnewton" => brm.itnewton, "itlinear" => brm.itlinear)
    catch err
        c["branch_matrixfree_error"] = sprint(showerror, err)
    end
    out[name] = c
end

let dims = (22, 22, 22), ls = (π, π, π)
    X, Y, Z = axes_of(dims, ls)
    s = [cos(x) * cos(y) + 0z for x in X, y in Y, z in Z]
    s .-= minimum(s); s ./= maximum(s); s .*= 1.2
    sh_case("sh3d_22", dims, ls, 0.1, 1.2, s)
end
# round 6: the same pairing (Pl = cholesky(L1), shift 0) on grids where the library's DEFAULT Arnoldi step is the stencil-free one
# (power-of-two extents >= 64: the x passes run as the fused LDS kernel) -- the reference example's own box at 64^3 and a 2-D grid.
# The 3-D factorisation is the expensive one (262 144 unknowns, 25-point stencil): wrapped, so that a machine that cannot hold it
# still emits the other cases.
try
    let dims = (64, 64, 64), ls = (π, π, π)
        X, Y, Z = axes_of(dims, ls)
        s = [cos(x) * cos(y) + 0z for x in X, y in Y, z in Z]
        s .-= minimum(s); s ./= maximum(s); s .*= 1.2
        sh_case("sh3d_64", dims, ls, 0.1, 1.2, s; branch_steps = 2)
    end
catch err
    out["sh3d_64_error"] = sprint(showerror, err)
end
let dims = (128, 64), ls = (12.5, 6.0)
    X, Y = axes_of(dims, ls)
    s = [cos(x) + cos(x / 2) * cos(sqrt(3) * y / 2) for x in X, y in Y]
    s .-= minimum(s); s ./= maximum(s); s .-= 0.25; s .*= 1.7
    sh_case("sh2d_128x64", dims, ls, -0.1, 1.3, s; branch_steps = 2)
end
let dims = (151, 100), ls = (8π, 4π / sqrt(3))
    X, Y = axes_of(dims, ls)
    s = [cos(x) + cos(x / 2) * cos(sqrt(3) * y / 2) for x in X, y in Y]
    s .-= minimum(s); s ./= maximum(s); s .-= 0.25; s .*= 1.7
    sh_case("sh2d_151x100", dims, ls, -0.1, 1.3, s; branch_steps = 3)
end

# ------------------------------------------------------------------------------------------------ cGL 2-D
function Fcgl(u, p)
    n = length(u) ÷ 2
    u1, u2 = u[1:n], u[n+1:end]
    ua = u1 .^ 2 .+ u2 .^ 2
    f1 = p.r .* u1 .- p.ν .* u2 .- ua .* (p.c3 .* u1 .- p.μ .* u2) .- p.c5 .* ua .^ 2 .* u1 .+ p.γ
    f2 = p.r .* u2 .+ p.ν .* u1 .- ua .* (p.c3 .* u2 .+ p.μ .* u1) .- p.c5 .* ua .^ 2 .* u2
    return p.Δ * u .+ vcat(f1, f2)
end
function Jcgl(u, p)
    n = length(u) ÷ 2
    u1, u2 = u[1:n], u[n+1:end]
    ua = u1 .^ 2 .+ u2 .^ 2
    f1u = p.r .- 2 .* u1 .* (p.c3 .* u1 .- p.μ .* u2) .- p.c3 .* ua .- 4 .* p.c5 .* ua .* u1 .^ 2 .- p.c5 .* ua .^ 2
    f1v = -p.ν .- 2 .* u2 .* (p.c3 .* u1 .- p.μ .* u2) .+ p.μ .* ua .- 4 .* p.c5 .* ua .* u1 .* u2
    f2u = p.ν .- 2 .* u1 .* (p.c3 .* u2 .+ p.μ .* u1) .- p.μ .* ua .- 4 .* p.c5 .* ua .* u1 .* u2
    f2v = p.r .- 2 .* u2 .* (p.c3 .* u2 .+ p.μ .* u1) .- p.c3 .* ua .- 4 .* p.c5 .* ua .* u2 .^ 2 .- p.c5 .* ua .^ 2
    return p.Δ + spdiagm(0 => vcat(f1u, f2v), n => f1v, -n => f2u)
end
let dims = (41, 21), ls = (π, π / 2)
    n = prod(dims)
    lap = laplacian(dims, ls; neumann = false)
    par = (r = 0.5, μ = 0.1, ν = 1.0, c3 = -1.0, c5 = 1.0, Δ = blockdiag(lap, lap), γ = 0.0)
    c = Dict{String, Any}("dims" => collect(dims), "ls" => collect(ls))
    u = 0.4 .* probe(5, 2n)
    du = probe(6, 2n)
    pr = merge(par, (r = 1.2,))
    c["F_probe"] = summary_of(Fcgl(u, pr))
    c["J_probe_du"] = summary_of(Jcgl(u, pr) * du)
    prob = BifurcationProblem(Fcgl, zeros(2n), par, (@optic _.r); J = Jcgl)
    eigls = EigArpack(1.0, :LM)
    vals, _, cv, _ = eigls(Jcgl(zeros(2n), par), 9)
    c["eig_trivial_r0.5"] = Dict("converged" => cv, "vals" => collect(vals))
    optn = NewtonPar(tol = 1e-9, verbose = false, eigsolver = eigls, max_iterations = 20)
    optc = ContinuationPar(dsmin = 0.001, dsmax = 0.15, ds = 0.001, p_max = 2.5, detect_bifurcation = 3, nev = 9,
                           newton_options = optn, max_steps = 60, n_inversion = 6, save_sol_every_step = 0)
    br = continuation(prob, PALC(), optc; verbosity = 0, normC = x -> norm(x, Inf))
    c["branch"] = Dict("param" => br.param, "n_unstable" => br.n_unstable, "n_imag" => br.n_imag,
                       "specialpoint" => [Dict("type" => string(sp.type), "param" => sp.param, "status" => string(sp.status),
                                               "interval" => collect(sp.interval), "step" => sp.step)
                                          for sp in br.specialpoint])
    out["cgl_41x21"] = c
end

target = joinpath(@__DIR__, "..", "tests", "golden", "julia_fixtures.json")
open(target, "w") do io
    write(io, json(out))
    write(io, "\n")
end
println("wrote ", target)
