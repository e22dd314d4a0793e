/*
 * bkhip.h -- C ABI of the MI355X-native Newton-Krylov corrector for BifurcationKit.jl's
 * pseudo-arclength continuation (PALC).
 *
 * This is the drop-in boundary: plain pointers and sizes, no torch / C++ types.  Each entry
 * point cites the reference interface (paths relative to the BifurcationKit.jl checkout) it
 * replaces.  Julia reaches it with `ccall` (julia/BifurcationKitHIP.jl); the
 * parity tests reach it with Python `ctypes` using the same call sequence.
 *
 * Conventions
 *   - every function returns an int status: 0 = ok, <0 = error (bk_last_error() has the text).
 *     Non-convergence of an iterative solver is NOT an error: it is reported through the
 *     `converged` out-parameter, like the reference which only `@debug`s it
 *     (src/Newton.jl:93, src/LinearSolver.jl:289).
 *   - `double*` vector arguments are DEVICE pointers (HIP) to fp64 data of the problem's LOCAL
 *     length (bk_problem_nlocal); scalars travel by value / host pointers.
 *   - all work is enqueued on the context's HIP stream; functions that return host scalars
 *     synchronise that stream before returning, the others are asynchronous.
 *   - state layout: flat index = i + Nx*(j + Ny*k), x fastest -- the order the reference's
 *     `kron` assembly and `vec` of an [x,y,z] comprehension produce (examples/SH3d.jl:38-39,77).
 *     Multi-GPU (3-D Swift-Hohenberg): z-slabs of contiguous planes, one slab per rank.
 */
#ifndef BKHIP_H
#define BKHIP_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct bk_ctx bk_ctx;         /* device, stream, scratch, (optional) communicator        */
typedef struct bk_problem bk_problem; /* a stencil PDE on a grid: residual F(u,p) and J(u,p)      */
typedef struct bk_op bk_op;           /* a linear operator handle = what `jacobian(prob,x,p)` returns */
typedef struct bk_precond bk_precond; /* a left preconditioner Pl (GMRESKrylovKit.Pl)              */

/* ------------------------------------------------------------------ context ---------------- */

#define BK_UNIQUE_ID_BYTES 128

int bk_version(void);
/* Layout version of the option structs below (bk_gmres_opts, bk_bordering_opts, bk_eig_opts, ...).  A binding compares it with
 * the BK_ABI_VERSION it was written against before its first call: the structs carry no size field, and a client built against
 * an older header would hand over shorter ones.                                                                  */
#define BK_ABI_VERSION 6
int bk_abi_version(void);
/* Create a single-GPU context on `device`; `stream` is the hipStream_t all work is enqueued on
 * (NULL = the default stream, which orders the library with the caller's own default-stream work). */
int bk_ctx_create(bk_ctx** ctx, int device, void* stream);
/* Multi-GPU context: one process per GPU, RCCL communicator built from a unique id that rank 0
 * obtained with bk_comm_unique_id() and broadcast out-of-band (torch.distributed / MPI).
 * No reference counterpart: the reference is single-process.             */
int bk_comm_unique_id(void* id128);
int bk_ctx_create_dist(bk_ctx** ctx, int device, void* stream, int rank, int nranks,
                       const void* id128);
/* Test-only communicator: collectives staged through host buffers and user callbacks (lets two
 * ranks share one GPU, or run over gloo).  allreduce(user, buf, n, op) with op 0=sum 1=max acts
 * in place on a host buffer; sendrecv(user, sendbuf, nsend, dst, recvbuf, nrecv, src) exchanges
 * host buffers with neighbour ranks (dst/src = -1: no peer).                                     */
typedef int (*bk_allreduce_fn)(void* user, double* buf, int n, int op);
typedef int (*bk_sendrecv_fn)(void* user, const double* sendbuf, size_t nsend, int dst,
                              double* recvbuf, size_t nrecv, int src);
int bk_ctx_create_hostcomm(bk_ctx** ctx, int device, void* stream, int rank, int nranks,
                           bk_allreduce_fn allreduce, bk_sendrecv_fn sendrecv, void* user);
/* THREADING of the callbacks.  A host-staged context ENQUEUES its collectives in the stream like an RCCL context does
 * (device-resident Arnoldi chunks, halo exchange under the interior z-chunks): each one is a stream-ordered hand-over to a
 * proxy thread the library owns, and the callbacks run ON THAT THREAD, never on the caller's (a Julia client needs
 * @cfunction callbacks that may be entered from a foreign thread: Julia >= 1.9).  `user` is passed through untouched.
 * Second lane (option two_lanes, solver: ls(J, rhs1, rhs2) with both solves in flight): its collectives are issued
 * concurrently with the context's own and must never be matched against them, so it needs a communicator of its own:
 * register one with bk_ctx_set_lane_comm (same callback types, its own `user`; e.g. a second gloo / MPI group).  Without
 * a registered lane communicator a host-staged context simply runs one lane.  RCCL contexts split their communicator
 * themselves (ncclCommSplit).  No reference counterpart.                                                              */
int bk_ctx_set_lane_comm(bk_ctx* ctx, bk_allreduce_fn allreduce, bk_sendrecv_fn sendrecv, void* user);
/* What the communicator of this context is: *kind 0 = none, 1 = RCCL, 2 = host-staged test communicator; *rank /
 * *nranks as the communicator itself reports them (RCCL: ncclCommUserRank / ncclCommCount -- the number of ranks RCCL
 * actually connected, bench.py prints it next to a multi-GPU number).  No reference counterpart.                    */
int bk_comm_info(bk_ctx* ctx, int* kind, int* rank, int* nranks);
/* Time the two collectives of the hot path on this context's communicator (collective call: every rank must make it):
 * what 0 = in-stream all-reduce (sum) of `count` doubles (count <= 256: the projections of one Arnoldi step),
 * what 1 = halo exchange of `count` doubles with each z-neighbour (ncclSend/ncclRecv group).  `reps` back-to-back calls
 * between two stream synchronisations; *us_per_call = host wall time / reps.  No reference counterpart.               */
int bk_comm_probe(bk_ctx* ctx, int what, size_t count, int reps, double* us_per_call);
int bk_ctx_destroy(bk_ctx* ctx);
const char* bk_last_error(bk_ctx* ctx);
int bk_ctx_sync(bk_ctx* ctx);
/* Tuning knobs ("sh_kernel": 0 gather / 1 streaming; "sh_zchunk"; "dgks_eta"; "dct_fft"; "dct_roundtrip";
 * "dct_gemm"; "halo_overlap"; ...): experiments and cross-checks, the defaults are the measured optimum.        */
int bk_ctx_set_option(bk_ctx* ctx, const char* key, double value);
int bk_ctx_get_option(bk_ctx* ctx, const char* key, double* value);
/* Per-kernel timing with HIP events on the context's stream (bench.py's roofline leg):
 * names: "jvp", "residual", "multidot", "multiaxpy", "dct_pass", "blas1", "combine"; multi-GPU also "halo",
 * "alltoall", "transpose".                                                                       */
int bk_prof_enable(bk_ctx* ctx, int on);
int bk_prof_reset(bk_ctx* ctx);
int bk_prof_get(bk_ctx* ctx, const char* name, double* total_ms, long long* calls,
                double* alg_bytes);

/* Diagnostics: with option "solver_trace" != 0 every Krylov linear solve appends its residual-estimate history (one
 * value per iteration: GMRES |y_k+1|, MINRES phibar) to a per-context log; a negative entry -(k) opens the k-th solve
 * and is followed by the initial residual norm.  Copies up to cap entries into buf (may be NULL), *n = entries
 * available; reset != 0 clears the log.  The reference's counterpart is the `verbose` / `log = true` switch of the
 * Krylov packages (src/LinearSolver.jl:169,202,244).                                                              */
int bk_solver_history(bk_ctx* ctx, double* buf, size_t cap, size_t* n, int reset);
/* Block log of the GMRES solves (ABI 6; option "gmres_block_log" != 0): one record of 11 doubles per Arnoldi block -- solve number
 * (from 1, since the last reset), first Hessenberg column j of the block, operator applications issued, steps the host algebra
 * accepted (fewer: the block was truncated at a small pivot), last pivot ratio of the block's Cholesky factor (the conditioning
 * margin against the truncation threshold 1e-8), the block's Newton shifts theta[0..3] (NaN: slot unused), residual estimate and
 * tolerance when the block was issued.  Same calling convention as bk_solver_history.  The library never writes to stderr.
 * Reference counterpart: `verbose` of the Krylov packages (src/LinearSolver.jl:169,202,244).                              */
int bk_solver_block_log(bk_ctx* ctx, double* buf, size_t cap, size_t* n, int reset);

/* ------------------------------------------------------------------ memory ----------------- */

int bk_malloc(bk_ctx* ctx, size_t n, double** out);
int bk_free(bk_ctx* ctx, double* p);
int bk_upload(bk_ctx* ctx, double* dst_dev, const double* src_host, size_t n);
int bk_download(bk_ctx* ctx, double* dst_host, const double* src_dev, size_t n);

/* ---------------------------------------- BLAS-1: the VectorInterface methods ---------------
 * of src/BorderedArrays.jl:86-217 on the `.u` part (the scalar `.p` part stays on the host).
 * Dots / norms are global (all-reduced over the communicator) and deterministic run-to-run.     */
int bk_vec_copy(bk_ctx* ctx, size_t n, const double* x, double* y);              /* _copyto!    :34 */
int bk_vec_zero(bk_ctx* ctx, size_t n, double* x);                               /* zerovector! :97 */
int bk_vec_scale(bk_ctx* ctx, size_t n, double a, double* x);                    /* scale!     :118 */
int bk_vec_axpby(bk_ctx* ctx, size_t n, double a, const double* x, double b, double* y);
                                                     /* y = a x + b y : VI.add!(y,x,a,b) :180-196 */
int bk_vec_dot(bk_ctx* ctx, size_t n, const double* x, const double* y, double* out); /* inner :213 */
int bk_vec_nrm2(bk_ctx* ctx, size_t n, const double* x, double* out);            /* norm     :55-70 */
int bk_vec_nrminf(bk_ctx* ctx, size_t n, const double* x, double* out);  /* norminf LinearSolver.jl:4 */
/* Fused Krylov-basis primitives (what the Gram-Schmidt loops inside KrylovKit / IterativeSolvers do with
 * k separate VI.inner / VI.add! passes).  V holds k vectors, V_i = V + i*ldv, k <= 64.
 *   multidot : out[i] = <V_i, w> (i < k), out[k] = <w, w>           -- one pass over V and w
 *   multiaxpy: dst = scale * (src + sum_i c[i] V_i); src may be NULL; *nrm2sq (may be NULL) = ||dst||^2 */
int bk_krylov_multidot(bk_ctx* ctx, size_t n, const double* V, size_t ldv, int k, const double* w,
                       double* out_host);
int bk_krylov_multiaxpy(bk_ctx* ctx, size_t n, const double* V, size_t ldv, int k, const double* c_host,
                        const double* src, double scale, double* dst, double* nrm2sq);

/* ------------------------------------------------------------------ problems --------------- */

enum {
    BK_PDE_SH = 1,    /* Swift-Hohenberg 2-D/3-D, Neumann-ghost: examples/SH3d.jl:16-53,
                         examples/SH2d-fronts.jl:13-34.  params = {l, nu}                           */
    BK_PDE_SH1D = 2,  /* 1-D cubic-quintic SH, Dirichlet: examples/SHpde_snaking.jl:16-25.
                         params = {lambda, nu}                                                      */
    BK_PDE_CGL2D = 3  /* 2-D complex Ginzburg-Landau, 2 stacked real fields, Dirichlet:
                         examples/cGL2d.jl:6-54,281-318.  params = {r, mu, nu, c3, c5, gamma}       */
};
#define BK_MAX_PARAMS 8

typedef struct {
    int pde;            /* BK_PDE_*                                                              */
    int ndim;           /* 1, 2 or 3                                                             */
    int n[3];           /* GLOBAL Nx, Ny, Nz (unused dims = 1)                                   */
    double l[3];        /* half-widths lx, ly, lz: h = 2 l / N (examples/SH3d.jl:18-20)           */
} bk_problem_desc;

int bk_problem_create(bk_ctx* ctx, const bk_problem_desc* desc, bk_problem** out);
int bk_problem_destroy(bk_problem* prob);
/* local (this rank's) vector length and the slab [lo, hi) of the slowest grid index it owns     */
int bk_problem_nlocal(bk_problem* prob, size_t* nlocal, int* slab_lo, int* slab_hi);
/* out = F(u, params): F_sh examples/SH3d.jl:44-47, R_SH SHpde_snaking.jl:19-23,
 * Fcgl! examples/cGL2d.jl:44-47                                                                  */
int bk_residual(bk_problem* prob, const double* u, const double* params, int nparams, double* out);
/* out = (F(u, p + eps) - F(u, p)) / eps for params[ipar] -- the finite-difference dF/dp of newton_palc and the Bordered
 * tangent (src/continuation/Palc.jl:239-240, src/continuation/Tangents.jl:77-82).  Every parameter of these problems
 * multiplies a pointwise term and the stencil part cancels identically, so the quotient is evaluated as
 * ((p + eps) - p)/eps * phi_p(u) in one pass without the ~1e-8 white rounding noise of the two-residual form (context
 * option "fd_dparam" = 0 selects the literal two-residual form).  out must not alias u.                            */
int bk_residual_dparam(bk_problem* prob, const double* u, const double* params, int nparams, int ipar, double eps,
                       double* out);
/* J(u, params) as an operator handle: the closure `dx -> dF_sh(x, p, dx)` of
 * examples/SH3d.jl:119 / the opaque Jacobian object of src/Problems.jl:98-101.  Like the Julia
 * closure it REFERENCES u (no copy): u must stay alive and unchanged while the handle is used. */
int bk_jacobian(bk_problem* prob, const double* u, const double* params, int nparams, bk_op** out);
/* the adjoint Jacobian J(u, p)' (JAd_at_xp of src/codim2/MinAugHopf.jl:66-80): the transposed pointwise block for
 * cGL2d, J itself for the (symmetric) Swift-Hohenberg problems.                                                   */
int bk_jacobian_adjoint(bk_problem* p, const double* u, const double* params, int nparams, bk_op** out);
int bk_op_destroy(bk_op* op);
/* out = a0*v + a1*J*v : _axpy_op, src/LinearSolver.jl:46-64 (a0=0,a1=1: apply, src/Utils.jl:191) */
int bk_op_apply(bk_op* op, const double* v, double a0, double a1, double* out);

/* ------------------------------------------------------------------ preconditioner ---------
 * Pl^-1 = ((I + Lap)^2 + shift I)^-1 applied exactly through the DCT-II diagonalisation of the
 * Neumann-ghost Laplacian.  shift = 0 is `Pl = cholesky(Symmetric(L1))` of examples/SH3d.jl:88;
 * shift = 1 is `lu(L1 + I)` of examples/SH2d-fronts.jl:121.                                      */
int bk_precond_sh_create(bk_problem* prob, double shift, bk_precond** out);
/* Pl^-1 = (Lap - c I)^-1 on both fields of a BK_PDE_CGL2D problem (Dirichlet 5-point Laplacian of
 * examples/cGL2d.jl:6-22, diagonalised by the DST-I), c > 0.  The reference has no iterative counterpart here: its
 * cGL2d runs use sparse LU (DefaultLS, and `EigArpack(1.0, :LM)` factorises J - sigma I inside ARPACK,
 * examples/cGL2d.jl:96, src/EigSolver.jl:85); this is the matrix-free replacement that keeps GMRES mesh-independent. */
int bk_precond_lap_create(bk_problem* prob, double c, bk_precond** out);
/* Pl^-1 = (Lap (x) I_2 + [[a, -b], [b, a]])^-1 on the stacked fields (u1, u2) of a BK_PDE_CGL2D problem: the 2x2 block is
 * inverted per sine mode.  With a = r, b = nu this IS the Jacobian of the trivial state u = 0 (Jcgl, examples/cGL2d.jl:57-79:
 * f1u = r, f1v = -nu, f2u = nu, f2v = r), i.e. the exact solve the reference gets from its sparse LU there; a = r - sigma
 * gives the shift-inverted operator of `EigArpack(sigma, :LM)` (cGL2d.jl:96).  b != 0 keeps every block invertible
 * (also at a Hopf point, where Lap + r I is singular); b = 0 requires a < 0.                                          */
int bk_precond_cgl_create(bk_problem* prob, double a, double b, bk_precond** out);
int bk_precond_destroy(bk_precond* pc);
int bk_precond_apply(bk_precond* pc, const double* v, double* out);   /* out = Pl \ v             */
/* out = a0 x + a1 Pl \ (J x): the `_linmap` closure GMRESKrylovKit hands to KrylovKit when it has a left preconditioner
 * (src/LinearSolver.jl:270-277), i.e. the operator every Arnoldi step of a preconditioned solve applies.  When `pl` is the
 * spectral preconditioner of J's own Swift-Hohenberg problem (Pl = L1 + s I, J = -L1 + diag g(u)) it is evaluated WITHOUT the
 * stencil, as (a0 - a1) x + a1 Pl \ ((g(u) + s) .* x) -- the form the solvers' Arnoldi steps use (context option
 * "gmres_stencil_free": 1 = where the transform kernels take the pointwise factor in, 2 = everywhere, 0 = never: stencil kernel,
 * then Pl).  *stencil_free (optional) reports which form ran.  out must not alias x.                                        */
int bk_precond_op_apply(bk_ctx* ctx, bk_precond* pl, bk_op* J, const double* x, double a0, double a1, double* out,
                        int* stencil_free);

/* ------------------------------------------------------------------ linear solver ----------
 * (ls::AbstractLinearSolver)(J, rhs; a0, a1) -> (x, success, niter): src/LinearSolver.jl:12.   */
enum {
    BK_GMRES_KRYLOVKIT = 0,        /* GMRESKrylovKit semantics, src/LinearSolver.jl:223-291:
                                      maxiter = restart cycles, tol = max(atol, rtol*||b||),
                                      niter = numops (operator applications)                       */
    BK_GMRES_ITERATIVESOLVERS = 1, /* GMRESIterativeSolvers semantics, :149-206: maxiter = inner
                                      iterations, tol = max(rtol*||r0||, atol), niter = iterations */
    BK_KRYLOV_MINRES = 3,          /* KrylovLS(KrylovAlg = :minres), :336-341 (symmetric solvers, "centered" SPD
                                      preconditioner M = Pl): Paige-Saunders MINRES on a0 + a1 J, stop on the
                                      estimated M^-1-norm of the residual <= atol + rtol*beta1; dim unused;
                                      maxiter = itmax; 8 vectors instead of a Krylov basis                        */
    BK_KRYLOV_CG = 4,              /* KrylovLS(KrylovAlg = :cg): preconditioned CG for SPD a0 + a1 J; stops without
                                      success on non-positive curvature                                            */
    BK_GMRES_KRYLOVJL = 2          /* KrylovLS / KrylovLSInplace with :gmres, :316-414: Krylov.jl stopping rule
                                      ||r|| <= atol + rtol*||r0||, dim = `memory` used as restart length,
                                      maxiter = itmax (inner iterations), niter = iterations; Pl = `M`      */
};
typedef struct {
    int flavor;      /* BK_GMRES_*                                                               */
    int dim;         /* Krylov dimension / restart (GMRESKrylovKit.dim, GMRESIterativeSolvers.restart) */
    int maxiter;
    double atol;
    double rtol;
    bk_precond* pr;  /* right preconditioner or NULL: GMRESIterativeSolvers.Pr (src/LinearSolver.jl:178,201 -- the solver iterates
                        on Pl^-1 (a0 I + a1 J) Pr^-1 y = Pl^-1 rhs and returns x = Pr^-1 y) and KrylovLS's `N = Pr` for the
                        non-symmetric methods (:343); an error with the KrylovKit flavor (GMRESKrylovKit has no such field,
                        :223-250); ignored by :minres / :cg exactly as the reference does ("we only pass centered
                        preconditioner", :339-341).  bk_gmres_default_opts sets NULL.                                      */
} bk_gmres_opts;
void bk_gmres_default_opts(bk_gmres_opts* o, int flavor);   /* the reference's defaults           */
/* Solve (a0 I + a1 J) x = rhs, x0 = 0.  `pl` may be NULL.  With `pl` the KrylovKit flavor solves
 * (a0 I + a1 Pl^-1 J) x = Pl^-1 rhs -- shift applied after the preconditioner, exactly the
 * reference's branch src/LinearSolver.jl:268-288.  x must not alias rhs; rhs is not modified.   */
int bk_gmres(bk_ctx* ctx, bk_op* J, const double* rhs, double* x, double a0, double a1,
             const bk_gmres_opts* opts, bk_precond* pl, int* converged, int* niter,
             double* resnorm);
/* (ls)(J, rhs1, rhs2): two sequential solves, flags ANDed, src/LinearSolver.jl:15-19.           */
int bk_gmres2(bk_ctx* ctx, bk_op* J, const double* rhs1, const double* rhs2, double* x1,
              double* x2, double a0, double a1, const bk_gmres_opts* opts, bk_precond* pl,
              int* converged, int niter[2]);

/* ------------------------------------------------------------------ bordered solvers -------
 * (lbs)(J, dR, dzu, dzp, R, n, xi_u, xi_p; shift, dotp) -> (dX, dl, success, itlinear):
 * src/LinearBorderSolver.jl:3-6.  Solves
 *     [ shift I + J      dR     ] [dX]   [R]
 *     [ xi_u dzu' S   xi_p dzp  ] [dl] = [n]       with dotp(x,y) = dotscale * <x,y>
 * (PALC passes dotscale = 1/N, xi_u = theta, xi_p = 1-theta: solve_bls_palc :16-36).            */
typedef struct {
    double tol;            /* BorderingBLS.tol             (:63)                                  */
    int check_precision;   /* BorderingBLS.check_precision (:66)                                  */
    int k;                 /* BorderingBLS.k               (:69)                                  */
    int kind;              /* which bordered solver the correctors (bk_newton_palc, bk_cont_*) use: 0 = BorderingBLS (the
                              fields above), 1 = MatrixFreeBLS (:326-335,424-437: ONE GMRES on the (N+1) operator; tol,
                              check_precision, k and the preconditioner are not used by it, as in the reference).
                              bk_bls_bordering itself ignores the field.                                           */
} bk_bordering_opts;
int bk_bls_bordering(bk_ctx* ctx, bk_op* J, const double* dR, const double* dzu, double dzp,
                     const double* R, double n, double xiu, double xip, int has_shift,
                     double shift, double dotscale, const bk_bordering_opts* bopts,
                     const bk_gmres_opts* lsopts, bk_precond* pl, double* dX, double* dl,
                     int* converged, int itlinear[2]);                   /* BorderingBLS :88-166 */
/* Iteration accounting with check_precision (k refinement passes, :111-121): the reference's BEC solves
 * (shift I + J) x = dR again on every pass (:134) and returns the counts of the LAST pass.  dR does not change between
 * the passes, so this implementation solves it once and reports that solve's count and flag for every pass -- the same
 * numbers the repeated (deterministic) solve would return, at one solve less per pass; itlinear[0] is the last pass's
 * solve with the refreshed right-hand side, as in the reference.                                                      */
int bk_bls_matrixfree(bk_ctx* ctx, bk_op* J, const double* dR, const double* dzu, double dzp,
                      const double* R, double n, double xiu, double xip, int has_shift,
                      double shift, double dotscale, const bk_gmres_opts* lsopts, double* dX,
                      double* dl, int* converged, int* itlinear);        /* MatrixFreeBLS :424-437 */

/* solve_bls_block(::BorderingBLS, ...) for an m-column border (normal forms, Bogdanov-Takens; :173-206):
 *     [ J   b ] [u1]   [rhst]      b, c: m device vectors each; d: m x m host, row-major d[i*m + j];
 *     [ c'  d ] [u2] = [rhsb]      plain inner products (VI.inner), no shift.
 * u1 (device, fresh buffer) and u2[m] (host) receive the solution, itlinear[m] the iteration counts of the m
 * border solves, *converged their AND (the flag of the J u = rhst solve is dropped, as in the reference).      */
#define BK_MAX_BORDER 8
int bk_bls_block_bordering(bk_ctx* ctx, bk_op* J, int m, const double* const* b, const double* const* c,
                           const double* d, const double* rhst, const double* rhsb,
                           const bk_gmres_opts* lsopts, bk_precond* pl, double* u1, double* u2,
                           int* converged, int* itlinear);

/* solve_bls_block(::MatrixFreeBLS, J, a, b, c, rhst, rhsb; shift, dotp) (:440-450): ONE GMRES on the (N + m) operator
 * MatrixFreeBLSmap (:338-352) acting on BorderedArray(u, p::Vector{m}):
 *     out.u = (J + shift) u + sum_i p_i a_i ,   out.p = c p + [dotscale <b_i, u>]_i          c: m x m row-major.   */
int bk_bls_block_matrixfree(bk_ctx* ctx, bk_op* J, int m, const double* const* a, const double* const* b,
                            const double* c, const double* rhst, const double* rhsb, int has_shift,
                            double shift, double dotscale, const bk_gmres_opts* lsopts, double* u1,
                            double* u2, int* converged, int* itlinear);

/* ------------------------------------------------------------------ eigensolver ------------
 * (eig::ShiftInvert)(J, nev) -> (vals, vecs, converged, niter): src/EigSolver.jl:246-266 with a
 * Krylov-Schur (KrylovKit.eigsolve-style) outer iteration; the SH3dEig of examples/SH3d.jl:96-113.
 * Eigenvalues come back sorted by decreasing real part (__sort_spectrum, src/EigSolver.jl:16-19).
 * vals_re / vals_im hold nev + 1 entries: *nvals = nev, or nev + 1 when the cut would split a complex-conjugate
 * pair (as ARPACK / KrylovKit do); entries that did not converge to max(tol, sqrt(eps)|mu|) are NaN.
 * vecs (may be NULL) receives up to nev + 1 device vectors of local length, stride ldvecs, real parts of
 * the eigenvectors (imaginary parts in vecs_im when not NULL; symmetric problems: zero).        */
typedef struct {
    double sigma;      /* shift                                                                    */
    int krylovdim;     /* examples/SH3d.jl:109: max(30, nev+30)                                   */
    int maxiter;       /* restarts                                                                 */
    double tol;        /* Ritz residual tolerance                                                  */
    int hermitian;     /* ishermitian = true -> Lanczos-like symmetric Rayleigh quotient          */
    unsigned long long seed;   /* start vector: rand(N) in the reference (SH3d.jl:109)            */
} bk_eig_opts;
int bk_eig_shiftinvert(bk_ctx* ctx, bk_op* J, int nev, const bk_eig_opts* eopts,
                       const bk_gmres_opts* lsopts, bk_precond* pl, double* vals_re,
                       double* vals_im, double* vecs, double* vecs_im, size_t ldvecs, int* nvals,
                       int* nconv, int* numops);
/* Start vector of the NEXT bk_eig_shiftinvert / bk_eig_krylovkit call on this context (one-shot; NULL restores the
 * default rand(N) of examples/SH3d.jl:109): the x0 argument of KrylovKit.eigsolve, EigKrylovKit.x0 (src/EigSolver.jl:
 * 143,160).  x0 (device, local length) must stay valid until that call returns.                                    */
int bk_eig_set_start_vector(bk_ctx* ctx, const double* x0);
/* (eig::EigKrylovKit)(J, nev), which = :LR (src/EigSolver.jl:117-166): Krylov-Schur on J itself, rightmost eigenvalues,
 * no inner solves (sigma ignored); same output conventions as bk_eig_shiftinvert; *numops = operator applications.   */
int bk_eig_krylovkit(bk_ctx* ctx, bk_op* J, int nev, const bk_eig_opts* eopts, double* vals_re, double* vals_im,
                     double* vecs, double* vecs_im, size_t ldvecs, int* nvals, int* nconv, int* numops);

/* ------------------------------------------------------------------ Newton correctors ------ */
/* Newton callback, the reference's `callback(state; fromNewton, kwargs...) -> Bool` (src/Newton.jl:88,111-113,
 * src/continuation/Palc.jl:235,294-297; e.g. cbMaxNorm src/Newton.jl:156-159): called before the first iteration
 * (step = 0, itlinear = 0), after every iteration, and once more when the loop has ended (its value is AND-ed into
 * `converged`).  x / fx are device vectors of the current iterate and residual; p the current continuation parameter
 * (newton_palc) or NaN (_newton); z0u / z0p the previous point of the branch (newton_palc) or NULL / NaN.
 * Return 0 to stop the iteration (a veto), non-zero to go on.                                                        */
typedef int (*bk_newton_callback)(void* user, const double* x, const double* fx, double residual, int step,
                                  int itlinear, double p, const double* z0u, double z0p, int from_newton);
typedef struct {
    double tol;           /* NewtonPar.tol            src/Newton.jl:19                            */
    int max_iterations;   /* NewtonPar.max_iterations :21                                         */
    int norm_inf;         /* normN: 1 = norminf (examples/SH3d.jl:166), 0 = 2-norm                */
    /* the remaining fields may be left zero (= the reference defaults)                                              */
    int linesearch;       /* NewtonPar.linesearch (:27): Armijo-type damping in newton_palc, Palc.jl:254-281        */
    double alpha;         /* NewtonPar.alpha  (:29), <= 0 -> 1                                                       */
    double alpha_min;     /* NewtonPar.alphamin (:31), <= 0 -> 1e-3                                                  */
    double max_residual;  /* > 0: built-in cbMaxNorm(max_residual) veto (src/Newton.jl:156-159), no host round trip  */
    bk_newton_callback callback;   /* may be NULL                                                                    */
    void* callback_user;
} bk_newton_opts;
#define BK_MAX_NEWTON_ITER 64
typedef struct {
    int converged;
    int itnewton;
    int itlinear;                                /* itlineartot, src/Newton.jl:60                 */
    double residuals[BK_MAX_NEWTON_ITER + 1];    /* NonLinearSolution.residuals                   */
} bk_newton_result;
/* _newton, src/Newton.jl:66-114: x is the initial guess on entry, the solution on exit.         */
int bk_newton(bk_ctx* ctx, bk_problem* prob, double* x, const double* params, int nparams,
              const bk_newton_opts* nopts, const bk_gmres_opts* lsopts, bk_precond* pl,
              bk_newton_result* res);
/* newton_palc, src/continuation/Palc.jl:187-305 (linesearch = false) with BorderingBLS:
 * (x, *p) = predictor on entry, corrected point on exit; (z0u, z0p) last point, (tauu, taup)
 * tangent; the continuation parameter is params[ipar].                                          */
int bk_newton_palc(bk_ctx* ctx, bk_problem* prob, double* x, double* p, const double* z0u,
                   double z0p, const double* tauu, double taup, double ds, double theta,
                   const double* params, int nparams, int ipar, double p_min, double p_max,
                   const bk_newton_opts* nopts, const bk_bordering_opts* bopts,
                   const bk_gmres_opts* lsopts, bk_precond* pl, bk_newton_result* res);

/* ------------------------------------------------------------------ complex shifts (Hopf) ----------
 * The Hopf normal form and the minimally augmented Hopf system call the same plugin surface with a COMPLEX shift on
 * complex vectors and a real Jacobian: ls(L, rhs; a0 = Complex(0, 2w), a1 = -1) (src/NormalForms.jl:1053),
 * bls(J, a, b, 0, 0, 1; shift = Complex(0, -w)) (src/codim2/MinAugHopf.jl:17, 72-76).  Complex device vectors are
 * (re, im) pairs; *_im inputs may be NULL (= 0).  The solve runs on the real-equivalent 2N system with the real GMRES
 * (iteration counts can differ from a complex-arithmetic GMRES; same residual tolerance).  SURVEY section 8(f) item 3. */
int bk_gmres_cshift(bk_ctx* ctx, bk_op* J, const double* rhs_re, const double* rhs_im, double* x_re,
                    double* x_im, double a0_re, double a0_im, double a1, const bk_gmres_opts* opts,
                    bk_precond* pl, int* converged, int* niter, double* resnorm);
/* BorderingBLS, one BEC pass (check_precision = false), dotp(x, y) = dotscale * conj(x).y; dl[2] = (re, im).           */
int bk_bls_bordering_cshift(bk_ctx* ctx, bk_op* J, const double* dR_re, const double* dR_im,
                            const double* dzu_re, const double* dzu_im, double dzp_re, double dzp_im,
                            const double* R_re, const double* R_im, double n_re, double n_im, double xiu,
                            double xip, double shift_re, double shift_im, double dotscale,
                            const bk_gmres_opts* lsopts, bk_precond* pl, double* dX_re, double* dX_im,
                            double dl[2], int* converged, int itlinear[2]);

/* ------------------------------------------------------------------ continuation step -----------
 * The body of `iterate` (src/Continuation.jl:458-504) as one call: corrector! (newton_palc), compute_eigenvalues!
 * + is_stable (src/Utils.jl:67-104, src/Bifurcations.jl:5-19), _step_size_control! (src/continuation/Contbase.jl:77-102),
 * gettangent! (Secant / Bordered, src/continuation/Tangents.jl:28-104) and the predictor (addtangent!, :8-15).
 * bk_cont_create takes the two converged points of src/Continuation.jl:349-456 ((u0, p0) and (u1, p1 = p0 + ds/eta))
 * and performs initialize! (Palc.jl:112-123): secant tangent, first predictor and, with detect != 0, the eigenvalues
 * at (u0, p0) (returned in *init).  SURVEY section 8(f) item 2.                                                   */
#define BK_MAX_NEV 62
typedef struct {
    double ds, dsmin, dsmax;   /* ContinuationPar.ds / dsmin / dsmax   (src/ContParameters.jl)                      */
    double a;                  /* step-size aggressiveness ContinuationPar.a                                        */
    double theta;              /* PALC.theta                                                                         */
    double p_min, p_max;       /* ContinuationPar.p_min / p_max                                                      */
    int tangent;               /* 0: Secant(), 1: Bordered()                                                         */
    int detect;                /* detect_bifurcation > 0: eigenvalues after every converged step                    */
    int nev;                   /* ContinuationPar.nev                                                                */
    double tol_stability;      /* ContinuationPar.tol_stability                                                      */
} bk_cont_opts;
typedef struct {
    int converged, itnewton, itlinear;           /* corrector (NonLinearSolution)                                   */
    double residuals[BK_MAX_NEWTON_ITER + 1];
    double p;                                    /* parameter of the current point z after the step                  */
    double ds_used, ds_next;                     /* arclength step of this corrector / of the next predictor         */
    int step;                                    /* number of accepted steps so far                                  */
    int stop;                                    /* 0 continue, 1 |ds| <= dsmin after a failed corrector, 2 the previous point reached
                                                    the boundary of [p_min, p_max]: nothing was done (`done`, Continuation.jl:254) */
    int n_unstable, n_imag, bifurcation;         /* is_stable counts; bifurcation = 1 when n_unstable changed        */
    int nvals, eig_converged, eig_numops;        /* eigensolver return values; vals sorted by decreasing real part   */
    double vals_re[BK_MAX_NEV + 1], vals_im[BK_MAX_NEV + 1];
    int tangent_converged;                       /* Bordered(): convergence flag of the tangent's bordered solve     */
    int natural;                                 /* 1: the predictor left [p_min, p_max] and the step was corrected by the Natural
                                                    corrector at the clamped parameter (Palc.jl:157-160, Natural.jl:38-58)      */
} bk_cont_step_result;
typedef struct bk_cont bk_cont;
int bk_cont_create(bk_ctx* ctx, bk_problem* prob, const double* params, int nparams, int ipar,
                   const double* u0, double p0, const double* u1, double p1, const bk_cont_opts* copts,
                   const bk_newton_opts* nopts, const bk_bordering_opts* bopts, const bk_gmres_opts* lsopts,
                   bk_precond* pl, const bk_eig_opts* eopts, const bk_gmres_opts* eig_lsopts,
                   bk_precond* eig_pl, bk_cont_step_result* init, bk_cont** out);
int bk_cont_step(bk_cont* c, bk_cont_step_result* res);
/* copies of the current point / tangent (device buffers of the local length, any may be NULL) */
int bk_cont_get(bk_cont* c, double* u, double* p, double* tauu, double* taup, double* ds);
int bk_cont_destroy(bk_cont* c);
/* copy(state), src/Continuation.jl (ContState copies used by the bisection and by the engine's own bookkeeping)        */
int bk_cont_clone(bk_cont* src, bk_cont** out);
/* locate_bifurcation!(iter, state), src/Bifurcations.jl:159-349: call right after a bk_cont_step whose result has
 * bifurcation = 1.  Bisects with the continuation step itself (step-size control off, ds halved / reversed at every
 * crossing) until n_inversion crossings, max_bisection_steps steps or |ds| < dsmin_bisection; leaves the state next to the
 * bifurcation point with the predictor rebuilt.  status: 0 none, 1 guess, 2 converged, 3 guessL; type: 0 none, 1 bp,
 * 2 hopf, 3 nd (the codim-1 classification of _get_bifurcation_type, :80-130).                                        */
typedef struct {
    double dsmin_bisection;          /* ContinuationPar.dsmin_bisection           (1e-16)                             */
    int n_inversion;                 /* ContinuationPar.n_inversion               (2, must be even)                   */
    int max_bisection_steps;         /* ContinuationPar.max_bisection_steps       (25)                                */
    double tol_bisection_eigenvalue; /* ContinuationPar.tol_bisection_eigenvalue  (1e-16)                             */
    int max_steps;                   /* ContinuationPar.max_steps: `done` also bounds the bisection's own step counter */
} bk_bisection_opts;
typedef struct {
    int status, type;
    double interval[2];              /* parameter interval that contains the bifurcation point                        */
    double p;                        /* parameter of the state on return                                             */
    int n_unstable[2], n_imag[2];    /* (current, previous) pairs of the state on return                              */
    int steps;                       /* continuation steps spent                                                      */
    int nvals;                       /* eigenvalues of the state on return (`_state.eigvals = state.eigvals`, :341)   */
    double vals_re[BK_MAX_NEV + 1], vals_im[BK_MAX_NEV + 1];
} bk_bisection_result;
int bk_cont_locate_bifurcation(bk_cont* c, const bk_bisection_opts* opts, bk_bisection_result* res);

/* ------------------------------------------------------------------ deflated Newton --------------
 * solve(prob, defOp, options, DeflatedProblemCustomLS()) (src/DeflationOperator.jl:340-355): Newton on M(u) F(u) with
 * M(u) = prod_i ( <u - root_i, u - root_i>^-power + alpha ) (accumulator_mean != 0: the mean, Val(:Mean)), dM by finite
 * differences of step delta (:160-169), every linear solve = two solves with the plain Jacobian (ls(J, rhs, Fu)) and the
 * Sherman-Morrison style recombination of :264-312.  x: guess on entry, solution on exit.  SURVEY section 8(f) item 4. */
int bk_newton_deflated(bk_ctx* ctx, bk_problem* prob, double* x, const double* params, int nparams,
                       const double* const* roots, int nroots, double power, double alpha,
                       int accumulator_mean, double delta, const bk_newton_opts* nopts,
                       const bk_gmres_opts* lsopts, bk_precond* pl, bk_newton_result* res);

#ifdef __cplusplus
}
#endif
#endif /* BKHIP_H */
