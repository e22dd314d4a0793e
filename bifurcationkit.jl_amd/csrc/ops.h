// Internal: operator / problem / preconditioner objects behind the opaque C handles, and the
// stencil launchers.
#pragma once

#include "common.h"

namespace bk {

int halo_exchange(bk_ctx* ctx, hipStream_t stream, const double* v, size_t plane, int nplanes, int width, double* halo_lo,
                  double* halo_hi);

int comm_alltoallv(bk_ctx* ctx, const double* sendbuf, const size_t* scount, const size_t* sdispl, double* recvbuf,
                   const size_t* rcount, const size_t* rdispl);

// ---- stencil launchers (stencil.hip) --------------------------------------------------------
struct ShArgs {             // Swift-Hohenberg 2-D/3-D, Neumann-ghost (mirror) boundaries
    int nx, ny, nz;         // local extents (nz = planes owned by this rank; 1 in 2-D)
    int nzg, zoff;          // global plane count, global index of local plane 0
    double ax, ay, az;      // 1/hx^2, 1/hy^2, 1/hz^2 (az = 0 in 2-D)
    double l, nu;           // parameters
    double a0, a1;          // out = a0*v + a1*( -L1 v + g(u) v )
    double ag = 0.0;        // ag_set: out = a0*v + a1*(-L1 v) + ag*g(u) v -- the stencil part and the pointwise part scaled
    bool ag_set = false;    //   separately (a shift folded through the preconditioner, solver.hip: ShiftPrecOp)
    int mode;               // 0: JVP, g = l + 2 nu u - 3 u^2 ; 1: residual (v == u), g = l + nu u - u^2
    const double* v;
    const double* u;        // JVP only
    double* out;
    const double* halo_lo;  // 2 planes below local plane 0 (multi-GPU interior boundary) or NULL
    const double* halo_hi;  // 2 planes above local plane nz-1 or NULL
    int part = 0;           // 0: all z-chunks; 1: the chunks that read no halo plane; 2: the two face chunks (halo overlap)
    // fused Lanczos step (3-D streaming kernel, part 0 only; sh_fused_dot_ok): out += addc * addv (addv may be NULL), and the
    // per-tile partial sums of v . out go to ctx->d_partials -- *dot_blocks receives their count for reduce_finish
    int* dot_blocks = nullptr;
    const double* addv = nullptr;
    double addc = 0.0;
};
int sh_apply(bk_ctx* ctx, const ShArgs& a);
bool sh_fused_dot_ok(bk_ctx* ctx, const ShArgs& a);

struct CglArgs {            // 2-D cubic-quintic complex Ginzburg-Landau, Dirichlet, SoA [u1; u2]
    int nx, ny;
    double ax, ay;
    double r, mu, nu, c3, c5, gamma;
    double a0, a1;
    int mode;               // 0: JVP, 1: residual (v == u)
    const double* v;
    const double* u;
    double* out;
};
int cgl_apply(bk_ctx* ctx, const CglArgs& a);

struct Sh1dArgs {           // 1-D cubic-quintic SH, Dirichlet, L1 = -(I + D)^2
    int nx;
    double ax;
    double lam, nu;
    double a0, a1;
    int mode;               // 0: JVP g = lam + 3 nu u^2 - 5 u^4 ; 1: residual g = lam + nu u^2 - u^4
    const double* v;
    const double* u;
    double* out;
};
int sh1d_apply(bk_ctx* ctx, const Sh1dArgs& a);
// out = c * phi_ipar(u): the parameter derivative of the pointwise part (npts = grid points per field)
int pde_dparam(bk_ctx* ctx, int pde, int ipar, size_t npts, double c, const double* u, double* out);

// ---- DCT preconditioner launchers (dct.hip) -------------------------------------------------
struct DctPlan;
int dct_plan_create(bk_ctx* ctx, int ndim, const int n[3], const double ainv[3], double shift, DctPlan** out);
// distributed (z-slab) variant: n = GLOBAL extents; this rank owns planes [zlo, zhi)
int dct_plan_create_dist(bk_ctx* ctx, const int n[3], const double ainv[3], double shift, int zlo, int zhi, DctPlan** out);
void dct_plan_destroy(DctPlan* p);
// dot_blocks (optional): when the merged middle pass runs as the fused LDS kernel it also leaves the per-tile partial sums
// of v . out (= sum over the spectrum of symbol * |v^|^2: the transforms are orthonormal) in ctx->d_partials and their count
// in *dot_blocks; 0 = not available on this path
// Pointwise work fused into the FIRST and LAST pass of one preconditioner application (the stencil-free Arnoldi step of
// solver.hip: Pl^-1 J = -I + Pl^-1 diag(g(u) + s)): the x-forward pass reads in[i] * (A + u[i] (B + C u[i])) -- one extra 8 B/point
// stream -- and the x-inverse pass stores ct * result[i] + cx * xadd[i] (one more, only when xadd != NULL).
struct DctFuse {
    const double* u = nullptr;
    double A = 1.0, B = 0.0, C = 0.0;
    const double* xadd = nullptr;
    double cx = 0.0, ct = 1.0;
    // round 6 (MINRES: bk_precond::apply_dot_pre_axpy): the x-forward pass transforms in[i] + cadd * add[i] and stores that combined
    // vector to store[i] (may be the input array itself) -- the pass y <- y + c r of the Lanczos recurrence rides in the transform
    const double* add = nullptr;
    double cadd = 0.0;
    double* store = nullptr;
};
int dct_apply(bk_ctx* ctx, DctPlan* p, const double* v, double* out, int* dot_blocks = nullptr, const DctFuse* fz = nullptr);
// axis pass of the LDS FFT kernels (dct_fast.hip).  fuse_scale 0: plain, 1: forward + inverse symbol, 2: forward,
// symbol, inverse in one pass.  split (distributed plan, y passes only): the output (forward) / input (inverse) side
// uses the all-to-all block layout: element (x, k, other) at kmap[k] + other * plane + x.
struct DctSplit {
    const unsigned* kmap;
    unsigned plane;
};
// z passes of the slab z-solve as forward / inverse halves (dct_slab.hip, dct.hip: dct_apply_slab): the forward half (inverse = 0)
// stores y^ = sym .* f^ and writes the face data of the capacitance system -- (u'y, w'y) of this rank's two faces, from the values of
// y = B^-1 f at the four planes next to them -- into the all-to-all send buffer face_y ([owner][4][Lr]); the inverse half (inverse = 1)
// reads the solution (nu_u, nu_w) of both faces from the receive buffer face_d, forms delta = -U nu, adds sym_k sum_p phi_k(p) delta_p
// in the z-spectral domain and transforms back.
struct DctSlabHalf {
    double* face_y = nullptr;
    const double* face_d = nullptr;
    const double* phi = nullptr;      // [2][nl] local DCT-II basis at planes 0, 1
    size_t Lr = 0;                    // lines per owner (even)
    double a = 0.0;                   // 1 / h_z^2
    bool has_bottom = false, has_top = false;
};
int dct_axis_fft(bk_ctx* ctx, int n0, int n1, int n2, int axis, int inverse, const double* twid, const double* in,
                 double* out, const double* symx, const double* symy, const double* symz, double shift, int fuse_scale,
                 const DctSplit* split = nullptr, int* dot_blocks = nullptr, const DctFuse* fz = nullptr,
                 const DctSlabHalf* sh = nullptr);
bool dct_slab_half_ok(bk_ctx* ctx, int n0, int n1, int nl, const double* a, const double* b);
bool dct_axis_fft_supported(int n);
bool dct_axis_fused_ok(bk_ctx* ctx, int n0, int n1, int n2, int axis, const double* in, const double* out, int fuse_scale);

}  // namespace bk

// ---- objects behind the opaque handles ---------------------------------------------------------
struct bk_problem {
    bk_ctx* ctx = nullptr;
    bk_problem_desc desc{};
    size_t nloc = 0;        // local vector length
    size_t plane = 0;       // doubles per slab plane
    int lo = 0, hi = 0;     // slab [lo, hi) of the slowest index
    double ainv[3] = {0, 0, 0};
    double* halo_lo = nullptr;
    double* halo_hi = nullptr;
    // ag != NULL (SH 2-D / 3-D only): out = a0 v + a1 (-L1 v) + *ag g(u) v
    int apply(int mode, const double* v, const double* u, const double* params, double a0, double a1, double* out,
              const double* ag = nullptr);
    // out = a0 v + a1 J(u) v + c r (r may be NULL), *dot = v . out, in ONE pass where the stencil kernel supports it
    // (*fused = 1), else *fused = 0 and nothing was done
    int jvp_axpy_dot(const double* v, const double* u, const double* params, double a0, double a1, double c, const double* r,
                     double* out, double* dot, int* fused);
    // dFdp = (F(u, p + eps) - F(u, p)) / eps for the parameter `ipar`; `f0` = F(u, p) when the caller already has it
    // (only read by the literal two-residual form, option "fd_dparam" = 0), `out` must not alias u or f0
    int dparam(const double* u, const double* params, int nparams, int ipar, double eps, const double* f0, double* out);
};

struct bk_op {              // a linear operator on (device vector [+ one host tail scalar])
    bk_ctx* ctx = nullptr;
    size_t n = 0;           // local device length
    int ntail = 0;          // 0, or 1 for bordered (N+1) operators
    virtual ~bk_op() {}
    // out = a0*x + a1*A(x)
    virtual int apply(const double* x, const double* xt, double a0, double a1, double* out, double* outt) = 0;   // xt / outt: ntail host scalars (bordered operators), else NULL
    // The Lanczos step of the symmetric solvers (unbordered operators): out = a0 x + a1 A x + c r (r may be NULL),
    // *dot = x . out.  Default: apply, then one fused axpy + dot pass; operators with a fused kernel override it.
    virtual int apply_axpy_dot(const double* x, double a0, double a1, double c, const double* r, double* out, double* dot);
    // Newton-basis blocks of GMRES (solver.hip: arnoldi_block) apply (a0 - theta) x + a1 A x with a different theta per step:
    // true if a0 != 0 costs no extra pass over the vectors
    virtual bool shift_is_free() const { return false; }
    // the Swift-Hohenberg Jacobian J = -L1 + diag(g(u)) with its two parts scaled separately,
    // out = a0 x + aL (-L1 x) + ag g(u) x; returns 1 if this operator is not of that form (nothing done)
    virtual int apply_parts(const double* x, double a0, double aL, double ag, double* out) { return 1; }
    virtual const bk_problem* sh_problem() const { return nullptr; }
    // ... and the state it is linearised at: g(u) = l + 2 nu u - 3 u^2 (false: not a Swift-Hohenberg Jacobian)
    virtual bool sh_state(const double** u, double* l, double* nu) const { return false; }
    // The explicit residual of a solve, r = b - (a0 + a1 A) x (KrylovKit: "to ensure that no numerical errors have accumulated";
    // restarts of the other flavors).  Operators that run their Arnoldi steps on an algebraically rearranged form (solver.hip:
    // ShiftPrecOp in stencil-free mode) evaluate THIS through the original operator chain, so the check stays independent.
    virtual int apply_check(const double* x, const double* xt, double a0, double a1, double* out, double* outt) {
        return apply(x, xt, a0, a1, out, outt);
    }
    // true: GMRES builds its Krylov space on A itself and applies (alpha0, alpha1) to the Hessenberg matrix whatever the flavor
    // (the space of alpha0 + alpha1 A is the space of A; the iterates are the same) -- for operators whose shift costs a stream
    virtual bool hessenberg_shift() const { return false; }
    // The shift of the blocks that have no Ritz values yet ("monomial" blocks p_{i+1} = (A - shift) p_i): 0 for an ordinary operator.
    // An operator that iterates on a rearranged form A = W + theta0 I of the operator W the solve is about (solver.hip: ShiftPrecOp,
    // T = Pl^-1 J + I) returns theta0 when option gmres_monomial_shift = 1 -- its first block then is the literal operator's first block --
    // and 0 by default: powers of T itself are the better-conditioned first block (solver.hip: gmres_core, "First block").
    virtual double monomial_shift() const { return 0.0; }
    // The offset theta0 of a rearranged operator A = W + theta0 I itself, whatever the first block does (option gmres_monomial_shift
    // only gates monomial_shift): the origin of the Leja order of the later Newton shifts (option gmres_leja_origin; solver.hip:
    // ritz_shifts) -- the order relative to T's own origin truncated a block per solve on running branches (measured in round 5,
    // DESIGN 3) -- so that the two options can be set independently
    virtual double rearranged_origin() const { return 0.0; }
};

struct bk_precond {
    bk_ctx* ctx = nullptr;
    size_t n = 0;
    virtual ~bk_precond() {}
    virtual int apply(const double* v, double* out) = 0;     // out = Pl \ v ; out may alias v
    // out = Pl \ v and *dot = v . out (out must not alias v).  Default: apply, then a dot pass; the spectral
    // preconditioner takes the dot from the spectrum (Parseval) inside its merged middle pass.
    virtual int apply_dot(const double* v, double* out, double* dot);
    // y <- y + c r, then out = Pl \ y and *dot = y . out (out must not alias y or r).  Default: the axpy pass, then apply_dot; the
    // spectral preconditioner lets the axpy ride in its x-forward transform pass (4 array streams instead of 3 + 2), the same values
    // bit for bit (the sum is formed without contraction, as v_axpbyz forms it).
    virtual int apply_dot_pre_axpy(double* y, double c, const double* r, double* out, double* dot);
    // true if this preconditioner is the exact inverse of L1 + *shift I of the problem `prob` (the spectral preconditioner)
    virtual bool is_l1_plus_shift(const bk_problem* prob, double* shift) const { return false; }
    // out = cx x + ct Pl \ (d .* x), d_i = A + u_i (B + C u_i)   (out must not alias x unless pw_fused_ok)
    // Default: a pointwise pass, apply, an axpby; the spectral preconditioner fuses both into its first / last transform pass.
    virtual int apply_pw(const double* x, const bk::DctFuse& d, double cx, double ct, double* out);
    virtual bool pw_fused_ok(const double* x, const double* u, const double* out) const { return false; }
    // the same for this rank's part of the plan alone (shape of the x passes, library-owned scratch; no caller pointer looked at) ...
    virtual bool pw_plan_ok() const { return false; }
    // ... and what the ranks agreed on, once per preconditioner (solver.hip: ShiftPrecOp::init_fold; -1: not asked yet).  The FORM of
    // the operator inside GMRES -- stencil-free or the halo-exchanging chain -- follows this rank-invariant flag only; a rank whose
    // caller vectors are not 16-byte aligned runs the same form with a separate pointwise pass (apply_pw's default), i.e. the same
    // sequence of collectives.
    int pw_agreed = -1;
};

namespace bk {

struct PdeJacobian : bk_op {          // J(u, params) of a bk_problem; references u
    bk_problem* prob;
    const double* u;
    double params[BK_MAX_PARAMS];
    bool adjoint = false;             // J' (cGL only; the SH Jacobians are symmetric)
    int apply(const double* x, const double* xt, double a0, double a1, double* out, double* outt) override;
    int apply_axpy_dot(const double* x, double a0, double a1, double c, const double* r, double* out, double* dot) override;
    bool shift_is_free() const override { return true; }          // a0 is a term of the stencil kernels' store stage
    int apply_parts(const double* x, double a0, double aL, double ag, double* out) override;
    const bk_problem* sh_problem() const override;
    bool sh_state(const double** u_, double* l, double* nu) const override;
};

// A preconditioner object for the second lane that shares the tables of `pl` (read-only) but has its own scratch arrays
// (t1, t2; the face buffers of a distributed plan's slab z-solve) and runs on `lane` -- distributed plans included: the lane
// has its own communicator (ncclCommSplit / bk_ctx_set_lane_comm); NULL for preconditioners of another kind.  Delete it after
// the solve.
bk_precond* precond_lane_shadow(bk_precond* pl, bk_ctx* lane);

// GMRES core on an operator (solver.hip)
struct GmresResult {
    int converged = 0;
    int niter = 0;
    double resnorm = 0.0;
};
// Solve (alpha0 + alpha1 A) x = b, x0 = 0.  Tails (bt in, xt out: A->ntail host scalars) only for bordered operators.
int gmres_core(bk_ctx* ctx, bk_op* A, const double* b, const double* bt, double* x, double* xt, double alpha0,
               double alpha1, const bk_gmres_opts& o, GmresResult* res);
// The public linear solve with optional left preconditioner (reference branch semantics)
int linsolve(bk_ctx* ctx, bk_op* J, const double* rhs, double* x, double a0, double a1, const bk_gmres_opts& o,
             bk_precond* pl, GmresResult* res);
// two independent solves with the same operator, concurrently on two lanes where supported (solver.hip)
int linsolve2(bk_ctx* ctx, bk_op* J, const double* rhs1, double* x1, const double* rhs2, double* x2, double a0, double a1,
              const bk_gmres_opts& o, bk_precond* pl, GmresResult* r1, GmresResult* r2);

}  // namespace bk
