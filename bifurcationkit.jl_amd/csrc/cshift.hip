// Complex-shift linear and bordered solves: the reference's Hopf machinery calls the plugin
// surface with a COMPLEX shift on complex vectors while the Jacobian stays real,
//   ls(L, rhs; a0 = Complex(0, 2w), a1 = -1)                               src/NormalForms.jl:1053
//   bls(J, a, b, 0, 0, 1; shift = Complex(0, -w))                          src/codim2/MinAugHopf.jl:17, 72-76
// Complex device vectors are (re, im) pairs of real vectors.  The solve runs on the real-equivalent system of size 2N
//   [ a0r + a1 M   -a0i      ] [xr]   [br]
//   [ a0i           a0r + a1 M ] [xi] = [bi] ,      M = J  or  Pl^-1 J (GMRESKrylovKit's Pl branch, src/LinearSolver.jl:268-288)
// with the real GMRES of solver.hip (same JVP / Gram-Schmidt kernels, vectors of length 2N).  The Krylov space is the
// real span of {A^k r}, not KrylovKit's complex span: iteration counts may differ from a complex-arithmetic GMRES, the
// returned solution satisfies the same system to the same (preconditioned-)residual tolerance.
#include <cmath>

#include "ops.h"

using namespace bk;

namespace {

struct ComplexShiftOp : bk_op {
    bk_op* J;
    bk_precond* P;
    double a0r, a0i, a1;
    int order;              // 0: a0 x + a1 Pl^-1 J x (KrylovKit flavor), 1: Pl^-1 (a0 x + a1 J x) (the others)
    size_t N;               // n = 2 N
    double* tmp;            // N doubles
    int half(const double* x, const double* y, double sgn, double b0, double b1, double* out) {
        // out = b0 x + b1 ( W x + sgn a0i Q y ),  W = a0r + a1 M,  Q = I (order 0) or Pl^-1 (order 1)
        bk_ctx* c = ctx;
        if (!P || order == 0) {
            if (!P) {
                BK_TRY(J->apply(x, nullptr, b0 + b1 * a0r, b1 * a1, out, nullptr));
            } else {
                BK_TRY(J->apply(x, nullptr, 0.0, 1.0, tmp, nullptr));
                BK_TRY(P->apply(tmp, tmp));
                BK_TRY(v_axpbyz(c, N, b0 + b1 * a0r, x, b1 * a1, tmp, out));
            }
            return v_axpby(c, N, b1 * sgn * a0i, y, 1.0, out);
        }
        // order 1: Pl^-1 (a0r x + a1 J x + sgn a0i y)
        BK_TRY(J->apply(x, nullptr, a0r, a1, tmp, nullptr));
        BK_TRY(v_axpby(c, N, sgn * a0i, y, 1.0, tmp));
        BK_TRY(P->apply(tmp, tmp));
        return v_axpbyz(c, N, b0, x, b1, tmp, out);
    }
    int apply(const double* x, const double*, double b0, double b1, double* out, double*) override {
        BK_TRY(half(x, x + N, -1.0, b0, b1, out));
        return half(x + N, x, +1.0, b0, b1, out + N);
    }
};

// (a0 + a1 J) x = rhs on stacked vectors [re; im] of length 2N; x must not alias rhs
int csolve(bk_ctx* ctx, bk_op* J, const double* rhs2, double* x2, double a0r, double a0i, double a1,
           const bk_gmres_opts& o, bk_precond* pl, GmresResult* res) {
    if (o.flavor >= BK_KRYLOV_MINRES) return set_error(ctx, "complex-shift solves need a GMRES flavor (the real-equivalent operator is not symmetric)");
    const size_t N = J->n;
    WsGuard ws(ctx);
    ComplexShiftOp W;
    W.ctx = ctx; W.n = 2 * N; W.ntail = 0; W.N = N;
    W.J = J; W.P = pl; W.a0r = a0r; W.a0i = a0i; W.a1 = a1;
    W.order = o.flavor == BK_GMRES_KRYLOVKIT ? 0 : 1;
    BK_TRY(ws.get(N, &W.tmp));
    const double* b = rhs2;
    if (pl) {
        double* prhs = nullptr;
        BK_TRY(ws.get(2 * N, &prhs));
        BK_TRY(pl->apply(rhs2, prhs));
        BK_TRY(pl->apply(rhs2 + N, prhs + N));
        b = prhs;
    }
    return gmres_core(ctx, &W, b, nullptr, x2, nullptr, 0.0, 1.0, o, res);
}

// complex inner product conj(x) . y on (re, im) pairs: (xr.yr + xi.yi) + i (xr.yi - xi.yr)
int cdot(bk_ctx* ctx, size_t N, const double* x2, const double* y2, double* re, double* im) {
    double a, b, c, d;
    BK_TRY(v_dot(ctx, N, x2, y2, &a));
    BK_TRY(v_dot(ctx, N, x2 + N, y2 + N, &b));
    BK_TRY(v_dot(ctx, N, x2, y2 + N, &c));
    BK_TRY(v_dot(ctx, N, x2 + N, y2, &d));
    *re = a + b;
    *im = c - d;
    return 0;
}

int stack(bk_ctx* ctx, size_t N, const double* re, const double* im, double* out2) {
    BK_TRY(v_copy(ctx, N, re, out2));
    return im ? v_copy(ctx, N, im, out2 + N) : v_zero(ctx, N, out2 + N);
}

}  // namespace

extern "C" {

int bk_gmres_cshift(bk_ctx* ctx, bk_op* J, const double* rhs_re, const double* rhs_im, double* x_re, double* x_im,
                    double a0_re, double a0_im, double a1, const bk_gmres_opts* opts, bk_precond* pl, int* converged,
                    int* niter, double* resnorm) {
    if (!ctx || !J || !rhs_re || !x_re || !x_im || !opts) return -1;
    if (J->ntail != 0) return set_error(ctx, "bk_gmres_cshift: operator must be unbordered");
    if (J->n % 2 != 0) return set_error(ctx, "bk_gmres_cshift: odd local length (the stacked halves must stay 16-B aligned)");
    const size_t N = J->n;
    WsGuard ws(ctx);
    double *b2 = nullptr, *x2 = nullptr;
    BK_TRY(ws.get(2 * N, &b2));
    BK_TRY(ws.get(2 * N, &x2));
    BK_TRY(stack(ctx, N, rhs_re, rhs_im, b2));
    GmresResult r;
    BK_TRY(csolve(ctx, J, b2, x2, a0_re, a0_im, a1, *opts, pl, &r));
    BK_TRY(v_copy(ctx, N, x2, x_re));
    BK_TRY(v_copy(ctx, N, x2 + N, x_im));
    if (converged) *converged = r.converged;
    if (niter) *niter = r.niter;
    if (resnorm) *resnorm = r.resnorm;
    return 0;
}

// BorderingBLS / BEC (src/LinearBorderSolver.jl:125-144) with a complex shift on complex data:
//   x1 = (shift + J)^-1 R,  dx = (shift + J)^-1 dR,  dl = (n - dotp(dzu, x1) xiu) / (dzp xip - dotp(dzu, dx) xiu),
//   dX = x1 - dl dx,   dotp(x, y) = dotscale * conj(x) . y.   One BEC pass (check_precision = false).
int bk_bls_bordering_cshift(bk_ctx* ctx, bk_op* J, const double* dR_re, const double* dR_im, const double* dzu_re,
                            const double* dzu_im, double dzp_re, double dzp_im, const double* R_re, const double* R_im,
                            double n_re, double n_im, double xiu, double xip, double shift_re, double shift_im,
                            double dotscale, const bk_gmres_opts* lsopts, bk_precond* pl, double* dX_re, double* dX_im,
                            double dl[2], int* converged, int itlinear[2]) {
    if (!ctx || !J || !dR_re || !dzu_re || !R_re || !lsopts || !dX_re || !dX_im || !dl) return -1;
    if (J->n % 2 != 0) return set_error(ctx, "bk_bls_bordering_cshift: odd local length (the stacked halves must stay 16-B aligned)");
    const size_t N = J->n;
    WsGuard ws(ctx);
    double *dR2 = nullptr, *dzu2 = nullptr, *R2 = nullptr, *x1 = nullptr, *dx = nullptr;
    BK_TRY(ws.get(2 * N, &dR2)); BK_TRY(ws.get(2 * N, &dzu2)); BK_TRY(ws.get(2 * N, &R2));
    BK_TRY(ws.get(2 * N, &x1)); BK_TRY(ws.get(2 * N, &dx));
    BK_TRY(stack(ctx, N, dR_re, dR_im, dR2));
    BK_TRY(stack(ctx, N, dzu_re, dzu_im, dzu2));
    BK_TRY(stack(ctx, N, R_re, R_im, R2));
    GmresResult r1, r2;
    BK_TRY(csolve(ctx, J, R2, x1, shift_re, shift_im, 1.0, *lsopts, pl, &r1));
    BK_TRY(csolve(ctx, J, dR2, dx, shift_re, shift_im, 1.0, *lsopts, pl, &r2));
    double ar, ai, br, bi;
    BK_TRY(cdot(ctx, N, dzu2, x1, &ar, &ai));
    BK_TRY(cdot(ctx, N, dzu2, dx, &br, &bi));
    const double nr = n_re - ar * dotscale * xiu, ni = n_im - ai * dotscale * xiu;
    const double dr = dzp_re * xip - br * dotscale * xiu, di = dzp_im * xip - bi * dotscale * xiu;
    const double den = dr * dr + di * di;
    if (den == 0.0) return set_error(ctx, "bk_bls_bordering_cshift: singular bordered system");
    const double lr = (nr * dr + ni * di) / den, li = (ni * dr - nr * di) / den;
    // dX = x1 - dl dx : re = x1r - (lr dxr - li dxi), im = x1i - (lr dxi + li dxr)
    BK_TRY(v_axpbyz(ctx, N, 1.0, x1, -lr, dx, dX_re));
    BK_TRY(v_axpby(ctx, N, li, dx + N, 1.0, dX_re));
    BK_TRY(v_axpbyz(ctx, N, 1.0, x1 + N, -lr, dx + N, dX_im));
    BK_TRY(v_axpby(ctx, N, -li, dx, 1.0, dX_im));
    dl[0] = lr; dl[1] = li;
    if (converged) *converged = r1.converged & r2.converged;
    if (itlinear) { itlinear[0] = r1.niter; itlinear[1] = r2.niter; }
    return 0;
}

}  // extern "C"
