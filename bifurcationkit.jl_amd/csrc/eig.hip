// Shift-invert Krylov-Schur eigensolver: (eig::ShiftInvert)(J, nev), src/EigSolver.jl:246-266, with the
// outer iteration of KrylovKit.eigsolve (Arnoldi / Lanczos + Krylov-Schur thick restart) as used by
// examples/SH3d.jl:96-113 (SH3dEig: sigma = 0.1, :LM, tol 1e-12, krylovdim max(30, nev+30), maxiter 20).
// Every operator application is one device-resident GMRES solve of (J - sigma I) x = v.
// The Krylov basis lives in HBM; the (<= 63 x 63) Rayleigh quotient is decomposed on the host (dense.h).
#include <algorithm>
#include <cmath>
#include <complex>
#include <vector>

#include "dense.h"
#include "ops.h"

namespace bk {

// exposed by solver.hip (same translation unit would be nicer; keep one definition)
int arnoldi_step_public(bk_ctx* ctx, bk_op* A, double* V, size_t ld, std::vector<double>& tails, int j, double* w,
                        double* h, double* beta, std::vector<double>* G = nullptr, int* gram_n = nullptr);

namespace {

// Jshift = du -> J(du) .- sigma .* du   (examples/SH3d.jl:106)
struct ShiftedOp : bk_op {
    bk_op* J;
    double sigma;
    int apply(const double* x, const double*, double b0, double b1, double* out, double*) override {
        return J->apply(x, nullptr, b0 - b1 * sigma, b1, out, nullptr);
    }
    // the inner MINRES / CG solves take the operator's fused Lanczos step where it has one
    int apply_axpy_dot(const double* x, double b0, double b1, double c, const double* r, double* out, double* dot) override {
        return J->apply_axpy_dot(x, b0 - b1 * sigma, b1, c, r, out, dot);
    }
    // still a Swift-Hohenberg Jacobian, J - sigma I = -L1 + diag(g - sigma): the preconditioned inner solves of ShiftInvert take the
    // folded shift / the stencil-free Arnoldi step of solver.hip (ShiftPrecOp) like the corrector's solves do
    bool shift_is_free() const override { return J->shift_is_free(); }
    const bk_problem* sh_problem() const override { return J->sh_problem(); }
    bool sh_state(const double** u, double* l, double* nu) const override {
        if (!J->sh_state(u, l, nu)) return false;
        *l -= sigma;                       // g(u) = l + 2 nu u - 3 u^2: the shift is a shift of l
        return true;
    }
    int apply_parts(const double* x, double a0, double aL, double ag, double* out) override {
        return J->apply_parts(x, a0 - ag * sigma, aL, ag, out);
    }
};

// A = du -> ls(Jshift, du)[1]   (examples/SH3d.jl:107).  The shift is folded into the operator BEFORE the
// (optionally preconditioned) solve, exactly as SH3dEig does; this sidesteps the reference's Pl+shift quirk
// (src/LinearSolver.jl:268-277 solves (a0 I + a1 Pl^-1 J), which is not a shift-invert of J).  Without a
// preconditioner it is identical to ShiftInvert's ls(J, rhs; a0 = -sigma, a1 = 1) (src/EigSolver.jl:259-261).
struct ShiftInvertOp : bk_op {
    ShiftedOp Js;
    bk_gmres_opts ls;
    bk_precond* pl;
    int solves = 0, failed = 0, inner_ops = 0;
    int apply(const double* x, const double*, double b0, double b1, double* out, double*) override {
        if (b0 != 0.0 || b1 != 1.0) return set_error(ctx, "ShiftInvertOp: only plain application is supported");
        GmresResult r;
        BK_TRY(linsolve(ctx, &Js, x, out, 0.0, 1.0, ls, pl, &r));
        solves += 1;
        inner_ops += r.niter;
        if (!r.converged) failed += 1;
        return 0;
    }
};

}  // namespace
}  // namespace bk

using namespace bk;
using dense::cplx;

// Krylov-Schur core on an operator `Aop`.  invert: Aop = (J - sigma)^-1, Ritz values are ordered by magnitude (:LM) and
// mapped back by 1/mu + sigma; otherwise Aop = J itself and the order is by real part (:LR, EigKrylovKit's default use).
static int eig_core(bk_ctx* ctx, bk_op* Aop, bool invert, int nev, const bk_eig_opts* eo, double* vals_re, double* vals_im,
                    double* vecs, double* vecs_im, size_t ldvecs, int* nvals_out, int* nconv_out, int* napplied) {
    // one-shot start vector (bk_eig_set_start_vector): the x0 of KrylovKit.eigsolve(A, x0, ...) -- EigKrylovKit.x0,
    // src/EigSolver.jl:143,160; the reference's SH3dEig passes rand(N) (examples/SH3d.jl:109), which stays the default
    const double* x0 = ctx->eig_x0;
    ctx->eig_x0 = nullptr;
    bk_op& A = *Aop;
    const size_t n = A.n;
    int m = eo->krylovdim;
    if (m > kMaxBasis - 1) m = kMaxBasis - 1;
    if (nev > m) return set_error(ctx, "eigensolver: nev=%d exceeds the Krylov dimension %d", nev, m);

    WsGuard ws(ctx);
    const size_t ld = (n + 31) / 32 * 32;
    double *V = nullptr, *w = nullptr;
    BK_TRY(ws.get(ld * (size_t)(m + 1), &V));
    BK_TRY(ws.get(ld, &w));
    std::vector<double> tails(m + 1, 0.0);

    // x0 = rand(N) (examples/SH3d.jl:109): deterministic in (seed, global index) so that the start vector
    // does not depend on the slab decomposition
    size_t goff = 0;
    {
        // global offset of this rank's slab = sum of lower ranks' lengths; slabs are contiguous
        double offs[1] = {0.0};
        if (ctx->nranks > 1) {
            std::vector<double> lens(ctx->nranks, 0.0);
            lens[ctx->rank] = (double)n;
            BK_TRY(comm_allreduce_host(ctx, lens.data(), ctx->nranks, 0));
            for (int r = 0; r < ctx->rank; ++r) offs[0] += lens[r];
        }
        goff = (size_t)offs[0];
    }
    if (x0) BK_TRY(v_copy(ctx, n, x0, V));
    else BK_TRY(v_fill_random(ctx, n, goff, eo->seed, V));
    double nrm;
    BK_TRY(v_nrm2(ctx, n, V, &nrm));
    if (!(nrm > 0.0)) return set_error(ctx, "eigensolver: the start vector is zero");
    BK_TRY(v_scale(ctx, n, 1.0 / nrm, V));

    // Outer orthogonalisation: the Gram-corrected single pass of the linear solvers (solver.hip: arnoldi_step; option eig_gram,
    // default on) -- KrylovKit runs Lanczos / Arnoldi with a full re-orthogonalisation against the kept basis here
    // (src/EigSolver.jl:157-160, `ishermitian` of examples/SH3d.jl:109); the measured Gram matrix G of the basis is rotated
    // with it at every thick restart (G <- Q'GQ), the newest vector's column is measured by the step that follows.
    const bool use_gram = ctx->opt("eig_gram", 1.0) != 0.0 && ctx->opt("gmres_gram", 1.0) != 0.0;
    const int ldg = kMaxBasis + 1;
    std::vector<double> G(use_gram ? (size_t)ldg * ldg : 0, 0.0);
    int gram_n = 0;
    dense::Mat H(m + 1, m);
    int k = 0, numiter = 0, nconv = 0, applied = 0;
    std::vector<cplx> mu;
    dense::CMat Y;
    std::vector<int> order;
    std::vector<double> resid;
    bool breakdown = false;
    int meff = m;
    std::vector<double> h(m + 2, 0.0);
    while (true) {
        numiter += 1;
        while (k < meff) {
            double beta = 0.0;
            BK_TRY(arnoldi_step_public(ctx, &A, V, ld, tails, k, w, h.data(), &beta, use_gram ? &G : nullptr, use_gram ? &gram_n : nullptr));
            applied += 1;
            for (int i = 0; i <= k; ++i) H(i, k) = h[i];
            H(k + 1, k) = beta;
            k += 1;
            if (beta == 0.0) { breakdown = true; meff = k; break; }
        }
        // Rayleigh quotient B = H[:meff,:meff], border b = H[meff,:meff]
        dense::Mat B(meff, meff);
        std::vector<double> b(meff, 0.0);
        for (int j = 0; j < meff; ++j) {
            for (int i = 0; i < meff; ++i) B(i, j) = H(i, j);
            b[j] = breakdown ? 0.0 : H(meff, j);
        }
        mu.assign(meff, cplx(0.0, 0.0));
        Y = dense::CMat(meff, meff);
        if (eo->hermitian) {
            std::vector<double> wv;
            dense::Mat Z;
            if (dense::jacobi_eigh(B, wv, Z) < 0) return set_error(ctx, "eig: Jacobi sweep did not converge");
            for (int j = 0; j < meff; ++j) {
                mu[j] = cplx(wv[j], 0.0);
                for (int i = 0; i < meff; ++i) Y(i, j) = cplx(Z(i, j), 0.0);
            }
        } else {
            if (dense::eig_general(B, mu, Y) != 0) return set_error(ctx, "eig: QR iteration did not converge");
        }
        order.resize(meff);
        for (int i = 0; i < meff; ++i) order[i] = i;
        if (invert) std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return std::abs(mu[x]) > std::abs(mu[y]); });  // :LM
        else std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return mu[x].real() > mu[y].real(); });              // :LR
        resid.assign(meff, 0.0);
        for (int jj = 0; jj < meff; ++jj) {
            const int j = order[jj];
            cplx s = 0.0;
            for (int i = 0; i < meff; ++i) s += b[i] * Y(i, j);
            resid[jj] = std::abs(s);
        }
        nconv = 0;
        while (nconv < meff && resid[nconv] < eo->tol) nconv += 1;
        if (nconv >= nev || numiter >= eo->maxiter || breakdown) break;
        // ---- Krylov-Schur thick restart: keep the leading Ritz directions
        int keep = (3 * meff + 2 * nconv) / 5;
        if (keep < nev) keep = std::min(nev, meff - 1);
        if (keep >= meff) keep = meff - 1;
        if (!eo->hermitian && keep < meff && std::fabs(mu[order[keep - 1]].imag()) > 0.0 &&
            std::abs(mu[order[keep - 1]] - std::conj(mu[order[keep]])) <=
                1e-8 * std::abs(mu[order[keep - 1]]))
            keep += 1;                       // never split a complex-conjugate pair
        dense::Mat C(meff, eo->hermitian ? keep : 2 * keep);
        for (int jj = 0; jj < keep; ++jj) {
            const int j = order[jj];
            for (int i = 0; i < meff; ++i) {
                C(i, eo->hermitian ? jj : 2 * jj) = Y(i, j).real();
                if (!eo->hermitian) C(i, 2 * jj + 1) = Y(i, j).imag();
            }
        }
        dense::Mat Q;
        int kq = dense::orthonormalize_columns(C, 1e-8, Q);
        if (kq >= meff && keep > 1) {
            // the kept directions fill the whole basis (nev close to the Krylov dimension and a conjugate pair at the
            // cut): drop the trailing one(s) instead of failing
            const int cols = eo->hermitian ? keep - 1 : 2 * (keep - 2 > 0 ? keep - 2 : 1);
            dense::Mat C2(meff, cols);
            for (int c = 0; c < cols; ++c)
                for (int i = 0; i < meff; ++i) C2(i, c) = C(i, c);
            kq = dense::orthonormalize_columns(C2, 1e-8, Q);
        }
        if (kq < 1 || kq >= meff) return set_error(ctx, "eig: restart basis has rank %d of %d", kq, meff);
        // V[0..kq) <- V[0..meff) Q  (in place), V[kq] <- V[meff]
        BK_TRY(v_basis_combine(ctx, n, V, ld, meff, Q.a.data(), kq, V, ld));
        BK_TRY(v_copy(ctx, n, V + (size_t)meff * ld, V + (size_t)kq * ld));
        if (use_gram) {
            if (gram_n >= meff) {
                // Gram matrix of the rotated vectors; column kq (the old V[meff]) is measured by the next step
                std::vector<double> GQ((size_t)meff * kq, 0.0), Gn((size_t)ldg * ldg, 0.0);
                for (int c = 0; c < kq; ++c)
                    for (int i = 0; i < meff; ++i) {
                        double s_ = 0.0;
                        for (int l = 0; l < meff; ++l) s_ += G[(size_t)i + (size_t)l * ldg] * Q(l, c);
                        GQ[(size_t)i + (size_t)c * meff] = s_;
                    }
                for (int c = 0; c < kq; ++c)
                    for (int r = 0; r < kq; ++r) {
                        double s_ = 0.0;
                        for (int i = 0; i < meff; ++i) s_ += Q(i, r) * GQ[(size_t)i + (size_t)c * meff];
                        Gn[(size_t)r + (size_t)c * ldg] = s_;
                    }
                G.swap(Gn);
                gram_n = kq;
            } else {
                gram_n = -1;                  // a column is missing (a fallback step): two passes for the rest of this solve
            }
        }
        dense::Mat Hn(m + 1, m);
        for (int c = 0; c < kq; ++c) {
            for (int r = 0; r < kq; ++r) {
                double s = 0.0;
                for (int i = 0; i < meff; ++i) {
                    double t = 0.0;
                    for (int l = 0; l < meff; ++l) t += B(i, l) * Q(l, c);
                    s += Q(i, r) * t;
                }
                Hn(r, c) = s;
            }
            double s = 0.0;
            for (int i = 0; i < meff; ++i) s += b[i] * Q(i, c);
            Hn(kq, c) = s;
        }
        H = Hn;
        k = kq;
    }
    // ---- back-transform 1/mu + sigma, sort by decreasing real part (__sort_spectrum, src/EigSolver.jl:16-19)
    // number of values returned: nev, or nev + 1 when the cut would split a complex-conjugate pair (what ARPACK's
    // dneupd and KrylovKit do for real non-symmetric problems) -- the caller's arrays hold nev + 1 entries
    int nout = 0;
    while (nout < nev && nout < meff) {
        const cplx a = mu[order[nout]];
        const bool pair = !eo->hermitian && nout + 1 < meff && std::fabs(a.imag()) > 1e-12 * std::abs(a) &&
                          std::abs(a - std::conj(mu[order[nout + 1]])) <= 1e-8 * std::abs(a);
        nout += pair ? 2 : 1;
    }
    std::vector<cplx> lam(nout);
    std::vector<int> sel(nout);
    // Only converged Ritz pairs are reported (ARPACK / KrylovKit's `converged` count): an unconverged Ritz value of the
    // inverse can sit anywhere, and 1/mu + sigma would then fake an unstable eigenvalue.  Unconverged slots are NaN.
    std::vector<char> okv(nout, 0);
    for (int jj = 0; jj < nout; ++jj) {
        sel[jj] = order[jj];
        // "usable": Ritz residual below tol, or below sqrt(eps)|mu| -- the level inexact inner solves (rtol 1e-9 in
        // examples/SH3d.jl:115 against tol 1e-12) leave behind; the eigenvalue is then accurate to that level
        okv[jj] = (breakdown || resid[jj] < std::max(eo->tol, 1.4901161193847656e-08 * std::abs(mu[order[jj]]))) ? 1 : 0;
        lam[jj] = okv[jj] ? (invert ? cplx(1.0, 0.0) / mu[order[jj]] + eo->sigma : mu[order[jj]]) : cplx(NAN, NAN);
    }
    std::vector<int> perm(nout);
    for (int i = 0; i < nout; ++i) perm[i] = i;
    std::stable_sort(perm.begin(), perm.end(), [&](int x, int y) {
        if (okv[x] != okv[y]) return okv[x] > okv[y];                 // converged first, NaNs last
        return okv[x] && lam[x].real() > lam[y].real();
    });
    for (int i = 0; i < nout; ++i) {
        vals_re[i] = lam[perm[i]].real();
        vals_im[i] = lam[perm[i]].imag();
    }
    for (int i = nout; i <= nev; ++i) { vals_re[i] = NAN; vals_im[i] = NAN; }
    if (nvals_out) *nvals_out = nout;
    if (vecs) {
        if (ldvecs < n) return set_error(ctx, "bk_eig_shiftinvert: ldvecs < local length");
        std::vector<double> Qr((size_t)meff * nout), Qi((size_t)meff * nout);
        for (int c = 0; c < nout; ++c)
            for (int i = 0; i < meff; ++i) {
                Qr[(size_t)i + (size_t)c * meff] = Y(i, sel[perm[c]]).real();
                Qi[(size_t)i + (size_t)c * meff] = Y(i, sel[perm[c]]).imag();
            }
        BK_TRY(v_basis_combine(ctx, n, V, ld, meff, Qr.data(), nout, vecs, ldvecs));
        if (vecs_im) BK_TRY(v_basis_combine(ctx, n, V, ld, meff, Qi.data(), nout, vecs_im, ldvecs));
    }
    if (nconv_out) *nconv_out = std::min(nconv, nout);      // strictly converged (resid < tol), as info.converged
    if (napplied) *napplied = applied;
    return 0;
}

extern "C" int bk_eig_set_start_vector(bk_ctx* ctx, const double* x0) {
    if (!ctx) return -1;
    ctx->eig_x0 = x0;
    return 0;
}

extern "C" int bk_eig_shiftinvert(bk_ctx* ctx, bk_op* J, int nev, const bk_eig_opts* eo, const bk_gmres_opts* lsopts,
                                  bk_precond* pl, double* vals_re, double* vals_im, double* vecs, double* vecs_im,
                                  size_t ldvecs, int* nvals_out, int* nconv_out, int* numops_out) {
    if (!ctx || !J || !eo || !lsopts || !vals_re || !vals_im || nev < 1) return -1;
    if (J->ntail != 0) return set_error(ctx, "bk_eig_shiftinvert: operator must be unbordered");
    ShiftInvertOp A;
    A.ctx = ctx; A.n = J->n; A.ntail = 0; A.ls = *lsopts; A.pl = pl;
    A.Js.ctx = ctx; A.Js.n = J->n; A.Js.ntail = 0; A.Js.J = J; A.Js.sigma = eo->sigma;
    BK_TRY(eig_core(ctx, &A, true, nev, eo, vals_re, vals_im, vecs, vecs_im, ldvecs, nvals_out, nconv_out, nullptr));
    if (numops_out) *numops_out = A.solves;
    ctx->opts["eig_last_inner_ops"] = (double)A.inner_ops;     // diagnostics: bk_ctx_get_option
    return 0;
}

// (eig::EigKrylovKit)(J, nev) with which = :LR (src/EigSolver.jl:117-166): KrylovKit.eigsolve on J itself -- the
// rightmost eigenvalues without a shift-invert solve (only practical on mildly stiff operators; the PDE examples use
// ShiftInvert).  eo->sigma is ignored.
extern "C" int bk_eig_krylovkit(bk_ctx* ctx, bk_op* J, int nev, const bk_eig_opts* eo, double* vals_re, double* vals_im,
                                double* vecs, double* vecs_im, size_t ldvecs, int* nvals_out, int* nconv_out,
                                int* numops_out) {
    if (!ctx || !J || !eo || !vals_re || !vals_im || nev < 1) return -1;
    if (J->ntail != 0) return set_error(ctx, "bk_eig_krylovkit: operator must be unbordered");
    return eig_core(ctx, J, false, nev, eo, vals_re, vals_im, vecs, vecs_im, ldvecs, nvals_out, nconv_out, numops_out);
}
