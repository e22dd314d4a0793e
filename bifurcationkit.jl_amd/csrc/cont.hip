// Native PALC continuation step: the body of `iterate` (src/Continuation.jl:458-504) as one library call --
// corrector! (newton_palc, Palc.jl:187-305), compute_eigenvalues! (Utils.jl:67-104) + is_stable
// (Bifurcations.jl:5-19), _step_size_control! (Contbase.jl:77-102), gettangent! (Secant Tangents.jl:28-54 /
// Bordered :71-104) and getpredictor! / addtangent! (Tangents.jl:8-15).  Everything is composed from the public
// entry points of bkhip.h, so a step issued here is call-for-call the sequence the Julia engine (or the Python
// mirror continuation.py) would issue through the plugin surface; what it removes is the host round trips
// between them.
#include <cmath>
#include <vector>

#include "ops.h"

using namespace bk;

struct bk_cont {
    bk_ctx* ctx = nullptr;
    bk_problem* prob = nullptr;
    size_t n = 0;
    double Nglob = 1.0;
    double params[BK_MAX_PARAMS];
    int nparams = 0, ipar = 0;
    bk_cont_opts co;
    bk_newton_opts no;
    bk_bordering_opts bo;
    bk_gmres_opts lo;
    bk_precond* pl = nullptr;
    bool has_eig = false;
    bk_eig_opts eo;
    bk_gmres_opts elo;
    bk_precond* epl = nullptr;
    // state (Continuation.jl ContState: z, z_old, tau, z_pred, ds, n_unstable, n_imag, step)
    double *zu = nullptr, *zoldu = nullptr, *tauu = nullptr, *predu = nullptr, *work = nullptr, *work2 = nullptr;
    double zp = 0, zoldp = 0, taup = 0, predp = 0, ds = 0;
    int n_unstable = -1, n_imag = -1, step = 0;
    // the rest of ContState that the bisection of locate_bifurcation! steers (src/Bifurcations.jl:159-349)
    int n_unstable_prev = -1, n_imag_prev = -1;       // n_unstable / n_imag are (current, previous) pairs in the reference
    bool stepsizecontrol = true, converged = true;
    int nvals = 0;
    double vals_re[BK_MAX_NEV + 1], vals_im[BK_MAX_NEV + 1];
    // context option "eig_thick_start" != 0 at bk_cont_create: the eigensolve of a step starts from the sum of the
    // previous step's Ritz vectors (the x0 of KrylovKit.eigsolve, EigKrylovKit.x0 src/EigSolver.jl:143,160) instead of
    // rand(N); the eigenvectors move slowly along the branch, so the Krylov-Schur iteration restarts (almost) converged
    bool thick = false, have_x0 = false;
    double* eigx0 = nullptr;
};

namespace {

int dot_theta(bk_cont* c, const double* u1, const double* u2, double p1, double p2, double* out) {
    double d;
    BK_TRY(v_dot(c->ctx, c->n, u1, u2, &d));               // DotTheta with NormalisedDot, Palc.jl:1-6, 35
    *out = d / c->Nglob * c->co.theta + p1 * p2 * (1.0 - c->co.theta);
    return 0;
}

double copysign1(double x) { return std::copysign(1.0, x); }

// _secant_tangent!, Tangents.jl:28-42: tau = (z1 - z0) * sign(ds) / ||z1 - z0||_theta
int secant_tangent(bk_cont* c, const double* z1u, double z1p, const double* z0u, double z0p, double ds) {
    BK_TRY(v_copy(c->ctx, c->n, z1u, c->tauu));
    BK_TRY(v_axpby(c->ctx, c->n, -1.0, z0u, 1.0, c->tauu));
    c->taup = z1p - z0p;
    double nn;
    BK_TRY(dot_theta(c, c->tauu, c->tauu, c->taup, c->taup, &nn));
    const double a = copysign1(ds) / std::sqrt(nn);
    BK_TRY(v_scale(c->ctx, c->n, a, c->tauu));
    c->taup *= a;
    return 0;
}

// gettangent!(::Bordered), Tangents.jl:71-104
int bordered_tangent(bk_cont* c, int* converged) {
    const double eps = 1.4901161193847656e-08;             // getdelta = sqrt(eps), src/Problems.jl:69-70
    double par[BK_MAX_PARAMS];
    for (int i = 0; i < c->nparams; ++i) par[i] = c->params[i];
    double* dFdl = c->work;
    double* f0 = c->work2;
    par[c->ipar] = c->zp;
    BK_TRY(c->prob->dparam(c->zu, par, c->nparams, c->ipar, eps, nullptr, dFdl));      // Tangents.jl:77-82
    BK_TRY(v_zero(c->ctx, c->n, f0));                      // rhs (0, 1), Tangents.jl:90-94
    bk_op* J = nullptr;
    BK_TRY(bk_jacobian(c->prob, c->zu, par, c->nparams, &J));
    double tp = 0.0;
    int cv = 0, it[2] = {0, 0};
    double* tu = c->predu;                                  // the predictor is rebuilt right after the tangent
    const int s = bk_bls_bordering(c->ctx, J, dFdl, c->tauu, c->taup, f0, 1.0, c->co.theta, 1.0 - c->co.theta, 0, 0.0,
                                   1.0 / c->Nglob, &c->bo, &c->lo, c->pl, tu, &tp, &cv, it);
    bk_op_destroy(J);
    if (s != 0) return s;
    double nn, sg;
    BK_TRY(dot_theta(c, tu, tu, tp, tp, &nn));
    BK_TRY(dot_theta(c, c->tauu, tu, c->taup, tp, &sg));
    const double a = copysign1(sg) / std::sqrt(nn);
    BK_TRY(v_copy(c->ctx, c->n, tu, c->tauu));
    BK_TRY(v_scale(c->ctx, c->n, a, c->tauu));
    c->taup = tp * a;
    if (converged) *converged = cv;
    return 0;
}

int predictor(bk_cont* c) {                                 // z_pred = z + ds * tau, Tangents.jl:8-15
    BK_TRY(v_copy(c->ctx, c->n, c->zu, c->predu));
    BK_TRY(v_axpby(c->ctx, c->n, c->ds, c->tauu, 1.0, c->predu));
    c->predp = c->zp + c->ds * c->taup;
    return 0;
}

// compute_eigenvalues (Utils.jl:67-104: nev_ = max(n_unstable + 5, nev)) + is_stable (Bifurcations.jl:5-19)
int eigen(bk_cont* c, bk_cont_step_result* r) {
    double par[BK_MAX_PARAMS];
    for (int i = 0; i < c->nparams; ++i) par[i] = c->params[i];
    par[c->ipar] = c->zp;
    // n = state.n_unstable[2]: the count BEFORE the last update (Utils.jl:78-79); -1 at the first two calls
    int nev = std::max(c->n_unstable_prev + 5, c->co.nev);
    nev = std::min(nev, BK_MAX_NEV);
    bk_eig_opts eo = c->eo;
    if (eo.krylovdim <= 0) eo.krylovdim = std::max(30, nev + 30);      // examples/SH3d.jl:109
    eo.krylovdim = std::min(eo.krylovdim, 63);
    nev = std::max(1, std::min(nev, eo.krylovdim - 2));                // room for a restart that keeps a conjugate pair
    bk_op* J = nullptr;
    BK_TRY(bk_jacobian(c->prob, c->zu, par, c->nparams, &J));
    int nvals = 0, nconv = 0, nops = 0;
    WsGuard ws(c->ctx);
    double* vecs = nullptr;
    const size_t ld = (c->n + 31) / 32 * 32;
    if (c->thick) {
        BK_TRY(ws.get(ld * (size_t)(nev + 1), &vecs));
        if (c->have_x0) BK_TRY(bk_eig_set_start_vector(c->ctx, c->eigx0));
    }
    const int s = bk_eig_shiftinvert(c->ctx, J, nev, &eo, &c->elo, c->epl, r->vals_re, r->vals_im, vecs, nullptr, ld,
                                     &nvals, &nconv, &nops);
    bk_op_destroy(J);
    if (s != 0) return s;
    if (c->thick && nvals > 0) {                                        // x0 of the next eigensolve: sum of the Ritz vectors
        double ones[BK_MAX_NEV + 1];
        for (int i = 0; i < nvals; ++i) ones[i] = 1.0;
        BK_TRY(v_multiaxpy(c->ctx, c->n, vecs, ld, nvals, ones, nullptr, 1.0, c->eigx0, nullptr));
        c->have_x0 = true;
    }
    int nu = 0, ni = 0;
    for (int i = 0; i < nvals; ++i) {                       // NaN (unconverged) compares false, as in the mirror
        if (r->vals_re[i] > c->co.tol_stability) {
            ++nu;
            if (std::fabs(r->vals_im[i]) > c->co.tol_stability) ++ni;
        }
    }
    r->nvals = nvals;
    r->eig_converged = nconv >= nev;
    r->eig_numops = nops;
    c->n_unstable_prev = c->n_unstable;                     // update_stability!, src/Continuation.jl:274-278
    c->n_imag_prev = c->n_imag;
    c->n_unstable = nu;
    c->n_imag = ni;
    c->nvals = nvals;
    for (int i = 0; i < nvals; ++i) { c->vals_re[i] = r->vals_re[i]; c->vals_im[i] = r->vals_im[i]; }
    return 0;
}

}  // namespace

extern "C" {

int bk_cont_create(bk_ctx* ctx, bk_problem* prob, const double* params, int nparams, int ipar, const double* u0,
                   double p0, const double* u1, double p1, const bk_cont_opts* copts, const bk_newton_opts* nopts,
                   const bk_bordering_opts* bopts, const bk_gmres_opts* lsopts, bk_precond* pl,
                   const bk_eig_opts* eopts, const bk_gmres_opts* eig_lsopts, bk_precond* eig_pl,
                   bk_cont_step_result* init, bk_cont** out) {
    if (!ctx || !prob || !params || !u0 || !u1 || !copts || !nopts || !bopts || !lsopts || !out) return -1;
    if (ipar < 0 || ipar >= nparams || nparams > BK_MAX_PARAMS) return set_error(ctx, "bad parameter index");
    if (copts->detect && (!eopts || !eig_lsopts)) return set_error(ctx, "bk_cont_create: detect needs eigensolver options");
    if (nopts->max_iterations > BK_MAX_NEWTON_ITER) return set_error(ctx, "max_iterations > %d", BK_MAX_NEWTON_ITER);
    bk_cont* c = new bk_cont();
    c->ctx = ctx; c->prob = prob; c->n = prob->nloc;
    for (int a = 0; a < prob->desc.ndim; ++a) c->Nglob *= prob->desc.n[a];
    if (prob->desc.pde == BK_PDE_CGL2D) c->Nglob *= 2.0;
    for (int i = 0; i < BK_MAX_PARAMS; ++i) c->params[i] = i < nparams ? params[i] : 0.0;
    c->nparams = nparams; c->ipar = ipar;
    c->co = *copts; c->no = *nopts; c->bo = *bopts; c->lo = *lsopts; c->pl = pl;
    c->has_eig = copts->detect != 0;
    if (c->has_eig) { c->eo = *eopts; c->elo = *eig_lsopts; c->epl = eig_pl; }
    c->thick = c->has_eig && ctx->opt("eig_thick_start", 0.0) != 0.0;
    double** bufs[] = {&c->zu, &c->zoldu, &c->tauu, &c->predu, &c->work, &c->work2, &c->eigx0};
    for (double** b : bufs) {
        if (b == &c->eigx0 && !c->thick) continue;
        if (hipMalloc(b, c->n * sizeof(double)) != hipSuccess) {
            bk_cont_destroy(c);
            return set_error(ctx, "bk_cont_create: device allocation failed");
        }
    }
    // initialize!, Palc.jl:112-123 after the two Newton solves of Continuation.jl:349-456
    int s = 0;
    c->ds = copts->ds;
    if ((s = v_copy(ctx, c->n, u0, c->zu)) || (s = v_copy(ctx, c->n, u0, c->zoldu))) { bk_cont_destroy(c); return s; }
    c->zp = c->zoldp = p0;
    bk_cont_step_result r0 = {};
    r0.converged = 1; r0.p = p0; r0.ds_used = c->ds; r0.n_unstable = -1; r0.n_imag = -1;
    if (c->has_eig && (s = eigen(c, &r0))) { bk_cont_destroy(c); return s; }
    r0.n_unstable = c->n_unstable; r0.n_imag = c->n_imag;
    if ((s = secant_tangent(c, u1, p1, u0, p0, c->ds)) || (s = predictor(c))) { bk_cont_destroy(c); return s; }
    r0.ds_next = c->ds;
    if (init) *init = r0;
    *out = c;
    return 0;
}

int bk_cont_destroy(bk_cont* c) {
    if (!c) return 0;
    double* bufs[] = {c->zu, c->zoldu, c->tauu, c->predu, c->work, c->work2, c->eigx0};
    for (double* b : bufs)
        if (b) (void)hipFree(b);
    delete c;
    return 0;
}

int bk_cont_step(bk_cont* c, bk_cont_step_result* r) {
    if (!c || !r) return -1;
    bk_ctx* ctx = c->ctx;
    *r = bk_cont_step_result{};
    r->n_unstable = c->n_unstable; r->n_imag = c->n_imag;
    r->p = c->zp; r->ds_used = c->ds; r->ds_next = c->ds;
    // done(it, state), src/Continuation.jl:254-257: the point that reached the boundary of [p_min, p_max] was the last one
    if (c->step > 0 && !(c->zp > c->co.p_min && c->zp < c->co.p_max)) {
        r->stop = 2;
        return 0;
    }
    double* x = c->work;
    BK_TRY(v_copy(ctx, c->n, c->predu, x));
    double p = c->predp;
    bk_newton_result nr = {};
    if (c->predp <= c->co.p_min || c->predp >= c->co.p_max) {
        // corrector!(::PALC) hands over to the Natural corrector at the clamped parameter (Palc.jl:157-160,
        // Natural.jl:38-58): plain Newton from z_pred.u at p = clamp(z_pred.p); the point lands ON the boundary
        c->predp = p = std::min(std::max(c->predp, c->co.p_min), c->co.p_max);
        double par[BK_MAX_PARAMS];
        for (int i = 0; i < c->nparams; ++i) par[i] = c->params[i];
        par[c->ipar] = p;
        BK_TRY(bk_newton(ctx, c->prob, x, par, c->nparams, &c->no, &c->lo, c->pl, &nr));
        r->natural = 1;
    } else {
        // corrector!: newton_palc from the predictor; z (the last point) and tau are inputs
        BK_TRY(bk_newton_palc(ctx, c->prob, x, &p, c->zu, c->zp, c->tauu, c->taup, c->ds, c->co.theta, c->params,
                              c->nparams, c->ipar, c->co.p_min, c->co.p_max, &c->no, &c->bo, &c->lo, c->pl, &nr));
    }
    r->converged = nr.converged; r->itnewton = nr.itnewton; r->itlinear = nr.itlinear;
    c->converged = nr.converged != 0;
    for (int i = 0; i <= nr.itnewton && i <= BK_MAX_NEWTON_ITER; ++i) r->residuals[i] = nr.residuals[i];
    const int prev_unst = c->n_unstable;
    if (nr.converged) {
        BK_TRY(v_copy(ctx, c->n, c->zu, c->zoldu));
        c->zoldp = c->zp;
        BK_TRY(v_copy(ctx, c->n, x, c->zu));
        c->zp = p;
        if (c->has_eig) {
            BK_TRY(eigen(c, r));
            if (prev_unst != -1 && c->n_unstable != prev_unst) r->bifurcation = 1;      // Bifurcations.jl:22-28
        }
        c->step += 1;
    }
    r->p = c->zp;
    r->n_unstable = c->n_unstable; r->n_imag = c->n_imag;
    r->step = c->step;
    // _step_size_control!, Contbase.jl:77-102 (skipped while a bisection drives ds itself: stepsizecontrol = false)
    double ds = c->ds;
    if (!c->stepsizecontrol) {
        // keep ds
    } else if (!nr.converged) {
        if (std::fabs(ds) <= c->co.dsmin) {
            r->stop = 1;
            return 0;
        }
        ds = std::copysign(std::max(std::fabs(ds) / 2.0, c->co.dsmin), ds);
    } else {
        const double Nmax = c->no.max_iterations;
        const double factor = (Nmax - nr.itnewton) / Nmax;
        ds = ds * (1.0 + c->co.a * factor * factor);
    }
    if (c->stepsizecontrol)
        ds = std::copysign(std::min(std::max(std::fabs(ds), c->co.dsmin), c->co.dsmax), ds);      // clamp_ds
    c->ds = ds;
    r->ds_next = ds;
    if (nr.converged) {                                                 // Palc.jl:140-143
        if (c->co.tangent == 0) {
            BK_TRY(secant_tangent(c, c->zu, c->zp, c->zoldu, c->zoldp, ds));
            r->tangent_converged = 1;
        } else {
            BK_TRY(bordered_tangent(c, &r->tangent_converged));
        }
    }
    BK_TRY(predictor(c));
    return 0;
}

static int cont_copy_state(bk_cont* dst, const bk_cont* src) {      // copyto!(dst, src) on ContState
    double* d[] = {dst->zu, dst->zoldu, dst->tauu, dst->predu};
    const double* from[] = {src->zu, src->zoldu, src->tauu, src->predu};
    for (int i = 0; i < 4; ++i) BK_TRY(v_copy(src->ctx, src->n, from[i], d[i]));
    dst->zp = src->zp; dst->zoldp = src->zoldp; dst->taup = src->taup; dst->predp = src->predp; dst->ds = src->ds;
    dst->n_unstable = src->n_unstable; dst->n_imag = src->n_imag; dst->step = src->step;
    dst->n_unstable_prev = src->n_unstable_prev; dst->n_imag_prev = src->n_imag_prev;
    dst->stepsizecontrol = src->stepsizecontrol; dst->converged = src->converged;
    dst->nvals = src->nvals;
    for (int i = 0; i < src->nvals; ++i) { dst->vals_re[i] = src->vals_re[i]; dst->vals_im[i] = src->vals_im[i]; }
    if (src->thick && dst->thick && src->have_x0) BK_TRY(v_copy(src->ctx, src->n, src->eigx0, dst->eigx0));
    dst->have_x0 = src->thick && dst->thick && src->have_x0;
    return 0;
}

int bk_cont_clone(bk_cont* src, bk_cont** out) {
    if (!src || !out) return -1;
    bk_cont* c = new bk_cont(*src);                           // options, solver settings, scalars
    c->zu = c->zoldu = c->tauu = c->predu = c->work = c->work2 = c->eigx0 = nullptr;
    double** bufs[] = {&c->zu, &c->zoldu, &c->tauu, &c->predu, &c->work, &c->work2, &c->eigx0};
    for (double** b : bufs) {
        if (b == &c->eigx0 && !c->thick) continue;
        if (hipMalloc(b, c->n * sizeof(double)) != hipSuccess) {
            bk_cont_destroy(c);
            return set_error(src->ctx, "bk_cont_clone: device allocation failed");
        }
    }
    const int s = cont_copy_state(c, src);
    if (s != 0) { bk_cont_destroy(c); return s; }
    *out = c;
    return 0;
}

// locate_bifurcation!(iter, state), src/Bifurcations.jl:159-349: bisection with the continuation step itself.  `c` is the
// state right after the step that changed n_unstable; on return it sits right after (status guess / converged) or right
// before (guessL) the bifurcation point, its (n_unstable, n_imag) pairs bracket the crossing, and the predictor is rebuilt.
int bk_cont_locate_bifurcation(bk_cont* c, const bk_bisection_opts* bo, bk_bisection_result* res) {
    if (!c || !bo || !res) return -1;
    *res = bk_bisection_result{};
    const int n2 = c->n_unstable, n1 = c->n_unstable_prev;
    if (n1 == -1 || n2 == -1 || std::fabs(c->ds) < c->co.dsmin) return 0;                 // status none
    bk_cont *after = nullptr, *state = nullptr, *before = nullptr;
    int s = bk_cont_clone(c, &after);
    if (!s) s = bk_cont_clone(c, &state);
    if (!s) s = bk_cont_clone(c, &before);
    auto cleanup = [&]() { bk_cont_destroy(after); bk_cont_destroy(state); bk_cont_destroy(before); };
    if (s) { cleanup(); return s; }
    std::swap(before->n_unstable, before->n_unstable_prev);
    std::swap(before->n_imag, before->n_imag_prev);
    std::swap(before->zp, before->zoldp);
    state->ds *= -1.0;
    state->step = 0;
    state->stepsizecontrol = false;
    int last_unst = n2;
    double interval[2] = {std::min(state->zp, state->zoldp), std::max(state->zp, state->zoldp)};
    int ind = interval[0] == state->zp ? 0 : 1;
    int n_inv = 0, steps = 0;
    bool have_next = true, first = true;
    while (true) {
        if (!state->converged) break;
        if (!have_next) break;
        // (the first pass looks at the initial state itself: same count -> ds is halved before the first backward step)
        (void)first;
        if (state->n_unstable == last_unst) state->ds /= 2.0;
        else { state->ds /= -2.0; n_inv += 1; ind = 1 - ind; }
        last_unst = state->n_unstable;
        if ((s = predictor(state))) break;                                                 // update_predictor!, Palc.jl:148-151
        if ((s = cont_copy_state(n_inv % 2 == 0 ? after : before, state))) break;
        if (state->step > 0) interval[ind] = state->zp;
        double rm = INFINITY;                                                               // rightmost: smallest |Re|
        for (int i = 0; i < state->nvals; ++i)
            if (!std::isnan(state->vals_re[i])) rm = std::min(rm, std::fabs(state->vals_re[i]));
        const bool located = rm < bo->tol_bisection_eigenvalue;
        if (!(std::fabs(state->ds) >= bo->dsmin_bisection && state->step < bo->max_bisection_steps &&
              n_inv < bo->n_inversion && !located))
            break;
        if (state->step > bo->max_steps) { have_next = false; continue; }                  // done(it, state), Continuation.jl:254
        bk_cont_step_result r;
        if ((s = bk_cont_step(state, &r))) break;
        steps += 1;
        have_next = r.stop == 0;
        first = false;
    }
    if (s) { cleanup(); return s; }
    const bk_cont* src;
    if (n_inv % 2 == 0) {
        res->status = n_inv >= bo->n_inversion ? 2 : 1;
        src = state;
        res->n_unstable[0] = state->n_unstable; res->n_unstable[1] = before->n_unstable;
        res->n_imag[0] = state->n_imag; res->n_imag[1] = before->n_imag;
        interval[0] = state->zp; interval[1] = before->zp;
    } else {
        res->status = 3;
        src = after;
        res->n_unstable[0] = after->n_unstable; res->n_unstable[1] = state->n_unstable;
        res->n_imag[0] = after->n_imag; res->n_imag[1] = state->n_imag;
        interval[0] = state->zp; interval[1] = after->zp;
    }
    // _copyto! of z_old, z_pred, z, tau and the eigenvalues; ds, step and the solver state of `c` stay
    {
        const double keep_ds = c->ds;
        const int keep_step = c->step;
        const bool keep_ssc = c->stepsizecontrol;
        s = cont_copy_state(c, src);
        c->ds = keep_ds; c->step = keep_step; c->stepsizecontrol = keep_ssc; c->converged = true;
    }
    c->n_unstable = res->n_unstable[0]; c->n_unstable_prev = res->n_unstable[1];
    c->n_imag = res->n_imag[0]; c->n_imag_prev = res->n_imag[1];
    if (!s) s = predictor(c);                                                               // update_predictor!(_state, iter)
    res->interval[0] = std::min(interval[0], interval[1]);
    res->interval[1] = std::max(interval[0], interval[1]);
    res->steps = steps;
    res->p = c->zp;
    res->nvals = c->nvals;
    for (int i = 0; i < c->nvals; ++i) { res->vals_re[i] = c->vals_re[i]; res->vals_im[i] = c->vals_im[i]; }
    {   // _get_bifurcation_type, codim-1 cases (Bifurcations.jl:95-130): 1 bp, 2 hopf, 3 nd
        const int dn = std::abs(res->n_unstable[0] - res->n_unstable[1]), di = std::abs(res->n_imag[0] - res->n_imag[1]);
        res->type = dn == 1 ? (di == 0 ? 1 : (di == 1 ? 2 : 3)) : (dn == 2 ? (di == 2 ? 2 : 3) : (dn > 2 ? 3 : 0));
    }
    cleanup();
    return s;
}

int bk_cont_get(bk_cont* c, double* u, double* p, double* tauu, double* taup, double* ds) {
    if (!c) return -1;
    if (u) BK_TRY(v_copy(c->ctx, c->n, c->zu, u));
    if (tauu) BK_TRY(v_copy(c->ctx, c->n, c->tauu, tauu));
    if (p) *p = c->zp;
    if (taup) *taup = c->taup;
    if (ds) *ds = c->ds;
    return 0;
}

}  // extern "C"

// ================================================================== deflated Newton
namespace {

struct Deflation {
    bk_ctx* ctx;
    size_t n;
    const double* const* roots;
    int nroots;
    double power, alpha, delta;
    int mean;
    double* tmp;
    // M(u) = acc_i ( <u - r_i, u - r_i>^-power + alpha ), src/DeflationOperator.jl:124-138 (in-place version)
    int M(const double* u, double* out) const {
        if (nroots == 0) { *out = 1.0; return 0; }
        double acc = 0.0;
        for (int i = 0; i < nroots; ++i) {
            double d;
            BK_TRY(v_axpbyz(ctx, n, 1.0, u, -1.0, roots[i], tmp));
            BK_TRY(v_dot(ctx, n, tmp, tmp, &d));
            const double m = 1.0 / std::pow(d, power) + alpha;
            acc = i == 0 ? m : (mean ? acc + m : acc * m);
        }
        if (mean) acc /= nroots;
        *out = acc;
        return 0;
    }
    // dM(u) . du by finite differences, Val(:dMwithTmp) :160-169 with autodiff = false; `up` is scratch
    int dM(const double* u, const double* du, double Mu, double* up, double* out) const {
        if (nroots == 0) { *out = 0.0; return 0; }
        double Mp;
        BK_TRY(v_axpbyz(ctx, n, 1.0, u, delta, du, up));
        BK_TRY(M(up, &Mp));
        *out = (Mp - Mu) / delta;
        return 0;
    }
};

}  // namespace

extern "C" {

// solve(prob, defOp, options, DeflatedProblemCustomLS()), src/DeflationOperator.jl:340-355: _newton (src/Newton.jl:66-114)
// on the deflated functional M(u) F(u); every linear solve is DeflatedProblemCustomLS (:264-312): two solves with the
// plain Jacobian (ls(J, rhs, Fu), src/LinearSolver.jl:15-19) recombined as h = (h1 - z h2) / M(u), z = dM.h1 / (M + dM.h2).
int bk_newton_deflated(bk_ctx* ctx, bk_problem* prob, double* x, const double* params, int nparams,
                       const double* const* roots, int nroots, double power, double alpha, int accumulator_mean,
                       double delta, const bk_newton_opts* no, const bk_gmres_opts* lsopts, bk_precond* pl,
                       bk_newton_result* res) {
    if (!ctx || !prob || !x || !params || !no || !lsopts || !res || (nroots > 0 && !roots)) return -1;
    if (no->max_iterations > BK_MAX_NEWTON_ITER) return set_error(ctx, "max_iterations > %d", BK_MAX_NEWTON_ITER);
    const size_t n = prob->nloc;
    WsGuard ws(ctx);
    double *Fu = nullptr, *fx = nullptr, *h1 = nullptr, *h2 = nullptr, *tmp = nullptr, *up = nullptr;
    BK_TRY(ws.get(n, &Fu)); BK_TRY(ws.get(n, &fx)); BK_TRY(ws.get(n, &h1)); BK_TRY(ws.get(n, &h2));
    BK_TRY(ws.get(n, &tmp)); BK_TRY(ws.get(n, &up));
    Deflation D{ctx, n, roots, nroots, power, alpha, delta, accumulator_mean, tmp};
    const bool inf = no->norm_inf != 0;
    auto deflated_residual = [&](double* Mu, double* r) -> int {      // (dfp)(u, par), :194-198
        BK_TRY(bk_residual(prob, x, params, nparams, Fu));
        BK_TRY(D.M(x, Mu));
        BK_TRY(v_copy(ctx, n, Fu, fx));
        BK_TRY(v_scale(ctx, n, *Mu, fx));
        return inf ? v_nrminf(ctx, n, fx, r) : v_nrm2(ctx, n, fx, r);
    };
    double Mu, r;
    BK_TRY(deflated_residual(&Mu, &r));
    int step = 0, itlin = 0;
    res->residuals[0] = r;
    while (step < no->max_iterations && r > no->tol) {
        bk_op* J = nullptr;
        BK_TRY(bk_jacobian(prob, x, params, nparams, &J));
        int s = 0, cv = 0, it[2] = {0, 0};
        if (nroots == 0) {
            int it1 = 0;
            double rn;
            s = bk_gmres(ctx, J, fx, h1, 0.0, 1.0, lsopts, pl, &cv, &it1, &rn);
            itlin += it1;
        } else {
            s = bk_gmres2(ctx, J, fx, Fu, h1, h2, 0.0, 1.0, lsopts, pl, &cv, it);
            itlin += it[0] + it[1];
        }
        bk_op_destroy(J);
        if (s != 0) return s;
        if (nroots > 0) {
            double z1, z2;
            BK_TRY(D.dM(x, h1, Mu, up, &z1));
            BK_TRY(D.dM(x, h2, Mu, up, &z2));
            const double z = z1 / (Mu + z2);
            BK_TRY(v_axpby(ctx, n, -z, h2, 1.0, h1));            // h = (h1 - z h2) / Mu
            BK_TRY(v_scale(ctx, n, 1.0 / Mu, h1));
        }
        BK_TRY(v_axpby(ctx, n, -1.0, h1, 1.0, x));                // x = minus!!(x, u), src/Newton.jl:97
        BK_TRY(deflated_residual(&Mu, &r));
        step += 1;
        res->residuals[step] = r;
    }
    res->converged = res->residuals[step] < no->tol;
    res->itnewton = step;
    res->itlinear = itlin;
    return 0;
}

}  // extern "C"
