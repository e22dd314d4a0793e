// bk_problem: grid description + slab decomposition + residual / Jacobian-operator entry points.
#include "ops.h"

using namespace bk;

int bk_problem::apply(int mode, const double* v, const double* u, const double* params, double a0, double a1,
                      double* out, const double* ag) {
    const bk_problem_desc& d = desc;
    if (d.pde == BK_PDE_SH) {
        ShArgs a;
        a.nx = d.n[0];
        a.ny = d.n[1];
        a.nz = d.ndim == 3 ? (hi - lo) : 1;
        a.nzg = d.ndim == 3 ? d.n[2] : 1;
        a.zoff = d.ndim == 3 ? lo : 0;
        a.ax = ainv[0]; a.ay = ainv[1]; a.az = d.ndim == 3 ? ainv[2] : 0.0;
        a.l = params[0]; a.nu = params[1];
        a.a0 = a0; a.a1 = a1; a.mode = mode;
        if (ag) { a.ag = *ag; a.ag_set = true; }
        a.v = v; a.u = u; a.out = out;
        a.halo_lo = halo_lo; a.halo_hi = halo_hi;
        if (ctx->nranks > 1) {
            // Overlap: the halo exchange (2 planes per face) runs on its own stream while the z-chunks that read no
            // halo plane are computed; the two face chunks follow once it has landed.  Both communicator kinds enqueue the
            // exchange (ncclSend / ncclRecv group; the host-staged communicator's proxy hand-over, context.hip).
            const bool overlap = ctx->opt("halo_overlap", 1.0) != 0.0;
            if (overlap) {
                if (!ctx->comm_stream) {
                    BK_HIP(ctx, hipStreamCreateWithFlags(&ctx->comm_stream, hipStreamNonBlocking));
                    BK_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_ready, hipEventDisableTiming));
                    BK_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_halo, hipEventDisableTiming));
                }
                BK_HIP(ctx, hipEventRecord(ctx->ev_ready, ctx->stream));          // v is complete on the compute stream
                BK_HIP(ctx, hipStreamWaitEvent(ctx->comm_stream, ctx->ev_ready, 0));
                BK_TRY(halo_exchange(ctx, ctx->comm_stream, v, plane, a.nz, 2, halo_lo, halo_hi));
                BK_HIP(ctx, hipEventRecord(ctx->ev_halo, ctx->comm_stream));
                a.part = 1;
                BK_TRY(sh_apply(ctx, a));
                BK_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_halo, 0));
                a.part = 2;
                return sh_apply(ctx, a);
            }
            {
                ProfScope ps(ctx, "halo", 32.0 * plane * 2);
                BK_TRY(halo_exchange(ctx, ctx->stream, v, plane, a.nz, 2, halo_lo, halo_hi));
            }
            if (ctx->opt("halo_split", 1.0) != 0.0) {
                a.part = 1;
                BK_TRY(sh_apply(ctx, a));
                a.part = 2;
            }
        }
        return sh_apply(ctx, a);
    }
    if (ag) return set_error(ctx, "bk_problem::apply: separately scaled parts are implemented for BK_PDE_SH only");
    if (d.pde == BK_PDE_CGL2D) {
        CglArgs a;
        a.nx = d.n[0]; a.ny = d.n[1];
        a.ax = ainv[0]; a.ay = ainv[1];
        a.r = params[0]; a.mu = params[1]; a.nu = params[2]; a.c3 = params[3]; a.c5 = params[4]; a.gamma = params[5];
        a.a0 = a0; a.a1 = a1; a.mode = mode;
        a.v = v; a.u = u; a.out = out;
        return cgl_apply(ctx, a);
    }
    if (d.pde == BK_PDE_SH1D) {
        Sh1dArgs a;
        a.nx = d.n[0];
        a.ax = ainv[0];
        a.lam = params[0]; a.nu = params[1];
        a.a0 = a0; a.a1 = a1; a.mode = mode;
        a.v = v; a.u = u; a.out = out;
        return sh1d_apply(ctx, a);
    }
    return set_error(ctx, "unknown pde kind %d", d.pde);
}

int bk_problem::jvp_axpy_dot(const double* v, const double* u, const double* params, double a0, double a1, double c,
                             const double* r, double* out, double* dot, int* fused) {
    *fused = 0;
    const bk_problem_desc& d = desc;
    if (d.pde != BK_PDE_SH || d.ndim != 3) return 0;
    // ranks (round 5): the same fused kernel over the slab with the halo planes in place -- the exchange runs in line first (this
    // launch covers every z-chunk, there is nothing to overlap it with), the per-tile partial sums are all-reduced by reduce_finish
    const bool ranks = ctx->nranks > 1;
    // (slabs of at least 8 planes, every rank alike: the decision must not differ between ranks -- d.n[2] / nranks is global)
    if (ranks && (ctx->opt("jvp_fused_dot_ranks", 1.0) == 0.0 || !halo_lo || !halo_hi || d.n[2] / ctx->nranks < 8)) return 0;
    ShArgs a;
    a.nx = d.n[0]; a.ny = d.n[1]; a.nz = hi - lo; a.nzg = d.n[2]; a.zoff = lo;
    a.ax = ainv[0]; a.ay = ainv[1]; a.az = ainv[2];
    a.l = params[0]; a.nu = params[1];
    a.a0 = a0; a.a1 = a1; a.mode = 0;
    a.v = v; a.u = u; a.out = out;
    a.halo_lo = ranks ? halo_lo : nullptr; a.halo_hi = ranks ? halo_hi : nullptr;
    a.addv = r; a.addc = r ? c : 0.0;
    if (ranks) {
        // The decision must not differ between ranks (the fused path exchanges the halo on ctx->stream, the other one on the
        // communication stream): it is taken from GLOBAL quantities only -- the options and the two slab heights a z-partition of
        // d.n[2] planes over nranks ranks can produce -- never from this rank's own height or pointers.  A vector that is
        // not 16-byte aligned (nothing the library hands out) is an error here instead of a silent change of path.
        ShArgs g = a;
        g.addv = nullptr;
        g.nz = d.n[2] / ctx->nranks;
        const bool ok_lo = sh_fused_dot_ok(ctx, g);
        g.nz = (d.n[2] + ctx->nranks - 1) / ctx->nranks;
        if (!ok_lo || !sh_fused_dot_ok(ctx, g)) return 0;
        if (r && ((uintptr_t)r & 15)) return set_error(ctx, "jvp_axpy_dot: on ranks the vectors must be 16-byte aligned");
    }
    if (!sh_fused_dot_ok(ctx, a)) return ranks ? set_error(ctx, "jvp_axpy_dot: slab height outside the partition's two heights") : 0;
    if (ranks) {
        ProfScope ps(ctx, "halo", 32.0 * plane * 2);
        BK_TRY(halo_exchange(ctx, ctx->stream, v, plane, a.nz, 2, halo_lo, halo_hi));
    }
    int nb = 0;
    a.dot_blocks = &nb;
    BK_TRY(sh_apply(ctx, a));
    BK_TRY(reduce_finish(ctx, nb, 1, 0));
    *dot = ctx->h_red[0];
    *fused = 1;
    return 0;
}

int bk_problem::dparam(const double* u, const double* params, int nparams, int ipar, double eps, const double* f0,
                       double* out) {
    if (ctx->opt("fd_dparam", 1.0) != 0.0) {
        // the scalar keeps the quotient's own rounding: ((p + eps) - p) / eps
        const volatile double pe = params[ipar] + eps;
        const double c = (pe - params[ipar]) / eps;
        const size_t npts = desc.pde == BK_PDE_CGL2D ? nloc / 2 : nloc;
        return pde_dparam(ctx, desc.pde, ipar, npts, c, u, out);
    }
    // literal form: two residual evaluations and a scaled difference (Palc.jl:239-240)
    double par[BK_MAX_PARAMS];
    for (int i = 0; i < nparams; ++i) par[i] = params[i];
    par[ipar] = params[ipar] + eps;
    BK_TRY(apply(1, u, u, par, 0.0, 1.0, out));
    if (f0) return v_axpby(ctx, nloc, -1.0 / eps, f0, 1.0 / eps, out);
    WsGuard ws(ctx);
    double* tmp = nullptr;
    BK_TRY(ws.get(nloc, &tmp));
    BK_TRY(apply(1, u, u, params, 0.0, 1.0, tmp));
    return v_axpby(ctx, nloc, -1.0 / eps, tmp, 1.0 / eps, out);
}

int PdeJacobian::apply(const double* x, const double*, double a0, double a1, double* out, double*) {
    // the SH Jacobians are symmetric (issymmetric = true, examples/SH3d.jl:123): only cGL has a distinct adjoint
    return prob->apply(adjoint && prob->desc.pde == BK_PDE_CGL2D ? 2 : 0, x, u, params, a0, a1, out);
}

int PdeJacobian::apply_parts(const double* x, double a0, double aL, double ag, double* out) {
    if (prob->desc.pde != BK_PDE_SH) return 1;
    BK_TRY(prob->apply(0, x, u, params, a0, aL, out, &ag));
    return 0;
}

const bk_problem* PdeJacobian::sh_problem() const { return prob->desc.pde == BK_PDE_SH ? prob : nullptr; }

int PdeJacobian::apply_axpy_dot(const double* x, double a0, double a1, double c, const double* r, double* out, double* dot) {
    int fused = 0;
    if (!(adjoint && prob->desc.pde == BK_PDE_CGL2D)) BK_TRY(prob->jvp_axpy_dot(x, u, params, a0, a1, c, r, out, dot, &fused));
    return fused ? 0 : bk_op::apply_axpy_dot(x, a0, a1, c, r, out, dot);
}

// defaults of the fused interfaces: the separate passes
int bk_op::apply_axpy_dot(const double* x, double a0, double a1, double c, const double* r, double* out, double* dot) {
    if (ntail != 0) return set_error(ctx, "apply_axpy_dot: unbordered operators only");
    BK_TRY(apply(x, nullptr, a0, a1, out, nullptr));
    return v_axpy_dot(ctx, n, c, r, out, x, dot);
}

int bk_precond::apply_dot_pre_axpy(double* y, double c, const double* r, double* out, double* dot) {
    BK_TRY(v_axpby(ctx, n, c, r, 1.0, y));
    return apply_dot(y, out, dot);
}

int bk_precond::apply_pw(const double* x, const bk::DctFuse& d, double cx, double ct, double* out) {
    // the separate passes: t = d .* x ; t = Pl \ t ; out = cx x + ct t
    if (cx == 0.0) {
        BK_TRY(v_pw_scale(ctx, n, x, d.u, d.A, d.B, d.C, out));
        BK_TRY(apply(out, out));
        return ct == 1.0 ? 0 : v_scale(ctx, n, ct, out);
    }
    if (out == x) return set_error(ctx, "apply_pw: out must not alias x");
    WsGuard ws(ctx);
    double* t = nullptr;
    BK_TRY(ws.get(n, &t));
    BK_TRY(v_pw_scale(ctx, n, x, d.u, d.A, d.B, d.C, t));
    BK_TRY(apply(t, t));
    return v_axpbyz(ctx, n, cx, x, ct, t, out);
}

bool PdeJacobian::sh_state(const double** u_, double* l, double* nu) const {
    if (prob->desc.pde != BK_PDE_SH) return false;
    *u_ = u; *l = params[0]; *nu = params[1];
    return true;
}

int bk_precond::apply_dot(const double* v, double* out, double* dot) {
    BK_TRY(apply(v, out));
    return v_dot(ctx, n, v, out, dot);
}

static int nparams_of(int pde) { return pde == BK_PDE_CGL2D ? 6 : 2; }

extern "C" {

int bk_problem_create(bk_ctx* ctx, const bk_problem_desc* desc, bk_problem** out) {
    if (!ctx || !desc || !out) return -1;
    const bk_problem_desc& d = *desc;
    if (d.pde == BK_PDE_SH && d.ndim != 2 && d.ndim != 3) return set_error(ctx, "BK_PDE_SH needs ndim 2 or 3");
    if (d.pde == BK_PDE_SH1D && d.ndim != 1) return set_error(ctx, "BK_PDE_SH1D needs ndim 1");
    if (d.pde == BK_PDE_CGL2D && d.ndim != 2) return set_error(ctx, "BK_PDE_CGL2D needs ndim 2");
    if (d.pde != BK_PDE_SH && d.pde != BK_PDE_SH1D && d.pde != BK_PDE_CGL2D) return set_error(ctx, "unknown pde kind");
    for (int a = 0; a < d.ndim; ++a)
        if (d.n[a] < 2 || !(d.l[a] > 0.0)) return set_error(ctx, "bad grid extent on axis %d", a);
    bk_problem* p = new bk_problem();
    p->ctx = ctx;
    p->desc = d;
    for (int a = d.ndim; a < 3; ++a) { p->desc.n[a] = 1; p->desc.l[a] = 1.0; }
    for (int a = 0; a < d.ndim; ++a) {
        const double h = 2.0 * d.l[a] / d.n[a];     // hx = 2lx/Nx, examples/SH3d.jl:18
        p->ainv[a] = 1.0 / (h * h);
    }
    if (ctx->nranks > 1) {
        if (!(d.pde == BK_PDE_SH && d.ndim == 3)) {
            delete p;
            return set_error(ctx, "multi-GPU decomposition is implemented for the 3-D Swift-Hohenberg problem only");
        }
        const int nz = d.n[2], R = ctx->nranks, r = ctx->rank;
        const int base = nz / R, rem = nz % R;
        p->lo = r * base + (r < rem ? r : rem);
        p->hi = p->lo + base + (r < rem ? 1 : 0);
        if (p->hi - p->lo < 2) { delete p; return set_error(ctx, "z-slab thinner than 2 planes"); }
        p->plane = (size_t)d.n[0] * d.n[1];
        p->nloc = p->plane * (size_t)(p->hi - p->lo);
        if (hipMalloc(&p->halo_lo, 2 * p->plane * sizeof(double)) != hipSuccess ||
            hipMalloc(&p->halo_hi, 2 * p->plane * sizeof(double)) != hipSuccess) {
            delete p;
            return set_error(ctx, "halo allocation failed");
        }
    } else {
        const int slow = d.ndim - 1;
        p->lo = 0;
        p->hi = d.n[slow];
        size_t n = 1;
        for (int a = 0; a < d.ndim; ++a) n *= (size_t)d.n[a];
        p->plane = n / (size_t)d.n[slow];
        p->nloc = n * (d.pde == BK_PDE_CGL2D ? 2 : 1);
    }
    *out = p;
    return 0;
}

int bk_problem_destroy(bk_problem* p) {
    if (!p) return 0;
    if (p->halo_lo) (void)hipFree(p->halo_lo);
    if (p->halo_hi) (void)hipFree(p->halo_hi);
    delete p;
    return 0;
}

int bk_problem_nlocal(bk_problem* p, size_t* nlocal, int* slab_lo, int* slab_hi) {
    if (!p) return -1;
    if (nlocal) *nlocal = p->nloc;
    if (slab_lo) *slab_lo = p->lo;
    if (slab_hi) *slab_hi = p->hi;
    return 0;
}

int bk_residual(bk_problem* p, const double* u, const double* params, int nparams, double* out) {
    if (!p || !u || !params || !out) return -1;
    if (nparams < nparams_of(p->desc.pde)) return set_error(p->ctx, "bk_residual: expected %d parameters", nparams_of(p->desc.pde));
    if (u == out) return set_error(p->ctx, "bk_residual: out must not alias u");
    return p->apply(1, u, u, params, 0.0, 1.0, out);
}

int bk_residual_dparam(bk_problem* p, const double* u, const double* params, int nparams, int ipar, double eps,
                       double* out) {
    if (!p || !u || !params || !out) return -1;
    if (nparams < nparams_of(p->desc.pde) || nparams > BK_MAX_PARAMS)
        return set_error(p->ctx, "bk_residual_dparam: expected %d parameters", nparams_of(p->desc.pde));
    if (ipar < 0 || ipar >= nparams_of(p->desc.pde)) return set_error(p->ctx, "bk_residual_dparam: bad parameter index");
    if (!(eps > 0.0)) return set_error(p->ctx, "bk_residual_dparam: eps must be positive");
    if (u == out) return set_error(p->ctx, "bk_residual_dparam: out must not alias u");
    return p->dparam(u, params, nparams, ipar, eps, nullptr, out);
}

int bk_jacobian(bk_problem* p, const double* u, const double* params, int nparams, bk_op** out) {
    if (!p || !u || !params || !out) return -1;
    if (nparams < nparams_of(p->desc.pde) || nparams > BK_MAX_PARAMS)
        return set_error(p->ctx, "bk_jacobian: expected %d parameters", nparams_of(p->desc.pde));
    PdeJacobian* J = new PdeJacobian();
    J->ctx = p->ctx;
    J->n = p->nloc;
    J->ntail = 0;
    J->prob = p;
    J->u = u;
    for (int i = 0; i < BK_MAX_PARAMS; ++i) J->params[i] = i < nparams ? params[i] : 0.0;
    *out = J;
    return 0;
}

int bk_jacobian_adjoint(bk_problem* p, const double* u, const double* params, int nparams, bk_op** out) {
    BK_TRY(bk_jacobian(p, u, params, nparams, out));
    static_cast<PdeJacobian*>(*out)->adjoint = true;
    return 0;
}

int bk_op_destroy(bk_op* op) {
    delete op;
    return 0;
}

int bk_op_apply(bk_op* op, const double* v, double a0, double a1, double* out) {
    if (!op || !v || !out) return -1;
    if (op->ntail != 0) return set_error(op->ctx, "bk_op_apply: bordered operators are internal");
    if (v == out) return set_error(op->ctx, "bk_op_apply: out must not alias v");
    return op->apply(v, nullptr, a0, a1, out, nullptr);
}

}  // extern "C"
