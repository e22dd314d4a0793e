// Internal header of libbkhip: context, error handling, kernel-launcher prototypes.
// gfx950 (MI355X / CDNA4) only.  See include/bkhip.h for the public C ABI.
#pragma once

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <map>
#include <set>
#include <string>
#include <utility>
#include <vector>

#include "../../include/bkhip.h"

namespace bk {

constexpr int kMaxBasis = 64;      // largest Krylov dimension + 1 the fused kernels are built for
constexpr int kRedSlots = 256;     // doubles in the reduction result buffers
constexpr int kRedBlocks = 1024;   // blocks of a reduction kernel (stage 1); stage 2 is one block
// values per workgroup the stage-1 kernels may leave in d_partials: multidot kMaxBasis + 2, the Gram variant 2 * 32 + 1, block_dots
// 4 * 8 + 36 = 68 (vecops.hip: static_asserts at the launch sites); every launcher checks grid * values against kPartialDoubles
constexpr int kPartialVals = 72;
constexpr size_t kPartialDoubles = (size_t)kRedBlocks * kPartialVals;
constexpr double kCancelTol = 1e-8;   // Pythagorean norm b^2 = |w|^2 - |h|^2 is trusted while b^2 > kCancelTol |w|^2 (host and device Arnoldi steps)
constexpr int kRecChunks = 16;     // most speculative Arnoldi steps per synchronisation (option gmres_chunk)

struct Coefs {                     // by-value kernel argument: coefficients of a fused multi-axpy
    double c[kMaxBasis];
};

struct ProfEntry {
    double ms = 0.0;
    long long calls = 0;
    double bytes = 0.0;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
};

enum CommKind { COMM_NONE = 0, COMM_RCCL = 1, COMM_HOST = 2 };

}  // namespace bk

struct CommProxy;                  // context.hip: proxy thread of the host-staged communicator (in-stream collectives)

struct bk_ctx {
    int device = 0;
    int num_cu = 256;              // compute units of the device (persistent-kernel grids)
    hipStream_t stream = nullptr;
    bool own_stream = false;
    // communicator
    int rank = 0, nranks = 1;
    bk::CommKind comm = bk::COMM_NONE;
    ncclComm_t nccl = nullptr;
    hipStream_t comm_stream = nullptr;   // halo exchange runs here, overlapped with the interior z-chunks of the JVP
    hipEvent_t ev_ready = nullptr, ev_halo = nullptr;
    void* blas = nullptr;          // rocblas_handle, created on first use by the dense transform passes (dct.hip)
    bk_allreduce_fn h_allreduce = nullptr;
    bk_sendrecv_fn h_sendrecv = nullptr;
    void* h_user = nullptr;
    CommProxy* proxy = nullptr;          // host-staged communicator: created with the first collective
    // communicator of the second lane (bk_ctx_set_lane_comm): the lane's collectives are issued concurrently with the
    // context's own and must never be matched against them
    bk_allreduce_fn lane_allreduce = nullptr;
    bk_sendrecv_fn lane_sendrecv = nullptr;
    void* lane_user = nullptr;
    // reduction scratch
    double* d_partials = nullptr;   // [kPartialDoubles]
    double* d_red = nullptr;        // [kRedSlots]
    double* h_red = nullptr;        // pinned [kRedSlots]
    double* h_red_dev = nullptr;    // its device-side address (mapped): single-rank reductions land in it directly
    double* h_rec = nullptr;        // pinned, mapped: records of the device-resident Arnoldi chunks (kRecChunks x (kMaxBasis+2))
    double* h_rec_dev = nullptr;
    // workspace pool (device buffers keyed by size in doubles)
    std::multimap<size_t, double*> pool_free;
    std::map<double*, size_t> pool_all;
    std::vector<double*> host_pinned;
    // options + profiling
    std::map<std::string, double> opts;
    bool prof = false;
    std::map<std::string, bk::ProfEntry> prof_entries;
    std::vector<hipEvent_t> event_pool;
    std::string err;
    // residual history of the linear solves since the last bk_solver_history_reset (option "solver_trace" != 0):
    // one entry per Krylov iteration (the solver's own residual estimate); a negative entry -(k+1) opens solve k
    std::vector<double> hist;
    int hist_solves = 0;
    // block log of the GMRES solves (option "gmres_block_log" != 0; bk_solver_block_log): kBlockLogRec doubles per Arnoldi block --
    // solve number, first column j, steps issued, steps accepted, last pivot ratio, theta[0..3] (NaN: slot unused), residual
    // estimate and tolerance when the block was issued
    static constexpr int kBlockLogRec = 11;
    std::vector<double> block_log;
    int block_log_solves = 0;
    // diagnostics counters of the GMRES solves (plain fields: the solver loop never looks a counter up by name); read through
    // bk_ctx_get_option under the names in diag_get, reset by bk_ctx_set_option(name, 0)
    struct Diag {
        double block_steps = 0.0, block_truncated = 0.0, block_unconsumed = 0.0;    // operator applications issued in blocks / void tails / speculated past convergence
        double last_orth_defect = 0.0, last_orth_estimate = 0.0;                   // option orth_probe
        double check_mismatch = 0.0;                                               // explicit residual checks that contradicted the Arnoldi estimate (stencil-free form)
    } diag;
    double* diag_slot(const std::string& key) {
        if (key == "gmres_block_steps") return &diag.block_steps;
        if (key == "gmres_block_truncated") return &diag.block_truncated;
        if (key == "gmres_block_unconsumed") return &diag.block_unconsumed;
        if (key == "gmres_last_orth_defect") return &diag.last_orth_defect;
        if (key == "gmres_last_orth_estimate") return &diag.last_orth_estimate;
        if (key == "gmres_check_mismatch") return &diag.check_mismatch;
        return nullptr;
    }
    std::vector<double> newton_shifts; // option gmres_newton_carry: Leja-ordered Ritz values of the last GMRES solve (solver.hip)
    int gmres_last_steps = 1 << 20;   // Arnoldi steps of the previous GMRES solve (speculation ramp of the device-resident chunks)
    const double* eig_x0 = nullptr;   // one-shot start vector of the next eigensolve (bk_eig_set_start_vector)
    // second execution lane (context.hip: ctx_lane): an independent context on the same device -- own non-blocking stream,
    // reduction buffers, workspace pool, profile -- on which the second of two independent linear solves runs concurrently
    // with the first (solver.hip: linsolve2).  Created on first use, destroyed with the context.
    bk_ctx* lane2 = nullptr;
    // Two lanes on RANKS (solver.hip: linsolve2): the (global problem size, Krylov dimension) pairs whose two solves have already run
    // once one after the other on their lanes, so that every pool -- workspace buffers, the proxy's staging buffers, streams, events,
    // kernel attributes -- holds what a concurrent pair of solves asks for.  Cleared whenever an option changes (bk_ctx_set_option).
    std::set<std::pair<size_t, int>> lanes_warm;

    double opt(const char* key, double dflt) const {
        auto it = opts.find(key);
        return it == opts.end() ? dflt : it->second;
    }
};

namespace bk {

int set_error(bk_ctx* ctx, const char* fmt, ...);

#define BK_HIP(ctx, call)                                                                     \
    do {                                                                                      \
        hipError_t e_ = (call);                                                               \
        if (e_ != hipSuccess)                                                                 \
            return bk::set_error((ctx), "%s:%d: %s failed: %s", __FILE__, __LINE__, #call,    \
                                 hipGetErrorString(e_));                                      \
    } while (0)

#define BK_NCCL(ctx, call)                                                                    \
    do {                                                                                      \
        ncclResult_t e_ = (call);                                                             \
        if (e_ != ncclSuccess)                                                                \
            return bk::set_error((ctx), "%s:%d: %s failed: %s", __FILE__, __LINE__, #call,    \
                                 ncclGetErrorString(e_));                                     \
    } while (0)

#define BK_TRY(call)             \
    do {                         \
        int s_ = (call);         \
        if (s_ != 0) return s_;  \
    } while (0)

// ---- second lane (context.hip) --------------------------------------------------------------
bk_ctx* ctx_lane(bk_ctx* ctx);                    // creates it on first use; NULL on failure (error on ctx)
void ctx_lane_merge(bk_ctx* ctx, bk_ctx* lane);   // after a joint solve: profile entries and solver history into ctx

// ---- workspace pool ---------------------------------------------------------------------
int ws_get(bk_ctx* ctx, size_t n, double** out);
void ws_put(bk_ctx* ctx, double* p);

struct WsGuard {    // RAII return-to-pool
    bk_ctx* ctx;
    std::vector<double*> ptrs;
    explicit WsGuard(bk_ctx* c) : ctx(c) {}
    ~WsGuard() { for (double* p : ptrs) ws_put(ctx, p); }
    int get(size_t n, double** out) {
        int s = ws_get(ctx, n, out);
        if (s == 0) ptrs.push_back(*out);
        return s;
    }
};

// ---- profiling --------------------------------------------------------------------------
struct ProfScope {  // records a start/stop event pair on the ctx stream around a launch sequence
    bk_ctx* ctx;
    ProfEntry* e = nullptr;
    hipEvent_t start = nullptr, stop = nullptr;
    ProfScope(bk_ctx* c, const char* name, double alg_bytes);
    ~ProfScope();
};

// ---- reductions (deterministic two-stage; global over the communicator) -----------------
// After a stage-1 kernel wrote d_partials[nblocks][nvals], reduce -> all-reduce -> host.
// op: 0 = sum, 1 = max.  On return ctx->h_red[0..nvals) holds the results (stream synchronised).
int reduce_finish(bk_ctx* ctx, int nblocks, int nvals, int op);
int comm_allreduce_host(bk_ctx* ctx, double* buf, int n, int op);   // small host-side all-reduce
// all-reduce of a device buffer ENQUEUED in `stream` (ncclAllReduce on RCCL ranks, a stream-ordered hand-over to the proxy thread
// on host-staged ranks): no host synchronisation
int comm_allreduce_dev(bk_ctx* ctx, hipStream_t stream, double* dbuf, int n, int op);
int ctx_sync(bk_ctx* ctx);           // hipStreamSynchronize(ctx->stream) + the error state of proxied collectives
void proxy_destroy(bk_ctx* ctx);

// ---- BLAS-1 launchers (vecops.hip) ---------------------------------------------------------
void blas_release(bk_ctx* ctx);    // dct.hip
int v_copy(bk_ctx* ctx, size_t n, const double* x, double* y);
int v_zero(bk_ctx* ctx, size_t n, double* x);
int v_scale(bk_ctx* ctx, size_t n, double a, double* x);
int v_axpby(bk_ctx* ctx, size_t n, double a, const double* x, double b, double* y);
// z = a x + b y (z may alias x or y)
int v_axpbyz(bk_ctx* ctx, size_t n, double a, const double* x, double b, const double* y, double* z);
// z = x .* (A + u (B + C u))   (z may alias x)
int v_pw_scale(bk_ctx* ctx, size_t n, const double* x, const double* u, double A, double B, double C, double* z);
int v_dot(bk_ctx* ctx, size_t n, const double* x, const double* y, double* out);
int v_dot2(bk_ctx* ctx, size_t n, const double* x, const double* y1, const double* y2, double* out2);
int v_nrm2(bk_ctx* ctx, size_t n, const double* x, double* out);
// *out = |x - y|_2 (global), nothing written
int v_diff_nrm2(bk_ctx* ctx, size_t n, const double* x, const double* y, double* out);
int v_axpy_dot(bk_ctx* ctx, size_t n, double c, const double* r, double* y, const double* z, double* out);
int v_minres_update(bk_ctx* ctx, size_t n, double cz, const double* z, double c1, const double* w1, double c2, const double* w2,
                    double* w, double phi, double* x);
int v_minres_update2(bk_ctx* ctx, size_t n, double cza, const double* za, double c1a, double c2a, double czb, const double* zb, double c1b,
                     double c2b, const double* m2, const double* m1, double* wa, double* wb, double phia, double phib, double* x);
int v_nrminf(bk_ctx* ctx, size_t n, const double* x, double* out);
// out[i] = <V_i, w> for i < k, out[k] = <w, w>;  V_i = V + i*ldv
int v_multidot(bk_ctx* ctx, size_t n, const double* V, size_t ldv, int k, const double* w, double* out);
// the same plus gram[0..k) = <V_i, V_{k-1}>: the Gram column of the newest basis vector, from the same pass
bool v_multidot_gram_ok(bk_ctx* ctx, size_t n, const double* V, size_t ldv, int k, const double* w);
int v_multidot_gram(bk_ctx* ctx, size_t n, const double* V, size_t ldv, int k, const double* w, double* out, double* gram);
// dst = scale * (src + sum_i c[i] V_i); src may be NULL (treated as 0); if nrm2sq != NULL the
// squared 2-norm of dst is returned (global).  dst may alias src.
int v_multiaxpy(bk_ctx* ctx, size_t n, const double* V, size_t ldv, int k, const double* c,
                const double* src, double scale, double* dst, double* nrm2sq);
// block Arnoldi step (sstep.h): the two streaming passes of s Arnoldi steps at once
bool v_block_ok(bk_ctx* ctx, size_t n, const double* V, size_t ldv);
int v_block_dots(bk_ctx* ctx, size_t n, const double* V, size_t ldv, int kold, int r0, int nr, double* D, double* T);
int v_block_axpy(bk_ctx* ctx, size_t n, double* V, size_t ldv, int k, int s, const double* Cm, const double* Tm);
// device-resident orthogonalisation step (vecops.hip): rec / coef are device buffers of kMaxBasis + 2 doubles
// gram != NULL: the Gram-corrected single-pass step (device Gram matrix, (kMaxBasis + 1)^2 doubles, owned by the caller)
int v_arnoldi_step_dev(bk_ctx* ctx, size_t n, double* V, size_t ldv, int k, const double* w, double eta, double orth_tol,
                       double* rec, double* coef, double* gram = nullptr);
// dst_j = sum_{i<m} Q(i,j) V_i for j < kout (Q host, column-major m x kout); dst may alias V
int v_basis_combine(bk_ctx* ctx, size_t n, const double* V, size_t ldv, int m, const double* Qhost, int kout,
                    double* dst, size_t lddst);
// uniform [0,1) pseudo-random fill, deterministic in (seed, global index)
int v_fill_random(bk_ctx* ctx, size_t n, size_t global_offset, unsigned long long seed, double* x);

}  // namespace bk
