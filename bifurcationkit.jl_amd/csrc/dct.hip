// Exact inverse of L1 + shift = (I + Lap)^2 + shift*I through the DCT-II diagonalisation of the
// Neumann-ghost Laplacian.
//
// Reference role: the left preconditioner `Pl = cholesky(Symmetric(L1))` of examples/SH3d.jl:88-93
// (shift = 0) and `lu(L1 + I)` of examples/SH2d-fronts.jl:121 (shift = 1): sparse direct factors of the
// assembled matrix, applied once per GMRES iteration.  The 1-D operator D (tridiag(1,-2,1)/h^2 with
// corner entries -1/h^2, examples/SH3d.jl:21-32) has eigenpairs
//     lambda_k = -(4/h^2) sin^2(pi k / 2N),   phi_k[n] = s_k cos(pi (2n+1) k / 2N),  k = 0..N-1
// (s_0 = sqrt(1/N), s_k = sqrt(2/N)), so with Phi = Phi_z (x) Phi_y (x) Phi_x
//     (L1 + shift)^-1 v = Phi diag( 1 / ((1 + lam_x + lam_y + lam_z)^2 + shift) ) Phi' v .
//
// Kernels:
//   dct_axis_direct   any N: one thread per output element, O(N) dot with a row of the (N x N)
//                     orthonormal DCT matrix.  Correctness baseline and the path for N that are not
//                     powers of two (the reference example runs 22^3).
//   dct_axis_fft      N = 2^q, 8 <= N <= 1024: one workgroup owns a tile of lines staged in LDS and runs a
//                     radix-2 Stockham FFT of the Makhoul-permuted line there; HBM traffic is one read and
//                     one write of the array per axis.  (dct_fast.hip)
//   spectral_scale    multiply by the inverse symbol.
#include <dlfcn.h>

#include <cmath>
#include <mutex>
#include <vector>

#include <rocblas/rocblas.h>   // types only: the library itself is dlopen()ed by the cross-check option (rocblas_api below)

#include "dct_core.h"
#include "ops.h"

namespace bk {

struct DctPlan {
    int ndim = 0;
    int n[3] = {1, 1, 1};
    double shift = 0.0;
    double* T[3] = {nullptr, nullptr, nullptr};    // T[k*N + n]  = s_k cos(pi (2n+1) k / 2N)   (forward rows)
    double* TT[3] = {nullptr, nullptr, nullptr};   // TT[n*N + k] = same, transposed
    double* lam[3] = {nullptr, nullptr, nullptr};  // eigenvalues lambda_k per axis
    double* twid[3] = {nullptr, nullptr, nullptr}; // fast path: twiddle tables
    unsigned* kmap = nullptr;                 // distributed, uniform y split: y index -> element offset in the block layout
    double* t1 = nullptr;
    double* t2 = nullptr;
    size_t total = 0;
    int kind = 0;                             // 0: DCT-II / SH symbol 1/((1+sum lam)^2 + shift); 1: DST-I / 1/(sum lam - shift);
                                              // 2: DST-I, 2x2 block symbol of the two cGL fields (blk_a, blk_b)
    double blk_a = 0.0, blk_b = 0.0;
    int batch = 1;                            // stacked fields sharing the transform (cGL: 2)
    // hand-written fp64-MFMA path of the sine transforms (dense_mfma.hip): per axis the half-size blocks Te, To, their
    // transposes, and the eigenvalues in the permuted spectral order [even k | odd k]
    double* mf[2][4] = {{nullptr, nullptr, nullptr, nullptr}, {nullptr, nullptr, nullptr, nullptr}};
    double* lamp[2] = {nullptr, nullptr};
    // distributed (z-slab) variant: transposes to y-slabs for the z pass
    bool dist = false;
    int R = 1, rank = 0;
    int zlo = 0, zhi = 0, ylo = 0, yhi = 0;
    std::vector<int> zcut, ycut;              // slab boundaries per rank (R+1 entries)
    std::vector<size_t> cnt_f, dsp_f, cnt_b, dsp_b;   // forward: send counts (to y-owners) / recv counts (from z-owners)
    double* lam_yloc = nullptr;               // lam[1] + ylo (view)
    // slab z-solve (dct_slab.hip): local DCT of length nl = nz / R + Woodbury correction over the slab faces
    bool slab_ok = false;
    int nl = 0;
    double az = 0.0;
    double* twid_loc = nullptr;               // twiddles of the length-nl transform
    double* lam_loc = nullptr;                // eigenvalues of the slab-local Neumann second difference
    double* phi_loc = nullptr;                // [2][nl] local DCT-II basis at planes 0, 1
    double* fsend = nullptr;                  // face data: [R][4][Lr] each
    double* frecv = nullptr;
    std::vector<size_t> cnt_s, dsp_s;
};

bool dense_mfma_supported(int n0, int n1, int nb);
void dense_mfma_tables(int N, const std::vector<double>& T, const std::vector<double>& lam, std::vector<double>& Te,
                       std::vector<double>& To, std::vector<double>& TeT, std::vector<double>& ToT, std::vector<double>& lamp);
int dense_mfma_pass(bk_ctx* ctx, int n0, int n1, int nb, int axis, int inverse, const double* const tab[4], const double* in,
                    double* out, double* work);

struct SlabK;
int slab_faces_gather(bk_ctx* ctx, const SlabK& P, const double* y, double* sbuf);
int slab_faces_solve(bk_ctx* ctx, const SlabK& P, const double* rbuf, double* out);
int slab_faces_correct(bk_ctx* ctx, const SlabK& P, const double* rbuf, double* f);
struct SlabK {                                // kernel argument of dct_slab.hip (same definition there)
    int nx, ny, nl, R, rank;
    size_t L, Lr;
    double a, shift;
    const double* lam0;
    const double* lam1;
    const double* lam_loc;
    const double* phi;
};

struct Cuts {                                 // slab boundaries by value (kernel argument), up to 64 ranks
    int c[65];
};

namespace {

// forward: out[k, r] = sum_n T[k][n] in[n, r];  inverse: out[n, r] = sum_k T[k][n] in[k, r]
// Array viewed as [n2][n1][n0] (n0 fastest); `axis` selects the transformed index.
__global__ void __launch_bounds__(256) dct_axis_direct(int n0, int n1, int n2, int axis, const double* __restrict__ M,
                                                       const double* __restrict__ in, double* __restrict__ out) {
    // M is laid out so that M[q*N + o] multiplies input index q for output index o
    const size_t total = (size_t)n0 * n1 * n2;
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int i0 = (int)(idx % n0);
    const int i1 = (int)((idx / n0) % n1);
    const int i2 = (int)(idx / ((size_t)n0 * n1));
    int N, o;
    size_t stride, base;
    if (axis == 0) { N = n0; o = i0; stride = 1; base = (size_t)n0 * (i1 + (size_t)n1 * i2); }
    else if (axis == 1) { N = n1; o = i1; stride = n0; base = i0 + (size_t)n0 * n1 * i2; }
    else { N = n2; o = i2; stride = (size_t)n0 * n1; base = i0 + (size_t)n0 * i1; }
    double acc = 0.0;
    for (int q = 0; q < N; ++q) acc = fma(M[(size_t)q * N + o], in[base + (size_t)q * stride], acc);
    out[idx] = acc;
}

__global__ void __launch_bounds__(256) spectral_scale_kernel(int n0, int n1, int n2, const double* __restrict__ lx,
                                                             const double* __restrict__ ly,
                                                             const double* __restrict__ lz, double shift,
                                                             double* __restrict__ a) {
    const size_t total = (size_t)n0 * n1 * n2;
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int i0 = (int)(idx % n0);
    const int i1 = (int)((idx / n0) % n1);
    const int i2 = (int)(idx / ((size_t)n0 * n1));
    const double s = 1.0 + lx[i0] + ly[i1] + (lz ? lz[i2] : 0.0);
    a[idx] = a[idx] / (s * s + shift);
}

// Dirichlet Laplacian symbol: a[idx] /= (lam_x + lam_y - c), the same for every stacked field
__global__ void __launch_bounds__(256) spectral_scale_lap_kernel(int n0, int n1, int nb, const double* __restrict__ lx,
                                                                 const double* __restrict__ ly, double c,
                                                                 double* __restrict__ a) {
    const size_t total = (size_t)n0 * n1 * nb;
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int i0 = (int)(idx % n0);
    const int i1 = (int)((idx / n0) % n1);
    a[idx] = a[idx] / (lx[i0] + ly[i1] - c);
}

// cGL block symbol: per sine mode the two stacked fields are coupled by [[m, -b], [b, m]], m = lam_x + lam_y + a;
// (x1, x2) <- [[m, b], [-b, m]] (x1, x2) / (m^2 + b^2).  b != 0 keeps the block invertible also where m = 0.
__global__ void __launch_bounds__(256) spectral_block_cgl_kernel(int n0, int n1, const double* __restrict__ lx,
                                                                 const double* __restrict__ ly, double a, double b,
                                                                 double* __restrict__ t) {
    const size_t n = (size_t)n0 * n1;
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= n) return;
    const int i0 = (int)(idx % n0);
    const int i1 = (int)(idx / n0);
    const double m = lx[i0] + ly[i1] + a;
    const double x1 = t[idx], x2 = t[idx + n];
    const double d = 1.0 / (m * m + b * b);
    t[idx] = (m * x1 + b * x2) * d;
    t[idx + n] = (m * x2 - b * x1) * d;
}

}  // namespace

namespace {

__device__ __forceinline__ int owner_of(const Cuts& c, int R, int i) {
    int r = 0;
    while (r + 1 < R && i >= c.c[r + 1]) ++r;
    return r;
}

// z-slab [nzl][ny][nx]  <->  per-destination blocks [d][zl][y - ycut[d]][x]   (dir 0: slab -> blocks)
__global__ void __launch_bounds__(256) slab_blocks_kernel(int nx, int ny, int nzl, int R, Cuts ycut, const double* in,
                                                          double* out, int dir) {
    const size_t total = (size_t)nx * ny * nzl;
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int x = (int)(idx % nx), y = (int)((idx / nx) % ny), zl = (int)(idx / ((size_t)nx * ny));
    const int d = owner_of(ycut, R, y);
    const int y0 = ycut.c[d], nyd = ycut.c[d + 1] - y0;
    const size_t off = (size_t)nx * nzl * y0 + ((size_t)zl * nyd + (y - y0)) * nx + x;   // blocks are laid out in rank order
    if (dir == 0) out[off] = in[idx];
    else out[idx] = in[off];
}

}  // namespace

// rocBLAS is NOT a link-time dependency of the library: its dgemm only serves the cross-check option dct_gemm = 2 (the dense
// transform passes ran on it in rounds 1-2; the product path is dense_mfma.hip).  The five entry points are resolved with
// dlopen("librocblas.so") the first time that option is taken; a missing library is an error of that option only.
namespace {
struct RocblasApi {
    void* lib = nullptr;
    rocblas_status (*create_handle)(rocblas_handle*) = nullptr;
    rocblas_status (*destroy_handle)(rocblas_handle) = nullptr;
    rocblas_status (*set_stream)(rocblas_handle, hipStream_t) = nullptr;
    rocblas_status (*dgemm)(rocblas_handle, rocblas_operation, rocblas_operation, rocblas_int, rocblas_int, rocblas_int, const double*,
                            const double*, rocblas_int, const double*, rocblas_int, const double*, double*, rocblas_int) = nullptr;
    rocblas_status (*dgemm_strided_batched)(rocblas_handle, rocblas_operation, rocblas_operation, rocblas_int, rocblas_int, rocblas_int,
                                            const double*, const double*, rocblas_int, rocblas_stride, const double*, rocblas_int,
                                            rocblas_stride, const double*, double*, rocblas_int, rocblas_stride, rocblas_int) = nullptr;
    bool ok = false;
};
RocblasApi& rocblas_api() {
    static RocblasApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        for (const char* name : {"librocblas.so", "librocblas.so.5", "librocblas.so.4", "/opt/rocm/lib/librocblas.so"}) {
            api.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (api.lib) break;
        }
        if (!api.lib) return;
        api.create_handle = reinterpret_cast<decltype(api.create_handle)>(dlsym(api.lib, "rocblas_create_handle"));
        api.destroy_handle = reinterpret_cast<decltype(api.destroy_handle)>(dlsym(api.lib, "rocblas_destroy_handle"));
        api.set_stream = reinterpret_cast<decltype(api.set_stream)>(dlsym(api.lib, "rocblas_set_stream"));
        api.dgemm = reinterpret_cast<decltype(api.dgemm)>(dlsym(api.lib, "rocblas_dgemm"));
        api.dgemm_strided_batched = reinterpret_cast<decltype(api.dgemm_strided_batched)>(dlsym(api.lib, "rocblas_dgemm_strided_batched"));
        api.ok = api.create_handle && api.destroy_handle && api.set_stream && api.dgemm && api.dgemm_strided_batched;
    });
    return api;
}
}  // namespace

void blas_release(bk_ctx* ctx) {
    if (ctx->blas) (void)rocblas_api().destroy_handle(static_cast<rocblas_handle>(ctx->blas));
    ctx->blas = nullptr;
}

// Dense transform along one axis, out[.., o, ..] = sum_q M[q*N + o] in[.., q, ..]: sizes without a fast transform
// (non-powers of two such as the reference's 22^3, the DST-I of the cGL Laplacian, N = 1024).  This IS a plain GEMM
// -- along x: Out(rows x N) = X(rows x N) M; along y / z: Out_plane = M' X_plane, batched over the planes -- so from
// N = 32 on it goes to rocBLAS' fp64 MFMA dgemm (the library call the design rules reserve for plain GEMMs); the
// one-thread-per-output kernel stays for tiny extents and as a cross-check (option dct_gemm = 0).
int dense_gemm_axis_pass(bk_ctx* ctx, int n0, int n1, int n2, int axis, const double* M, const double* MT, const double* in,
                         double* out);
// MT = M' (row-major).  Option dct_gemm: 1 (default) the hand-written fp64-MFMA product (dense_mfma.hip), 2 rocBLAS dgemm
// (the cross-check; rounds 1-2 ran on it), 0 the one-thread-per-output kernel.
static int dense_axis_pass(bk_ctx* ctx, int n0, int n1, int n2, int axis, const double* M, const double* MT, const double* in,
                           double* out) {
    const int N = axis == 0 ? n0 : (axis == 1 ? n1 : n2);
    const size_t total = (size_t)n0 * n1 * n2;
    const int gemm = (int)ctx->opt("dct_gemm", 1.0);
    if (N >= 32 && gemm == 1) return dense_gemm_axis_pass(ctx, n0, n1, n2, axis, M, MT, in, out);
    if (N < 32 || gemm == 0) {
        hipLaunchKernelGGL(dct_axis_direct, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream, n0, n1, n2, axis,
                           M, in, out);
        BK_HIP(ctx, hipGetLastError());
        return 0;
    }
    RocblasApi& rb = rocblas_api();
    if (!rb.ok) return set_error(ctx, "dct_gemm = 2: librocblas.so could not be loaded (the cross-check needs it; the product path does not)");
    if (!ctx->blas) {
        rocblas_handle h = nullptr;
        if (rb.create_handle(&h) != rocblas_status_success) return set_error(ctx, "rocblas_create_handle failed");
        ctx->blas = h;
    }
    rocblas_handle h = static_cast<rocblas_handle>(ctx->blas);
    if (rb.set_stream(h, ctx->stream) != rocblas_status_success) return set_error(ctx, "rocblas_set_stream failed");
    const double one = 1.0, zero = 0.0;
    rocblas_status st;
    if (axis == 0) {
        // row-major Out(rows x N) = X M  <=>  column-major Out'(N x rows) = M' X', and the row-major M pointer read
        // column-major IS M'
        const size_t rows = total / N;
        st = rb.dgemm(h, rocblas_operation_none, rocblas_operation_none, N, (rocblas_int)rows, N, &one, M, N, in, N, &zero,
                           out, N);
    } else {
        // per plane (row-major [N][inner]): Out = M' X  <=>  column-major Out'(inner x N) = X'(inner x N) M
        const size_t inner = axis == 1 ? (size_t)n0 : (size_t)n0 * n1;
        const size_t batch = axis == 1 ? (size_t)n2 : 1;
        st = rb.dgemm_strided_batched(h, rocblas_operation_none, rocblas_operation_transpose, (rocblas_int)inner, N, N,
                                           &one, in, (rocblas_int)inner, (rocblas_stride)(inner * N), M, N, 0, &zero, out,
                                           (rocblas_int)inner, (rocblas_stride)(inner * N), (rocblas_int)batch);
    }
    if (st != rocblas_status_success) return set_error(ctx, "rocblas dgemm failed (%d)", (int)st);
    return 0;
}

// Twiddle tables of the LDS FFT kernels (dct_fast.hip), in the slot layout of dct_core.h (twi: one padding slot per 16
// entries): w[j] = exp(-2 pi i j / N), j < N/2 (FFT), followed by the Makhoul post-twiddles e[k] = exp(-i pi k / 2N), k <= N/2.
std::vector<double> dct_twiddle_table(int N) {
    const int twl = dctc::tw_len(N), ewl = dctc::ew_len(N);
    std::vector<double> tw(2 * (size_t)(twl + ewl), 0.0);
    for (int j = 0; j < N / 2; ++j) {
        tw[2 * (size_t)dctc::twi(j)] = std::cos(2.0 * M_PI * j / N);
        tw[2 * (size_t)dctc::twi(j) + 1] = -std::sin(2.0 * M_PI * j / N);
    }
    for (int k = 0; k <= N / 2; ++k) {
        tw[2 * (size_t)(twl + dctc::twi(k))] = std::cos(M_PI * k / (2.0 * N));
        tw[2 * (size_t)(twl + dctc::twi(k)) + 1] = -std::sin(M_PI * k / (2.0 * N));
    }
    return tw;
}

int dct_plan_create(bk_ctx* ctx, int ndim, const int n[3], const double ainv[3], double shift, DctPlan** out) {
    DctPlan* p = new DctPlan();
    p->ndim = ndim;
    p->shift = shift;
    p->total = 1;
    for (int a = 0; a < 3; ++a) { p->n[a] = a < ndim ? n[a] : 1; p->total *= (size_t)p->n[a]; }
    for (int a = 0; a < ndim; ++a) {
        const int N = p->n[a];
        std::vector<double> T((size_t)N * N), TT((size_t)N * N), lam(N);
        for (int k = 0; k < N; ++k) {
            const double sk = k == 0 ? std::sqrt(1.0 / N) : std::sqrt(2.0 / N);
            for (int q = 0; q < N; ++q) {
                // reduce the argument mod 4N exactly before calling cos: (2q+1)k can be large
                const long long arg = ((long long)(2 * q + 1) * k) % (4LL * N);
                const double c = sk * std::cos(M_PI * (double)arg / (2.0 * N));
                T[(size_t)k * N + q] = c;
                TT[(size_t)q * N + k] = c;
            }
            const double sn = std::sin(M_PI * k / (2.0 * N));
            lam[k] = -4.0 * ainv[a] * sn * sn;
        }
        if (hipMalloc(&p->T[a], sizeof(double) * N * N) != hipSuccess ||
            hipMalloc(&p->TT[a], sizeof(double) * N * N) != hipSuccess ||
            hipMalloc(&p->lam[a], sizeof(double) * N) != hipSuccess) {
            dct_plan_destroy(p);
            return set_error(ctx, "dct plan: allocation failed");
        }
        (void)hipMemcpy(p->T[a], T.data(), sizeof(double) * N * N, hipMemcpyHostToDevice);
        (void)hipMemcpy(p->TT[a], TT.data(), sizeof(double) * N * N, hipMemcpyHostToDevice);
        (void)hipMemcpy(p->lam[a], lam.data(), sizeof(double) * N, hipMemcpyHostToDevice);
        if (dct_axis_fft_supported(N)) {
            const std::vector<double> tw = dct_twiddle_table(N);
            if (hipMalloc(&p->twid[a], sizeof(double) * tw.size()) != hipSuccess) {
                dct_plan_destroy(p);
                return set_error(ctx, "dct plan: allocation failed");
            }
            (void)hipMemcpy(p->twid[a], tw.data(), sizeof(double) * tw.size(), hipMemcpyHostToDevice);
        }
    }
    if (hipMalloc(&p->t1, sizeof(double) * p->total) != hipSuccess ||
        hipMalloc(&p->t2, sizeof(double) * p->total) != hipSuccess) {
        dct_plan_destroy(p);
        return set_error(ctx, "dct plan: scratch allocation failed");
    }
    *out = p;
    return 0;
}

// DST-I plan for the Dirichlet 5-point Laplacian of examples/cGL2d.jl:6-22: the 1-D operator tridiag(1,-2,1)/h^2 has
// eigenvectors sqrt(2/(N+1)) sin(pi (k+1)(n+1)/(N+1)) and eigenvalues -(4/h^2) sin^2(pi (k+1) / 2(N+1)).
int dst_plan_create(bk_ctx* ctx, const int n[2], const double ainv[2], double c, int batch, DctPlan** out) {
    DctPlan* p = new DctPlan();
    p->ndim = 2;
    p->kind = 1;
    p->batch = batch;
    p->shift = c;
    p->n[0] = n[0]; p->n[1] = n[1]; p->n[2] = batch;
    p->total = (size_t)n[0] * n[1] * batch;
    for (int a = 0; a < 2; ++a) {
        const int N = n[a];
        std::vector<double> T((size_t)N * N), lam(N);
        const double s = std::sqrt(2.0 / (N + 1));
        for (int k = 0; k < N; ++k) {
            for (int q = 0; q < N; ++q) {
                const long long arg = ((long long)(k + 1) * (q + 1)) % (2LL * (N + 1));
                T[(size_t)k * N + q] = s * std::sin(M_PI * (double)arg / (N + 1));
            }
            const double sn = std::sin(M_PI * (k + 1) / (2.0 * (N + 1)));
            lam[k] = -4.0 * ainv[a] * sn * sn;
        }
        if (hipMalloc(&p->T[a], sizeof(double) * N * N) != hipSuccess || hipMalloc(&p->lam[a], sizeof(double) * N) != hipSuccess) {
            dct_plan_destroy(p);
            return set_error(ctx, "dst plan: allocation failed");
        }
        (void)hipMemcpy(p->T[a], T.data(), sizeof(double) * N * N, hipMemcpyHostToDevice);
        (void)hipMemcpy(p->lam[a], lam.data(), sizeof(double) * N, hipMemcpyHostToDevice);
        if (dense_mfma_supported(n[0], n[1], batch)) {
            std::vector<double> tb[4], lp;
            dense_mfma_tables(N, T, lam, tb[0], tb[1], tb[2], tb[3], lp);
            bool ok = hipMalloc(&p->lamp[a], sizeof(double) * N) == hipSuccess;
            for (int t = 0; t < 4 && ok; ++t) ok = hipMalloc(&p->mf[a][t], sizeof(double) * tb[t].size()) == hipSuccess;
            if (!ok) {
                dct_plan_destroy(p);
                return set_error(ctx, "dst plan: allocation failed");
            }
            (void)hipMemcpy(p->lamp[a], lp.data(), sizeof(double) * N, hipMemcpyHostToDevice);
            for (int t = 0; t < 4; ++t) (void)hipMemcpy(p->mf[a][t], tb[t].data(), sizeof(double) * tb[t].size(), hipMemcpyHostToDevice);
        }
    }
    if (hipMalloc(&p->t1, sizeof(double) * p->total) != hipSuccess || hipMalloc(&p->t2, sizeof(double) * p->total) != hipSuccess) {
        dct_plan_destroy(p);
        return set_error(ctx, "dst plan: scratch allocation failed");
    }
    *out = p;
    return 0;
}

// (Lap - c)^-1 on every stacked field: the sine matrix is symmetric and its own inverse
static int dst_apply(bk_ctx* ctx, DctPlan* p, const double* v, double* out) {
    const int n0 = p->n[0], n1 = p->n[1], nb = p->batch;
    const unsigned grid = (unsigned)((p->total + 255) / 256);
    if (p->mf[0][0] && p->mf[1][0] && ctx->opt("dst_mfma", 1.0) != 0.0) {
        // the hand-written path (dense_mfma.hip): fold + half-size fp64-MFMA products; the spectrum lives in the permuted
        // order [even k | odd k] along both axes between the forward and the inverse passes, the symbol kernels read the
        // permuted eigenvalue tables.  `out` (which may alias v) is free as soon as the first pass has read v.
        auto mpass = [&](int a, int inverse, const double* in, double* o) -> int {
            ProfScope ps(ctx, "dct_pass", 16.0 * p->total);
            const double* tab[4] = {p->mf[a][0], p->mf[a][1], p->mf[a][2], p->mf[a][3]};
            return dense_mfma_pass(ctx, n0, n1, nb, a, inverse, tab, in, o, p->t1);
        };
        BK_TRY(mpass(0, 0, v, p->t2));
        BK_TRY(mpass(1, 0, p->t2, out));
        if (p->kind == 2) {
            const unsigned g2 = (unsigned)(((size_t)n0 * n1 + 255) / 256);
            hipLaunchKernelGGL(spectral_block_cgl_kernel, dim3(g2), dim3(256), 0, ctx->stream, n0, n1, p->lamp[0], p->lamp[1],
                               p->blk_a, p->blk_b, out);
        } else {
            hipLaunchKernelGGL(spectral_scale_lap_kernel, dim3(grid), dim3(256), 0, ctx->stream, n0, n1, nb, p->lamp[0], p->lamp[1],
                               p->shift, out);
        }
        BK_HIP(ctx, hipGetLastError());
        BK_TRY(mpass(1, 1, out, p->t2));
        return mpass(0, 1, p->t2, out);
    }
    auto pass = [&](int a, const double* in, double* o) -> int {
        ProfScope ps(ctx, "dct_pass", 16.0 * p->total);
        return dense_axis_pass(ctx, n0, n1, nb, a, p->T[a], p->T[a], in, o);       // the sine matrix is symmetric
    };
    BK_TRY(pass(0, v, p->t1));
    BK_TRY(pass(1, p->t1, p->t2));
    if (p->kind == 2) {
        const unsigned g2 = (unsigned)(((size_t)n0 * n1 + 255) / 256);
        hipLaunchKernelGGL(spectral_block_cgl_kernel, dim3(g2), dim3(256), 0, ctx->stream, n0, n1, p->lam[0], p->lam[1],
                           p->blk_a, p->blk_b, p->t2);
    } else {
        hipLaunchKernelGGL(spectral_scale_lap_kernel, dim3(grid), dim3(256), 0, ctx->stream, n0, n1, nb, p->lam[0], p->lam[1],
                           p->shift, p->t2);
    }
    BK_HIP(ctx, hipGetLastError());
    BK_TRY(pass(1, p->t2, p->t1));
    BK_TRY(pass(0, p->t1, out));
    return 0;
}

void dct_plan_destroy(DctPlan* p) {
    if (!p) return;
    for (int a = 0; a < 3; ++a) {
        if (p->T[a]) (void)hipFree(p->T[a]);
        if (p->TT[a]) (void)hipFree(p->TT[a]);
        if (p->lam[a]) (void)hipFree(p->lam[a]);
        if (p->twid[a]) (void)hipFree(p->twid[a]);
        if (a == 0 && p->kmap) (void)hipFree(p->kmap);
    }
    if (p->t1) (void)hipFree(p->t1);
    if (p->t2) (void)hipFree(p->t2);
    for (int a = 0; a < 2; ++a) {
        if (p->lamp[a]) (void)hipFree(p->lamp[a]);
        for (int t = 0; t < 4; ++t)
            if (p->mf[a][t]) (void)hipFree(p->mf[a][t]);
    }
    double* extra[] = {p->twid_loc, p->lam_loc, p->phi_loc, p->fsend, p->frecv};
    for (double* e : extra)
        if (e) (void)hipFree(e);
    delete p;
}

static int dct_apply_slab(bk_ctx* ctx, DctPlan* p, const double* v, double* out, const DctFuse* fz = nullptr);

int dct_apply(bk_ctx* ctx, DctPlan* p, const double* v, double* out, int* dot_blocks, const DctFuse* fz) {
    if (dot_blocks) *dot_blocks = 0;
    if (p->slab_ok) return dct_apply_slab(ctx, p, v, out, fz);      // cost-model emulation (option dct_slab_emulate), timing only
    const int n0 = p->n[0], n1 = p->n[1], n2 = p->n[2];
    const unsigned grid = (unsigned)((p->total + 255) / 256);
    const bool use_fft = ctx->opt("dct_fft", 1.0) != 0.0;
    // forward along each axis: v -> t1 -> t2 -> ... ; then scale; then inverse in reverse order
    const double* src = v;
    double* bufs[2] = {p->t1, p->t2};
    int cur = 0;
    auto axis_pass = [&](int a, int inverse, const double* in, double* o, int fuse) -> int {
        // the fused pointwise work (DctFuse) rides in the first pass (input side: in == v) and the last one (output side: o == out)
        const DctFuse* f = (fz && a == 0 && ((!inverse && in == v && (fz->u || fz->add)) || (inverse && o == out && fz->xadd))) ? fz : nullptr;
        // one read + one write of the array per axis pass (+ the fused stream; + the stored sum of the pre-axpy form)
        ProfScope ps(ctx, "dct_pass", (f ? (!inverse && fz->add ? 32.0 : 24.0) : 16.0) * p->total);
        if (use_fft && p->twid[a]) {
            return dct_axis_fft(ctx, n0, n1, n2, a, inverse, p->twid[a], in, o, p->lam[0], p->lam[1],
                                p->ndim == 3 ? p->lam[2] : nullptr, p->shift, fuse, nullptr, fuse == 2 ? dot_blocks : nullptr, f);
        }
        if (f) return set_error(ctx, "dct_apply: fused pointwise work on a dense transform pass");
        // forward: out[k] = sum_n T[k][n] in[n]  -> M[q=n][o=k] = TT ; inverse: out[n] = sum_k T[k][n] in[k] -> M = T
        return dense_axis_pass(ctx, n0, n1, n2, a, inverse ? p->T[a] : p->TT[a], inverse ? p->TT[a] : p->T[a], in, o);
    };
    // the last forward pass applies the inverse symbol while storing when it runs on the fast path
    const int last = p->ndim - 1;
    const bool fused = use_fft && p->twid[last] != nullptr;
    const bool roundtrip = fused && ctx->opt("dct_roundtrip", 1.0) != 0.0;
    if (roundtrip) {
        // forward on axes 0..last-1, ONE fused pass on the last axis (forward, symbol, inverse in LDS), inverse back
        for (int a = 0; a < last; ++a) {
            BK_TRY(axis_pass(a, 0, src, bufs[cur], 0));
            src = bufs[cur];
            cur ^= 1;
        }
        double* dst = (last == 0) ? out : bufs[cur];
        BK_TRY(axis_pass(last, 0, src, dst, 2));
        src = dst;
        cur ^= 1;
        for (int a = last - 1; a >= 0; --a) {
            double* d2 = (a == 0) ? out : bufs[cur];
            BK_TRY(axis_pass(a, 1, src, d2, 0));
            src = d2;
            cur ^= 1;
        }
        return 0;
    }
    for (int a = 0; a < p->ndim; ++a) {
        BK_TRY(axis_pass(a, 0, src, bufs[cur], (a == last && fused) ? 1 : 0));
        src = bufs[cur];
        cur ^= 1;
    }
    // src now holds the spectrum (in bufs[cur^1])
    double* spec = bufs[cur ^ 1];
    if (!fused) {
        hipLaunchKernelGGL(spectral_scale_kernel, dim3(grid), dim3(256), 0, ctx->stream, n0, n1, n2, p->lam[0], p->lam[1],
                           p->ndim == 3 ? p->lam[2] : nullptr, p->shift, spec);
        BK_HIP(ctx, hipGetLastError());
    }
    src = spec;
    for (int a = p->ndim - 1; a >= 0; --a) {
        double* dst = (a == 0) ? out : bufs[cur];
        BK_TRY(axis_pass(a, 1, src, dst, 0));
        src = dst;
        cur ^= 1;
    }
    return 0;
}

static int slab_tables_create(bk_ctx* ctx, DctPlan* p, int nl, bool even, double az);

int dct_plan_create_dist(bk_ctx* ctx, const int n[3], const double ainv[3], double shift, int zlo, int zhi, DctPlan** out) {
    if (ctx->nranks > 64) return set_error(ctx, "distributed DCT: at most 64 ranks");
    DctPlan* p = nullptr;
    BK_TRY(dct_plan_create(ctx, 3, n, ainv, shift, &p));       // tables for the GLOBAL extents (scratch resized below)
    p->dist = true;
    p->R = ctx->nranks; p->rank = ctx->rank;
    const int R = p->R;
    p->zcut.assign(R + 1, 0);
    p->ycut.assign(R + 1, 0);
    for (int r = 0; r <= R; ++r) {
        auto cut = [&](int N) { const int base = N / R, rem = N % R; return r * base + (r < rem ? r : rem); };
        p->zcut[r] = cut(n[2]);
        p->ycut[r] = cut(n[1]);
    }
    if (p->zcut[p->rank] != zlo || p->zcut[p->rank + 1] != zhi) {
        dct_plan_destroy(p);
        return set_error(ctx, "distributed DCT: slab [%d,%d) does not match the balanced decomposition", zlo, zhi);
    }
    p->zlo = zlo; p->zhi = zhi; p->ylo = p->ycut[p->rank]; p->yhi = p->ycut[p->rank + 1];
    const size_t nx = n[0], nzl = zhi - zlo, nyl = p->yhi - p->ylo;
    const size_t loc_z = nx * n[1] * nzl, loc_y = nx * nyl * n[2];
    p->cnt_f.resize(R); p->dsp_f.resize(R); p->cnt_b.resize(R); p->dsp_b.resize(R);
    for (int r = 0; r < R; ++r) {
        p->cnt_f[r] = nx * nzl * (size_t)(p->ycut[r + 1] - p->ycut[r]);   // my z-planes, rank r's y-rows
        p->dsp_f[r] = nx * nzl * (size_t)p->ycut[r];
        p->cnt_b[r] = nx * nyl * (size_t)(p->zcut[r + 1] - p->zcut[r]);   // rank r's z-planes, my y-rows
        p->dsp_b[r] = nx * nyl * (size_t)p->zcut[r];
    }
    (void)hipFree(p->t1); (void)hipFree(p->t2);
    p->t1 = p->t2 = nullptr;
    const size_t cap = loc_z > loc_y ? loc_z : loc_y;
    p->total = cap;
    if (hipMalloc(&p->t1, sizeof(double) * cap) != hipSuccess || hipMalloc(&p->t2, sizeof(double) * cap) != hipSuccess) {
        dct_plan_destroy(p);
        return set_error(ctx, "distributed DCT: scratch allocation failed");
    }
    p->lam_yloc = p->lam[1] + p->ylo;
    if (n[1] % R == 0 && nx * (size_t)n[1] * nzl < ((size_t)1 << 29)) {
        // every rank owns nyl = ny / R rows: block r = [zl][yl][x] starts at nx * nzl * nyl * r, so the element
        // (x, y, zl) of the z-slab sits at kmap[y] + zl * (nyl * nx) + x with kmap[y] = nx * (nzl * nyl * r + yl)
        std::vector<unsigned> km(n[1]);
        for (int y = 0; y < n[1]; ++y) {
            const size_t r = (size_t)y / nyl, yl = (size_t)y % nyl;
            km[y] = (unsigned)(nx * (nzl * nyl * r + yl));
        }
        if (hipMalloc(&p->kmap, sizeof(unsigned) * km.size()) != hipSuccess) {
            dct_plan_destroy(p);
            return set_error(ctx, "distributed DCT: kmap allocation failed");
        }
        (void)hipMemcpy(p->kmap, km.data(), sizeof(unsigned) * km.size(), hipMemcpyHostToDevice);
    }
    // slab z-solve: equal power-of-two slabs, lines divisible among the ranks, a positive shift (B_r SPD, well conditioned)
    if (slab_tables_create(ctx, p, n[2] / R, n[2] % R == 0, ainv[2]) != 0) { dct_plan_destroy(p); return -1; }
    *out = p;
    return 0;
}

static int slab_tables_create(bk_ctx* ctx, DctPlan* p, int nl, bool even, double az) {
    {
        const int R = p->R;
        const size_t nx = p->n[0];
        const int* n = p->n;
        const double shift = p->shift;
        const double ainv[3] = {0.0, 0.0, az};
        const size_t L = nx * (size_t)n[1];
        if (even && nl >= 8 && dct_axis_fft_supported(nl) && L % R == 0 && shift > 0.0 && R - 1 <= 15) {
            const std::vector<double> tw = dct_twiddle_table(nl);
            std::vector<double> lam(nl), phi(2 * (size_t)nl);
            for (int k = 0; k < nl; ++k) {
                const double sn = std::sin(M_PI * k / (2.0 * nl));
                lam[k] = -4.0 * ainv[2] * sn * sn;
                const double sk = k == 0 ? std::sqrt(1.0 / nl) : std::sqrt(2.0 / nl);
                phi[k] = sk * std::cos(M_PI * (double)k / (2.0 * nl));                 // plane 0
                phi[nl + k] = sk * std::cos(M_PI * 3.0 * (double)k / (2.0 * nl));      // plane 1
            }
            const size_t fb = 4 * L;
            if (hipMalloc(&p->twid_loc, sizeof(double) * tw.size()) != hipSuccess ||
                hipMalloc(&p->lam_loc, sizeof(double) * nl) != hipSuccess ||
                hipMalloc(&p->phi_loc, sizeof(double) * 2 * nl) != hipSuccess ||
                hipMalloc(&p->fsend, sizeof(double) * fb) != hipSuccess ||
                hipMalloc(&p->frecv, sizeof(double) * fb) != hipSuccess) {
                // the plan stays with its owner (dct_plan_create_dist / the preconditioner object), which destroys it once
                return set_error(ctx, "distributed DCT: slab tables allocation failed");
            }
            (void)hipMemcpy(p->twid_loc, tw.data(), sizeof(double) * tw.size(), hipMemcpyHostToDevice);
            (void)hipMemcpy(p->lam_loc, lam.data(), sizeof(double) * nl, hipMemcpyHostToDevice);
            (void)hipMemcpy(p->phi_loc, phi.data(), sizeof(double) * 2 * nl, hipMemcpyHostToDevice);
            p->nl = nl;
            p->az = ainv[2];
            p->cnt_s.assign(R, 4 * (L / R));
            p->dsp_s.resize(R);
            for (int r = 0; r < R; ++r) p->dsp_s[r] = (size_t)r * 4 * (L / R);
            p->slab_ok = true;
        }
    }
    return 0;
}

// Slab path of the distributed apply (dct_slab.hip): x, y passes on the z-slab; B^-1 = the fused z pass of length nl on the
// slab; face data to the line owners, the capacitance solves, the corrections back; B^-1 again on the corrected right-hand
// side; inverse y, x.  Two small all-to-alls (4 doubles per line each way) instead of two transposes of the array.
static int dct_apply_slab(bk_ctx* ctx, DctPlan* p, const double* v, double* out, const DctFuse* fz) {
    const int nx = p->n[0], ny = p->n[1], nl = p->nl;
    const double n3 = (double)nx * ny * nl;
    auto pass = [&](int axis, int inverse, const double* tw, const double* in, double* o, int fuse, const double* l2) -> int {
        const DctFuse* f = (fz && axis == 0 && ((!inverse && in == v && fz->u) || (inverse && o == out && fz->xadd))) ? fz : nullptr;
        ProfScope ps(ctx, "dct_pass", (f ? 24.0 : 16.0) * n3);
        return dct_axis_fft(ctx, nx, ny, nl, axis, inverse, tw, in, o, p->lam[0], p->lam[1], l2, p->shift, fuse, nullptr, nullptr, f);
    };
    SlabK K;
    K.nx = nx; K.ny = ny; K.nl = nl; K.R = p->R; K.rank = p->rank;
    K.L = (size_t)nx * ny; K.Lr = K.L / p->R;
    K.a = p->az; K.shift = p->shift;
    K.lam0 = p->lam[0]; K.lam1 = p->lam[1]; K.lam_loc = p->lam_loc; K.phi = p->phi_loc;
    double *a = p->t1, *b = p->t2;
    BK_TRY(pass(0, 0, p->twid[0], v, a, 0, nullptr));
    BK_TRY(pass(1, 0, p->twid[1], a, b, 0, nullptr));               // b = f (z-solve right-hand side, spectral in x, y)
    if (ctx->opt("dct_slab_split", 1.0) != 0.0 && dct_slab_half_ok(ctx, nx, ny, nl, b, a) && K.Lr % 2 == 0) {
        // Round 5: the z solve as forward / inverse HALVES.  M^-1 f = B^-1 (f + delta) with delta supported on the four planes next to
        // the faces, and B^-1 = Phi diag(sym) Phi': the forward half leaves y^ = sym .* Phi' f and, from sums over the spectrum, the
        // values of y = B^-1 f at those planes (all the capacitance system needs); the correction sym_k sum_p phi_k(p) delta_p is a
        // pointwise update of y^ inside the inverse half.  Two single transforms instead of two round trips (the same 32 B/point of
        // traffic), and the face gather / correction kernels are gone: the halves write and read the all-to-all buffers themselves.
        DctSlabHalf hf;
        hf.phi = p->phi_loc; hf.Lr = K.Lr; hf.a = K.a; hf.has_bottom = K.rank > 0; hf.has_top = K.rank < K.R - 1;
        hf.face_y = p->fsend; hf.face_d = p->frecv;
        {
            ProfScope ps(ctx, "dct_pass", 16.0 * n3);
            BK_TRY(dct_axis_fft(ctx, nx, ny, nl, 2, 0, p->twid_loc, b, a, p->lam[0], p->lam[1], p->lam_loc, p->shift, 0, nullptr, nullptr,
                                nullptr, &hf));                     // a = y^; fsend = this rank's face data, per line owner
        }
        { ProfScope ps(ctx, "alltoall", 8.0 * 4.0 * K.L);
        BK_TRY(comm_alltoallv(ctx, p->fsend, p->cnt_s.data(), p->dsp_s.data(), p->frecv, p->cnt_s.data(), p->dsp_s.data())); }
        {
            ProfScope ps(ctx, "blas1", 8.0 * 8.0 * K.L);
            BK_TRY(slab_faces_solve(ctx, K, p->frecv, p->fsend));
        }
        { ProfScope ps(ctx, "alltoall", 8.0 * 4.0 * K.L);
        BK_TRY(comm_alltoallv(ctx, p->fsend, p->cnt_s.data(), p->dsp_s.data(), p->frecv, p->cnt_s.data(), p->dsp_s.data())); }
        {
            ProfScope ps(ctx, "dct_pass", 16.0 * n3);
            BK_TRY(dct_axis_fft(ctx, nx, ny, nl, 2, 1, p->twid_loc, a, b, p->lam[0], p->lam[1], p->lam_loc, p->shift, 0, nullptr, nullptr,
                                nullptr, &hf));                     // b = Phi (y^ + sym .* Phi' delta) = M^-1 f, delta = -U nu from frecv
        }
        BK_TRY(pass(1, 1, p->twid[1], b, a, 0, nullptr));
        BK_TRY(pass(0, 1, p->twid[0], a, out, 0, nullptr));
        return 0;
    }
    BK_TRY(pass(2, 0, p->twid_loc, b, a, 2, p->lam_loc));          // a = B^-1 f
    {
        ProfScope ps(ctx, "blas1", 8.0 * 8.0 * K.L);
        BK_TRY(slab_faces_gather(ctx, K, a, p->fsend));
    }
    { ProfScope ps(ctx, "alltoall", 8.0 * 4.0 * K.L);
    BK_TRY(comm_alltoallv(ctx, p->fsend, p->cnt_s.data(), p->dsp_s.data(), p->frecv, p->cnt_s.data(), p->dsp_s.data())); }
    {
        ProfScope ps(ctx, "blas1", 8.0 * 8.0 * K.L);
        BK_TRY(slab_faces_solve(ctx, K, p->frecv, p->fsend));
    }
    { ProfScope ps(ctx, "alltoall", 8.0 * 4.0 * K.L);
    BK_TRY(comm_alltoallv(ctx, p->fsend, p->cnt_s.data(), p->dsp_s.data(), p->frecv, p->cnt_s.data(), p->dsp_s.data())); }
    {
        ProfScope ps(ctx, "blas1", 8.0 * 12.0 * K.L);
        BK_TRY(slab_faces_correct(ctx, K, p->frecv, b));
    }
    BK_TRY(pass(2, 0, p->twid_loc, b, a, 2, p->lam_loc));          // a = B^-1 (f - U nu) = M^-1 f
    BK_TRY(pass(1, 1, p->twid[1], a, b, 0, nullptr));
    BK_TRY(pass(0, 1, p->twid[0], b, out, 0, nullptr));
    return 0;
}

// Distributed apply: x and y passes on the z-slab [nzl][ny][nx]; all-to-all of per-rank blocks [zl][yl_r][x]; the
// received buffer IS the y-slab array [nz][nyl][nx] (block s holds the planes zcut[s]..zcut[s+1] in order), so the z pass
// (forward, symbol, inverse) runs on it directly as an axis-2 pass; all-to-all back; inverse y and x passes.  With a
// uniform y split and the fused y kernel the forward y pass writes the block layout itself and the inverse y pass reads
// it (DctSplit), so no pack / unpack kernel runs at all; otherwise slab_blocks_kernel packs / unpacks.
static int dct_apply_dist(bk_ctx* ctx, DctPlan* p, const double* v, double* out, const DctFuse* fz = nullptr) {
    if (p->slab_ok && ctx->opt("dct_fft", 1.0) != 0.0 && p->twid[0] && p->twid[1] && ctx->opt("dct_dist_slab", 1.0) != 0.0)
        return dct_apply_slab(ctx, p, v, out, fz);
    const int nx = p->n[0], ny = p->n[1], nz = p->n[2];
    const int nzl = p->zhi - p->zlo, nyl = p->yhi - p->ylo;
    const size_t loc_z = (size_t)nx * ny * nzl, loc_y = (size_t)nx * nyl * nz;
    const bool use_fft = ctx->opt("dct_fft", 1.0) != 0.0;
    Cuts yc;
    for (int r = 0; r <= p->R; ++r) yc.c[r] = p->ycut[r];
    auto pass = [&](int n0, int n1, int n2, int axis, int which, int inverse, const double* in, double* o, int fuse,
                    const double* l0, const double* l1, const double* l2, const DctSplit* split) -> int {
        // `which` = index of the global axis being transformed (selects tables)
        const DctFuse* f = (fz && which == 0 && ((!inverse && in == v && fz->u) || (inverse && o == out && fz->xadd))) ? fz : nullptr;
        ProfScope ps(ctx, "dct_pass", (f ? 24.0 : 16.0) * (double)n0 * n1 * n2);
        if (use_fft && p->twid[which])
            return dct_axis_fft(ctx, n0, n1, n2, axis, inverse, p->twid[which], in, o, l0, l1, l2, p->shift, fuse, split, nullptr, f);
        if (f) return set_error(ctx, "dct_apply_dist: fused pointwise work on a dense transform pass");
        return dense_axis_pass(ctx, n0, n1, n2, axis, inverse ? p->T[which] : p->TT[which], inverse ? p->TT[which] : p->T[which], in, o);
    };
    double *a = p->t1, *b = p->t2;
    const bool direct = p->kmap && use_fft && p->twid[1] && ctx->opt("dct_dist_direct", 1.0) != 0.0 &&
                        dct_axis_fused_ok(ctx, nx, ny, nzl, 1, a, b, 0);
    const DctSplit sp = {p->kmap, (unsigned)((size_t)nyl * nx)};
    // forward x, y on the z-slab; the send buffer (blocks in rank order) ends up in `b`
    BK_TRY(pass(nx, ny, nzl, 0, 0, 0, v, a, 0, nullptr, nullptr, nullptr, nullptr));
    if (direct) {
        BK_TRY(pass(nx, ny, nzl, 1, 1, 0, a, b, 0, nullptr, nullptr, nullptr, &sp));
    } else {
        BK_TRY(pass(nx, ny, nzl, 1, 1, 0, a, b, 0, nullptr, nullptr, nullptr, nullptr));
        ProfScope ps(ctx, "transpose", 16.0 * loc_z);
        hipLaunchKernelGGL(slab_blocks_kernel, dim3((unsigned)((loc_z + 255) / 256)), dim3(256), 0, ctx->stream, nx, ny, nzl, p->R, yc, b, a, 0);
        BK_HIP(ctx, hipGetLastError());
        double* t = a; a = b; b = t;
    }
    { ProfScope ps(ctx, "alltoall", 8.0 * loc_z);
    BK_TRY(comm_alltoallv(ctx, b, p->cnt_f.data(), p->dsp_f.data(), a, p->cnt_b.data(), p->dsp_b.data())); }
    // z pass on the y-slab a = [nz][nyl][nx]: axis 2 of (nx, nyl, nz); symbol indices (kx, ky_local, kz)
    const bool fused = use_fft && p->twid[2] != nullptr;
    if (fused && ctx->opt("dct_roundtrip", 1.0) != 0.0) {
        BK_TRY(pass(nx, nyl, nz, 2, 2, 0, a, b, 2, p->lam[0], p->lam_yloc, p->lam[2], nullptr));  // forward, symbol, inverse in LDS
    } else {
        BK_TRY(pass(nx, nyl, nz, 2, 2, 0, a, b, fused ? 1 : 0, p->lam[0], p->lam_yloc, p->lam[2], nullptr));
        if (!fused) {
            hipLaunchKernelGGL(spectral_scale_kernel, dim3((unsigned)((loc_y + 255) / 256)), dim3(256), 0, ctx->stream, nx, nyl, nz,
                               p->lam[0], p->lam_yloc, p->lam[2], p->shift, b);
            BK_HIP(ctx, hipGetLastError());
        }
        BK_TRY(pass(nx, nyl, nz, 2, 2, 1, b, a, 0, nullptr, nullptr, nullptr, nullptr));
        double* t = a; a = b; b = t;
    }
    // all-to-all back: b (y-slab) -> a (blocks), then inverse y (reading the blocks, or after the unpack) and x
    { ProfScope ps(ctx, "alltoall", 8.0 * loc_y);
    BK_TRY(comm_alltoallv(ctx, b, p->cnt_b.data(), p->dsp_b.data(), a, p->cnt_f.data(), p->dsp_f.data())); }
    if (direct) {
        BK_TRY(pass(nx, ny, nzl, 1, 1, 1, a, b, 0, nullptr, nullptr, nullptr, &sp));
    } else {
        { ProfScope ps(ctx, "transpose", 16.0 * loc_z);
        hipLaunchKernelGGL(slab_blocks_kernel, dim3((unsigned)((loc_z + 255) / 256)), dim3(256), 0, ctx->stream, nx, ny, nzl, p->R, yc, a, b, 1);
        BK_HIP(ctx, hipGetLastError()); }
        BK_TRY(pass(nx, ny, nzl, 1, 1, 1, b, a, 0, nullptr, nullptr, nullptr, nullptr));
        double* t = a; a = b; b = t;
    }
    BK_TRY(pass(nx, ny, nzl, 0, 0, 1, b, out, 0, nullptr, nullptr, nullptr, nullptr));
    return 0;
}

int dct_slab_emulate_tables(bk_ctx* ctx, DctPlan* p, double az) { return slab_tables_create(ctx, p, p->n[2], true, az); }

namespace {
struct ShDctPrecond : bk_precond {
    DctPlan* plan = nullptr;
    // the grid whose L1 this plan diagonalises, by VALUE (bk_precond_sh_create): the problem object may be destroyed or its
    // address reused while the preconditioner lives on
    bool has_grid = false;
    bk_problem_desc grid{};
    int grid_lo = 0, grid_hi = 0;
    bool is_l1_plus_shift(const bk_problem* pr, double* shift) const override {
        if (!plan || plan->kind != 0 || !has_grid || !pr) return false;
        const bk_problem_desc &a = pr->desc, &b = grid;
        if (a.pde != b.pde || a.ndim != b.ndim || pr->lo != grid_lo || pr->hi != grid_hi) return false;
        for (int d = 0; d < a.ndim; ++d)
            if (a.n[d] != b.n[d] || a.l[d] != b.l[d]) return false;
        *shift = plan->shift;
        return true;
    }
    bool shadow = false;                      // second-lane view: the tables belong to the original, only the scratch
                                              // arrays (t1, t2; fsend, frecv of the slab z-solve) are its own
    ~ShDctPrecond() override {
        if (!shadow) { dct_plan_destroy(plan); return; }
        if (plan) { ws_put(ctx, plan->t1); ws_put(ctx, plan->t2); ws_put(ctx, plan->fsend); ws_put(ctx, plan->frecv); delete plan; }
    }
    int apply(const double* v, double* out) override {
        if (plan->kind >= 1) return dst_apply(ctx, plan, v, out);
        return plan->dist ? dct_apply_dist(ctx, plan, v, out) : dct_apply(ctx, plan, v, out);
    }
    // the x passes of this plan run as the fused LDS kernel on these buffers: the pointwise work of apply_pw can ride in them
    bool pw_fused_ok(const double* x, const double* u, const double* out) const override {
        if (!plan || plan->kind != 0 || plan->ndim < 2 || !plan->twid[0] || ctx->opt("dct_fft", 1.0) == 0.0) return false;
        if (((uintptr_t)x | (uintptr_t)u | (uintptr_t)out) & 15) return false;
        const int nzl = plan->dist ? plan->zhi - plan->zlo : (plan->slab_ok ? plan->nl : plan->n[2]);
        return dct_axis_fused_ok(ctx, plan->n[0], plan->n[1], nzl, 0, x, plan->t1, 0) &&
               dct_axis_fused_ok(ctx, plan->n[0], plan->n[1], nzl, 0, plan->t2, const_cast<double*>(out), 0);
    }
    bool pw_plan_ok() const override { return plan && plan->t1 && plan->t2 && pw_fused_ok(plan->t1, plan->t1, plan->t2); }
    int apply_pw(const double* x, const DctFuse& d, double cx, double ct, double* out) override {
        if (!pw_fused_ok(x, d.u, out) || ctx->opt("dct_fuse_pw", 1.0) == 0.0) return bk_precond::apply_pw(x, d, cx, ct, out);
        DctFuse f = d;
        f.xadd = nullptr; f.cx = 0.0; f.ct = 1.0;
        const bool scale_after = cx == 0.0 && ct != 1.0;              // (rare: no x term but a scale -- not worth a kernel variant)
        if (cx != 0.0) { f.xadd = x; f.cx = cx; f.ct = ct; }
        BK_TRY(plan->dist ? dct_apply_dist(ctx, plan, x, out, &f) : dct_apply(ctx, plan, x, out, nullptr, &f));
        return scale_after ? v_scale(ctx, n, ct, out) : 0;
    }
    int apply_dot_pre_axpy(double* y, double c, const double* r, double* out, double* dot) override {
        // (plan->slab_ok: the cost model's slab emulation routes through dct_apply_slab, which only knows the stencil-free operator's fusions)
        if (plan->kind >= 1 || plan->dist || plan->slab_ok || plan->ndim < 2 || ctx->nranks != 1 || ctx->opt("minres_fuse_axpy", 1.0) == 0.0 ||
            ctx->opt("dct_roundtrip", 1.0) == 0.0 || (((uintptr_t)r) & 15) || !pw_fused_ok(y, y, out))
            return bk_precond::apply_dot_pre_axpy(y, c, r, out, dot);
        DctFuse f;
        f.add = r; f.cadd = c; f.store = y;
        int nb = 0;
        BK_TRY(dct_apply(ctx, plan, y, out, &nb, &f));
        if (nb == 0) return v_dot(ctx, n, y, out, dot);       // the merged middle did not run as the fused kernel
        BK_TRY(reduce_finish(ctx, nb, 1, 0));
        *dot = ctx->h_red[0];
        return 0;
    }
    int apply_dot(const double* v, double* out, double* dot) override {
        if (plan->kind >= 1 || plan->dist || plan->ndim < 2 || ctx->nranks != 1) return bk_precond::apply_dot(v, out, dot);
        int nb = 0;
        BK_TRY(dct_apply(ctx, plan, v, out, &nb));
        if (nb == 0) return v_dot(ctx, n, v, out, dot);       // the merged middle did not run as the fused kernel
        BK_TRY(reduce_finish(ctx, nb, 1, 0));
        *dot = ctx->h_red[0];
        return 0;
    }
};
}  // namespace

bk_precond* precond_lane_shadow(bk_precond* pl, bk_ctx* lane) {
    ShDctPrecond* P = dynamic_cast<ShDctPrecond*>(pl);
    if (!P || !P->plan) return nullptr;
    DctPlan* q = new DctPlan(*P->plan);        // shares every table pointer; never handed to dct_plan_destroy
    q->t1 = q->t2 = q->fsend = q->frecv = nullptr;
    bool ok = ws_get(lane, q->total, &q->t1) == 0 && ws_get(lane, q->total, &q->t2) == 0;
    if (ok && P->plan->fsend) {                // slab z-solve: the face buffers are scratch too (4 doubles per line each)
        const size_t fb = 4 * (size_t)q->n[0] * (size_t)q->n[1];
        ok = ws_get(lane, fb, &q->fsend) == 0 && ws_get(lane, fb, &q->frecv) == 0;
    }
    if (!ok) {
        ws_put(lane, q->t1); ws_put(lane, q->t2); ws_put(lane, q->fsend); ws_put(lane, q->frecv);
        delete q;
        return nullptr;
    }
    ShDctPrecond* S = new ShDctPrecond();
    S->ctx = lane; S->n = P->n; S->plan = q; S->shadow = true;
    S->has_grid = P->has_grid; S->grid = P->grid; S->grid_lo = P->grid_lo; S->grid_hi = P->grid_hi;
    S->pw_agreed = P->pw_agreed;
    return S;
}

}  // namespace bk

using namespace bk;

extern "C" {

int bk_precond_sh_create(bk_problem* prob, double shift, bk_precond** out) {
    if (!prob || !out) return -1;
    bk_ctx* ctx = prob->ctx;
    if (prob->desc.pde != BK_PDE_SH) return set_error(ctx, "bk_precond_sh_create: Swift-Hohenberg 2-D/3-D only");
    ShDctPrecond* P = new ShDctPrecond();
    P->ctx = ctx;
    P->n = prob->nloc;
    P->has_grid = true; P->grid = prob->desc; P->grid_lo = prob->lo; P->grid_hi = prob->hi;
    int s = ctx->nranks > 1
                ? dct_plan_create_dist(ctx, prob->desc.n, prob->ainv, shift, prob->lo, prob->hi, &P->plan)
                : dct_plan_create(ctx, prob->desc.ndim, prob->desc.n, prob->ainv, shift, &P->plan);
    if (s != 0) { delete P; return s; }
    const int emu = (int)ctx->opt("dct_slab_emulate", 0.0);
    if (ctx->nranks == 1 && emu > 1 && prob->desc.ndim == 3) {
        // cost-model runs (bench.py --size-z): this single-rank grid plays the z-slab of rank 1 of `emu`; the local kernels of
        // the slab z-solve run exactly as there, the two face all-to-alls degenerate to local copies.  The RESULT is not the
        // preconditioner of this grid -- timing only.
        P->plan->R = emu; P->plan->rank = 1;
        if ((s = dct_slab_emulate_tables(ctx, P->plan, prob->ainv[2])) != 0) { delete P; return s; }
    }
    *out = P;
    return 0;
}

int bk_precond_lap_create(bk_problem* prob, double c, bk_precond** out) {
    if (!prob || !out) return -1;
    bk_ctx* ctx = prob->ctx;
    if (prob->desc.pde != BK_PDE_CGL2D) return set_error(ctx, "bk_precond_lap_create: cGL2d problems only");
    if (!(c > 0.0)) return set_error(ctx, "bk_precond_lap_create: c must be positive (Lap - c I is then definite)");
    ShDctPrecond* P = new ShDctPrecond();
    P->ctx = ctx;
    P->n = prob->nloc;
    int s = dst_plan_create(ctx, prob->desc.n, prob->ainv, c, 2, &P->plan);
    if (s != 0) { delete P; return s; }
    *out = P;
    return 0;
}

int bk_precond_cgl_create(bk_problem* prob, double a, double b, bk_precond** out) {
    if (!prob || !out) return -1;
    bk_ctx* ctx = prob->ctx;
    if (prob->desc.pde != BK_PDE_CGL2D) return set_error(ctx, "bk_precond_cgl_create: cGL2d problems only");
    ShDctPrecond* P = new ShDctPrecond();
    P->ctx = ctx;
    P->n = prob->nloc;
    int s = dst_plan_create(ctx, prob->desc.n, prob->ainv, 0.0, 2, &P->plan);
    if (s != 0) { delete P; return s; }
    // singular only if b = 0 and -a is an eigenvalue of the Laplacian: refuse the b = 0, a >= 0 corner outright
    if (b == 0.0 && !(a < 0.0)) { delete P; return set_error(ctx, "bk_precond_cgl_create: b = 0 needs a < 0 (Lap + a I definite)"); }
    P->plan->kind = 2;
    P->plan->blk_a = a;
    P->plan->blk_b = b;
    *out = P;
    return 0;
}

int bk_precond_destroy(bk_precond* pc) {
    delete pc;
    return 0;
}

int bk_precond_apply(bk_precond* pc, const double* v, double* out) {
    if (!pc || !v || !out) return -1;
    return pc->apply(v, out);
}

}  // extern "C"
