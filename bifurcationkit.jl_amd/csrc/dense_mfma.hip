// Dense sine transforms of the cGL2d preconditioners on the fp64 matrix cores, hand-written (round 3).
//
// The Dirichlet Laplacian of examples/cGL2d.jl:6-22 is diagonalised by the DST-I, S[j][k] = sqrt(2/(N+1)) sin(pi (j+1)(k+1)/(N+1)),
// and N = 1024 has no radix-2 FFT (length 2(N+1) = 2050), so the transform is a dense N x N product per axis -- rounds 1-2 ran
// it as four rocBLAS dgemm calls per application (4.3 GFLOP each).  Two things replace them:
//
//  * the reflection symmetry S[N-1-j][k] = (-1)^k S[j][k] halves the work: with e_j = x_j + x_{N-1-j}, o_j = x_j - x_{N-1-j}
//    (j < N/2) the even outputs are Se' e and the odd ones So' o with the (N/2) x (N/2) blocks Se[j][q] = S[j][2q],
//    So[j][q] = S[j][2q+1]; the inverse is x_j = A_j + B_j, x_{N-1-j} = A_j - B_j with A = Se ye, B = So yo.  The spectrum
//    is kept in the permuted order [even k | odd k] between the passes (the symbol kernels read permuted eigenvalue tables),
//    so nothing is ever interleaved;
//  * gemm_f64_kernel: C(M x N) = A(M x K) B(K x N), row-major, 128 x 64 tile per workgroup, eight wavefronts of 32 x 32,
//    v_mfma_f64_16x16x4_f64 (A: lane -> (row lane & 15, k lane >> 4), B: (k lane >> 4, col lane & 15), D: row (lane >> 4)
//    + 4 r, col lane & 15), operands staged through double-buffered LDS with the next chunk's global loads in flight
//    during the 8 x 4 MFMAs of the current one.  A 64 x 32 wave tile issues 8 MFMAs (64 cycles each) per 6 LDS reads: the
//    kernel is bound by the matrix pipe, which is the point.
#include <cmath>
#include <vector>

#include <algorithm>

#include "ops.h"

namespace bk {

namespace {

typedef double v4d __attribute__((ext_vector_type(4)));

constexpr int BM = 128, BN = 64, BK = 16, PAD = 4;

struct GemmBatch {
    const double* A;
    const double* B;
    double* C;
};
struct GemmP {
    int M, N, K, lda, ldb, ldc;
    GemmBatch b[4];
};

constexpr int GT = 512;                // threads: 8 wavefronts of 32 x 32 (two per SIMD: one's LDS latency hides behind the other's MFMAs)

__global__ void __launch_bounds__(GT) gemm_f64_kernel(GemmP P) {
    __shared__ __attribute__((aligned(16))) double As[2][BK][BM + PAD];
    __shared__ __attribute__((aligned(16))) double Bs[2][BK][BN + PAD];
    const GemmBatch gb = P.b[blockIdx.z];
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = (wave >> 1) * 32, wn = (wave & 1) * 32;
    // staging maps: A tile 128 rows x 16 k (a thread takes 4 consecutive k of one row), B tile 16 k x 64 cols (2 consecutive cols)
    const int ar = tid >> 2, ak = (tid & 3) * 4;
    const int bk = tid >> 5, bn = (tid & 31) * 2;
    const double* Ag = gb.A + (size_t)(m0 + ar) * P.lda + ak;
    const double* Bg = gb.B + (size_t)bk * P.ldb + n0 + bn;
    double2 ra[2], rb;
    auto gload = [&](int k0) {
        ra[0] = *reinterpret_cast<const double2*>(Ag + k0);
        ra[1] = *reinterpret_cast<const double2*>(Ag + k0 + 2);
        rb = *reinterpret_cast<const double2*>(Bg + (size_t)k0 * P.ldb);
    };
    auto sstore = [&](int buf) {
        As[buf][ak][ar] = ra[0].x; As[buf][ak + 1][ar] = ra[0].y;
        As[buf][ak + 2][ar] = ra[1].x; As[buf][ak + 3][ar] = ra[1].y;
        *reinterpret_cast<double2*>(&Bs[buf][bk][bn]) = rb;
    };
    v4d acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = (v4d){0.0, 0.0, 0.0, 0.0};
    const int li = lane & 15, lk = lane >> 4;
    gload(0);
    sstore(0);
    __syncthreads();
    int cur = 0;
    for (int k0 = 0; k0 < P.K; k0 += BK) {
        const bool more = k0 + BK < P.K;
        if (more) gload(k0 + BK);
#pragma unroll
        for (int kk = 0; kk < BK; kk += 4) {
            double a[2], b[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) a[i] = As[cur][kk + lk][wm + i * 16 + li];
#pragma unroll
            for (int j = 0; j < 2; ++j) b[j] = Bs[cur][kk + lk][wn + j * 16 + li];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        if (more) {
            sstore(cur ^ 1);
            __syncthreads();
            cur ^= 1;
        }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                gb.C[(size_t)(m0 + wm + i * 16 + lk + 4 * r) * P.ldc + n0 + wn + j * 16 + li] = acc[i][j][r];
}

// The same product for ANY extents (the dense DCT-II passes of the Swift-Hohenberg preconditioner on grids that are not a
// power of two, and the sine transforms of cGL grids that do not fit the tiles above): scalar guarded loads that fill the
// tile's overhang with zeros, guarded stores, a strided batch in blockIdx.z.  Same tiles, same MFMA schedule.
struct GemmS {
    int M, N, K, lda, ldb, ldc;
    const double* A;
    const double* B;
    double* C;
    size_t sA, sB, sC;
};

__global__ void __launch_bounds__(GT) gemm_f64_any_kernel(GemmS P) {
    __shared__ __attribute__((aligned(16))) double As[2][BK][BM + PAD];
    __shared__ __attribute__((aligned(16))) double Bs[2][BK][BN + PAD];
    const double* A = P.A + (size_t)blockIdx.z * P.sA;
    const double* B = P.B + (size_t)blockIdx.z * P.sB;
    double* C = P.C + (size_t)blockIdx.z * P.sC;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = (wave >> 1) * 32, wn = (wave & 1) * 32;
    const int ar = tid >> 2, ak = (tid & 3) * 4;
    const int bk = tid >> 5, bn = (tid & 31) * 2;
    const bool arow = m0 + ar < P.M;
    const double* Ag = A + (size_t)(arow ? m0 + ar : 0) * P.lda;
    double ra[4], rb[2];
    auto gload = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int kk = k0 + ak + i;
            ra[i] = (arow && kk < P.K) ? Ag[kk] : 0.0;
        }
        const int kb = k0 + bk;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = n0 + bn + j;
            rb[j] = (kb < P.K && col < P.N) ? B[(size_t)kb * P.ldb + col] : 0.0;
        }
    };
    auto sstore = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 4; ++i) As[buf][ak + i][ar] = ra[i];
        Bs[buf][bk][bn] = rb[0]; Bs[buf][bk][bn + 1] = rb[1];
    };
    v4d acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = (v4d){0.0, 0.0, 0.0, 0.0};
    const int li = lane & 15, lk = lane >> 4;
    gload(0);
    sstore(0);
    __syncthreads();
    int cur = 0;
    for (int k0 = 0; k0 < P.K; k0 += BK) {
        const bool more = k0 + BK < P.K;
        if (more) gload(k0 + BK);
#pragma unroll
        for (int kk = 0; kk < BK; kk += 4) {
            double a[2], b[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) a[i] = As[cur][kk + lk][wm + i * 16 + li];
#pragma unroll
            for (int j = 0; j < 2; ++j) b[j] = Bs[cur][kk + lk][wn + j * 16 + li];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        if (more) {
            sstore(cur ^ 1);
            __syncthreads();
            cur ^= 1;
        }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = m0 + wm + i * 16 + lk + 4 * r, col = n0 + wn + j * 16 + li;
                if (row < P.M && col < P.N) C[(size_t)row * P.ldc + col] = acc[i][j][r];
            }
}

// x fold: X[rows][N] -> W[2][rows][N/2] (e, o);  unfold: W -> X
__global__ void __launch_bounds__(256) fold_x_kernel(size_t rows, int N, const double* __restrict__ X, double* __restrict__ W) {
    const int h = N >> 1;
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= rows * (size_t)h) return;
    const size_t r = idx / h;
    const int j = (int)(idx - r * h);
    const double a = X[r * N + j], b = X[r * N + (N - 1 - j)];
    W[idx] = a + b;
    W[rows * (size_t)h + idx] = a - b;
}
__global__ void __launch_bounds__(256) unfold_x_kernel(size_t rows, int N, const double* __restrict__ W, double* __restrict__ X) {
    const int h = N >> 1;
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= rows * (size_t)h) return;
    const size_t r = idx / h;
    const int j = (int)(idx - r * h);
    const double a = W[idx], b = W[rows * (size_t)h + idx];
    X[r * N + j] = a + b;
    X[r * N + (N - 1 - j)] = a - b;
}
// y fold: X[nb][N1][n0] -> W[2][nb][N1/2][n0]
__global__ void __launch_bounds__(256) fold_y_kernel(int n0, int N1, int nb, const double* __restrict__ X, double* __restrict__ W) {
    const int h = N1 >> 1;
    const size_t per = (size_t)h * n0, tot = per * nb;
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= tot) return;
    const size_t f = idx / per, rem = idx - f * per;
    const int j = (int)(rem / n0), x = (int)(rem - (size_t)j * n0);
    const double* Xf = X + f * (size_t)N1 * n0;
    const double a = Xf[(size_t)j * n0 + x], b = Xf[(size_t)(N1 - 1 - j) * n0 + x];
    W[idx] = a + b;
    W[tot + idx] = a - b;
}
__global__ void __launch_bounds__(256) unfold_y_kernel(int n0, int N1, int nb, const double* __restrict__ W, double* __restrict__ X) {
    const int h = N1 >> 1;
    const size_t per = (size_t)h * n0, tot = per * nb;
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= tot) return;
    const size_t f = idx / per, rem = idx - f * per;
    const int j = (int)(rem / n0), x = (int)(rem - (size_t)j * n0);
    double* Xf = X + f * (size_t)N1 * n0;
    const double a = W[idx], b = W[tot + idx];
    Xf[(size_t)j * n0 + x] = a + b;
    Xf[(size_t)(N1 - 1 - j) * n0 + x] = a - b;
}

}  // namespace

bool dense_mfma_supported(int n0, int n1, int nb) {
    // tile shapes: x pass M = nb n1 rows (BM), N = K = n0 / 2 (BN, BK); y pass M = K = n1 / 2, N = n0
    return n0 % 2 == 0 && n1 % 2 == 0 && (n0 / 2) % BN == 0 && (n0 / 2) % BK == 0 && (n1 / 2) % BM == 0 && (n1 / 2) % BK == 0 &&
           ((size_t)nb * n1) % BM == 0 && n0 % BN == 0 && nb >= 1 && nb <= 2;
}

// Host tables of one axis from the symmetric N x N transform T (row-major): Te[j][q] = T[j][2q], To[j][q] = T[j][2q+1]
// (j, q < N/2), their transposes, and the eigenvalues in the permuted order [even k | odd k].
void dense_mfma_tables(int N, const std::vector<double>& T, const std::vector<double>& lam, std::vector<double>& Te,
                       std::vector<double>& To, std::vector<double>& TeT, std::vector<double>& ToT, std::vector<double>& lamp) {
    const int h = N / 2;
    Te.assign((size_t)h * h, 0.0); To = Te; TeT = Te; ToT = Te;
    lamp.assign(N, 0.0);
    for (int j = 0; j < h; ++j)
        for (int q = 0; q < h; ++q) {
            const double e = T[(size_t)j * N + 2 * q], o = T[(size_t)j * N + 2 * q + 1];
            Te[(size_t)j * h + q] = e; TeT[(size_t)q * h + j] = e;
            To[(size_t)j * h + q] = o; ToT[(size_t)q * h + j] = o;
        }
    for (int q = 0; q < h; ++q) { lamp[q] = lam[2 * q]; lamp[h + q] = lam[2 * q + 1]; }
}

// One axis pass on the array [nb][n1][n0] (n0 fastest).  Forward: in (physical along the axis) -> out (permuted spectrum);
// inverse: permuted spectrum -> physical.  `work` holds n0 n1 nb doubles; in / out / work pairwise distinct.
// tab = {Te, To, TeT, ToT} of the axis (device).
int dense_mfma_pass(bk_ctx* ctx, int n0, int n1, int nb, int axis, int inverse, const double* const tab[4], const double* in,
                    double* out, double* work) {
    const size_t total = (size_t)n0 * n1 * nb;
    GemmP P;
    unsigned gx, gy, gz;
    if (axis == 0) {
        const int h = n0 / 2;
        const size_t rows = (size_t)n1 * nb;
        P.M = (int)rows; P.N = h; P.K = h;
        gx = h / BN; gy = (unsigned)(rows / BM); gz = 2;
        if (!inverse) {
            hipLaunchKernelGGL(fold_x_kernel, dim3((unsigned)((rows * h + 255) / 256)), dim3(256), 0, ctx->stream, rows, n0, in, work);
            P.lda = h; P.ldb = h; P.ldc = n0;
            for (int p = 0; p < 2; ++p) P.b[p] = {work + (size_t)p * rows * h, tab[p], out + (size_t)p * h};
            hipLaunchKernelGGL(gemm_f64_kernel, dim3(gx, gy, gz), dim3(GT), 0, ctx->stream, P);
        } else {
            P.lda = n0; P.ldb = h; P.ldc = h;
            for (int p = 0; p < 2; ++p) P.b[p] = {in + (size_t)p * h, tab[2 + p], work + (size_t)p * rows * h};
            hipLaunchKernelGGL(gemm_f64_kernel, dim3(gx, gy, gz), dim3(GT), 0, ctx->stream, P);
            hipLaunchKernelGGL(unfold_x_kernel, dim3((unsigned)((rows * h + 255) / 256)), dim3(256), 0, ctx->stream, rows, n0, work, out);
        }
    } else {
        const int h = n1 / 2;
        const size_t per = (size_t)h * n0;                     // one (parity, field) block of W
        P.M = h; P.N = n0; P.K = h;
        gx = n0 / BN; gy = h / BM; gz = 2 * nb;
        if (!inverse) {
            hipLaunchKernelGGL(fold_y_kernel, dim3((unsigned)((per * nb + 255) / 256)), dim3(256), 0, ctx->stream, n0, n1, nb, in, work);
            P.lda = h; P.ldb = n0; P.ldc = n0;
            for (int p = 0; p < 2; ++p)
                for (int f = 0; f < nb; ++f)
                    P.b[p * nb + f] = {tab[2 + p], work + ((size_t)p * nb + f) * per, out + (size_t)f * n1 * n0 + (size_t)p * per};
            hipLaunchKernelGGL(gemm_f64_kernel, dim3(gx, gy, gz), dim3(GT), 0, ctx->stream, P);
        } else {
            P.lda = h; P.ldb = n0; P.ldc = n0;
            for (int p = 0; p < 2; ++p)
                for (int f = 0; f < nb; ++f)
                    P.b[p * nb + f] = {tab[p], in + (size_t)f * n1 * n0 + (size_t)p * per, work + ((size_t)p * nb + f) * per};
            hipLaunchKernelGGL(gemm_f64_kernel, dim3(gx, gy, gz), dim3(GT), 0, ctx->stream, P);
            hipLaunchKernelGGL(unfold_y_kernel, dim3((unsigned)((per * nb + 255) / 256)), dim3(256), 0, ctx->stream, n0, n1, nb, work, out);
        }
    }
    (void)total;
    BK_HIP(ctx, hipGetLastError());        // (launch errors are sticky until read: one check covers the fold / gemm / unfold launches above)
    return 0;
}

// One dense axis pass of the array [n2][n1][n0] (n0 fastest) with the N x N matrix M of the axis: along x
// Out(rows x N) = X(rows x N) M, along y / z Out_plane(N x inner) = M' X_plane, batched over the planes.  MT = M' (row-major).
int dense_gemm_axis_pass(bk_ctx* ctx, int n0, int n1, int n2, int axis, const double* M, const double* MT, const double* in,
                         double* out) {
    GemmS P;
    unsigned gz = 1;
    if (axis == 0) {
        const size_t rows = (size_t)n1 * n2;
        if (rows > 0x7fffffffu) return set_error(ctx, "dense_gemm_axis_pass: too many rows");
        P.M = (int)rows; P.N = n0; P.K = n0;
        P.A = in; P.lda = n0; P.B = M; P.ldb = n0; P.C = out; P.ldc = n0;
        P.sA = P.sB = P.sC = 0;
    } else {
        const int N = axis == 1 ? n1 : n2;
        const size_t inner = axis == 1 ? (size_t)n0 : (size_t)n0 * n1;
        if (inner > 0x7fffffffu) return set_error(ctx, "dense_gemm_axis_pass: plane too large");
        P.M = N; P.N = (int)inner; P.K = N;
        P.A = MT; P.lda = N; P.B = in; P.ldb = (int)inner; P.C = out; P.ldc = (int)inner;
        P.sA = 0; P.sB = P.sC = inner * (size_t)N;
        gz = axis == 1 ? (unsigned)n2 : 1u;
    }
    // grid y (row tiles) and z (planes) are 16-bit launch dimensions: longer extents go in slices of 65535 (the
    // rocBLAS path this kernel replaced took such shapes; an error here would have been a regression)
    const unsigned gx = (unsigned)((P.N + BN - 1) / BN);
    const size_t gy_all = ((size_t)P.M + BM - 1) / BM;
    const int M_all = P.M;
    const double *A0 = P.A, *B0 = P.B;
    double* C0 = P.C;
    for (unsigned z0 = 0; z0 < gz; z0 += 65535u) {
        const unsigned gzc = std::min(gz - z0, 65535u);
        for (size_t y0 = 0; y0 < gy_all; y0 += 65535) {
            const unsigned gyc = (unsigned)std::min<size_t>(gy_all - y0, 65535);
            P.M = (int)std::min<size_t>((size_t)M_all - y0 * BM, (size_t)gyc * BM);
            P.A = A0 + y0 * BM * (size_t)P.lda + (size_t)z0 * P.sA;
            P.B = B0 + (size_t)z0 * P.sB;
            P.C = C0 + y0 * BM * (size_t)P.ldc + (size_t)z0 * P.sC;
            hipLaunchKernelGGL(gemm_f64_any_kernel, dim3(gx, gyc, gzc), dim3(GT), 0, ctx->stream, P);
            BK_HIP(ctx, hipGetLastError());
        }
    }
    return 0;
}

}  // namespace bk
