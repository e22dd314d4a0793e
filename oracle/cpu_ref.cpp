// ORACLE / CPU BASELINE (test infrastructure -- never linked into the product): C++/OpenMP restatement of the
// REFERENCE'S OWN FORMULATION of one PALC corrector pass for the 3-D Swift-Hohenberg example, the second CPU baseline
// SURVEY section 8(d) asks for ("CSR SpMV with the assembled L1, MGS GMRES, all host cores").  Not Julia, not the product's
// algorithm: the operator is an assembled sparse matrix applied by SpMV, exactly as the example does it.
//
//   Laplacian / L1 = (I + Lap)^2 assembled as CSR          examples/SH3d.jl:16-41, :85   (A*A by sparse product)
//   F(u) = -L1 u + l u + nu u^2 - u^3,  dF(u) du            examples/SH3d.jl:44-53
//   GMRES(m): KrylovKit.linsolve semantics (ModifiedGramSchmidt2, restart cycles, numops)
//                                                           src/LinearSolver.jl:254-291; restated in oracle/krylov.py
//   Pl: exact (L1 + shift)^-1 through dense orthonormal DCT-II matrices per axis (stand-in for the sparse direct factor
//       `lu(L1 + I)` of examples/SH2d-fronts.jl:121, which has no C++ counterpart in this image)
//   BorderingBLS (BEC, check_precision = false)             src/LinearBorderSolver.jl:88-144
//   newton_palc, one iteration from the predictor           src/continuation/Palc.jl:187-305
//
// Usage: cpu_ref nx ny nz lx ly lz l nu shift ds theta u0.bin p0 u1.bin p1 [steps [dump_prefix [mode]]]
//   (u0, u1: raw float64, x fastest) -> one JSON line {seconds_per_step, threads, itlinear, residuals, p}
//   dump_prefix: also write <prefix>{xp,res,jtau,x1,x}.bin -- the predictor, F(predictor), J(predictor) tau, the solution of
//   J x1 = F of the first bordered solve and the corrected state -- for the generic-state GPU parity test
//   (tests/test_gpu_fullsize.py::test_generic_state_against_the_cpp_restatement).
//   mode (default "assembled"):
//     "factored"  L1 v is applied as A (A v) with the 7-point CSR matrix A = I + Lap instead of the assembled product L1 = A*A
//                 (the same operator -- examples/SH3d.jl:85 forms L1 as that matrix product -- at 7 instead of 25 stored entries
//                 per row: 11 GB instead of 40+ GB at 512^3); everything else as above
//     "apply"     factored, and NO solves: for the 512^3 parity test (tests/test_gpu_fullsize.py).  Writes xp, res, jtau as above,
//                 reads <prefix>v.bin and writes plv = Pl^-1 v, reads <prefix>x1g.bin (a solution of J x1 = F(predictor) computed
//                 elsewhere) and reports the TRUE preconditioned residual |Pl^-1 (J x1g - F)| / |Pl^-1 F| in the JSON line
// The numbers are checked against the NumPy oracle (oracle/palc.py) in tests/test_oracle.py.
#include <omp.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

typedef std::vector<double> vec;

struct Csr {
    int n = 0;
    std::vector<long> ptr;
    std::vector<int> col;
    std::vector<double> val;
    void mul(const vec& x, vec& y) const {
#pragma omp parallel for schedule(static)
        for (int i = 0; i < n; ++i) {
            double s = 0.0;
            for (long k = ptr[i]; k < ptr[i + 1]; ++k) s += val[k] * x[col[k]];
            y[i] = s;
        }
    }
};

// A = I + Dx (+) Dy (+) Dz with D = tridiag(1,-2,1)/h^2, D[0,0] = D[end,end] = -1/h^2 (Neumann ghost), h = 2l/N
static Csr assemble_A(const int n[3], const double l[3]) {
    Csr A;
    const int nx = n[0], ny = n[1], nz = n[2];
    A.n = nx * ny * nz;
    A.ptr.assign(A.n + 1, 0);
    double a[3];
    for (int d = 0; d < 3; ++d) { const double h = 2.0 * l[d] / n[d]; a[d] = 1.0 / (h * h); }
    const long stride[3] = {1, nx, (long)nx * ny};
    for (int pass = 0; pass < 2; ++pass) {
        long nnz = 0;
        for (int k = 0; k < nz; ++k)
            for (int j = 0; j < ny; ++j)
                for (int i = 0; i < nx; ++i) {
                    const int row = i + nx * (j + ny * k);
                    const int idx[3] = {i, j, k};
                    // gather (col, val) sorted by column: -z, -y, -x, diag, +x, +y, +z
                    int cols[7]; double vals[7]; int m = 0;
                    double diag = 1.0;
                    for (int d = 0; d < 3; ++d) {
                        const bool edge = idx[d] == 0 || idx[d] == n[d] - 1;
                        diag += (n[d] == 1) ? 0.0 : (edge ? -a[d] : -2.0 * a[d]);
                    }
                    for (int d = 2; d >= 0; --d)
                        if (idx[d] > 0) { cols[m] = row - (int)stride[d]; vals[m++] = a[d]; }
                    cols[m] = row; vals[m++] = diag;
                    for (int d = 0; d < 3; ++d)
                        if (idx[d] < n[d] - 1) { cols[m] = row + (int)stride[d]; vals[m++] = a[d]; }
                    if (pass == 1)
                        for (int q = 0; q < m; ++q) { A.col[nnz + q] = cols[q]; A.val[nnz + q] = vals[q]; }
                    nnz += m;
                    A.ptr[row + 1] = nnz;
                }
        if (pass == 0) { A.col.resize(nnz); A.val.resize(nnz); }
    }
    return A;
}

// C = A * A, row by row: a row of A has <= 7 entries, so a row of the product gathers <= 49 (column, value) pairs; they are sorted by
// column and merged in a fixed-size buffer (two passes: count, then fill -- no per-row allocations: 16.7 M rows take seconds)
static Csr spgemm(const Csr& A) {
    Csr C;
    C.n = A.n;
    C.ptr.assign(A.n + 1, 0);
    auto row = [&](int i, int* cols, double* vals) -> int {
        int m = 0;
        for (long k = A.ptr[i]; k < A.ptr[i + 1]; ++k) {
            const int j = A.col[k];
            const double v = A.val[k];
            for (long q = A.ptr[j]; q < A.ptr[j + 1]; ++q) {
                const int c = A.col[q];
                const double p = v * A.val[q];
                int t = m;                                   // insertion into the sorted prefix; equal columns accumulate
                while (t > 0 && cols[t - 1] > c) --t;
                if (t > 0 && cols[t - 1] == c) { vals[t - 1] += p; continue; }
                for (int s = m; s > t; --s) { cols[s] = cols[s - 1]; vals[s] = vals[s - 1]; }
                cols[t] = c; vals[t] = p;
                ++m;
            }
        }
        return m;
    };
    std::vector<int> cnt(A.n);
#pragma omp parallel for schedule(static)
    for (int i = 0; i < A.n; ++i) {
        int cols[64]; double vals[64];
        cnt[i] = row(i, cols, vals);
    }
    for (int i = 0; i < A.n; ++i) C.ptr[i + 1] = C.ptr[i] + cnt[i];
    C.col.resize(C.ptr[A.n]); C.val.resize(C.ptr[A.n]);
#pragma omp parallel for schedule(static)
    for (int i = 0; i < A.n; ++i) {
        int cols[64]; double vals[64];
        const int m = row(i, cols, vals);
        for (int t = 0; t < m; ++t) { C.col[C.ptr[i] + t] = cols[t]; C.val[C.ptr[i] + t] = vals[t]; }
    }
    return C;
}

static double dot(const vec& a, const vec& b) {
    double s = 0.0;
#pragma omp parallel for reduction(+ : s) schedule(static)
    for (long i = 0; i < (long)a.size(); ++i) s += a[i] * b[i];
    return s;
}
static double nrm2(const vec& a) { return std::sqrt(dot(a, a)); }
static double nrminf(const vec& a) {
    double s = 0.0;
#pragma omp parallel for reduction(max : s) schedule(static)
    for (long i = 0; i < (long)a.size(); ++i) s = std::max(s, std::fabs(a[i]));
    return s;
}
static void axpy(double a, const vec& x, vec& y) {
#pragma omp parallel for schedule(static)
    for (long i = 0; i < (long)x.size(); ++i) y[i] += a * x[i];
}

struct Problem {
    int n[3];
    double l[3];
    Csr L1;                     // assembled A*A (mode "assembled") ...
    Csr A;                      // ... or its factor, applied twice (modes "factored" / "apply")
    bool factored = false;
    double nu;
    mutable vec tmp, tmp2;
    void mulL1(const vec& x, vec& y) const {
        if (!factored) { L1.mul(x, y); return; }
        A.mul(x, tmp2);
        A.mul(tmp2, y);
    }
    void F(const vec& u, double lpar, vec& out) const {           // SH3d.jl:44-47
        mulL1(u, tmp);
#pragma omp parallel for schedule(static)
        for (long i = 0; i < (long)u.size(); ++i) out[i] = -tmp[i] + lpar * u[i] + nu * u[i] * u[i] - u[i] * u[i] * u[i];
    }
    void dF(const vec& u, double lpar, const vec& du, vec& out) const {     // SH3d.jl:50-53
        mulL1(du, tmp);
#pragma omp parallel for schedule(static)
        for (long i = 0; i < (long)u.size(); ++i) out[i] = -tmp[i] + (lpar + 2.0 * nu * u[i] - 3.0 * u[i] * u[i]) * du[i];
    }
};

// exact (L1 + shift)^-1: dense orthonormal DCT-II per axis, symbol 1/((1 + lx + ly + lz)^2 + shift)
struct Precond {
    int n[3];
    std::vector<double> T[3], Tt[3], lam[3];
    double shift;
    mutable vec a, b;
    void init(const int n_[3], const double l[3], double s) {
        shift = s;
        dense_only = getenv("CPU_REF_DENSE_DCT") != nullptr;
        for (int d = 0; d < 3; ++d) {
            n[d] = n_[d];
            const int N = n[d];
            const double h = 2.0 * l[d] / N;
            T[d].resize((size_t)N * N); Tt[d].resize((size_t)N * N); lam[d].resize(N);
            for (int k = 0; k < N; ++k) {
                lam[d][k] = -(4.0 / (h * h)) * std::pow(std::sin(M_PI * k / (2.0 * N)), 2);
                const double s_ = k == 0 ? std::sqrt(1.0 / N) : std::sqrt(2.0 / N);
                for (int j = 0; j < N; ++j) {
                    T[d][(size_t)k * N + j] = s_ * std::cos(M_PI * k * (j + 0.5) / N);
                    Tt[d][(size_t)j * N + k] = T[d][(size_t)k * N + j];
                }
            }
        }
        a.resize((size_t)n[0] * n[1] * n[2]); b.resize(a.size());
    }
    // out[.., k, ..] = sum_j M(k, j) in[.., j, ..] along axis d, M = T (forward) or T' (inverse), organised as axpy
    // sweeps over contiguous memory (a blocked GEMM): W[j][k] = M(k, j)
    void axis(int d, bool inverse, const vec& in, vec& out) const {
        const int N = n[d];
        const long total = (long)n[0] * n[1] * n[2];
        const std::vector<double>& W = inverse ? T[d] : Tt[d];    // W[j*N + k]
        if (d == 0) {
            const long lines = total / N;
#pragma omp parallel for schedule(static)
            for (long ln = 0; ln < lines; ++ln) {
                const double* x = &in[ln * N];
                double* y = &out[ln * N];
                for (int k = 0; k < N; ++k) y[k] = 0.0;
                for (int j = 0; j < N; ++j) {
                    const double xj = x[j];
                    const double* w = &W[(size_t)j * N];
                    for (int k = 0; k < N; ++k) y[k] += w[k] * xj;
                }
            }
            return;
        }
        const long inner = d == 1 ? n[0] : (long)n[0] * n[1];
        const long outer = total / (inner * N);
        const long chunk = std::min<long>(inner, 1024);
        const long nchunks = (inner + chunk - 1) / chunk;
#pragma omp parallel for collapse(2) schedule(static)
        for (long o = 0; o < outer; ++o)
            for (long c = 0; c < nchunks; ++c) {
                const long i0 = c * chunk, len = std::min(chunk, inner - i0);
                const double* x = &in[o * N * inner + i0];
                double* y = &out[o * N * inner + i0];
                for (int k = 0; k < N; ++k) {
                    double* yk = y + k * inner;
                    for (long i = 0; i < len; ++i) yk[i] = 0.0;
                }
                for (int j = 0; j < N; ++j) {
                    const double* xj = x + j * inner;
                    const double* w = &W[(size_t)j * N];
                    for (int k = 0; k < N; ++k) {
                        double* yk = y + k * inner;
                        const double wk = w[k];
                        for (long i = 0; i < len; ++i) yk[i] += wk * xj[i];
                    }
                }
            }
    }
    // Power-of-two extents: the same orthonormal DCT-II / DCT-III through an N-point complex FFT of the Makhoul-permuted line
    // (v_n = x_2n, v_{N-1-n} = x_{2n+1}; C_k = Re(w_k V_k), w_k = exp(-i pi k / 2N); inverse: V_k = conj(w_k) (C_k - i C_{N-k})),
    // LB lines at a time so that the butterflies vectorise across lines: O(N log N) instead of the O(N^2) dense sweep -- at 256^3 /
    // 512^3 the dense form is ~30x the work of everything else in this file.  `fast_ok` selects it per axis; CPU_REF_DENSE_DCT=1
    // forces the dense form (the two are compared by the selftest mode, tests/test_oracle.py).
    static constexpr int LB = 32;
    bool dense_only = false;
    bool fast_ok(int d) const { const int N = n[d]; return !dense_only && N >= 8 && (N & (N - 1)) == 0; }
    void axis_fast(int d, bool inverse, const vec& in, vec& out) const {
        const int N = n[d];
        int bits = 0;
        while ((1 << bits) < N) ++bits;
        const long total = (long)n[0] * n[1] * n[2];
        const long inner = d == 0 ? 1 : (d == 1 ? n[0] : (long)n[0] * n[1]);      // stride of the transform index
        const long outer = total / (inner * N);
        // lines: (o, i) -> base o * N * inner + i (d >= 1), or line index * N (d == 0); blocks of LB lines with consecutive i
        const long nlines = total / N;
        const long nblocks = (nlines + LB - 1) / LB;
        std::vector<double> wr(N / 2), wi(N / 2), er(N), ei(N);
        for (int k = 0; k < N / 2; ++k) { wr[k] = std::cos(2.0 * M_PI * k / N); wi[k] = -std::sin(2.0 * M_PI * k / N); }
        for (int k = 0; k < N; ++k) { er[k] = std::cos(M_PI * k / (2.0 * N)); ei[k] = -std::sin(M_PI * k / (2.0 * N)); }
        std::vector<int> rev(N);
        for (int i = 0; i < N; ++i) { int r = 0; for (int b = 0; b < bits; ++b) r |= ((i >> b) & 1) << (bits - 1 - b); rev[i] = r; }
        const double s0 = std::sqrt(1.0 / N), s2 = std::sqrt(2.0 / N);
        (void)outer;
#pragma omp parallel
        {
            std::vector<double> re((size_t)N * LB), im((size_t)N * LB), c((size_t)(N + 1) * LB);
#pragma omp for schedule(static)
            for (long blk = 0; blk < nblocks; ++blk) {
                const long l0 = blk * LB;
                const int L = (int)std::min<long>(LB, nlines - l0);
                // element (line l, index j) lives at addr(l) + j * inner
                long addr[LB];
                for (int l = 0; l < L; ++l) {
                    const long ln = l0 + l;
                    addr[l] = d == 0 ? ln * N : (ln / inner) * N * inner + (ln % inner);
                }
                if (!inverse) {
                    for (int j = 0; j < N; ++j) {                 // Makhoul permutation straight into bit-reversed order
                        const int m = (j & 1) ? N - 1 - (j >> 1) : (j >> 1);
                        double* r = &re[(size_t)rev[m] * LB];
                        double* q = &im[(size_t)rev[m] * LB];
                        for (int l = 0; l < L; ++l) { r[l] = in[addr[l] + (long)j * inner]; q[l] = 0.0; }
                    }
                } else {
                    for (int k = 0; k < N; ++k) {
                        const double sk = 1.0 / (k == 0 ? s0 : s2);
                        double* ck = &c[(size_t)k * LB];
                        for (int l = 0; l < L; ++l) ck[l] = in[addr[l] + (long)k * inner] * sk;
                    }
                    for (int l = 0; l < L; ++l) c[(size_t)N * LB + l] = 0.0;
                    for (int k = 0; k < N; ++k) {                 // V_k = conj(w_k) (C_k - i C_{N-k})
                        const double* ck = &c[(size_t)k * LB];
                        const double* cn = &c[(size_t)(N - k) * LB];
                        double* r = &re[(size_t)rev[k] * LB];
                        double* q = &im[(size_t)rev[k] * LB];
                        const double a_ = er[k], b_ = -ei[k];     // conj(w_k) = a_ + i b_
                        for (int l = 0; l < L; ++l) { r[l] = a_ * ck[l] + b_ * cn[l]; q[l] = b_ * ck[l] - a_ * cn[l]; }
                    }
                }
                // radix-2 decimation in time on bit-reversed input; inverse: conjugate twiddles
                for (int sb = 0; sb < bits; ++sb) {
                    const int half = 1 << sb, step = N >> (sb + 1);
                    for (int g = 0; g < N; g += 2 * half)
                        for (int t = 0; t < half; ++t) {
                            const double tr = wr[(size_t)t * step], ti = inverse ? -wi[(size_t)t * step] : wi[(size_t)t * step];
                            double* ar = &re[(size_t)(g + t) * LB]; double* ai = &im[(size_t)(g + t) * LB];
                            double* br = &re[(size_t)(g + t + half) * LB]; double* bi = &im[(size_t)(g + t + half) * LB];
                            for (int l = 0; l < LB; ++l) {
                                const double xr = tr * br[l] - ti * bi[l], xi = tr * bi[l] + ti * br[l];
                                br[l] = ar[l] - xr; bi[l] = ai[l] - xi;
                                ar[l] += xr; ai[l] += xi;
                            }
                        }
                }
                if (!inverse) {
                    for (int k = 0; k < N; ++k) {                 // X_k = s_k Re(w_k V_k)
                        const double sk = k == 0 ? s0 : s2;
                        const double* r = &re[(size_t)k * LB];
                        const double* q = &im[(size_t)k * LB];
                        for (int l = 0; l < L; ++l) out[addr[l] + (long)k * inner] = sk * (er[k] * r[l] - ei[k] * q[l]);
                    }
                } else {
                    const double invN = 1.0 / N;
                    for (int m = 0; m < N; ++m) {                 // x_2n = v_n, x_{2n+1} = v_{N-1-n}
                        const int j = m < N / 2 ? 2 * m : 2 * (N - 1 - m) + 1;
                        const double* r = &re[(size_t)m * LB];
                        for (int l = 0; l < L; ++l) out[addr[l] + (long)j * inner] = r[l] * invN;
                    }
                }
            }
        }
    }
    void axis_any(int d, bool inverse, const vec& in, vec& out) const {
        if (fast_ok(d)) axis_fast(d, inverse, in, out);
        else axis(d, inverse, in, out);
    }
    void apply(const vec& v, vec& out) const {
        axis_any(0, false, v, a); axis_any(1, false, a, b); axis_any(2, false, b, a);
#pragma omp parallel for schedule(static)
        for (long i = 0; i < (long)a.size(); ++i) {
            const int ix = (int)(i % n[0]), iy = (int)((i / n[0]) % n[1]), iz = (int)(i / ((long)n[0] * n[1]));
            const double sy = 1.0 + lam[0][ix] + lam[1][iy] + lam[2][iz];
            a[i] /= sy * sy + shift;
        }
        axis_any(2, true, a, b); axis_any(1, true, b, a); axis_any(0, true, a, out);
    }
};

// GMRESKrylovKit with Pl (src/LinearSolver.jl:268-288): operator dx -> Pl^-1 (J dx), rhs Pl^-1 b; KrylovKit GMRES with
// ModifiedGramSchmidt2, restart cycles, numops counting (restated in oracle/krylov.py: gmres_krylovkit)
struct Gmres {
    int m = 30, maxiter = 150;
    double atol = 1e-12, rtol = 1e-9;
    const Problem* prob; const Precond* P;
    const vec* u; double lpar;
    mutable vec t;
    void op(const vec& x, vec& y) const { prob->dF(*u, lpar, x, t); P->apply(t, y); }
    static void givens(double f, double g, double& c, double& s, double& r) {
        if (g == 0.0) { c = 1; s = 0; r = f; }
        else if (f == 0.0) { c = 0; s = 1; r = g; }
        else { r = std::hypot(f, g); c = f / r; s = g / r; }
    }
    void mgs2(vec& w, const std::vector<vec>& V, int k, std::vector<double>& h) const {
        h.assign(k, 0.0);
        for (int i = 0; i < k; ++i) { h[i] = dot(V[i], w); axpy(-h[i], V[i], w); }
        for (int i = 0; i < k; ++i) { const double s = dot(V[i], w); axpy(-s, V[i], w); h[i] += s; }
    }
    int solve(const vec& rhs, vec& x, int* numops_out) const {
        const size_t n = rhs.size();
        t.resize(n);
        vec b(n), r(n), w(n);
        P->apply(rhs, b);
        std::fill(x.begin(), x.end(), 0.0);
        r = b;
        int numops = 1;
        double beta = nrm2(r);
        const double tol = std::max(atol, rtol * nrm2(b));
        if (beta < tol) { *numops_out = numops; return 1; }
        std::vector<vec> V(m + 1, vec(n));
        std::vector<double> R((size_t)m * m, 0.0), y(m + 1), cs(m), sn(m), h, col;
        double nrm = 0.0;
        auto start = [&]() {
            for (size_t i = 0; i < n; ++i) V[0][i] = r[i] / beta;
            op(V[0], w);
            mgs2(w, V, 1, h);
            nrm = nrm2(w);
        };
        start();
        numops += 1;
        int numiter = 0;
        while (numiter < maxiter) {
            numiter += 1;
            std::fill(y.begin(), y.end(), 0.0);
            y[0] = beta;
            int k = 1;
            double rr;
            givens(h[0], nrm, cs[0], sn[0], rr);
            R[0] = rr;
            y[1] = -sn[0] * y[0]; y[0] = cs[0] * y[0];
            beta = std::fabs(y[1]);
            while (beta > tol && k < m) {
                for (size_t i = 0; i < n; ++i) V[k][i] = w[i] / nrm;
                op(V[k], w);
                numops += 1;
                mgs2(w, V, k + 1, h);
                nrm = nrm2(w);
                k += 1;
                col = h;
                for (int i = 0; i < k - 1; ++i) {
                    const double tt = cs[i] * col[i] + sn[i] * col[i + 1];
                    col[i + 1] = -sn[i] * col[i] + cs[i] * col[i + 1];
                    col[i] = tt;
                }
                givens(col[k - 1], nrm, cs[k - 1], sn[k - 1], rr);
                col[k - 1] = rr;
                for (int i = 0; i < k; ++i) R[(size_t)i * m + (k - 1)] = col[i];
                y[k] = -sn[k - 1] * y[k - 1]; y[k - 1] = cs[k - 1] * y[k - 1];
                beta = std::fabs(y[k]);
            }
            std::vector<double> yk(k);
            for (int i = k - 1; i >= 0; --i) {
                double s = y[i];
                for (int j = i + 1; j < k; ++j) s -= R[(size_t)i * m + j] * yk[j];
                yk[i] = s / R[(size_t)i * m + i];
            }
            for (int i = 0; i < k; ++i) axpy(yk[i], V[i], x);
            if (beta > tol) {
                for (size_t i = 0; i < n; ++i) V[k][i] = w[i] / nrm;
                std::vector<double> z(k + 1, 0.0);
                z[k] = 1.0;
                for (int i = k - 1; i >= 0; --i) {
                    const double tt = cs[i] * z[i] - sn[i] * z[i + 1];
                    z[i + 1] = sn[i] * z[i] + cs[i] * z[i + 1];
                    z[i] = tt;
                }
                std::fill(r.begin(), r.end(), 0.0);
                for (int i = 0; i <= k; ++i) axpy(y[k] * z[i], V[i], r);
            } else {
                op(x, r);
                numops += 1;
#pragma omp parallel for schedule(static)
                for (long i = 0; i < (long)n; ++i) r[i] = b[i] - r[i];
                beta = nrm2(r);
                if (beta < tol) { *numops_out = numops; return 1; }
            }
            if (numiter < maxiter) { beta = nrm2(r); start(); numops += 1; }
        }
        *numops_out = numops;
        return 0;
    }
};

static vec read_bin(const char* path, size_t n) {
    vec v(n);
    FILE* f = fopen(path, "rb");
    if (!f || fread(v.data(), sizeof(double), n, f) != n) { fprintf(stderr, "cannot read %s\n", path); exit(2); }
    fclose(f);
    return v;
}

int main(int argc, char** argv) {
    if (argc < 16) { fprintf(stderr, "usage: cpu_ref nx ny nz lx ly lz l nu shift ds theta u0.bin p0 u1.bin p1 [steps [dump_prefix]]\n"); return 2; }
    Problem pb;
    for (int d = 0; d < 3; ++d) { pb.n[d] = atoi(argv[1 + d]); pb.l[d] = atof(argv[4 + d]); }
    const double lpar0 = atof(argv[7]);
    pb.nu = atof(argv[8]);
    const double shift = atof(argv[9]), ds = atof(argv[10]), theta = atof(argv[11]);
    const size_t N = (size_t)pb.n[0] * pb.n[1] * pb.n[2];
    const vec u0 = read_bin(argv[12], N), u1 = read_bin(argv[14], N);
    const double p0 = atof(argv[13]), p1 = atof(argv[15]);
    const int steps = argc > 16 ? atoi(argv[16]) : 1;
    const char* dump = argc > 17 ? argv[17] : nullptr;
    const std::string mode = argc > 18 ? argv[18] : "assembled";
    if (mode == "selftest") {
        // fast (FFT) vs dense transform passes of the preconditioner on this grid: max |difference| of Pl^-1 u0, relative
        Precond Pf, Pd;
        Pf.init(pb.n, pb.l, shift);
        Pd.init(pb.n, pb.l, shift);
        Pd.dense_only = true;
        vec a_(N), b_(N);
        Pf.apply(u0, a_);
        Pd.apply(u0, b_);
        double dmax = 0.0;
        for (size_t i = 0; i < N; ++i) dmax = std::max(dmax, std::fabs(a_[i] - b_[i]));
        printf("{\"mode\": \"selftest\", \"fast_axes\": [%d, %d, %d], \"dct_fast_vs_dense_rel\": %.3e}\n", (int)Pf.fast_ok(0), (int)Pf.fast_ok(1),
               (int)Pf.fast_ok(2), dmax / nrminf(b_));
        return 0;
    }
    if (mode != "assembled" && mode != "factored" && mode != "apply") { fprintf(stderr, "unknown mode %s\n", mode.c_str()); return 2; }
    pb.factored = mode != "assembled";
    auto write_bin = [&](const char* tag, const vec& v) {
        if (!dump) return;
        const std::string path = std::string(dump) + tag + ".bin";
        FILE* f = fopen(path.c_str(), "wb");
        if (!f || fwrite(v.data(), sizeof(double), v.size(), f) != v.size()) { fprintf(stderr, "cannot write %s\n", path.c_str()); exit(2); }
        fclose(f);
    };
    (void)lpar0;
    auto t_setup = std::chrono::steady_clock::now();
    if (pb.factored) {
        pb.A = assemble_A(pb.n, pb.l);
        pb.tmp2.resize(N);
    } else {
        Csr A = assemble_A(pb.n, pb.l);
        pb.L1 = spgemm(A);                                         // L1 = A*A, SH3d.jl:85
    }
    pb.tmp.resize(N);
    Precond P;
    P.init(pb.n, pb.l, shift);
    const double setup_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_setup).count();
    // secant tangent (Tangents.jl:28-42) and predictor
    vec tau(N);
    for (size_t i = 0; i < N; ++i) tau[i] = u1[i] - u0[i];
    double taup = p1 - p0;
    const double nn = std::sqrt(theta * dot(tau, tau) / N + (1 - theta) * taup * taup);
    const double sc = std::copysign(1.0, ds) / nn;
    for (auto& v : tau) v *= sc;
    taup *= sc;
    vec xp(N);
    for (size_t i = 0; i < N; ++i) xp[i] = u0[i] + ds * tau[i];
    const double pp = p0 + ds * taup;
    const double eps = 1.4901161193847656e-08;
    double res0 = 0, res1 = 0, pnew = 0;
    int itlin = 0, itl[2] = {0, 0};
    vec x(N), res_f(N), dFdp(N), x1(N), dx(N);
    const double dz0 = dot(u0, tau);
    auto Nfun = [&](const vec& xx, double p) { return (theta * dot(xx, tau) / N + (1 - theta) * (p - p0) * taup - ds) - theta * dz0 / N; };
    if (mode == "apply") {
        if (!dump) { fprintf(stderr, "mode apply needs a dump prefix\n"); return 2; }
        x = xp;
        pb.F(x, pp, res_f);
        write_bin("xp", x);
        write_bin("res", res_f);
        vec t(N), w(N);
        pb.dF(x, pp, tau, t);
        write_bin("jtau", t);
        {
            const vec v = read_bin((std::string(dump) + "v.bin").c_str(), N);
            P.apply(v, w);
            write_bin("plv", w);
        }
        double rel = -1.0, nb = 0.0;
        {
            const vec xg = read_bin((std::string(dump) + "x1g.bin").c_str(), N);
            pb.dF(x, pp, xg, t);                                   // J x1g
#pragma omp parallel for schedule(static)
            for (long i = 0; i < (long)N; ++i) t[i] -= res_f[i];
            P.apply(t, w);
            const double nr = nrm2(w);
            P.apply(res_f, w);
            nb = nrm2(w);
            rel = nr / nb;
        }
        printf("{\"mode\": \"apply\", \"threads\": %d, \"n\": %zu, \"setup_seconds\": %.3f, \"residual_inf\": %.17g, \"p_pred\": %.17g, "
               "\"tau_p\": %.17g, \"true_residual_rel\": %.17g, \"norm_pl_rhs\": %.17g, \"nnz_A\": %ld}\n",
               omp_get_max_threads(), N, setup_s, std::max(nrminf(res_f), std::fabs(Nfun(x, pp))), pp, taup, rel, nb, (long)pb.A.ptr[pb.A.n]);
        return 0;
    }
    auto t0 = std::chrono::steady_clock::now();
    for (int s = 0; s < steps; ++s) {                              // newton_palc, one iteration (Palc.jl:237-295)
        x = xp;
        double p = pp;
        pb.F(x, p, res_f);
        double res_n = Nfun(x, p);
        res0 = std::max(nrminf(res_f), std::fabs(res_n));
        pb.F(x, p + eps, dFdp);
#pragma omp parallel for schedule(static)
        for (long i = 0; i < (long)N; ++i) dFdp[i] = (dFdp[i] - res_f[i]) / eps;
        Gmres g;
        g.prob = &pb; g.P = &P; g.u = &x; g.lpar = p;
        int it1 = 0, it2 = 0;
        if (dump && s == 0) {
            write_bin("xp", x);
            write_bin("res", res_f);
            vec jt(N);
            pb.dF(x, p, tau, jt);
            write_bin("jtau", jt);
        }
        g.solve(res_f, x1, &it1);                                  // BEC: x1 = J^-1 R, dx = J^-1 dR  (:134-136)
        if (dump && s == 0) write_bin("x1", x1);
        g.solve(dFdp, dx, &it2);
        itlin = it1 + it2;
        itl[0] = it1; itl[1] = it2;
        const double xiu = theta, xip = 1 - theta;
        const double dl = (res_n - dot(tau, x1) / N * xiu) / (taup * xip - dot(tau, dx) / N * xiu);
        axpy(-dl, dx, x1);                                         // dX = x1 - dl dx
        axpy(-1.0, x1, x);
        p = p - dl;
        pb.F(x, p, res_f);
        res_n = Nfun(x, p);
        res1 = std::max(nrminf(res_f), std::fabs(res_n));
        pnew = p;
        if (dump && s == 0) write_bin("x", x);
    }
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() / steps;
    printf("{\"seconds_per_step\": %.6f, \"setup_seconds\": %.3f, \"threads\": %d, \"n\": %zu, \"itlinear\": %d, "
           "\"itlinear_each\": [%d, %d], \"residuals\": [%.17g, %.17g], \"p\": %.17g, \"p_pred\": %.17g, \"tau_p\": %.17g, \"nnz_L1\": %ld}\n",
           dt, setup_s, omp_get_max_threads(), N, itlin, itl[0], itl[1], res0, res1, pnew, pp, taup, (long)(pb.factored ? pb.A.ptr[pb.A.n] : pb.L1.ptr[pb.L1.n]));
    return 0;
}
