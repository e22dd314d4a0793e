// Host replay of the block Arnoldi step (bifurcationkit.jl_amd/csrc/sstep.h) on a dense matrix: the two streaming passes of
// vecops.hip (block_dots_kernel / block_axpy_kernel) are plain loops here, the coefficient algebra is the library's header.
// stdin: n m s, then A (n rows), then b, then optionally ns and ns Newton shifts (used cyclically inside every block).  stdout: status per block ("ok <defect of the folded solution update>"), then H ((m+1) x m, row-major), then Q ((m+1) x n).
#include <cstdio>
#include <vector>
#include "../../bifurcationkit.jl_amd/csrc/sstep.h"
using namespace bk::sstep;
int main() {
    int n, m, s;
    if (scanf("%d %d %d", &n, &m, &s) != 3) return 2;
    std::vector<double> A((size_t)n * n), b(n);
    for (auto& x : A) if (scanf("%lf", &x) != 1) return 2;
    for (auto& x : b) if (scanf("%lf", &x) != 1) return 2;
    int ns = 0;
    std::vector<double> shifts;
    if (scanf("%d", &ns) == 1 && ns > 0) { shifts.resize(ns); for (auto& x : shifts) if (scanf("%lf", &x) != 1) return 2; }
    const int ldg = kMaxK + 1, ldh = m + 2;
    std::vector<double> Q((size_t)(m + 1) * n, 0.0), G((size_t)ldg * ldg, 0.0), H((size_t)ldh * m, 0.0);
    double nb = 0.0;
    for (double x : b) nb += x * x;
    nb = std::sqrt(nb);
    for (int i = 0; i < n; ++i) Q[i] = b[i] / nb;
    auto dot = [&](const double* x, const double* y) { double v = 0.0; for (int i = 0; i < n; ++i) v += x[i] * y[i]; return v; };
    int j = 0, gram_n = 0, fails = 0;
    double fold_defect = 0.0;
    while (j < m) {
        const int sb = s < m - j ? s : m - j, k = j + 1, u = k - gram_n, ko = k - u, nr = u + sb;
        double theta[kS] = {0.0, 0.0, 0.0, 0.0};
        for (int i = 0; i < sb && ns > 0; ++i) theta[i] = shifts[i % ns];
        for (int i = 0; i < sb; ++i) {                       // p_{i+1} = (A - theta_i) p_i into the next basis slots
            const double* x = &Q[(size_t)(j + i) * n];
            double* y = &Q[(size_t)(j + i + 1) * n];
            for (int r = 0; r < n; ++r) { double v = -theta[i] * x[r]; for (int c = 0; c < n; ++c) v += A[(size_t)r * n + c] * x[c]; y[r] = v; }
        }
        std::vector<double> D((size_t)(ko > 0 ? ko : 1) * kR, 0.0), T(kTri, 0.0);
        const double* rhs0 = &Q[(size_t)ko * n];
        for (int i = 0; i < ko; ++i)
            for (int r = 0; r < nr; ++r) D[(size_t)i * kR + r] = dot(&Q[(size_t)i * n], rhs0 + (size_t)r * n);
        for (int r = 0; r < nr; ++r)
            for (int c = r; c < nr; ++c) T[tri(r, c)] = dot(rhs0 + (size_t)r * n, rhs0 + (size_t)c * n);
        double Cm[kMaxK * kS], Tm[kS * kS];
        int got = 0;
        const int st = block_coefficients(k, u, sb, D.data(), T.data(), G.data(), ldg, H.data(), ldh, Cm, Tm, &got, nullptr, ns > 0 ? theta : nullptr);
        gram_n = k;
        if (st != 0) { ++fails; printf("block at %d failed\n", j); return 0; }
        if (got < sb) { printf("block at %d truncated to %d\n", j, got); return 0; }
        std::vector<double> out((size_t)sb * n, 0.0);
        for (int q = 0; q < sb; ++q)
            for (int e = 0; e < n; ++e) {
                double v = 0.0;
                for (int r = 0; r < sb; ++r) v += Tm[r * kS + q] * Q[(size_t)(k + r) * n + e];
                for (int i = 0; i < k; ++i) v += Cm[i * kS + q] * Q[(size_t)i * n + e];
                out[(size_t)q * n + e] = v;
            }
        if (j + sb >= m) {
            // the last block: a solution update with coefficients on c = 1 .. sb of its vectors, once through the explicit new
            // vectors and once through fold_solution_coefficients on [Q, P] (the deferred update pass of solver.hip)
            for (int c = 1; c <= sb; ++c) {
                std::vector<double> yk(k + c), cf(k + c);
                for (int i = 0; i < k + c; ++i) yk[i] = ((i % 3) - 0.8) / (1.0 + 0.3 * i);
                fold_solution_coefficients(k, c, Cm, Tm, yk.data(), cf.data());
                double worst = 0.0, big = 0.0;
                for (int e = 0; e < n; ++e) {
                    double a = 0.0, f = 0.0;
                    for (int i = 0; i < k; ++i) { a += yk[i] * Q[(size_t)i * n + e]; f += cf[i] * Q[(size_t)i * n + e]; }
                    for (int q = 0; q < c; ++q) { a += yk[k + q] * out[(size_t)q * n + e]; f += cf[k + q] * Q[(size_t)(k + q) * n + e]; }
                    worst = std::fmax(worst, std::fabs(a - f)); big = std::fmax(big, std::fabs(a));
                }
                fold_defect = std::fmax(fold_defect, worst / big);
            }
        }
        for (int q = 0; q < sb; ++q)
            for (int e = 0; e < n; ++e) Q[(size_t)(k + q) * n + e] = out[(size_t)q * n + e];
        j += sb;
    }
    printf("ok %.3e\n", fold_defect);
    for (int a = 0; a <= m; ++a) { for (int c = 0; c < m; ++c) printf("%.17g ", H[(size_t)a + (size_t)c * ldh]); printf("\n"); }
    for (int a = 0; a <= m; ++a) { for (int e = 0; e < n; ++e) printf("%.17g ", Q[(size_t)a * n + e]); printf("\n"); }
    return 0;
}
