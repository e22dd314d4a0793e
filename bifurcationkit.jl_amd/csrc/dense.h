// Small dense host-side linear algebra for the projected (Rayleigh-quotient) problems of the
// Krylov-Schur eigensolver and the GMRES least-squares update.  Sizes are <= 64, so clarity beats
// speed.  Header-only, no HIP: unit-tested on the CPU (tests/test_dense_host.py builds it with g++).
//
// Role in the reference: KrylovKit.eigsolve / ArnoldiMethod / LinearAlgebra.eigen do this through
// LAPACK (src/EigSolver.jl:42-49, 157-160, 204-225); none of that source is in the reference tree.
#pragma once

#include <algorithm>
#include <cmath>
#include <complex>
#include <vector>

namespace bk {
namespace dense {

typedef std::complex<double> cplx;

// Column-major helpers: A(i,j) = a[i + j*n]
struct Mat {
    int n = 0, m = 0;
    std::vector<double> a;
    Mat() {}
    Mat(int n_, int m_) : n(n_), m(m_), a((size_t)n_ * m_, 0.0) {}
    double& operator()(int i, int j) { return a[(size_t)i + (size_t)j * n]; }
    double operator()(int i, int j) const { return a[(size_t)i + (size_t)j * n]; }
};
struct CMat {
    int n = 0, m = 0;
    std::vector<cplx> a;
    CMat() {}
    CMat(int n_, int m_) : n(n_), m(m_), a((size_t)n_ * m_, cplx(0.0, 0.0)) {}
    cplx& operator()(int i, int j) { return a[(size_t)i + (size_t)j * n]; }
    cplx operator()(int i, int j) const { return a[(size_t)i + (size_t)j * n]; }
};

// Cyclic Jacobi for a real symmetric matrix.  On return w = eigenvalues (ascending), columns of Z the
// orthonormal eigenvectors.  Returns the number of sweeps used (<0: no convergence).
inline int jacobi_eigh(const Mat& Ain, std::vector<double>& w, Mat& Z) {
    const int n = Ain.n;
    Mat A = Ain;
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < i; ++j) { const double s = 0.5 * (A(i, j) + A(j, i)); A(i, j) = s; A(j, i) = s; }
    Z = Mat(n, n);
    for (int i = 0; i < n; ++i) Z(i, i) = 1.0;
    int sweep = 0;
    for (; sweep < 100; ++sweep) {
        double off = 0.0, diag = 0.0;
        for (int i = 0; i < n; ++i) {
            diag += A(i, i) * A(i, i);
            for (int j = 0; j < i; ++j) off += 2.0 * A(i, j) * A(i, j);
        }
        if (off <= 1e-30 * (diag + off) || off == 0.0) break;   // ||offdiag||_F <= 1e-15 ||A||_F
        for (int p = 0; p < n - 1; ++p)
            for (int q = p + 1; q < n; ++q) {
                const double apq = A(p, q);
                if (apq == 0.0) continue;
                const double theta = (A(q, q) - A(p, p)) / (2.0 * apq);
                const double t = (theta >= 0.0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
                const double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < n; ++k) {
                    const double akp = A(k, p), akq = A(k, q);
                    A(k, p) = c * akp - s * akq;
                    A(k, q) = s * akp + c * akq;
                }
                for (int k = 0; k < n; ++k) {
                    const double apk = A(p, k), aqk = A(q, k);
                    A(p, k) = c * apk - s * aqk;
                    A(q, k) = s * apk + c * aqk;
                }
                for (int k = 0; k < n; ++k) {
                    const double zkp = Z(k, p), zkq = Z(k, q);
                    Z(k, p) = c * zkp - s * zkq;
                    Z(k, q) = s * zkp + c * zkq;
                }
            }
    }
    std::vector<int> idx(n);
    for (int i = 0; i < n; ++i) idx[i] = i;
    std::sort(idx.begin(), idx.end(), [&](int x, int y) { return A(x, x) < A(y, y); });
    w.resize(n);
    Mat Zs(n, n);
    for (int j = 0; j < n; ++j) {
        w[j] = A(idx[j], idx[j]);
        for (int i = 0; i < n; ++i) Zs(i, j) = Z(i, idx[j]);
    }
    Z = Zs;
    return sweep < 100 ? sweep : -1;
}

// Eigen-decomposition of a general real matrix through a complex Schur form (Householder Hessenberg
// reduction + explicit single-shift QR with Wilkinson shifts).  w = eigenvalues, columns of Y = unit
// 2-norm eigenvectors.  Returns 0 on success, -1 if the QR iteration did not converge.
inline int eig_general(const Mat& Ain, std::vector<cplx>& w, CMat& Y) {
    const int n = Ain.n;
    CMat H(n, n), Z(n, n);
    {
        // real Householder reduction to Hessenberg form, Q accumulated
        Mat A = Ain, Q(n, n);
        for (int i = 0; i < n; ++i) Q(i, i) = 1.0;
        std::vector<double> v(n);
        for (int k = 0; k + 2 < n; ++k) {
            double alpha = 0.0;
            for (int i = k + 1; i < n; ++i) alpha += A(i, k) * A(i, k);
            alpha = std::sqrt(alpha);
            if (alpha == 0.0) continue;
            if (A(k + 1, k) > 0.0) alpha = -alpha;
            for (int i = 0; i < n; ++i) v[i] = 0.0;
            v[k + 1] = A(k + 1, k) - alpha;
            for (int i = k + 2; i < n; ++i) v[i] = A(i, k);
            double vn = 0.0;
            for (int i = k + 1; i < n; ++i) vn += v[i] * v[i];
            if (vn == 0.0) continue;
            // A <- (I - 2 v v'/vn) A (I - 2 v v'/vn)
            for (int j = 0; j < n; ++j) {
                double s = 0.0;
                for (int i = k + 1; i < n; ++i) s += v[i] * A(i, j);
                s *= 2.0 / vn;
                for (int i = k + 1; i < n; ++i) A(i, j) -= s * v[i];
            }
            for (int i = 0; i < n; ++i) {
                double s = 0.0;
                for (int j = k + 1; j < n; ++j) s += A(i, j) * v[j];
                s *= 2.0 / vn;
                for (int j = k + 1; j < n; ++j) A(i, j) -= s * v[j];
            }
            for (int i = 0; i < n; ++i) {
                double s = 0.0;
                for (int j = k + 1; j < n; ++j) s += Q(i, j) * v[j];
                s *= 2.0 / vn;
                for (int j = k + 1; j < n; ++j) Q(i, j) -= s * v[j];
            }
        }
        for (int j = 0; j < n; ++j)
            for (int i = 0; i < n; ++i) {
                H(i, j) = (i <= j + 1) ? cplx(A(i, j), 0.0) : cplx(0.0, 0.0);
                Z(i, j) = cplx(Q(i, j), 0.0);
            }
    }
    const double eps = 2.220446049250313e-16;
    double hnorm = 0.0;
    for (int j = 0; j < n; ++j)
        for (int i = 0; i <= std::min(j + 1, n - 1); ++i) hnorm = std::max(hnorm, std::abs(H(i, j)));
    if (hnorm == 0.0) hnorm = 1.0;
    int hi = n - 1, iter = 0, total = 0;
    std::vector<double> cs(n);
    std::vector<cplx> sn(n);
    while (hi > 0) {
        // find the active block [lo..hi]
        int lo = hi;
        while (lo > 0) {
            double s = std::abs(H(lo - 1, lo - 1)) + std::abs(H(lo, lo));
            if (s == 0.0) s = hnorm;
            if (std::abs(H(lo, lo - 1)) <= eps * s) { H(lo, lo - 1) = 0.0; break; }
            --lo;
        }
        if (lo == hi) { --hi; iter = 0; continue; }
        if (++total > 60 * n + 200) return -1;
        ++iter;
        // Wilkinson shift from the trailing 2x2 (exceptional shifts every 10 iterations)
        cplx mu;
        if (iter % 10 == 0) {
            mu = H(hi, hi) + cplx(std::abs(H(hi, hi - 1)), 0.0);
        } else {
            const cplx a = H(hi - 1, hi - 1), b = H(hi - 1, hi), c = H(hi, hi - 1), d = H(hi, hi);
            const cplx tr = a + d, det = a * d - b * c;
            const cplx disc = std::sqrt(tr * tr * 0.25 - det);
            const cplx e1 = tr * 0.5 + disc, e2 = tr * 0.5 - disc;
            mu = (std::abs(e1 - d) < std::abs(e2 - d)) ? e1 : e2;
        }
        for (int k = lo; k <= hi; ++k) H(k, k) -= mu;
        for (int k = lo; k < hi; ++k) {       // QR: zero the subdiagonal with Givens rotations
            const cplx a = H(k, k), b = H(k + 1, k);
            const double r = std::sqrt(std::norm(a) + std::norm(b));
            double c;
            cplx s;
            if (r == 0.0) { c = 1.0; s = 0.0; }
            else if (std::abs(a) == 0.0) { c = 0.0; s = std::conj(b) / r; }
            else { c = std::abs(a) / r; s = (a / std::abs(a)) * std::conj(b) / r; }
            cs[k] = c; sn[k] = s;
            for (int j = k; j < n; ++j) {
                const cplx x = H(k, j), y = H(k + 1, j);
                H(k, j) = c * x + s * y;
                H(k + 1, j) = -std::conj(s) * x + c * y;
            }
        }
        for (int k = lo; k < hi; ++k) {       // RQ: apply the adjoint rotations from the right
            const double c = cs[k];
            const cplx s = sn[k];
            const int imax = std::min(k + 2, hi);
            for (int i = 0; i <= imax; ++i) {
                const cplx x = H(i, k), y = H(i, k + 1);
                H(i, k) = c * x + std::conj(s) * y;
                H(i, k + 1) = -s * x + c * y;
            }
            for (int i = 0; i < n; ++i) {
                const cplx x = Z(i, k), y = Z(i, k + 1);
                Z(i, k) = c * x + std::conj(s) * y;
                Z(i, k + 1) = -s * x + c * y;
            }
        }
        for (int k = lo; k <= hi; ++k) H(k, k) += mu;
    }
    w.resize(n);
    for (int i = 0; i < n; ++i) w[i] = H(i, i);
    // eigenvectors of the triangular factor by back substitution, then rotate with Z
    Y = CMat(n, n);
    std::vector<cplx> y(n);
    for (int i = 0; i < n; ++i) {
        for (int j = 0; j < n; ++j) y[j] = 0.0;
        y[i] = 1.0;
        for (int j = i - 1; j >= 0; --j) {
            cplx s = 0.0;
            for (int k = j + 1; k <= i; ++k) s += H(j, k) * y[k];
            cplx d = H(j, j) - H(i, i);
            if (std::abs(d) < eps * hnorm) d = eps * hnorm;
            y[j] = -s / d;
        }
        double nrm = 0.0;
        for (int r = 0; r < n; ++r) {
            cplx s = 0.0;
            for (int k = 0; k <= i; ++k) s += Z(r, k) * y[k];
            Y(r, i) = s;
            nrm += std::norm(s);
        }
        nrm = std::sqrt(nrm);
        if (nrm > 0.0)
            for (int r = 0; r < n; ++r) Y(r, i) /= nrm;
    }
    return 0;
}

// Real Givens rotation: [c s; -s c] [f; g] = [r; 0]
inline void givens(double f, double g, double& c, double& s, double& r) {
    if (g == 0.0) { c = 1.0; s = 0.0; r = f; }
    else if (f == 0.0) { c = 0.0; s = 1.0; r = g; }
    else { r = std::hypot(f, g); c = f / r; s = g / r; }
}

// Orthonormalise the columns of C (n x m, column-major) with modified Gram-Schmidt (two passes),
// dropping columns whose remainder falls below `droptol` times their original norm.  Returns the kept
// column count; Q receives them.
inline int orthonormalize_columns(const Mat& C, double droptol, Mat& Q) {
    const int n = C.n, m = C.m;
    std::vector<std::vector<double>> cols;
    for (int j = 0; j < m; ++j) {
        std::vector<double> v(n);
        double n0 = 0.0;
        for (int i = 0; i < n; ++i) { v[i] = C(i, j); n0 += v[i] * v[i]; }
        n0 = std::sqrt(n0);
        if (n0 == 0.0) continue;
        for (int pass = 0; pass < 2; ++pass)
            for (auto& q : cols) {
                double s = 0.0;
                for (int i = 0; i < n; ++i) s += q[i] * v[i];
                for (int i = 0; i < n; ++i) v[i] -= s * q[i];
            }
        double n1 = 0.0;
        for (int i = 0; i < n; ++i) n1 += v[i] * v[i];
        n1 = std::sqrt(n1);
        if (n1 <= droptol * n0) continue;
        for (int i = 0; i < n; ++i) v[i] /= n1;
        cols.push_back(v);
    }
    Q = Mat(n, (int)cols.size());
    for (int j = 0; j < (int)cols.size(); ++j)
        for (int i = 0; i < n; ++i) Q(i, j) = cols[j][i];
    return (int)cols.size();
}

}  // namespace dense
}  // namespace bk
