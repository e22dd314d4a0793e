// Host algebra of the BLOCK Arnoldi step of gmres_core (solver.hip): s Arnoldi steps whose orthogonalisation reads the Krylov
// basis ONCE for the projections and ONCE for the update, instead of once each PER STEP.  Header-only, no HIP: replayed on the
// CPU by tests/cpp/sstep_check.cpp (tests/test_sstep_host.py) against a textbook Arnoldi process.
//
// Role in the reference: the orthogonalisation inside KrylovKit.linsolve / IterativeSolvers.gmres! (src/LinearSolver.jl:198,
// 256-291; package sources are not in the reference tree).  Those packages orthogonalise every new vector against the whole basis
// with 2k .. 4k BLAS-1 passes; round 3 made that two fused passes per step; this makes it two fused passes per s steps.
//
// One block, starting from an (approximately) orthonormal basis Q = [q_0 .. q_j], k = j + 1, with the raw Hessenberg columns
// 0 .. j-1 known (A Q[:, :j] = Q H_old):
//   1. p_1 = (A - th_0) q_j, p_2 = (A - th_1) p_1, ..., p_s = (A - th_{s-1}) p_{s-1}   (s operator applications, nothing in
//      between; th = 0: the monomial block, th = Ritz values in Leja order: the Newton block, see Conditioning)
//   2. ONE pass over the basis and the block (vecops.hip: block_dots_kernel):  Aq = Q'P,  Gp = P'P,  and the Gram columns
//      Q'Q_u of the u basis vectors created by the PREVIOUS block, which nobody has measured yet
//   3. here: C = G^-1 Aq -- the coefficients of the ORTHOGONAL projection of P onto span(Q) under the MEASURED Gram matrix G
//      (the block analogue of round 3's Gram-corrected single pass: the defect of earlier vectors is measured and projected
//      out, it never accumulates); S = Gp - C'Aq = (P - QC)'(P - QC); R = chol(S); Q_new = (P - Q C) R^-1
//   4. ONE pass (block_axpy_kernel): the s new basis vectors, in place over P
//   5. here: the s new Hessenberg columns.  With B = [q_j, p_1 .. p_{s-1}] and P = [p_1 .. p_s] the block satisfies
//      A B = P + B diag(th) exactly (th = 0 in what follows; the shifts add Bc diag(th) to the right-hand side); in coordinates of Q+ = [Q, Q_new]:  P = Q+ Pc, Pc = [C; R],  B = Q+ Bc, Bc = [e_j, Pc[:, :s-1]].  Splitting
//      B into its components along q_0 .. q_{j-1} (where A is known: H_old) and along q_j, q_{j+1} .. q_{j+s-1} (rows U of Bc,
//      upper triangular with U_00 = 1, U_ii = R_{i-1,i-1} > 0):   H_new U = Pc - [H_old; 0] Bc[:j, :]   =>  H_new.
// Conditioning.  The monomial block [A q, A^2 q, ..] loses independence at the rate GMRES converges: the part of p_q outside
// span(Q, p_1 .. p_{q-1}) -- the Cholesky pivot -- shrinks, relative to |p_q|^2, by about the square of the residual reduction
// per step (measured on the 512^3 corrector: 5e-3, 5e-5, 7e-7, 3e-10 down the first block), and the orthonormality INSIDE the
// block is (rounding of the dots, ~1e-13) / (smallest pivot ratio).  So the block is TRUNCATED where the ratio falls below
// kPivotTol: the leading s_eff columns are a valid smaller block, the trailing operator applications are discarded and the
// caller shrinks its next blocks (solver.hip).  Shifts at Ritz values of the operator (Leja order; solver.hip takes them from the
// Hessenberg matrix at hand) remove the directions GMRES has already resolved from the block's vectors and lift the pivots by 3
// to 6 orders (same corrector: 4e-1, 2e-2, 1e-2, 4e-4), wherever a shift costs no extra pass (bk_op::shift_is_free).  s_eff = 0 (w in span(Q) to working precision, or a non-positive Gram matrix):
// return 1, the caller repeats the step on the single-vector path with its explicit cancellation branch.
#pragma once

#include <cmath>

namespace bk {
namespace sstep {

constexpr int kS = 4;                 // largest block
constexpr int kR = 8;                 // right-hand vectors of one block_dots launch: u unmeasured + s new, u, s <= 4
constexpr int kTri = kR * (kR + 1) / 2;
constexpr int kMaxK = 64;             // basis vectors
constexpr double kPivotTol = 1e-8;    // smallest accepted (pivot^2 / column norm^2) of chol(S): in-block orthonormality ~1e-5

// packed upper triangle of the kR x kR matrix of dots among the right-hand vectors, r <= c
inline int tri(int r, int c) { return r * kR - r * (r - 1) / 2 + (c - r); }

// In:  k basis vectors of which the last u are unmeasured; s new block vectors.
//      D[i * kR + r], i < k - u : <q_i, rhs_r>      rhs = [q_{k-u} .. q_{k-1}, p_1 .. p_s]  (nr = u + s)
//      T[tri(r, c)]             : <rhs_r, rhs_c>
//      G (ldg x ldg, column-major): measured Gram matrix, valid for the first k - u vectors; completed here
//      Hraw (ldh x *, column-major): raw Hessenberg, columns 0 .. k-2 valid (column c has c + 2 entries)
//      theta (optional): the s shifts of the block, p_{q+1} = (A - theta[q]) p_q
// Out: *s_eff <= s accepted columns (see Conditioning above); Cm[i * kS + q] = -(C R^-1)(i, q), i < k;  Tm[r * kS + q] =
//      R^-1(r, q) (upper), both for q < *s_eff;  columns k-1 .. k+*s_eff-2 of Hraw
// Returns 0, or 1 if not even one column is acceptable (nothing but G was modified).
constexpr double kGrowRatio = 1e-4;   // the next block may be one step longer if the last accepted pivot ratio is above this
inline int block_coefficients(int k, int u, int s_in, const double* D, const double* T, double* G, int ldg, double* Hraw, int ldh,
                              double* Cm, double* Tm, int* s_eff, double* last_ratio = nullptr, const double* theta = nullptr) {
    int s = s_in;
    *s_eff = 0;
    double ratio = 1.0;
    const int ko = k - u, j = k - 1;
    if (k < 1 || k > kMaxK || u < 0 || u > kS || u > k || s < 1 || s > kS || u + s > kR) return 1;
    auto g = [&](int a, int b) -> double& { return G[(size_t)a + (size_t)b * ldg]; };
    // Gram columns of the unmeasured vectors
    for (int t = 0; t < u; ++t) {
        const int c = ko + t;
        for (int i = 0; i < ko; ++i) { g(i, c) = D[i * kR + t]; g(c, i) = D[i * kR + t]; }
        for (int t2 = 0; t2 <= t; ++t2) { const double v = T[tri(t2, t)]; g(ko + t2, c) = v; g(c, ko + t2) = v; }
    }
    // Aq = Q'P, Gp = P'P
    double Aq[kMaxK][kS], Gp[kS][kS];
    for (int q = 0; q < s; ++q) {
        for (int i = 0; i < ko; ++i) Aq[i][q] = D[i * kR + u + q];
        for (int t = 0; t < u; ++t) Aq[ko + t][q] = T[tri(t, u + q)];
        for (int q2 = 0; q2 <= q; ++q2) { Gp[q2][q] = T[tri(u + q2, u + q)]; Gp[q][q2] = Gp[q2][q]; }
    }
    // C = G^-1 Aq through the Cholesky factor of G (near the identity)
    static thread_local double L[kMaxK][kMaxK];
    for (int a = 0; a < k; ++a) {
        for (int b = 0; b <= a; ++b) {
            double v = g(a, b);
            for (int c = 0; c < b; ++c) v -= L[a][c] * L[b][c];
            if (a == b) {
                if (!(v > 0.0)) return 1;
                L[a][a] = std::sqrt(v);
            } else {
                L[a][b] = v / L[b][b];
            }
        }
    }
    double C[kMaxK][kS];
    for (int q = 0; q < s; ++q) {
        double y[kMaxK];
        for (int a = 0; a < k; ++a) {
            double v = Aq[a][q];
            for (int c = 0; c < a; ++c) v -= L[a][c] * y[c];
            y[a] = v / L[a][a];
        }
        for (int a = k - 1; a >= 0; --a) {
            double v = y[a];
            for (int c = a + 1; c < k; ++c) v -= L[c][a] * C[c][q];
            C[a][q] = v / L[a][a];
        }
    }
    // S = Gp - C'Aq, R = chol(S) upper
    double S[kS][kS], R[kS][kS] = {{0.0}};
    for (int a = 0; a < s; ++a)
        for (int b = a; b < s; ++b) {
            double v = Gp[a][b];
            for (int i = 0; i < k; ++i) v -= 0.5 * (C[i][a] * Aq[i][b] + C[i][b] * Aq[i][a]);
            S[a][b] = v; S[b][a] = v;
        }
    for (int a = 0; a < s; ++a) {
        for (int b = a; b < s; ++b) {
            double v = S[a][b];
            for (int c = 0; c < a; ++c) v -= R[c][a] * R[c][b];
            if (a == b) {
                if (!(v > kPivotTol * Gp[a][a])) {          // truncate the block here
                    if (a == 0) return 1;
                    s = a;
                    break;
                }
                R[a][a] = std::sqrt(v);
                ratio = v / Gp[a][a];
            } else {
                R[a][b] = v / R[a][a];
            }
        }
    }
    *s_eff = s;
    if (last_ratio) *last_ratio = ratio;     // pivot ratio of the last accepted column (the caller sizes its next block by it)
    // Tm = R^-1 (upper), Cm = -C R^-1
    double Ri[kS][kS] = {{0.0}};
    for (int c = 0; c < s; ++c) {
        Ri[c][c] = 1.0 / R[c][c];
        for (int r = c - 1; r >= 0; --r) {
            double v = 0.0;
            for (int t = r + 1; t <= c; ++t) v -= R[r][t] * Ri[t][c];
            Ri[r][c] = v / R[r][r];
        }
    }
    for (int r = 0; r < kS; ++r)
        for (int q = 0; q < kS; ++q) Tm[r * kS + q] = (r < s && q < s) ? Ri[r][q] : 0.0;
    for (int i = 0; i < k; ++i)
        for (int q = 0; q < kS; ++q) {
            double v = 0.0;
            if (q < s)
                for (int t = 0; t <= q; ++t) v -= C[i][t] * Ri[t][q];
            Cm[i * kS + q] = v;
        }
    // Hessenberg columns j .. j + s - 1 (k + s rows): H_new U = Pc - [H_old; 0] Bc[:j, :]
    auto h = [&](int a, int b) -> double& { return Hraw[(size_t)a + (size_t)b * ldh]; };
    auto Pc = [&](int a, int q) -> double { return a < k ? C[a][q] : (a - k <= q ? R[a - k][q] : 0.0); };     // (k + s) x s
    auto Bc = [&](int a, int q) -> double { return q == 0 ? (a == j ? 1.0 : 0.0) : Pc(a, q - 1); };
    double rhs[kMaxK + kS][kS];
    for (int q = 0; q < s; ++q)
        for (int a = 0; a < k + s; ++a) {
            double v = Pc(a, q) + (theta ? theta[q] * Bc(a, q) : 0.0);
            if (a < k)
                for (int c = (a > 0 ? a - 1 : 0); c < j; ++c) v -= h(a, c) * Bc(c, q);       // H_old is upper Hessenberg
            rhs[a][q] = v;
        }
    auto U = [&](int r, int q) -> double { return r == 0 ? Bc(j, q) : Bc(k + r - 1, q); };
    for (int q = 0; q < s; ++q) {
        const double d = U(q, q);
        for (int a = 0; a < k + s; ++a) {
            double v = rhs[a][q];
            for (int q2 = 0; q2 < q; ++q2) v -= h(a, j + q2) * U(q2, q);
            h(a, j + q) = v / d;
        }
        // exact zeros below the subdiagonal (they are rounding noise of the reconstruction)
        for (int a = j + q + 2; a < k + s; ++a) h(a, j + q) = 0.0;
    }
    return 0;
}

// Solution update through a block whose update pass has not run (solver.hip: PendingBlock).  The basis is [Q (k0 vectors), Q_new]
// with Q_new = Q Cm + P Tm still unformed (slots k0 .. hold the raw P); the solution takes coefficients yk[0 .. k0 + c) with c <=
// s_eff of the block's vectors:  sum_i yk[i] q_i + sum_q yk[k0 + q] qnew_q  =  sum_i cf[i] q_i + sum_r cf[k0 + r] p_r,
//   cf[i] = yk[i] + sum_{q < c} Cm(i, q) yk[k0 + q],   cf[k0 + r] = sum_{r <= q < c} Tm(r, q) yk[k0 + q]     (Tm upper triangular:
// the first c columns of Q_new only involve p_0 .. p_{c-1}).  ONE multiaxpy over [Q, P] replaces the update pass + the multiaxpy.
inline void fold_solution_coefficients(int k0, int c, const double* Cm, const double* Tm, const double* yk, double* cf) {
    for (int i = 0; i < k0; ++i) {
        double v = yk[i];
        for (int q = 0; q < c; ++q) v += Cm[i * kS + q] * yk[k0 + q];
        cf[i] = v;
    }
    for (int r = 0; r < c; ++r) {
        double v = 0.0;
        for (int q = r; q < c; ++q) v += Tm[r * kS + q] * yk[k0 + q];
        cf[k0 + r] = v;
    }
}

}  // namespace sstep
}  // namespace bk
