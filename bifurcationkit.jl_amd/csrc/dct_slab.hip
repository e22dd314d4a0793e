// Communication-light distributed z-solve of the spectral preconditioner (multi-GPU, z-slabs).
//
// After the local x / y DCTs every (kx, ky) line is the system M w = f in z, M = (c I + D)^2 + s I with
// c = 1 + lam_x + lam_y and D the Neumann-ghost second difference over ALL nz planes.  Instead of transposing the array to
// y-slabs (two all-to-alls of the whole array per application), each rank keeps its z-slab and uses
//     M = B + U G U',      B = blockdiag_r((c I + D_nl)^2 + s I)   (Neumann-ghost closure at every slab face),
// where B^-1 is the EXISTING fused z pass run on the slab (local DCT-II of length nl = nz / R, symbol, inverse) and the
// difference is a rank-2 term per face p | q:  w = e_p - e_q,  u = a e_{p-1} + (c - a)(e_p - e_q) - a e_{q+1},
// G = [[0, -a], [-a, 2 a^2]] (a = 1/h_z^2).  Woodbury:
//     M^-1 f = B^-1 (f - U nu),     (G^-1 + U' B^-1 U) nu = U' B^-1 f,
// a block-tridiagonal system (2 x 2 blocks, R - 1 block rows) per line whose entries are six nl-term sums of the local DCT
// basis at the planes next to the faces.  Lines are dealt out to the ranks for that small solve, so one application moves
// 4 doubles per line out and 4 back (16 MiB per rank at 512^3) instead of 2 x 112 MiB -- derivation, stability study and the
// NumPy restatement: oracle/slab_zsolve.py (tests/test_oracle.py: 1e-11 relative over the whole range of c).
// No reference counterpart (the reference is single-process).
#include <cmath>
#include <vector>

#include "ops.h"

namespace bk {

constexpr int kSlabMaxFaces = 15;      // R <= 16: the block-Thomas recurrences of a line live in registers

struct SlabK {
    int nx, ny, nl, R, rank;
    size_t L, Lr;                      // lines (nx * ny) and lines per owner
    double a, shift;
    const double* lam0;
    const double* lam1;
    const double* lam_loc;             // [nl] eigenvalues of the slab-local Neumann second difference
    const double* phi;                 // [2][nl] local DCT-II basis at planes 0 and 1
};

namespace {

// partial face data of this rank: [4][Lr] per line owner -- (u'y, w'y) of the bottom face (this rank is the q side) and of
// the top face (the p side)
__global__ void __launch_bounds__(256) slab_face_gather(SlabK P, const double* __restrict__ y, double* __restrict__ sbuf) {
    const size_t line = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (line >= P.L) return;
    const int ix = (int)(line % P.nx), iy = (int)(line / P.nx);
    const double c = 1.0 + P.lam0[ix] + P.lam1[iy];
    const size_t pl = P.L;
    const double y0 = y[line], y1 = y[pl + line];
    const double yt1 = y[(size_t)(P.nl - 2) * pl + line], yt0 = y[(size_t)(P.nl - 1) * pl + line];
    const size_t d = line / P.Lr, l = line - d * P.Lr;
    double* o = sbuf + d * 4 * P.Lr + l;
    const bool hasb = P.rank > 0, hast = P.rank < P.R - 1;
    o[0] = hasb ? -(c - P.a) * y0 - P.a * y1 : 0.0;
    o[P.Lr] = hasb ? -y0 : 0.0;
    o[2 * P.Lr] = hast ? P.a * yt1 + (c - P.a) * yt0 : 0.0;
    o[3 * P.Lr] = hast ? yt0 : 0.0;
}

struct M2 { double a, b, c, d; };      // [[a, b], [c, d]]
__device__ __forceinline__ M2 inv2(M2 m) {
    const double r = 1.0 / (m.a * m.d - m.b * m.c);
    return {m.d * r, -m.b * r, -m.c * r, m.a * r};
}
__device__ __forceinline__ M2 mul2(M2 x, M2 y) {
    return {x.a * y.a + x.b * y.c, x.a * y.b + x.b * y.d, x.c * y.a + x.d * y.c, x.c * y.b + x.d * y.d};
}
__device__ __forceinline__ M2 tr2(M2 m) { return {m.a, m.c, m.b, m.d}; }

// the capacitance system of the lines this rank owns: rbuf holds [src][4][Lr]; out [dst][4][Lr] = (nu of dst's bottom face,
// nu of dst's top face)
__global__ void __launch_bounds__(256) slab_reduced_solve(SlabK P, const double* __restrict__ rbuf, double* __restrict__ out) {
    const size_t l = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (l >= P.Lr) return;
    const size_t line = (size_t)P.rank * P.Lr + l;
    const int ix = (int)(line % P.nx), iy = (int)(line / P.nx);
    const double c = 1.0 + P.lam0[ix] + P.lam1[iy];
    const double a = P.a;
    // face block of B_r^-1: A[z][z'] = sum_m phi_m(z) phi_m(z') sym_m,  C: the same with (-1)^m   (z, z' in {0, 1})
    double A00 = 0, A01 = 0, A11 = 0, C00 = 0, C01 = 0, C11 = 0;
    for (int m = 0; m < P.nl; ++m) {
        const double t = c + P.lam_loc[m];
        const double sym = 1.0 / (t * t + P.shift);
        const double p0 = P.phi[m], p1 = P.phi[P.nl + m];
        const double q00 = p0 * p0 * sym, q01 = p0 * p1 * sym, q11 = p1 * p1 * sym;
        A00 += q00; A01 += q01; A11 += q11;
        if (m & 1) { C00 -= q00; C01 -= q01; C11 -= q11; }
        else { C00 += q00; C01 += q01; C11 += q11; }
    }
    // P_top (rows: planes nl-2, nl-1; columns u, w) = [[a, 0], [c - a, 1]];  P_bot (rows: planes 0, 1) = [[-(c-a), -1], [-a, 0]]
    const double ca = c - a;
    // P_top' G_tt P_top with G_tt = [[A11, A01], [A01, A00]]
    const double t_uu = a * (A11 * a + A01 * ca) + ca * (A01 * a + A00 * ca);
    const double t_uw = a * A01 + ca * A00;
    const double t_ww = A00;
    // P_bot' G_bb P_bot with G_bb = [[A00, A01], [A01, A11]]
    const double b_uu = ca * (A00 * ca + A01 * a) + a * (A01 * ca + A11 * a);
    const double b_uw = ca * A00 + a * A01;
    const double b_ww = A00;
    const M2 D0 = {-2.0 + t_uu + b_uu, -1.0 / a + t_uw + b_uw, -1.0 / a + t_uw + b_uw, t_ww + b_ww};
    // E = P_bot' G_bt P_top with G_bt (rows planes 0, 1; columns planes nl-2, nl-1) = [[C01, C00], [C11, C01]]
    const double r0u = C01 * a + C00 * ca, r0w = C00;        // (G_bt P_top) row of plane 0
    const double r1u = C11 * a + C01 * ca, r1w = C01;        // row of plane 1
    const M2 E = {-ca * r0u - a * r1u, -ca * r0w - a * r1w, -r0u, -r0w};
    const M2 Et = tr2(E);
    const int nf = P.R - 1;
    M2 Dk[kSlabMaxFaces];
    double gu[kSlabMaxFaces], gw[kSlabMaxFaces];
    for (int i = 0; i < nf; ++i) {
        // g_i = top part from rank i + bottom part from rank i + 1
        const double* top = rbuf + (size_t)i * 4 * P.Lr + l;
        const double* bot = rbuf + (size_t)(i + 1) * 4 * P.Lr + l;
        double g0 = top[2 * P.Lr] + bot[0], g1 = top[3 * P.Lr] + bot[P.Lr];
        if (i == 0) {
            Dk[0] = D0;
        } else {
            const M2 W = mul2(Et, inv2(Dk[i - 1]));
            const M2 WE = mul2(W, E);
            Dk[i] = {D0.a - WE.a, D0.b - WE.b, D0.c - WE.c, D0.d - WE.d};
            g0 -= W.a * gu[i - 1] + W.b * gw[i - 1];
            g1 -= W.c * gu[i - 1] + W.d * gw[i - 1];
        }
        gu[i] = g0; gw[i] = g1;
    }
    double nu_u = 0.0, nu_w = 0.0;                            // nu_{i+1} during the back substitution
    for (int i = nf - 1; i >= 0; --i) {
        double r0 = gu[i], r1 = gw[i];
        if (i < nf - 1) { r0 -= E.a * nu_u + E.b * nu_w; r1 -= E.c * nu_u + E.d * nu_w; }
        const M2 Di = inv2(Dk[i]);
        nu_u = Di.a * r0 + Di.b * r1;
        nu_w = Di.c * r0 + Di.d * r1;
        // face i is the top face of rank i and the bottom face of rank i + 1
        double* ot = out + (size_t)i * 4 * P.Lr + l;
        double* ob = out + (size_t)(i + 1) * 4 * P.Lr + l;
        ot[2 * P.Lr] = nu_u; ot[3 * P.Lr] = nu_w;
        ob[0] = nu_u; ob[P.Lr] = nu_w;
    }
    // unused slots (rank 0 has no bottom face, rank R-1 no top face)
    out[l] = 0.0; out[P.Lr + l] = 0.0;
    double* last = out + (size_t)(P.R - 1) * 4 * P.Lr + l;
    last[2 * P.Lr] = 0.0; last[3 * P.Lr] = 0.0;
}

// f <- f - U nu on the four planes next to this rank's faces; rbuf = [owner][4][Lr]
__global__ void __launch_bounds__(256) slab_face_correct(SlabK P, const double* __restrict__ rbuf, double* __restrict__ f) {
    const size_t line = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (line >= P.L) return;
    const int ix = (int)(line % P.nx), iy = (int)(line / P.nx);
    const double c = 1.0 + P.lam0[ix] + P.lam1[iy];
    const size_t d = line / P.Lr, l = line - d * P.Lr;
    const double* nu = rbuf + d * 4 * P.Lr + l;
    const size_t pl = P.L;
    if (P.rank > 0) {                                         // bottom face: this rank is the q side, planes 0, 1
        const double nu_u = nu[0], nu_w = nu[P.Lr];
        f[line] += (c - P.a) * nu_u + nu_w;
        f[pl + line] += P.a * nu_u;
    }
    if (P.rank < P.R - 1) {                                   // top face: the p side, planes nl-2, nl-1
        const double nu_u = nu[2 * P.Lr], nu_w = nu[3 * P.Lr];
        f[(size_t)(P.nl - 2) * pl + line] -= P.a * nu_u;
        f[(size_t)(P.nl - 1) * pl + line] -= (c - P.a) * nu_u + nu_w;
    }
}

}  // namespace

int slab_faces_gather(bk_ctx* ctx, const SlabK& P, const double* y, double* sbuf) {
    hipLaunchKernelGGL(slab_face_gather, dim3((unsigned)((P.L + 255) / 256)), dim3(256), 0, ctx->stream, P, y, sbuf);
    BK_HIP(ctx, hipGetLastError());
    return 0;
}
int slab_faces_solve(bk_ctx* ctx, const SlabK& P, const double* rbuf, double* out) {
    hipLaunchKernelGGL(slab_reduced_solve, dim3((unsigned)((P.Lr + 255) / 256)), dim3(256), 0, ctx->stream, P, rbuf, out);
    BK_HIP(ctx, hipGetLastError());
    return 0;
}
int slab_faces_correct(bk_ctx* ctx, const SlabK& P, const double* rbuf, double* f) {
    hipLaunchKernelGGL(slab_face_correct, dim3((unsigned)((P.L + 255) / 256)), dim3(256), 0, ctx->stream, P, rbuf, f);
    BK_HIP(ctx, hipGetLastError());
    return 0;
}

}  // namespace bk
