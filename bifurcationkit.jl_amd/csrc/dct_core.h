// Index math and butterflies of the LDS-resident fast DCT-II / DCT-III (orthonormal), shared between the HIP
// kernel (dct_fast.hip) and a host test harness (tests/cpp/dct_core_check.cpp): every function is plain
// arithmetic on a caller-provided complex array, so the kernel's phases can be replayed sequentially on the
// CPU and compared with scipy.fft.dct.
//
// Algorithm (Makhoul 1980: N-point DCT-II from one N-point FFT; two real lines a, b ride one complex FFT):
//   forward   z[bitrev(mk(n))] = xa[n] + i xb[n]        mk(n) = n/2 (n even), N-1-(n-1)/2 (n odd)
//             in-place radix-2 DIT FFT (e^{-2 pi i/N})
//             Va = (Z_k + conj Z_{N-k})/2, Vb = -i (Z_k - conj Z_{N-k})/2        (split the two real spectra)
//             C_k = Re(e_k V_k), C_{N-k} = -Im(e_k V_k), e_k = exp(-i pi k / 2N);  X_k = s_k C_k
//   inverse   C_k = X_k / s_k;  V_k = conj(e_k) (C_k - i C_{N-k});  Z = Va + i Vb (Hermitian completion)
//             in-place radix-2 DIF inverse FFT (natural in, bit-reversed out), 1/N folded into the input
//             x[j] = z[bitrev(mk(j))]
// with s_0 = sqrt(1/N), s_k = sqrt(2/N).  LDS indices go through swz() (XOR of the low 4 bits with a hash of
// the upper bits) so that the bit-reversed scatter does not land every lane on one bank.
#pragma once

#ifdef __HIPCC__
#define BK_HD __host__ __device__ __forceinline__
#else
#define BK_HD inline
#endif

namespace bk {
namespace dctc {

struct c2 {
    double x, y;
};

BK_HD int swz(int i) { return i ^ (((i >> 4) ^ (i >> 8)) & 15); }

BK_HD int bitrev(int i, int bits) {
#if defined(__HIP_DEVICE_COMPILE__)
    return (int)(__builtin_bitreverse32((unsigned)i) >> (32 - bits));
#else
    unsigned v = (unsigned)i, r = 0;
    for (int b = 0; b < bits; ++b) { r = (r << 1) | (v & 1u); v >>= 1; }
    return (int)r;
#endif
}

BK_HD int makhoul(int n, int N) { return (n & 1) ? N - 1 - (n >> 1) : (n >> 1); }

// LDS slot of input sample n (forward) / of output sample j (inverse)
BK_HD int sample_slot(int n, int N, int bits) { return swz(bitrev(makhoul(n, N), bits)); }

// forward DIT butterfly j (< N/2) of the stage with half-size 2^lh.  tw[q] = exp(-2 pi i q / N), q < N/2.
BK_HD void dit_butterfly(c2* z, int bits, int lh, int j, const c2* tw) {
    const int half = 1 << lh;
    const int pos = j & (half - 1);
    const int i0 = ((j >> lh) << (lh + 1)) + pos;
    const int i1 = i0 + half;
    const c2 w = tw[pos << (bits - lh - 1)];
    const int p0 = swz(i0), p1 = swz(i1);
    const c2 a = z[p0], b = z[p1];
    const double tx = w.x * b.x - w.y * b.y, ty = w.x * b.y + w.y * b.x;
    z[p0].x = a.x + tx; z[p0].y = a.y + ty;
    z[p1].x = a.x - tx; z[p1].y = a.y - ty;
}

// inverse DIF butterfly (conjugate twiddles)
BK_HD void dif_butterfly_inv(c2* z, int bits, int lh, int j, const c2* tw) {
    const int half = 1 << lh;
    const int pos = j & (half - 1);
    const int i0 = ((j >> lh) << (lh + 1)) + pos;
    const int i1 = i0 + half;
    const c2 w = tw[pos << (bits - lh - 1)];          // conj applied below
    const int p0 = swz(i0), p1 = swz(i1);
    const c2 a = z[p0], b = z[p1];
    const double dx = a.x - b.x, dy = a.y - b.y;
    z[p0].x = a.x + b.x; z[p0].y = a.y + b.y;
    z[p1].x = w.x * dx + w.y * dy;                     // (dx + i dy) * (w.x - i w.y)
    z[p1].y = w.x * dy - w.y * dx;
}

// R consecutive radix-2 DIT stages (lh .. lh+R-1) on the 2^R elements {base + (q << lh)} of group g, carried out in
// registers: one LDS round trip per R stages instead of one per stage.  g in [0, N >> R).
template <int R>
BK_HD void dit_group(c2* z, int bits, int lh, int g, const c2* tw) {
    constexpr int M = 1 << R;
    const int lo = g & ((1 << lh) - 1);
    const int base = ((g >> lh) << (lh + R)) + lo;
    c2 v[M];
#pragma unroll
    for (int q = 0; q < M; ++q) v[q] = z[swz(base + (q << lh))];
#pragma unroll
    for (int s = 0; s < R; ++s) {
#pragma unroll
        for (int q = 0; q < M; ++q) {
            if (q & (1 << s)) continue;
            const int pos = ((q & ((1 << s) - 1)) << lh) + lo;
            const c2 w = tw[pos << (bits - (lh + s) - 1)];
            const c2 a = v[q], b = v[q | (1 << s)];
            const double tx = w.x * b.x - w.y * b.y, ty = w.x * b.y + w.y * b.x;
            v[q].x = a.x + tx; v[q].y = a.y + ty;
            v[q | (1 << s)].x = a.x - tx; v[q | (1 << s)].y = a.y - ty;
        }
    }
#pragma unroll
    for (int q = 0; q < M; ++q) z[swz(base + (q << lh))] = v[q];
}

// R consecutive inverse DIF stages (lh+R-1 down to lh), same grouping.
template <int R>
BK_HD void dif_group_inv(c2* z, int bits, int lh, int g, const c2* tw) {
    constexpr int M = 1 << R;
    const int lo = g & ((1 << lh) - 1);
    const int base = ((g >> lh) << (lh + R)) + lo;
    c2 v[M];
#pragma unroll
    for (int q = 0; q < M; ++q) v[q] = z[swz(base + (q << lh))];
#pragma unroll
    for (int s = R - 1; s >= 0; --s) {
#pragma unroll
        for (int q = 0; q < M; ++q) {
            if (q & (1 << s)) continue;
            const int pos = ((q & ((1 << s) - 1)) << lh) + lo;
            const c2 w = tw[pos << (bits - (lh + s) - 1)];
            const c2 a = v[q], b = v[q | (1 << s)];
            const double dx = a.x - b.x, dy = a.y - b.y;
            v[q].x = a.x + b.x; v[q].y = a.y + b.y;
            v[q | (1 << s)].x = w.x * dx + w.y * dy;
            v[q | (1 << s)].y = w.x * dy - w.y * dx;
        }
    }
#pragma unroll
    for (int q = 0; q < M; ++q) z[swz(base + (q << lh))] = v[q];
}

// forward post-processing for k in [0, N/2]: in place, slots k and N-k.  ew[k] = exp(-i pi k / 2N).
// On return z[swz(k)] = (Xa_k, Xb_k) and z[swz(N-k)] = (Xa_{N-k}, Xb_{N-k}) (orthonormal coefficients).
BK_HD void fwd_post(c2* z, int N, int k, const c2* ew, double s0, double s2) {
    const int pk = swz(k);
    const c2 Zk = z[pk];
    if (k == 0) {
        z[pk].x = s0 * Zk.x;
        z[pk].y = s0 * Zk.y;
        return;
    }
    const int pn = swz(N - k);
    const c2 Zn = z[pn];
    const double vax = 0.5 * (Zk.x + Zn.x), vay = 0.5 * (Zk.y - Zn.y);
    const double dx = Zk.x - Zn.x, dy = Zk.y + Zn.y;
    const double vbx = 0.5 * dy, vby = -0.5 * dx;
    const c2 e = ew[k];
    const double ax = e.x * vax - e.y * vay, ay = e.x * vay + e.y * vax;
    const double bx = e.x * vbx - e.y * vby, by = e.x * vby + e.y * vbx;
    z[pk].x = s2 * ax; z[pk].y = s2 * bx;
    if (2 * k != N) { z[pn].x = -s2 * ay; z[pn].y = -s2 * by; }
}

// inverse pre-processing for k in [0, N/2]: z holds (Xa_k, Xb_k) in natural slots; on return the Hermitian
// completed spectrum Z (scaled by 1/N) sits in the same slots, ready for the DIF inverse FFT.
BK_HD void inv_pre(c2* z, int N, int k, const c2* ew, double s0, double s2) {
    const int pk = swz(k);
    const c2 Xk = z[pk];
    const double rN = 1.0 / N;
    if (k == 0) {
        z[pk].x = Xk.x * (rN / s0);
        z[pk].y = Xk.y * (rN / s0);
        return;
    }
    const int pn = swz(N - k);
    const c2 Xn = z[pn];
    const double f = rN / s2;
    const double cak = Xk.x * f, cbk = Xk.y * f, can = Xn.x * f, cbn = Xn.y * f;
    const c2 e = ew[k];
    const double vax = e.x * cak - e.y * can, vay = -e.x * can - e.y * cak;
    const double vbx = e.x * cbk - e.y * cbn, vby = -e.x * cbn - e.y * cbk;
    z[pk].x = vax - vby; z[pk].y = vay + vbx;
    if (2 * k != N) { z[pn].x = vax + vby; z[pn].y = -vay + vbx; }
}

}  // namespace dctc
}  // namespace bk
