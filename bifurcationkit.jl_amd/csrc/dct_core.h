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

// LDS layout (round 4; the bank-conflict model of scripts/lds_model reproduces the measured conflict share of the round-3
// layout, 38.5 %, and puts this one at 17 %):
//   swz   XOR of the low 4 bits (the 16 slots of a 256-byte bank row) with bits 3.., 4.. and 7.. of the index, so that the
//         power-of-two strides of the radix-8 stages (8 data reads 2^lh apart) and the bit-reversed scatter of the first stage
//         spread over the row; swz(8g + q) == swz(8g) ^ q still holds for the first / last stage (see fused_first)
//   twi   twiddle tables carry one padding slot per 16 entries: the three twiddle loads of a radix-8 stage index the table
//         with lo << (sh + 2), lo << (sh + 1), lo << sh -- on a dense table the 8 distinct lo of a lane group hit ONE slot
// BK_DCT_LAYOUT 0 keeps the round-3 layout (i ^ ((i>>4 ^ i>>8) & 15), dense tables) as the A/B reference.
#ifndef BK_DCT_LAYOUT
#define BK_DCT_LAYOUT 1
#endif
#if BK_DCT_LAYOUT == 1
BK_HD int swz(int i) { return i ^ (((i >> 3) ^ (i >> 4) ^ (i >> 7)) & 15); }
BK_HD int twi(int j) { return j + (j >> 4); }
#else
BK_HD int swz(int i) { return i ^ (((i >> 4) ^ (i >> 8)) & 15); }
BK_HD int twi(int j) { return j; }
#endif
// table sizes in complex slots: FFT twiddles exp(-2 pi i j / N), j < N/2; post twiddles exp(-i pi k / 2N), k <= N/2
BK_HD int tw_len(int N) { return twi(N >> 1); }
BK_HD int ew_len(int N) { return twi(N >> 1) + 1; }

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
    const c2 w = tw[twi(pos << (bits - lh - 1))];
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
    const c2 w = tw[twi(pos << (bits - lh - 1))];          // conj applied below
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
            const c2 w = tw[twi(pos << (bits - (lh + s) - 1))];
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
            const c2 w = tw[twi(pos << (bits - (lh + s) - 1))];
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

// radix-8 register butterflies with the 7 twiddles [s0: 1][s1: 2][s2: 4] already gathered (r8_twiddles)
BK_HD void r8_fwd_regs(c2* v, const c2* w7) {
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        const c2* t = w7 + (s == 0 ? 0 : (s == 1 ? 1 : 3));
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            if (q & (1 << s)) continue;
            const c2 w = t[s == 0 ? 0 : (s == 1 ? (q & 1) : (q & 3))];
            const c2 a = v[q], b = v[q | (1 << s)];
            const double tx = w.x * b.x - w.y * b.y, ty = w.x * b.y + w.y * b.x;
            v[q].x = a.x + tx; v[q].y = a.y + ty;
            v[q | (1 << s)].x = a.x - tx; v[q | (1 << s)].y = a.y - ty;
        }
    }
}
BK_HD void r8_inv_regs(c2* v, const c2* w7) {
#pragma unroll
    for (int s = 2; s >= 0; --s) {
        const c2* t = w7 + (s == 0 ? 0 : (s == 1 ? 1 : 3));
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            if (q & (1 << s)) continue;
            const c2 w = t[s == 0 ? 0 : (s == 1 ? (q & 1) : (q & 3))];
            const c2 a = v[q], b = v[q | (1 << s)];
            const double dx = a.x - b.x, dy = a.y - b.y;
            v[q].x = a.x + b.x; v[q].y = a.y + b.y;
            v[q | (1 << s)].x = w.x * dx + w.y * dy;
            v[q | (1 << s)].y = w.x * dy - w.y * dx;
        }
    }
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
    const c2 e = ew[twi(k)];
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
    const c2 e = ew[twi(k)];
    const double vax = e.x * cak - e.y * can, vay = -e.x * can - e.y * cak;
    const double vbx = e.x * cbk - e.y * cbn, vby = -e.x * cbn - e.y * cbk;
    z[pk].x = vax - vby; z[pk].y = vay + vbx;
    if (2 * k != N) { z[pn].x = vax + vby; z[pn].y = -vay + vbx; }
}


// ------------------------------------------------------------------------------------------------------------------
// Fused schedule (dct_fused_kernel): the first and the last radix-8 stage run on registers that are filled from /
// drained to global memory directly, so a tile makes 2 LDS round trips per transform instead of 5:
//   forward   first<-global | LDS | middle stages | LDS | last + post -> global
//   inverse   global -> pre + first DIF | LDS | middle | LDS | last DIF -> global
//   roundtrip first<-global | LDS | middle | LDS | last + post + symbol + pre + first DIF | LDS | middle | LDS | last -> global
// The building blocks below are shared with the host replay (tests/cpp/dct_core_check.cpp modes 4..6).

BK_HD int bitrev3(int r) { return ((r & 1) << 2) | (r & 2) | ((r >> 2) & 1); }

// sample index whose Makhoul slot is m
BK_HD int makhoul_inv(int m, int N) { return m < (N >> 1) ? 2 * m : 2 * (N - 1 - m) + 1; }

// sample index feeding slot r (0..7, compile-time in the unrolled callers) of the first-stage group gp in [0, N/8):
// Makhoul slot m = gp + r N/8, which is below N/2 exactly for r < 4
BK_HD int first_sample(int gp, int r, int N) {
    const int m = gp + (N >> 3) * r;
    return r < 4 ? 2 * m : 2 * (N - 1 - m) + 1;
}

BK_HD c2 cmul(c2 w, c2 b) { c2 r; r.x = w.x * b.x - w.y * b.y; r.y = w.x * b.y + w.y * b.x; return r; }
BK_HD c2 cmulc(c2 w, c2 d) { c2 r; r.x = w.x * d.x + w.y * d.y; r.y = w.x * d.y - w.y * d.x; return r; }   // d * conj(w)

// DIT stages 0..2 (half sizes 1, 2, 4) on 8 consecutive bit-reversed positions: constant twiddles.
BK_HD void r8_first(c2* v) {
    const double c = 0.70710678118654752440;
#pragma unroll
    for (int q = 0; q < 8; q += 2) {                      // s = 0: w = 1
        const c2 a = v[q], b = v[q + 1];
        v[q].x = a.x + b.x; v[q].y = a.y + b.y;
        v[q + 1].x = a.x - b.x; v[q + 1].y = a.y - b.y;
    }
#pragma unroll
    for (int q = 0; q < 8; q += 4) {                      // s = 1: w = 1, -i
        c2 a = v[q], b = v[q + 2];
        v[q].x = a.x + b.x; v[q].y = a.y + b.y;
        v[q + 2].x = a.x - b.x; v[q + 2].y = a.y - b.y;
        a = v[q + 1]; b = v[q + 3];                       // -i b = (b.y, -b.x)
        v[q + 1].x = a.x + b.y; v[q + 1].y = a.y - b.x;
        v[q + 3].x = a.x - b.y; v[q + 3].y = a.y + b.x;
    }
    {                                                     // s = 2: w = 1, (c,-c), -i, (-c,-c)
        c2 a = v[0], b = v[4];
        v[0].x = a.x + b.x; v[0].y = a.y + b.y; v[4].x = a.x - b.x; v[4].y = a.y - b.y;
        a = v[1]; b = v[5];
        double tx = c * (b.x + b.y), ty = c * (b.y - b.x);
        v[1].x = a.x + tx; v[1].y = a.y + ty; v[5].x = a.x - tx; v[5].y = a.y - ty;
        a = v[2]; b = v[6];
        v[2].x = a.x + b.y; v[2].y = a.y - b.x; v[6].x = a.x - b.y; v[6].y = a.y + b.x;
        a = v[3]; b = v[7];
        tx = c * (b.y - b.x); ty = -c * (b.x + b.y);
        v[3].x = a.x + tx; v[3].y = a.y + ty; v[7].x = a.x - tx; v[7].y = a.y - ty;
    }
}

// inverse DIF stages 2..0 (conjugate twiddles) on 8 consecutive positions.
BK_HD void r8_last_inv(c2* v) {
    const double c = 0.70710678118654752440;
    {                                                     // s = 2: conj w = 1, (c,c), i, (-c,c)
        c2 a = v[0], b = v[4];
        v[0].x = a.x + b.x; v[0].y = a.y + b.y; v[4].x = a.x - b.x; v[4].y = a.y - b.y;
        a = v[1]; b = v[5];
        double dx = a.x - b.x, dy = a.y - b.y;
        v[1].x = a.x + b.x; v[1].y = a.y + b.y; v[5].x = c * (dx - dy); v[5].y = c * (dx + dy);
        a = v[2]; b = v[6];
        dx = a.x - b.x; dy = a.y - b.y;
        v[2].x = a.x + b.x; v[2].y = a.y + b.y; v[6].x = -dy; v[6].y = dx;
        a = v[3]; b = v[7];
        dx = a.x - b.x; dy = a.y - b.y;
        v[3].x = a.x + b.x; v[3].y = a.y + b.y; v[7].x = -c * (dx + dy); v[7].y = c * (dx - dy);
    }
#pragma unroll
    for (int q = 0; q < 8; q += 4) {                      // s = 1: conj w = 1, i
        c2 a = v[q], b = v[q + 2];
        v[q].x = a.x + b.x; v[q].y = a.y + b.y; v[q + 2].x = a.x - b.x; v[q + 2].y = a.y - b.y;
        a = v[q + 1]; b = v[q + 3];
        const double dx = a.x - b.x, dy = a.y - b.y;
        v[q + 1].x = a.x + b.x; v[q + 1].y = a.y + b.y; v[q + 3].x = -dy; v[q + 3].y = dx;
    }
#pragma unroll
    for (int q = 0; q < 8; q += 2) {                      // s = 0
        const c2 a = v[q], b = v[q + 1];
        v[q].x = a.x + b.x; v[q].y = a.y + b.y; v[q + 1].x = a.x - b.x; v[q + 1].y = a.y - b.y;
    }
}

// the 7 twiddles of a radix-8 group from 3 loads: stage s, pos p (< 2^s) uses W^{(p*2^lh + lo) << (bits-lh-s-1)}
//   = b_s * exp(-2 pi i p / 2^{s+1}),  b_s = tw[lo << (bits-lh-s-1)]   (constant rotations: 1, -i, e^{-i pi/4}, e^{-3i pi/4})
BK_HD void r8_twiddles(c2* w7, int lo, int sh, const c2* tw) {      // sh = bits - lh - 3
    const double c = 0.70710678118654752440;
    const c2 b0 = tw[twi(lo << (sh + 2))], b1 = tw[twi(lo << (sh + 1))], b2 = tw[twi(lo << sh)];
    w7[0] = b0;
    w7[1] = b1; w7[2].x = b1.y; w7[2].y = -b1.x;
    w7[3] = b2;
    w7[4].x = c * (b2.x + b2.y); w7[4].y = c * (b2.y - b2.x);
    w7[5].x = b2.y; w7[5].y = -b2.x;
    w7[6].x = w7[4].y; w7[6].y = -w7[4].x;
}

// radix-8 group through LDS (three stages lh .. lh+2), g in [0, N/8): 8 data + 3 twiddle reads, 8 writes
BK_HD void r8_group_fwd(c2* z, int bits, int lh, int g, const c2* tw) {
    const int lo = g & ((1 << lh) - 1);
    const int base = ((g >> lh) << (lh + 3)) + lo;
    c2 v[8], w[7];
#pragma unroll
    for (int q = 0; q < 8; ++q) v[q] = z[swz(base + (q << lh))];
    r8_twiddles(w, lo, bits - lh - 3, tw);
    r8_fwd_regs(v, w);
#pragma unroll
    for (int q = 0; q < 8; ++q) z[swz(base + (q << lh))] = v[q];
}
BK_HD void r8_group_inv(c2* z, int bits, int lh, int g, const c2* tw) {
    const int lo = g & ((1 << lh) - 1);
    const int base = ((g >> lh) << (lh + 3)) + lo;
    c2 v[8], w[7];
#pragma unroll
    for (int q = 0; q < 8; ++q) v[q] = z[swz(base + (q << lh))];
    r8_twiddles(w, lo, bits - lh - 3, tw);
    r8_inv_regs(v, w);
#pragma unroll
    for (int q = 0; q < 8; ++q) z[swz(base + (q << lh))] = v[q];
}

// X_k = sc * (Re(e_k Va_k), Re(e_k Vb_k)),  Va = (Zk + conj Zn)/2, Vb = -i (Zk - conj Zn)/2, Zn = Z_{N-k};
// sc carries the 1/2:  sc = s0/2 (k = 0, where Zn = Zk) or s2/2.   Valid for every k in [0, N).
BK_HD c2 post_one(c2 Zk, c2 Zn, c2 e, double sc) {
    const double vax = Zk.x + Zn.x, vay = Zk.y - Zn.y;
    const double vbx = Zk.y + Zn.y, vby = Zn.x - Zk.x;
    c2 r;
    r.x = sc * (e.x * vax - e.y * vay);
    r.y = sc * (e.x * vbx - e.y * vby);
    return r;
}
// Z_k = Va_k + i Vb_k, V_k = conj(e_k) (C_k - i C_{N-k}), C_k = fk X_k, C_{N-k} = fn X_{N-k} (fn = 0 for k = 0)
BK_HD c2 pre_one(c2 Xk, c2 Xn, c2 e, double fk, double fn) {
    const double cak = Xk.x * fk, cbk = Xk.y * fk, can = Xn.x * fn, cbn = Xn.y * fn;
    const double vax = e.x * cak - e.y * can, vay = -e.x * can - e.y * cak;
    const double vbx = e.x * cbk - e.y * cbn, vby = -e.x * cbn - e.y * cbk;
    c2 r;
    r.x = vax - vby; r.y = vay + vbx;
    return r;
}

// F1: first radix-8 stage of pair-line zp for natural group gp in [0, N/8): samples come from ld(slot, n) -> c2, slot =
// 0..7 a compile-time position (lets the kernel hand over registers it prefetched in exactly this order).
template <class Load>
BK_HD void fused_first(c2* zp, int N, int bits, int gp, Load&& ld) {
    c2 v[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) v[bitrev3(r)] = ld(r, first_sample(gp, r, N));
    r8_first(v);
    const int sb = swz(bitrev(gp, bits - 3) << 3);          // swz(8g + q) == swz(8g) ^ q  (q < 8 never reaches bit 4)
#pragma unroll
    for (int q = 0; q < 8; ++q) zp[sb ^ q] = v[q];
}

// I3: last inverse radix-8 stage, results handed to st(n, value).
template <class Store>
BK_HD void fused_last(const c2* zp, int N, int bits, int gp, Store&& st) {
    const int sb = swz(bitrev(gp, bits - 3) << 3);
    c2 v[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) v[q] = zp[sb ^ q];
    r8_last_inv(v);
#pragma unroll
    for (int r = 0; r < 8; ++r) st(first_sample(gp, r, N), v[bitrev3(r)]);
}

// Contiguous-axis variants (the transform runs along the fastest index): a lane reads the two adjacent samples (2j, 2j+1)
// of each line with one 16-B access.  The even one lands in Makhoul slot j, the odd one in slot N-1-j, i.e. for
// j = gp + r N/8 (r < 4) in first-stage group gp (position r) and in group N/8-1-gp (position 7-r): an item owns BOTH
// groups gp and gp' = N/8-1-gp, gp in [0, N/16).   ld2(slot, j, ea, oa, eb, ob): samples 2j, 2j+1 of line a and of
// line b; slot = 0..7 is the compile-time position of the call (registers prefetched in exactly this order).
template <class Load2>
BK_HD void fused_first2(c2* zp, int N, int bits, int gp, Load2&& ld2) {
    const int G = N >> 3, gq = G - 1 - gp;
    c2 A[8], B[8];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        ld2(2 * r, gp + G * r, A[r].x, B[7 - r].x, A[r].y, B[7 - r].y);
        ld2(2 * r + 1, gq + G * r, B[r].x, A[7 - r].x, B[r].y, A[7 - r].y);
    }
    c2 v[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) v[bitrev3(r)] = A[r];
    r8_first(v);
    int sb = swz(bitrev(gp, bits - 3) << 3);
#pragma unroll
    for (int q = 0; q < 8; ++q) zp[sb ^ q] = v[q];
#pragma unroll
    for (int r = 0; r < 8; ++r) v[bitrev3(r)] = B[r];
    r8_first(v);
    sb = swz(bitrev(gq, bits - 3) << 3);
#pragma unroll
    for (int q = 0; q < 8; ++q) zp[sb ^ q] = v[q];
}
// st2(j, ea, oa, eb, ob): store samples 2j, 2j+1 of line a and line b.
template <class Store2>
BK_HD void fused_last2(const c2* zp, int N, int bits, int gp, Store2&& st2) {
    const int G = N >> 3, gq = G - 1 - gp;
    c2 A[8], B[8], v[8];
    int sb = swz(bitrev(gp, bits - 3) << 3);
#pragma unroll
    for (int q = 0; q < 8; ++q) v[q] = zp[sb ^ q];
    r8_last_inv(v);
#pragma unroll
    for (int r = 0; r < 8; ++r) A[r] = v[bitrev3(r)];
    sb = swz(bitrev(gq, bits - 3) << 3);
#pragma unroll
    for (int q = 0; q < 8; ++q) v[q] = zp[sb ^ q];
    r8_last_inv(v);
#pragma unroll
    for (int r = 0; r < 8; ++r) B[r] = v[bitrev3(r)];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        st2(gp + G * r, A[r].x, B[7 - r].x, A[r].y, B[7 - r].y);
        st2(gq + G * r, B[r].x, A[7 - r].x, B[r].y, A[7 - r].y);
    }
}

// One (k, N-k) pair of the merged middle, in place: x = element k, y = element N-k, 0 < k < N, k != N/2.
// e_{N-k} = -i conj(e_k) = (-e_k.y, -e_k.x).
// DOT (MODE 2): the caller also wants sum_k sym(k) |X_k|^2 per line -- with orthonormal transforms that is x . (M^-1 x).
// pacc collects the pairs UNSCALED (times hs2^2 at the end), sacc the two self-paired indices (already scaled).
template <int MODE, bool UPPER, bool DOT, class Sym>      // UPPER: k > N/2 -- the table holds k <= N/2 only
BK_HD void mid_pair(c2& x, c2& y, int k, int N, const c2* ew, double hs2, double f2, Sym&& sym, c2& pacc) {
    const c2 t = ew[twi(UPPER ? N - k : k)];
    c2 e, en;
    if (UPPER) { en = t; e.x = -t.y; e.y = -t.x; }
    else { e = t; en.x = -t.y; en.y = -t.x; }
    c2 X = x, Y = y;
    if (MODE == 2) {
        // round trip: the forward scale hs2 and the inverse scale f2 are folded into the symbol (8 multiplications per
        // pair instead of 16); post_one / pre_one run unscaled
        const double c = hs2 * f2;
        X = post_one(x, y, e, 1.0); Y = post_one(y, x, en, 1.0);
        c2 f = sym(k);
        if (DOT) { pacc.x = fma(X.x * X.x, f.x, pacc.x); pacc.y = fma(X.y * X.y, f.y, pacc.y); }
        X.x *= f.x * c; X.y *= f.y * c;
        f = sym(N - k);
        if (DOT) { pacc.x = fma(Y.x * Y.x, f.x, pacc.x); pacc.y = fma(Y.y * Y.y, f.y, pacc.y); }
        Y.x *= f.x * c; Y.y *= f.y * c;
        x = pre_one(X, Y, e, 1.0, 1.0); y = pre_one(Y, X, en, 1.0, 1.0);
        return;
    }
    if (MODE != 1) { X = post_one(x, y, e, hs2); Y = post_one(y, x, en, hs2); }
    if (MODE != 0) { x = pre_one(X, Y, e, f2, f2); y = pre_one(Y, X, en, f2, f2); }
    else { x = X; y = Y; }
}
// k = 0 or k = N/2: the partner is the element itself (k = 0: fn = 0, scales s0).
template <int MODE, bool DOT, class Sym>
BK_HD void mid_single(c2& x, int k, const c2* ew, double hs, double fk, double fn, Sym&& sym, c2& sacc) {
    const c2 e = ew[twi(k)];
    c2 X = x;
    if (MODE != 1) X = post_one(x, x, e, hs);
    if (MODE == 2) {
        const c2 f = sym(k);
        if (DOT) { sacc.x = fma(X.x * X.x, f.x, sacc.x); sacc.y = fma(X.y * X.y, f.y, sacc.y); }
        X.x *= f.x; X.y *= f.y;
    }
    if (MODE != 0) x = pre_one(X, X, e, fk, fn);
    else x = X;
}

// Middle item t in [0, N/16): owns the two top groups ga = t, gb = N/8 - t (t = 0: the two self-paired groups 0 and
// N/16), i.e. every spectral index k together with N-k.
//   MODE 0: LDS -> top DIT stage -> post -> st(k, X_k)
//   MODE 1: ld(slot, k) (slot 0..7: group a, 8..15: group b) -> pre -> top inverse DIF stage -> LDS
//   MODE 2: LDS -> top DIT -> post -> X_k *= sym(k) (per line) -> pre -> top inverse DIF -> LDS
//   DOT (MODE 2): dacc.x / dacc.y += this item's share of sum_k sym(k) |X_k|^2 of line a / line b
template <int MODE, bool DOT, class Load, class Store, class Sym>
BK_HD void fused_mid(c2* zp, int N, int t, const c2* tw, const c2* ew, double s0, double s2, Load&& ld, Store&& st,
                     Sym&& sym, c2& dacc) {
    const int G = N >> 3;
    const bool self = t == 0;
    const int ga = t, gb = self ? (G >> 1) : G - t;
    c2 va[8], vb[8];
    if (MODE != 1) {
#pragma unroll
        for (int q = 0; q < 8; ++q) va[q] = zp[swz(ga + q * G)];
        { c2 w[7]; r8_twiddles(w, ga, 0, tw); r8_fwd_regs(va, w); }
#pragma unroll
        for (int q = 0; q < 8; ++q) vb[q] = zp[swz(gb + q * G)];
        { c2 w[7]; r8_twiddles(w, gb, 0, tw); r8_fwd_regs(vb, w); }
    } else {
#pragma unroll
        for (int q = 0; q < 8; ++q) { va[q] = ld(q, ga + q * G); vb[q] = ld(8 + q, gb + q * G); }
    }
    const double rN = 1.0 / N, f2 = rN / s2, f0 = rN / s0, hs2 = 0.5 * s2;
    c2 pacc, sacc;
    pacc.x = pacc.y = sacc.x = sacc.y = 0.0;
    if (self) {
        mid_single<MODE, DOT>(va[0], 0, ew, 0.5 * s0, f0, 0.0, sym, sacc);
        mid_single<MODE, DOT>(va[4], N >> 1, ew, hs2, f2, f2, sym, sacc);
#pragma unroll
        for (int q = 1; q < 4; ++q) mid_pair<MODE, false, DOT>(va[q], va[8 - q], q * G, N, ew, hs2, f2, sym, pacc);
#pragma unroll
        for (int q = 0; q < 4; ++q) mid_pair<MODE, false, DOT>(vb[q], vb[7 - q], gb + q * G, N, ew, hs2, f2, sym, pacc);
    } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) mid_pair<MODE, false, DOT>(va[q], vb[7 - q], ga + q * G, N, ew, hs2, f2, sym, pacc);
#pragma unroll
        for (int q = 4; q < 8; ++q) mid_pair<MODE, true, DOT>(va[q], vb[7 - q], ga + q * G, N, ew, hs2, f2, sym, pacc);
    }
    if (MODE == 2 && DOT) {
        dacc.x += fma(hs2 * hs2, pacc.x, sacc.x);
        dacc.y += fma(hs2 * hs2, pacc.y, sacc.y);
    }
    if (MODE == 0) {
#pragma unroll
        for (int q = 0; q < 8; ++q) { st(ga + q * G, va[q]); st(gb + q * G, vb[q]); }
        return;
    }
    { c2 w[7]; r8_twiddles(w, ga, 0, tw); r8_inv_regs(va, w); }
#pragma unroll
    for (int q = 0; q < 8; ++q) zp[swz(ga + q * G)] = va[q];
    { c2 w[7]; r8_twiddles(w, gb, 0, tw); r8_inv_regs(vb, w); }
#pragma unroll
    for (int q = 0; q < 8; ++q) zp[swz(gb + q * G)] = vb[q];
}

// ------------------------------------------------------------------------------------------------------------------
// Lane-pair split of the round-trip merged middle (round 6; dct_fused_kernel with 512 lanes per tile).  The item t of fused_mid
// <2, DOT> is shared by TWO adjacent lanes: h = 0 owns the top group ga = t, h = 1 the group gb = N/8 - t (t = 0: the self-paired
// groups 0 and N/16) -- 8 complex values per lane instead of 16, which is what lets the kernel run 4 waves per SIMD inside 128
// VGPRs.  Every spectral index k meets N-k in the pair (va[q], vb[7-q]): after the top DIT stage the lanes swap their upper halves
// v[4..7] (one DPP move per register on the device), lane 0 then holds (va[0..3], vb[4..7]) = the pairs k = ga + qG, q < 4, lane 1
// (va[4..7], vb[0..3]) = the pairs q >= 4 (k > N/2: the table holds k <= N/2 only); they swap back and each runs the top inverse
// stage of its own group.  Same arithmetic per value as fused_mid -- the host replay (tests/cpp/dct_core_check.cpp) compares the
// two bit for bit.  Three phases with the exchanges between them:
//   mid_half_fwd    LDS -> top DIT stage of the lane's group
//   [p[i] <- partner's v[4 + i]]
//   mid_half_pairs  post -> symbol -> pre on the lane's four (k, N-k) pairs (t = 0: on its own self-paired group, p untouched)
//   [r[i] <- partner's p[i];  t != 0: v[4 + i] = r[i]]
//   mid_half_inv    top inverse DIF stage -> LDS
BK_HD int mid_half_group(int N, int t, int h) { const int G = N >> 3; return h == 0 ? t : (t == 0 ? (G >> 1) : G - t); }

BK_HD void mid_half_fwd(const c2* zp, int N, int t, int h, const c2* tw, c2* v) {
    const int G = N >> 3, g = mid_half_group(N, t, h);
#pragma unroll
    for (int q = 0; q < 8; ++q) v[q] = zp[swz(g + q * G)];
    c2 w[7];
    r8_twiddles(w, g, 0, tw);
    r8_fwd_regs(v, w);
}

// one (k, N-k) pair, round trip, `upper` (k > N/2) a run-time flag: the arithmetic of mid_pair<2, UPPER, DOT>
template <bool DOT, class Sym>
BK_HD void mid_pair_rt(c2& x, c2& y, int k, int N, bool upper, const c2* ew, double hs2, double f2, Sym&& sym, c2& pacc) {
    const c2 t = ew[twi(upper ? N - k : k)];
    c2 e, en;
    e.x = upper ? -t.y : t.x; e.y = upper ? -t.x : t.y;
    en.x = upper ? t.x : -t.y; en.y = upper ? t.y : -t.x;
    const double c = hs2 * f2;
    c2 X = post_one(x, y, e, 1.0), Y = post_one(y, x, en, 1.0);
    c2 f = sym(k);
    if (DOT) { pacc.x = fma(X.x * X.x, f.x, pacc.x); pacc.y = fma(X.y * X.y, f.y, pacc.y); }
    X.x *= f.x * c; X.y *= f.y * c;
    f = sym(N - k);
    if (DOT) { pacc.x = fma(Y.x * Y.x, f.x, pacc.x); pacc.y = fma(Y.y * Y.y, f.y, pacc.y); }
    Y.x *= f.x * c; Y.y *= f.y * c;
    x = pre_one(X, Y, e, 1.0, 1.0); y = pre_one(Y, X, en, 1.0, 1.0);
}

template <bool DOT, class Sym>
BK_HD void mid_half_pairs(c2* v, c2* p, int N, int t, int h, const c2* ew, double s0, double s2, Sym&& sym, c2& dacc) {
    const int G = N >> 3;
    const double rN = 1.0 / N, f2 = rN / s2, f0 = rN / s0, hs2 = 0.5 * s2;
    c2 pacc, sacc;
    pacc.x = pacc.y = sacc.x = sacc.y = 0.0;
    if (t == 0) {
        if (h == 0) {
            mid_single<2, DOT>(v[0], 0, ew, 0.5 * s0, f0, 0.0, sym, sacc);
            mid_single<2, DOT>(v[4], N >> 1, ew, hs2, f2, f2, sym, sacc);
#pragma unroll
            for (int q = 1; q < 4; ++q) mid_pair<2, false, DOT>(v[q], v[8 - q], q * G, N, ew, hs2, f2, sym, pacc);
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) mid_pair<2, false, DOT>(v[q], v[7 - q], (G >> 1) + q * G, N, ew, hs2, f2, sym, pacc);
        }
    } else {
        // h = 0: (va[i], vb[7 - i]) = (v[i], p[3 - i]);  h = 1: (va[4 + i], vb[3 - i]) = (p[i], v[3 - i]).  The lanes of a pair run in
        // lockstep: every choice between them is a select on VALUES (never on addresses -- the register arrays must stay registers)
        const bool up = h != 0;
        c2 X[4], Y[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const c2 vi = v[i], pi = p[i];
            X[i].x = up ? pi.x : vi.x; X[i].y = up ? pi.y : vi.y;
            Y[3 - i].x = up ? vi.x : pi.x; Y[3 - i].y = up ? vi.y : pi.y;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) mid_pair_rt<DOT>(X[i], Y[i], t + (up ? 4 + i : i) * G, N, up, ew, hs2, f2, sym, pacc);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const c2 xi = X[i], yi = Y[3 - i];
            v[i].x = up ? yi.x : xi.x; v[i].y = up ? yi.y : xi.y;
            p[i].x = up ? xi.x : yi.x; p[i].y = up ? xi.y : yi.y;
        }
    }
    if (DOT) {
        dacc.x += fma(hs2 * hs2, pacc.x, sacc.x);
        dacc.y += fma(hs2 * hs2, pacc.y, sacc.y);
    }
}

BK_HD void mid_half_inv(c2* zp, int N, int t, int h, const c2* tw, c2* v) {
    const int G = N >> 3, g = mid_half_group(N, t, h);
    c2 w[7];
    r8_twiddles(w, g, 0, tw);
    r8_inv_regs(v, w);
#pragma unroll
    for (int q = 0; q < 8; ++q) zp[swz(g + q * G)] = v[q];
}

}  // namespace dctc
}  // namespace bk
