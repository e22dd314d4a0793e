// EXPERIMENT (not part of the product library; staged for measurement in the next round): the merged middle of the z round
// trip (bifurcationkit.jl_amd/csrc/dct_core.h: fused_mid<2>) split across a PAIR of lanes, so that a 16-line x 512-point
// tile can run on 512 lanes -- one radix-8 group per lane instead of two -- at half the registers per lane.
//
// fused_mid's item t owns the two top groups ga = t and gb = N/8 - t because the post-twiddle pairs the spectral index
// k = ga + q G with N - k = gb + (7 - q) G.  Here lane h = 0 owns ga, lane h = 1 owns gb; own element i pairs with the
// PARTNER lane's element 7 - i, so per step i < 4 a lane sends (v[7-i], v[i]), receives the partner's same two, forms its two
// post-twiddled, symbol-scaled values, sends those back the same way and applies the pre-twiddle.  t = 0 is self-paired:
// both of its lanes run the same steps with "received = sent" (the lane of group 0 with rotated registers; see SplitRole).
//
// The helpers below are plain arithmetic (BK_HD): host_check.cpp replays a pair of lanes in lockstep against fused_mid<2>.
#pragma once
#include "../../bifurcationkit.jl_amd/csrc/dct_core.h"

namespace bk {
namespace dctc {

// e_k = exp(-i pi k / 2N) for any k in (0, N) from the half table ew[0 .. N/2]:  e_{N-k} = -i conj(e_k)
BK_HD c2 ek_any(int k, int N, const c2* ew) {
    c2 e;
    if (k <= (N >> 1)) e = ew[k];
    else { const c2 t = ew[N - k]; e.x = -t.y; e.y = -t.x; }
    return e;
}

// own value Z_k, partner value Z_{N-k}: unscaled post-twiddle, times symbol(k) * c (c = forward scale * inverse scale).
// DOT: pacc += symbol(k) |X_k|^2 with X unscaled (the caller multiplies by hs2^2 once).
// e_k from the half table when the caller knows statically which half k lies in.  In the exchange steps own element I < 4
// is ALWAYS in the lower half (k = g + I G < N/2; k = 0 .. 3G for the rotated group-0 lane) and own element 7 - I always in
// the upper one (k = N/2 exactly, group 0's u[4], is covered by both forms: -i conj(e_{N/2}) = e_{N/2}).
template <bool UPPER>
BK_HD c2 ek_half(int k, int N, const c2* ew) {
    if (!UPPER) return ew[k];
    const c2 t = ew[N - k];
    c2 e; e.x = -t.y; e.y = -t.x;
    return e;
}

template <bool DOT, bool UPPER, class Sym>
BK_HD c2 split_post_sym(c2 own, c2 partner, int k, int N, const c2* ew, double c, Sym&& sym, c2& pacc, double wgt = 1.0) {
    const c2 e = ek_half<UPPER>(k, N, ew);
    c2 X = post_one(own, partner, e, 1.0);
    const c2 f = sym(k);
    const double tx = X.x * f.x, ty = X.y * f.y;          // shared by the dot and the scaling
    if (DOT) {
        double dx = tx * X.x, dy = ty * X.y;
#ifdef __HIP_DEVICE_COMPILE__
        // pin the evaluation point: left alone the compiler sinks the four products and the accumulation to the end of the
        // exchange steps and keeps their operands alive (measured: +18 VGPRs per step, 232 B of scratch at the 128 cap)
        asm volatile("" : "+v"(dx), "+v"(dy));
#endif
        pacc.x = fma(dx, wgt, pacc.x); pacc.y = fma(dy, wgt, pacc.y);
#ifdef __HIP_DEVICE_COMPILE__
        asm volatile("" : "+v"(pacc.x), "+v"(pacc.y));
#endif
    }
    X.x = tx * c; X.y = ty * c;
    return X;
}

template <bool UPPER>
BK_HD c2 split_pre(c2 Xown, c2 Xpartner, int k, int N, const c2* ew) { return pre_one(Xown, Xpartner, ek_half<UPPER>(k, N, ew), 1.0, 1.0); }

// top forward radix-8 of group g: v[q] = Z_{g + q G} afterwards; w = the group's 7 twiddles (kept for the inverse)
BK_HD void split_load_fwd(const c2* zp, int N, int g, const c2* tw, c2* v, c2* w) {
    const int G = N >> 3;
#pragma unroll
    for (int q = 0; q < 8; ++q) v[q] = zp[swz(g + q * G)];
    r8_twiddles(w, g, 0, tw);
    r8_fwd_regs(v, w);
}
BK_HD void split_inv_store(c2* zp, int N, int g, c2* v, const c2* w) {
    const int G = N >> 3;
    r8_inv_regs(v, w);
#pragma unroll
    for (int q = 0; q < 8; ++q) zp[swz(g + q * G)] = v[q];
}

// ---- the exchange steps, the same code for every lane.  Roles: a general lane (t != 0) pairs own element i with the PARTNER
// lane's element 7 - i; the self-paired lane (t = 0, h = 1: group N/16) with its OWN element 7 - i ("received = sent"); the
// lane of group 0 (t = 0, h = 0; flag sp) holds its registers rotated, u = [v1, v2, v3, v0 | v4 .. v7], so that steps
// 0..2 are its pairs (q, 8 - q) and step 3 its two single indices k = 0 (partner = itself, no C_{N-k} term, half weight
// in the dot because its forward scale is s0 instead of s2) and k = N/2 (partner = itself).  Step I (compile time):
//   phase 1: own a = u[I], b = u[7-I], received ra / rb = the partner lane's a / b  ->  Xa, Xb (post-twiddle, symbol, scale)
//   phase 2: received qa / qb = the partner lane's Xa / Xb                           ->  u[I], u[7-I] (pre-twiddle)
// component-wise select (a struct-valued ?: may be lowered to a select of ADDRESSES, i.e. through memory)
BK_HD c2 csel(bool c, c2 a, c2 b) { c2 r; r.x = c ? a.x : b.x; r.y = c ? a.y : b.y; return r; }

struct SplitRole {
    bool self, sp;      // t == 0;  t == 0 && h == 0
    int g, G, N;        // own top group, N / 8, transform length
};
template <int I>
BK_HD int split_ki(const SplitRole& r) { return r.sp ? ((I + 1) & 3) * r.G : r.g + I * r.G; }
template <int I>
BK_HD int split_kj(const SplitRole& r) { return r.g + (7 - I) * r.G; }

// (two halves, so that the caller can keep them apart in the instruction schedule: interleaved they cost 30+ VGPRs)
template <bool DOT, int I, class Sym>
BK_HD c2 split_phase1a(const SplitRole& r, c2 a, c2 b, c2 rb, const c2* ew, double c, Sym&& sym, c2& pacc) {
    c2 Pa = csel(r.self, b, rb);                          // partner of own u[I]: the other lane's u[7-I] (self-paired: own)
    if (I == 3) Pa = csel(r.sp, a, Pa);                   // k = 0 pairs with itself
    const double wgt = (I == 3 && r.sp) ? 0.5 : 1.0;    // k = 0 carries the forward scale s0, not s2
    const c2 Xa = split_post_sym<DOT, false>(a, Pa, split_ki<I>(r), r.N, ew, c, sym, pacc, wgt);
    return Xa;
}
template <bool DOT, int I, class Sym>
BK_HD c2 split_phase1b(const SplitRole& r, c2 a, c2 b, c2 ra, const c2* ew, double c, Sym&& sym, c2& pacc) {
    c2 Pb = csel(r.self, a, ra);
    if (I == 3) Pb = csel(r.sp, b, Pb);                   // k = N/2 pairs with itself
    return split_post_sym<DOT, true>(b, Pb, split_kj<I>(r), r.N, ew, c, sym, pacc);
}
template <bool DOT, int I, class Sym>
BK_HD void split_phase1(const SplitRole& r, c2 a, c2 b, c2 ra, c2 rb, const c2* ew, double c, Sym&& sym, c2& pacc, c2& Xa, c2& Xb) {
    Xa = split_phase1a<DOT, I>(r, a, b, rb, ew, c, sym, pacc);
    Xb = split_phase1b<DOT, I>(r, a, b, ra, ew, c, sym, pacc);
}
template <int I>
BK_HD void split_phase2(const SplitRole& r, c2 Xa, c2 Xb, c2 qa, c2 qb, const c2* ew, c2& ua, c2& ub) {
    c2 Qa = csel(r.self, Xb, qb), Qb = csel(r.self, Xa, qa);
    if (I == 3) {
        c2 zero;
        zero.x = zero.y = 0.0;
        Qa = csel(r.sp, zero, Qa);
        Qb = csel(r.sp, Xb, Qb);
    }
    ua = split_pre<false>(Xa, Qa, split_ki<I>(r), r.N, ew);
    ub = split_pre<true>(Xb, Qb, split_kj<I>(r), r.N, ew);
}
// register rotation of the group-0 lane (and back)
BK_HD void split_rotate_in(bool sp, c2* v) {
    const c2 v0 = v[0], v1 = v[1], v2 = v[2], v3 = v[3];
    v[0] = csel(sp, v1, v0); v[1] = csel(sp, v2, v1); v[2] = csel(sp, v3, v2); v[3] = csel(sp, v0, v3);
}
BK_HD void split_rotate_out(bool sp, c2* u) {
    const c2 u0 = u[0], u1 = u[1], u2 = u[2], u3 = u[3];
    u[0] = csel(sp, u3, u0); u[1] = csel(sp, u0, u1); u[2] = csel(sp, u1, u2); u[3] = csel(sp, u2, u3);
}

}  // namespace dctc
}  // namespace bk
