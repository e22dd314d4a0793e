// Host replay of the fast-DCT phases of bifurcationkit.jl_amd/csrc/dct_core.h for ONE pair of lines.
// stdin: "mode N" (mode 0/1 = forward/inverse radix-2, 2/3 = the same with grouped radix-8 stages, 4/5/6 = fused schedule forward / inverse / forward-symbol-inverse, 7/8/9 = the same with the contiguous-axis outer stages, 10 = 6 plus the spectral dot products, printed as one more line, 11 / 12 = 6 / 10 with the merged middle split over lane pairs (mid_half_*: two "lanes" replayed in lockstep, exchanges as array swaps)) then N values of line a, N values of line b.  stdout: the two transformed lines.
#include <cmath>
#include <cstdio>
#include <vector>
#ifndef BK_DCT_CORE_H
#define BK_DCT_CORE_H "../../bifurcationkit.jl_amd/csrc/dct_core.h"
#endif
#include BK_DCT_CORE_H
#ifndef BK_DCT_TWI              // table index map of the header under test (the candidate layout pads the tables: -DBK_DCT_TWI=twi)
#define BK_DCT_TWI(j) (j)
#endif
using namespace bk::dctc;
int main() {
    int inverse, N, grouped = 0, fused = 0;
    if (scanf("%d %d", &inverse, &N) != 2) return 2;
    int contiguous = 0, want_dot = 0;
    int split = 0;
    if (inverse == 11 || inverse == 12) { split = 1; inverse = inverse == 11 ? 6 : 10; }   // 11 / 12: lane-pair split of the round trip's merged middle
    if (inverse == 10) { want_dot = 1; inverse = 6; }          // 10: fused roundtrip that also returns sum_k sym(k) X_k^2 per line
    if (inverse >= 7) { contiguous = 1; inverse -= 3; }        // 7/8/9: the same with the contiguous-axis first / last stage
    if (inverse >= 4) { fused = inverse - 3; inverse = 0; }   // 4: fused forward, 5: fused inverse, 6: fused roundtrip
    if (inverse >= 2) { grouped = 1; inverse -= 2; }     // modes 2/3: radix-8 grouped stages
    int bits = 0;
    while ((1 << bits) < N) ++bits;
    std::vector<double> a(N), b(N);
    for (auto& x : a) if (scanf("%lf", &x) != 1) return 2;
    for (auto& x : b) if (scanf("%lf", &x) != 1) return 2;
    std::vector<c2> tw(BK_DCT_TWI(N / 2) + 1), ew(BK_DCT_TWI(N / 2 + 1) + 1), z(N + 16);
    for (int j = 0; j < N / 2; ++j) { tw[BK_DCT_TWI(j)].x = std::cos(2.0 * M_PI * j / N); tw[BK_DCT_TWI(j)].y = -std::sin(2.0 * M_PI * j / N); }
    for (int k = 0; k <= N / 2; ++k) { ew[BK_DCT_TWI(k)].x = std::cos(M_PI * k / (2.0 * N)); ew[BK_DCT_TWI(k)].y = -std::sin(M_PI * k / (2.0 * N)); }
    const double s0 = std::sqrt(1.0 / N), s2 = std::sqrt(2.0 / N);
    if (fused) {
        // modes 4/5/6: the fused schedule of dct_fused_kernel (first / last radix-8 stage on registers fed from "global")
        std::vector<c2> out(N);
        std::vector<c2>& ewf = ew;                          // half table k <= N/2
        auto ldin = [&](int, int n) { c2 r; r.x = a[n]; r.y = b[n]; return r; };
        auto stout = [&](int n, c2 v) { out[n] = v; };
        auto sym = [&](int k) { c2 r; r.x = 1.0 / (1.0 + 0.01 * k); r.y = 1.0 / (2.0 + 0.02 * k * k); return r; };
        auto middle_fwd = [&]() {
            for (int lh = 3; lh < bits - 3;) {
                const int R = bits - 3 - lh >= 3 ? 3 : bits - 3 - lh;
                for (int g = 0; g < (N >> R); ++g) {
                    if (R == 3) r8_group_fwd(z.data(), bits, lh, g, tw.data());
                    else if (R == 2) dit_group<2>(z.data(), bits, lh, g, tw.data());
                    else dit_group<1>(z.data(), bits, lh, g, tw.data());
                }
                lh += R;
            }
        };
        auto middle_inv = [&]() {
            for (int top = bits - 3; top > 3;) {
                const int R = top - 3 >= 3 ? 3 : top - 3;
                const int lh = top - R;
                for (int g = 0; g < (N >> R); ++g) {
                    if (R == 3) r8_group_inv(z.data(), bits, lh, g, tw.data());
                    else if (R == 2) dif_group_inv<2>(z.data(), bits, lh, g, tw.data());
                    else dif_group_inv<1>(z.data(), bits, lh, g, tw.data());
                }
                top -= R;
            }
        };
        if (fused != 2) {                                  // forward or roundtrip: first stage from the samples
            if (contiguous)
                for (int gp = 0; gp < N / 16; ++gp)
                    fused_first2(z.data(), N, bits, gp, [&](int, int j, double& ea, double& oa, double& eb, double& ob) {
                        ea = a[2 * j]; oa = a[2 * j + 1]; eb = b[2 * j]; ob = b[2 * j + 1];
                    });
            else
            for (int gp = 0; gp < N / 8; ++gp) fused_first(z.data(), N, bits, gp, ldin);
            middle_fwd();
        }
        c2 dacc;
        dacc.x = dacc.y = 0.0;
        if (split && fused == 3) {
            for (int t = 0; t < N / 16; ++t) {
                c2 v[2][8], p[2][4], r[2][4];
                for (int h = 0; h < 2; ++h) mid_half_fwd(z.data(), N, t, h, tw.data(), v[h]);
                for (int h = 0; h < 2; ++h) for (int i = 0; i < 4; ++i) p[h][i] = v[h ^ 1][4 + i];        // exchange 1
                for (int h = 0; h < 2; ++h) {
                    if (want_dot) mid_half_pairs<true>(v[h], p[h], N, t, h, ewf.data(), s0, s2, sym, dacc);
                    else mid_half_pairs<false>(v[h], p[h], N, t, h, ewf.data(), s0, s2, sym, dacc);
                }
                for (int h = 0; h < 2; ++h) for (int i = 0; i < 4; ++i) r[h][i] = p[h ^ 1][i];            // exchange 2
                for (int h = 0; h < 2; ++h) {
                    if (t != 0) for (int i = 0; i < 4; ++i) v[h][4 + i] = r[h][i];
                    mid_half_inv(z.data(), N, t, h, tw.data(), v[h]);
                }
            }
        } else
        for (int t = 0; t < N / 16; ++t) {
            if (fused == 1) fused_mid<0, false>(z.data(), N, t, tw.data(), ewf.data(), s0, s2, ldin, stout, sym, dacc);
            else if (fused == 2) fused_mid<1, false>(z.data(), N, t, tw.data(), ewf.data(), s0, s2, ldin, stout, sym, dacc);
            else if (want_dot) fused_mid<2, true>(z.data(), N, t, tw.data(), ewf.data(), s0, s2, ldin, stout, sym, dacc);
            else fused_mid<2, false>(z.data(), N, t, tw.data(), ewf.data(), s0, s2, ldin, stout, sym, dacc);
        }
        if (fused != 1) {
            middle_inv();
            if (contiguous)
                for (int gp = 0; gp < N / 16; ++gp)
                    fused_last2(z.data(), N, bits, gp, [&](int j, double ea, double oa, double eb, double ob) {
                        out[2 * j].x = ea; out[2 * j + 1].x = oa; out[2 * j].y = eb; out[2 * j + 1].y = ob;
                    });
            else
            for (int gp = 0; gp < N / 8; ++gp) fused_last(z.data(), N, bits, gp, stout);
        }
        for (int k = 0; k < N; ++k) printf("%.17g %.17g\n", out[k].x, out[k].y);
        if (want_dot) printf("%.17g %.17g\n", dacc.x, dacc.y);
        return 0;
    }
    if (!inverse) {
        for (int n = 0; n < N; ++n) { const int p = sample_slot(n, N, bits); z[p].x = a[n]; z[p].y = b[n]; }
        if (inverse == 0 && grouped) {
            for (int lh = 0; lh < bits;) {
                const int R = bits - lh >= 3 ? 3 : bits - lh;
                for (int g = 0; g < (N >> R); ++g) {
                    if (R == 3) dit_group<3>(z.data(), bits, lh, g, tw.data());
                    else if (R == 2) dit_group<2>(z.data(), bits, lh, g, tw.data());
                    else dit_group<1>(z.data(), bits, lh, g, tw.data());
                }
                lh += R;
            }
        } else
        for (int lh = 0; lh < bits; ++lh)
            for (int j = 0; j < N / 2; ++j) dit_butterfly(z.data(), bits, lh, j, tw.data());
        for (int k = 0; k <= N / 2; ++k) fwd_post(z.data(), N, k, ew.data(), s0, s2);
        for (int k = 0; k < N; ++k) printf("%.17g %.17g\n", z[swz(k)].x, z[swz(k)].y);
    } else {
        for (int k = 0; k < N; ++k) { z[swz(k)].x = a[k]; z[swz(k)].y = b[k]; }
        for (int k = 0; k <= N / 2; ++k) inv_pre(z.data(), N, k, ew.data(), s0, s2);
        if (grouped) {
            for (int top = bits; top > 0;) {
                const int R = top >= 3 ? 3 : top;
                const int lh = top - R;
                for (int g = 0; g < (N >> R); ++g) {
                    if (R == 3) dif_group_inv<3>(z.data(), bits, lh, g, tw.data());
                    else if (R == 2) dif_group_inv<2>(z.data(), bits, lh, g, tw.data());
                    else dif_group_inv<1>(z.data(), bits, lh, g, tw.data());
                }
                top -= R;
            }
        } else
        for (int lh = bits - 1; lh >= 0; --lh)
            for (int j = 0; j < N / 2; ++j) dif_butterfly_inv(z.data(), bits, lh, j, tw.data());
        for (int j = 0; j < N; ++j) { const int p = sample_slot(j, N, bits); printf("%.17g %.17g\n", z[p].x, z[p].y); }
    }
    return 0;
}
