// Host replay of the lane-pair split of the merged middle (dct_split.h) against fused_mid<2> of the product header, for
// one pair of lines: both run the fused round-trip schedule (first stage, LDS middle stages, merged middle, inverse middle
// stages, last stage) and must agree to rounding, including the spectral dot.   Usage: host_check N   (64 .. 1024)
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "dct_split.h"
using namespace bk::dctc;

int main(int argc, char** argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 512;
    int bits = 0;
    while ((1 << bits) < N) ++bits;
    const int G = N >> 3;
    std::vector<double> a(N), b(N);
    unsigned s = 12345u + N;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (double)(s >> 8) / (1 << 24) - 0.5; };
    for (auto& x : a) x = rnd();
    for (auto& x : b) x = rnd();
    std::vector<c2> tw(N / 2), ew(N / 2 + 1);
    for (int j = 0; j < N / 2; ++j) { tw[j].x = std::cos(2.0 * M_PI * j / N); tw[j].y = -std::sin(2.0 * M_PI * j / N); }
    for (int k = 0; k <= N / 2; ++k) { ew[k].x = std::cos(M_PI * k / (2.0 * N)); ew[k].y = -std::sin(M_PI * k / (2.0 * N)); }
    const double s0 = std::sqrt(1.0 / N), s2 = std::sqrt(2.0 / N);
    auto sym = [&](int k) { c2 r; r.x = 1.0 / (1.0 + 0.01 * k); r.y = 1.0 / (2.0 + 0.02 * k * k); return r; };
    auto ldin = [&](int, int n) { c2 r; r.x = a[n]; r.y = b[n]; return r; };
    auto nold = [](int, int) { c2 r; r.x = r.y = 0.0; return r; };
    auto nost = [](int, c2) {};
    auto forward = [&](std::vector<c2>& z) {
        for (int gp = 0; gp < N / 8; ++gp) fused_first(z.data(), N, bits, gp, ldin);
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
    auto inverse = [&](std::vector<c2>& z, std::vector<c2>& out) {
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
        for (int gp = 0; gp < N / 8; ++gp) fused_last(z.data(), N, bits, gp, [&](int n, c2 v) { out[n] = v; });
    };
    // reference: the product's merged middle
    std::vector<c2> zr(N + 16), outr(N);
    forward(zr);
    c2 dref; dref.x = dref.y = 0.0;
    for (int t = 0; t < N / 16; ++t) fused_mid<2, true>(zr.data(), N, t, tw.data(), ew.data(), s0, s2, nold, nost, sym, dref);
    inverse(zr, outr);
    // split: two lanes per item in lockstep
    std::vector<c2> zs(N + 16), outs(N);
    forward(zs);
    const double rN = 1.0 / N, f2 = rN / s2, hs2 = 0.5 * s2, cc = hs2 * f2;
    (void)s0;
    c2 dsum; dsum.x = dsum.y = 0.0;
    for (int t = 0; t < N / 16; ++t) {
        const bool self = t == 0;
        c2 v[2][8], w[2][7], pacc[2];
        SplitRole role[2];
        for (int h = 0; h < 2; ++h) {
            role[h].self = self; role[h].sp = self && h == 0;
            role[h].g = h == 0 ? t : (self ? (G >> 1) : G - t); role[h].G = G; role[h].N = N;
            pacc[h].x = pacc[h].y = 0.0;
            split_load_fwd(zs.data(), N, role[h].g, tw.data(), v[h], w[h]);
            split_rotate_in(role[h].sp, v[h]);
        }
        // the four exchange steps, both lanes in lockstep: "received" = what the other lane holds at that point
#define STEP(I)                                                                                                           \
        {                                                                                                                \
            c2 Xa[2], Xb[2];                                                                                             \
            for (int h = 0; h < 2; ++h)                                                                                  \
                split_phase1<true, I>(role[h], v[h][I], v[h][7 - I], v[1 - h][I], v[1 - h][7 - I], ew.data(), cc, sym, pacc[h], Xa[h], Xb[h]); \
            for (int h = 0; h < 2; ++h)                                                                                  \
                split_phase2<I>(role[h], Xa[h], Xb[h], Xa[1 - h], Xb[1 - h], ew.data(), v[h][I], v[h][7 - I]);           \
        }
        STEP(0) STEP(1) STEP(2) STEP(3)
#undef STEP
        for (int h = 0; h < 2; ++h) {
            split_rotate_out(role[h].sp, v[h]);
            split_inv_store(zs.data(), N, role[h].g, v[h], w[h]);
            dsum.x += hs2 * hs2 * pacc[h].x;
            dsum.y += hs2 * hs2 * pacc[h].y;
        }
    }
    inverse(zs, outs);
    double worst = 0.0, scale = 0.0;
    for (int n = 0; n < N; ++n) {
        worst = std::fmax(worst, std::fmax(std::fabs(outs[n].x - outr[n].x), std::fabs(outs[n].y - outr[n].y)));
        scale = std::fmax(scale, std::fmax(std::fabs(outr[n].x), std::fabs(outr[n].y)));
    }
    const double ddot = std::fmax(std::fabs(dsum.x - dref.x) / std::fabs(dref.x), std::fabs(dsum.y - dref.y) / std::fabs(dref.y));
    printf("N=%d  max|split - fused_mid| = %.3e (scale %.3e)  dot rel. diff %.3e\n", N, worst, scale, ddot);
    return (worst <= 1e-13 * scale && ddot <= 1e-13) ? 0 : 1;
}
