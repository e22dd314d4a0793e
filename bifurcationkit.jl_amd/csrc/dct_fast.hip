// Fast DCT-II / DCT-III axis passes for the spectral preconditioner (dct.hip): one workgroup stages a tile of
// LT lines of length N = 2^bits in LDS (two real lines per complex sequence), runs the FFT of dct_core.h there
// (radix-8 register groups: three radix-2 stages per LDS round trip), and writes the tile back -- HBM traffic is exactly one read and one write of the array per
// axis pass (16 B/point), everything else happens in LDS.
//
// Tiling.  The array is [n2][n1][n0] with n0 fastest.
//   axis 0: a tile is LT consecutive rows (each row N = n0 contiguous doubles): global accesses are coalesced
//           along the row.
//   axis 1/2: a tile is LT consecutive x for one fixed (z) / (y): element n of line L sits at
//           base + L + n*stride, so a wavefront reads LT contiguous doubles for several n -- 128-B segments for
//           LT = 16 -- and the transform direction never has to be contiguous in memory.
// The last forward pass can apply the inverse symbol 1/((1 + lam_x + lam_y + lam_z)^2 + shift) while storing
// (fuse_scale), which removes the separate scaling pass.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <mutex>
#include <vector>

#include "ops.h"

// The library is built with -ffp-contract=off (the stencil / Krylov kernels are compared with the CPU oracle to tight
// tolerances); the FFT butterflies here are VALU-issue bound and gain from fused multiply-adds, and the DCT parity
// tolerance (1e-13 relative) does not depend on the contraction.
#pragma clang fp contract(fast)
#include "dct_core.h"

namespace bk {

namespace {

using dctc::c2;

struct FftK {
    int n0, n1, n2, axis, N, bits, LT, ltbits, inverse;   // ltbits = log2(LT) if LT is a power of two, else -1
    const double* in;
    double* out;
    const double* twid;       // dct_twiddle_table(N): tw_len(N) complex exp(-2 pi i q/N), then ew_len(N) complex exp(-i pi k/2N)
    const double* lam0;
    const double* lam1;
    const double* lam2;       // may be NULL (2-D)
    double shift;
    double* dotp;             // DOT kernels: per-workgroup partial sums
    int fuse_scale;           // 1: multiply by the inverse symbol while storing the forward result
    int roundtrip;            // 1: forward, inverse symbol, inverse -- all in LDS, one read + one write of the array
    int tiles_x;              // axis >= 1: number of LT-wide tiles along x
    int pairvec;              // axis >= 1 and 16-B aligned pairs: the two lines of a pair are loaded / stored as one double2
    const unsigned* kmap;     // axis-1 fused passes of the distributed plan: element offset of index k in the per-rank
    unsigned split_plane;     //   block layout (kmap[k] + other * split_plane + x); split 1: on the output, 2: on the input
    int split;
    long long* trace;         // debug (option dct_trace): per-tile phase timestamps, 8 per workgroup, or NULL
    int fast;                 // full tiles, power-of-two shapes, < 2^31 elements: incremental addressing (host-checked)
    int nt_load, nt_store;    // fused kernel: non-temporal hint on the tile loads / stores (every element is touched once)
    int ntiles;               // fused kernel: tiles of the pass (= workgroups)
    int xmap;                 // fused kernel: XCD-contiguous slot -> tile map (ntiles % 8 == 0)
    int stagger, stag_cu;     // fused kernel: start offset of the workgroups that fill every CU's second slot (stag_cu = CUs)
    // FZ kernels (x passes, ops.h: DctFuse): forward -- every sample is multiplied by fzA + u (fzB + fzC u), u = fz_v at the
    // sample's own index; inverse -- the stored value is fz_ct * result + fz_cx * fz_v[same index]
    const double* fz_v;
    double fzA, fzB, fzC, fz_cx, fz_ct;
    double* fz_store;         // FZS (forward): the sample + fz_cx * fz_v[same index] is what gets transformed, and it is stored here
    // SLAB kernels (z passes of the slab z-solve as forward / inverse HALVES, dct.hip: dct_apply_slab): the forward half stores
    // y^_k = sym_k f^_k and, from sums over the spectrum with the local basis phi, the values of y = B^-1 f at the four planes next to
    // the slab faces (planes 0, 1, nl-2, nl-1); the inverse half adds sym_k * sum_p phi_k(p) delta_p(line) -- the Woodbury correction,
    // applied in the z-spectral domain -- before it transforms back
    // The face data goes straight into / comes straight from the all-to-all buffers of the capacitance solve ([owner][4][Lr], dct_slab.hip):
    // the forward half writes (u'y, w'y) of this rank's bottom and top face, the inverse half reads (nu_u, nu_w) of both faces.
    double* face_y;
    const double* face_d;
    const double* phi;        // [2][N]: local DCT-II basis at planes 0 and 1 (phi_k(N-1-z) = (-1)^k phi_k(z))
    unsigned face_Lr;         // lines per owner
    double slab_a;            // 1 / h_z^2
    int slab_hasb, slab_hast; // this rank has a bottom / a top neighbour
};

template <int NT>
__global__ void __launch_bounds__(NT) dct_fft_kernel(FftK P) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int N = P.N, bits = P.bits, LT = P.LT, half = N >> 1;
    const int npairs = LT >> 1;
    const int pstride = N + 1;                                // complex elements per pair (+1: bank skew)
    c2* z = reinterpret_cast<c2*>(smem);
    c2* tw = z + (size_t)npairs * pstride;                    // N/2 FFT twiddles (dct_core.h: twi layout)
    const int twl = dctc::tw_len(N);
    const c2* ew = reinterpret_cast<const c2*>(P.twid) + twl;    // post twiddles stay in global (read once)
    const int tid = threadIdx.x;

    // ---- tile decode
    size_t base, lstride, estride;
    int nlines;                                               // valid lines in this tile
    int ti0 = 0, ti1 = 0, ti2 = 0;                            // 3-D index of (line 0, element 0)
    if (P.axis == 0) {
        const size_t rows = (size_t)P.n1 * P.n2;
        const size_t r0 = (size_t)blockIdx.x * LT;
        nlines = (int)min((size_t)LT, rows - r0);
        base = r0 * P.n0; lstride = P.n0; estride = 1;
        ti1 = (int)(r0 % P.n1); ti2 = (int)(r0 / P.n1);
    } else {
        const int tx = blockIdx.x % P.tiles_x;
        const int other = blockIdx.x / P.tiles_x;             // i2 (axis 1) or i1 (axis 2)
        const int x0 = tx * LT;
        nlines = min(LT, P.n0 - x0);
        lstride = 1;
        ti0 = x0;
        if (P.axis == 1) { base = x0 + (size_t)P.n0 * P.n1 * other; estride = P.n0; ti2 = other; }
        else { base = x0 + (size_t)P.n0 * other; estride = (size_t)P.n0 * P.n1; ti1 = other; }
    }

    for (int q = tid; q < twl; q += NT) tw[q] = reinterpret_cast<const c2*>(P.twid)[q];

    // ---- load: one work item = one complex LDS element = the same sample n of the two lines (2p, 2p+1) of a pair.
    // axis 0: consecutive items walk along n (both rows coalesced); axis >= 1: consecutive items are consecutive pairs,
    // i.e. consecutive x -- one 16-B load per item, 128-B segments per n.  U items per lane are in flight before the
    // first LDS write (with two workgroups per CU the memory-level parallelism has to come from each lane).
    const int total = LT * N;
    const int nitems = npairs << bits;
    const int pbits = P.ltbits >= 1 ? P.ltbits - 1 : -1;       // log2(npairs) when LT is a power of two
    auto decode = [&](int q, int& pr, int& n) {
        if (P.axis == 0) { pr = q >> bits; n = q & (N - 1); }
        else if (pbits >= 0) { pr = q & (npairs - 1); n = q >> pbits; }
        else { pr = q % npairs; n = q / npairs; }
    };
    const bool fwd_slots = !(P.inverse && !P.roundtrip);     // which slot order the input goes to
    if (P.fast) {
        // Incremental addressing (no per-item decode): nitems / NT == 8 items per lane.
        if (P.axis == 0) {
            // NT is a multiple of N: lane owns one sample index n and walks over the pairs
            const int n = tid & (N - 1);
            const int slot = fwd_slots ? dctc::sample_slot(n, N, bits) : dctc::swz(n);
            const int pr0 = tid >> bits, dpr = NT >> bits;
            const double* src = P.in + base + n + (size_t)(2 * pr0) * P.n0;
            const size_t inc = (size_t)(2 * dpr) * P.n0;
            c2 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (pr0 + u * dpr < npairs) { v[u].x = src[u * inc]; v[u].y = src[u * inc + P.n0]; }
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (pr0 + u * dpr < npairs) z[(pr0 + u * dpr) * pstride + slot] = v[u];
        } else {
            // lane owns one pair (two adjacent x) and walks along the transform direction
            const int pr = tid & (npairs - 1);
            const int n0_ = tid >> pbits, dn = NT >> pbits;
            const double* src = P.in + base + 2 * pr + (size_t)n0_ * estride;
            const size_t inc = (size_t)dn * estride;
            c2 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (n0_ + u * dn < N) { const double2 t = *reinterpret_cast<const double2*>(src + u * inc); v[u].x = t.x; v[u].y = t.y; }
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int n = n0_ + u * dn;
                if (n < N) z[pr * pstride + (fwd_slots ? dctc::sample_slot(n, N, bits) : dctc::swz(n))] = v[u];
            }
        }
    } else
    {
        constexpr int U = 8;
        for (int q0 = tid; q0 < nitems; q0 += NT * U) {
            c2 v[U];
            int dst[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int q = q0 + u * NT;
                int pr, n;
                decode(q, pr, n);
                const bool inside = q < nitems;
                const int L0 = 2 * pr;
                const double* src = P.in + base + (size_t)L0 * lstride + (size_t)n * estride;
                v[u].x = 0.0; v[u].y = 0.0;
                if (inside) {
                    if (P.pairvec && L0 + 1 < nlines) {
                        const double2 t = *reinterpret_cast<const double2*>(src);
                        v[u].x = t.x; v[u].y = t.y;
                    } else {
                        if (L0 < nlines) v[u].x = src[0];
                        if (L0 + 1 < nlines) v[u].y = src[lstride];
                    }
                }
                const int slot = (P.inverse && !P.roundtrip) ? dctc::swz(n) : dctc::sample_slot(n, N, bits);
                dst[u] = inside ? pr * pstride + slot : -1;
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (dst[u] >= 0) z[dst[u]] = v[u];
        }
    }
    __syncthreads();

    const double s0 = sqrt(1.0 / N), s2 = sqrt(2.0 / N);
    // FFT stages, three radix-2 stages per LDS round trip (radix-8 groups held in registers)
    auto run_groups = [&](int lh, int R, bool inv) {
        const int gbits = bits - R;                            // log2(groups per pair)
        const int ngr = npairs << gbits;
        for (int w = tid; w < ngr; w += NT) {
            c2* zp = z + (size_t)(w >> gbits) * pstride;
            const int g = w & ((1 << gbits) - 1);
            if (!inv) {
                if (R == 3) dctc::dit_group<3>(zp, bits, lh, g, tw);
                else if (R == 2) dctc::dit_group<2>(zp, bits, lh, g, tw);
                else dctc::dit_group<1>(zp, bits, lh, g, tw);
            } else {
                if (R == 3) dctc::dif_group_inv<3>(zp, bits, lh, g, tw);
                else if (R == 2) dctc::dif_group_inv<2>(zp, bits, lh, g, tw);
                else dctc::dif_group_inv<1>(zp, bits, lh, g, tw);
            }
        }
        __syncthreads();
    };
    // pre/post twiddle phase: work item (pair, k) for k in [0, N/2); k = 0 also handles k = N/2
    auto run_twiddle = [&](bool inv) {
        const int nw = npairs << (bits - 1);
        for (int w = tid; w < nw; w += NT) {
            c2* zp = z + (size_t)(w >> (bits - 1)) * pstride;
            const int k = w & (half - 1);
            if (!inv) {
                dctc::fwd_post(zp, N, k, ew, s0, s2);
                if (k == 0) dctc::fwd_post(zp, N, half, ew, s0, s2);
            } else {
                dctc::inv_pre(zp, N, k, ew, s0, s2);
                if (k == 0) dctc::inv_pre(zp, N, half, ew, s0, s2);
            }
        }
        __syncthreads();
    };
    auto symbol_inv = [&](int L, int n) -> double {
        int i0, i1, i2;
        if (P.axis == 0) { const size_t r = (size_t)ti1 + (size_t)ti2 * P.n1 + L; i0 = n; i1 = (int)(r % P.n1); i2 = (int)(r / P.n1); }
        else if (P.axis == 1) { i0 = ti0 + L; i1 = n; i2 = ti2; }
        else { i0 = ti0 + L; i1 = ti1; i2 = n; }
        const double sy = 1.0 + P.lam0[i0] + P.lam1[i1] + (P.lam2 ? P.lam2[i2] : 0.0);
        return 1.0 / (sy * sy + P.shift);
    };
    auto forward = [&]() {
        for (int lh = 0; lh < bits;) {
            const int R = bits - lh >= 3 ? 3 : bits - lh;
            run_groups(lh, R, false);
            lh += R;
        }
        run_twiddle(false);
    };
    auto inverse = [&]() {
        run_twiddle(true);
        for (int top = bits; top > 0;) {
            const int R = top >= 3 ? 3 : top;
            run_groups(top - R, R, true);
            top -= R;
        }
    };
    bool out_is_samples = P.inverse != 0;                     // which slot order the result sits in
    if (P.roundtrip) {
        forward();
        for (int q = tid; q < nitems; q += NT) {             // spectrum sits in natural slots swz(k)
            int pr, n;
            decode(q, pr, n);
            const int L0 = 2 * pr;
            if (L0 >= nlines) continue;
            c2 e = z[pr * pstride + dctc::swz(n)];
            e.x *= symbol_inv(L0, n);
            if (L0 + 1 < nlines) e.y *= symbol_inv(L0 + 1, n);
            z[pr * pstride + dctc::swz(n)] = e;
        }
        __syncthreads();
        inverse();
        out_is_samples = true;
    } else if (!P.inverse) {
        forward();
    } else {
        inverse();
    }

    // ---- store (same item -> address map as the load)
    if (P.fast && !(P.fuse_scale && !P.roundtrip)) {
        if (P.axis == 0) {
            const int n = tid & (N - 1);
            const int slot = out_is_samples ? dctc::sample_slot(n, N, bits) : dctc::swz(n);
            const int pr0 = tid >> bits, dpr = NT >> bits;
            double* dstp = P.out + base + n + (size_t)(2 * pr0) * P.n0;
            const size_t inc = (size_t)(2 * dpr) * P.n0;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (pr0 + u * dpr < npairs) {
                    const c2 e = z[(pr0 + u * dpr) * pstride + slot];
                    dstp[u * inc] = e.x;
                    dstp[u * inc + P.n0] = e.y;
                }
            }
        } else {
            const int pr = tid & (npairs - 1);
            const int n0_ = tid >> pbits, dn = NT >> pbits;
            double* dstp = P.out + base + 2 * pr + (size_t)n0_ * estride;
            const size_t inc = (size_t)dn * estride;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int n = n0_ + u * dn;
                if (n < N) {
                    const c2 e = z[pr * pstride + (out_is_samples ? dctc::sample_slot(n, N, bits) : dctc::swz(n))];
                    *reinterpret_cast<double2*>(dstp + u * inc) = make_double2(e.x, e.y);
                }
            }
        }
    } else
    for (int q = tid; q < nitems; q += NT) {
        int pr, n;
        decode(q, pr, n);
        const int L0 = 2 * pr;
        if (L0 >= nlines) continue;
        const int slot = out_is_samples ? dctc::sample_slot(n, N, bits) : dctc::swz(n);
        c2 e = z[pr * pstride + slot];
        const bool two = L0 + 1 < nlines;
        if (P.fuse_scale && !P.roundtrip) {
            e.x *= symbol_inv(L0, n);
            if (two) e.y *= symbol_inv(L0 + 1, n);
        }
        double* dstp = P.out + base + (size_t)L0 * lstride + (size_t)n * estride;
        if (P.pairvec && two) {
            *reinterpret_cast<double2*>(dstp) = make_double2(e.x, e.y);
        } else {
            dstp[0] = e.x;
            if (two) dstp[lstride] = e.y;
        }
    }
    (void)total;
}

// Fused schedule for axis >= 1 passes over full power-of-two tiles (host-checked: P.fast, bits >= 6, pairvec): the
// first radix-8 stage takes its inputs straight from global memory and the last one writes straight back, the top
// stage of the forward FFT is merged with the DCT post-processing (and, for the roundtrip pass, with the symbol, the
// inverse pre-processing and the top stage of the inverse FFT) on registers holding every index k together with
// N-k.  A tile therefore makes 2 LDS round trips per transform instead of 5 (dct_core.h: fused_first / fused_mid /
// fused_last).  Work items: (pair, natural group) for the outer stages, (pair, t) for the merged middle.
// 1/d for d > 0 in the normal range: hardware reciprocal + two Newton steps (<= 1 ulp), a third of the cost of the
// IEEE division sequence; the symbol is evaluated twice per grid point and pass, which made the divisions the largest
// single VALU item of the roundtrip pass.
__device__ __forceinline__ double rcp_nr(double d, int steps = 2) {
    double y = __builtin_amdgcn_rcp(d);
    double e = __builtin_fma(-d, y, 1.0);
    y = __builtin_fma(y, e, y);
    if (steps < 2) return y;
    e = __builtin_fma(-d, y, 1.0);
    return __builtin_fma(y, e, y);
}

// Workgroup barrier that orders LDS accesses only: __syncthreads() also drains every outstanding global load
// (s_waitcnt vmcnt(0)), which would put the latency of the prefetched next tile back on the critical path.  Global
// memory needs no ordering inside the fused kernel: a lane consumes its own loads before its LDS writes, and tiles are
// owned by one workgroup.
__device__ __forceinline__ void lds_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// AX0: the transform runs along the contiguous index (a tile is LT consecutive rows); lanes then walk along the row
// (dct_core.h: fused_first2 / fused_last2) instead of across the LT lines.
// One tile per workgroup.  The tile's global loads (first-stage samples; MODE 1: the spectral pairs of the merged
// middle) are requested into registers before the twiddle tables are staged, so that the two latencies overlap.
// (A persistent variant -- two workgroups per CU walking over the tiles with the next tile's samples prefetched across
// the LDS phases -- was measured 5-12 % slower in every pass, also with only 2 tiles per workgroup: docs/history.md section 4;
// it lived in this file up to commit 96342eb.)  NTM: non-temporal tile loads / stores (every element is touched once).
// DOT (MODE 2): the per-tile partial sum of sum_k symbol(k) |v^_k|^2 -- by Parseval (orthonormal transforms on every axis)
// the dot product v . (M^-1 v) of this tile's lines -- goes to P.dotp[workgroup]; costs no memory traffic.
// FZ (AX0 only): the pointwise work of the stencil-free preconditioned operator rides in the x passes (ops.h: DctFuse) -- one more
// 8 B/point read stream in the pass, same tile pipeline.
// SLAB (axis 2 only): 1 = forward half (MODE 0), 2 = inverse half (MODE 1) of the slab z-solve -- see FftK.
// value of the neighbouring lane (lane ^ 1): two DPP moves (quad_perm [1, 0, 3, 2]), no LDS traffic
__device__ __forceinline__ double lane_xor1(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_mov_dpp(lo, 0xB1, 0xF, 0xF, true);
    hi = __builtin_amdgcn_mov_dpp(hi, 0xB1, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ c2 lane_xor1(c2 v) { c2 r; r.x = lane_xor1(v.x); r.y = lane_xor1(v.y); return r; }

// NT = 512 (round 6, the z round trip only): one first- / last-stage item per lane and the merged middle split over lane PAIRS
// (dct_core.h: mid_half_*), 8 complex values per lane instead of 16 -- the tile's phases are latency-bound (one wave per SIMD and
// tile with 256 lanes: 38 % of the wave cycles issue, profiles/r5_sq_stall_breakdown.txt), so the same two tiles per CU now bring 4
// waves per SIMD inside 128 VGPRs.  (Round 3's 512-lane instantiation of the unsplit kernel spilled 264 B per lane and ran 2x slower.)
// FZS (round 6, with FZ, MODE 0): the second stream is ADDED -- sample + fz_cx * fz_v -- instead of entering a pointwise factor, and the
// sum is stored to fz_store (the MINRES recurrence's y <- y + c r riding in the preconditioner's first pass: DctFuse::add).  A separate
// instantiation: the kernels of the corrector's hot path are compiled exactly as before.
template <int NT, int MODE, bool AX0, bool NTM, bool DOT = false, bool FZ = false, int SLAB = 0, bool FZS = false>   // MODE 0: forward, 1: inverse, 2: forward - symbol - inverse (AX0: 0 / 1 only)
__global__ void __launch_bounds__(NT, NT == 512 ? 4 : 2) dct_fused_kernel(FftK P) {
    constexpr bool SPLIT = NT == 512;
    static_assert(!FZS || (FZ && MODE == 0), "FZS: the x-forward pass only");
    static_assert(!SPLIT || (MODE == 2 && !AX0 && !FZ && SLAB == 0), "512 lanes: the z / y round trip only");
    static_assert(!FZ || (AX0 && MODE != 2), "FZ: x passes only");
    static_assert(SLAB == 0 || (!AX0 && !FZ && !DOT && ((SLAB == 1 && MODE == 0) || (SLAB == 2 && MODE == 1))), "SLAB: z halves only");
    __shared__ double dsum[NT / 64];
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int N = P.N, bits = P.bits, G = N >> 3;
    const int npairs = P.LT >> 1, pbits = P.ltbits - 1;
    const int pstride = N + 1;
    c2* z = reinterpret_cast<c2*>(smem);
    const int twl = dctc::tw_len(N), ewl = dctc::ew_len(N);
    c2* tw = z + (size_t)npairs * pstride;                    // N/2 FFT twiddles (dct_core.h: twi layout, twl slots)
    c2* ew = tw + twl;                                        // N/2 + 1 post twiddles exp(-i pi k / 2N), k <= N/2 (ewl slots)
    double* lamk = reinterpret_cast<double*>(ew + ewl + 1);   // MODE 2 / SLAB: eigenvalues along the transform axis
    double* phik = lamk + N;                                  // SLAB: [2][N] local basis at planes 0, 1
    const int tid = threadIdx.x;
    const int nfirst = AX0 ? npairs * (G >> 1) : npairs * G;  // work items of the outer stages ...
    const int nmid = npairs * (G >> 1);                       // ... and of the merged middle
    const int hbits = bits - 4;                               // log2(G / 2)
    // element stride along the transform axis; the host guarantees that the array is < 4 GiB, so that every access is
    // (uniform tile base) + (32-bit per-lane byte offset) -- one VGPR per address instead of two
    const unsigned estride = AX0 ? 1u : (P.axis == 1 ? (unsigned)P.n0 : (unsigned)P.n0 * (unsigned)P.n1);
    const unsigned lstride = AX0 ? (unsigned)P.n0 : 1u;       // line a -> line b of a pair

    // tile -> (x0, other, element offsets of its input / output)
    auto tile_x0 = [&](int tile) { return AX0 ? 0 : (tile % P.tiles_x) * P.LT; };
    auto tile_other = [&](int tile) { return AX0 ? 0 : tile / P.tiles_x; };     // i2 (axis 1) or i1 (axis 2)
    auto tile_base = [&](int tile) {
        const size_t x0 = (size_t)tile_x0(tile), other = (size_t)tile_other(tile);
        return AX0 ? (size_t)tile * P.LT * P.n0 : (P.axis == 1 ? x0 + (size_t)P.n0 * P.n1 * other : x0 + (size_t)P.n0 * other);
    };
    // distributed plan: one side of the y pass lives in the all-to-all block layout (dct.hip, dct_apply_dist)
    auto tile_sbase = [&](int tile) { return (size_t)tile_x0(tile) + (size_t)tile_other(tile) * P.split_plane; };
    auto tile_in = [&](int tile) { return P.in + (!AX0 && P.split == 2 ? tile_sbase(tile) : tile_base(tile)); };
    auto tile_out = [&](int tile) { return P.out + (!AX0 && P.split == 1 ? tile_sbase(tile) : tile_base(tile)); };
    double* gout = nullptr;
    typedef double nt_d2 __attribute__((ext_vector_type(2)));
    auto ld16 = [&](const double* p) {
        c2 r;
        if (NTM) { const nt_d2 t = __builtin_nontemporal_load(reinterpret_cast<const nt_d2*>(p)); r.x = t.x; r.y = t.y; }
        else { const double2 t = *reinterpret_cast<const double2*>(p); r.x = t.x; r.y = t.y; }
        return r;
    };
    auto st16 = [&](double* p, double a, double b) {
        if (NTM) { nt_d2 t; t.x = a; t.y = b; __builtin_nontemporal_store(t, reinterpret_cast<nt_d2*>(p)); }
        else *reinterpret_cast<double2*>(p) = make_double2(a, b);
    };
    auto ldfrom = [&](const double* g, unsigned el) {
        return ld16(reinterpret_cast<const double*>(reinterpret_cast<const char*>(g) + (size_t)(el * 8u)));
    };
    auto stg = [&](unsigned el, c2 v) {
        st16(reinterpret_cast<double*>(reinterpret_cast<char*>(gout) + (size_t)(el * 8u)), v.x, v.y);
    };
    // AX0 accessors: one double of line a / b (merged middle), or the adjacent samples (2j, 2j+1) of both lines
    auto st1 = [&](unsigned el, c2 v) { gout[el] = v.x; gout[el + lstride] = v.y; };
    auto stamp = [&](int i) {
        if (P.trace && tid == 0) P.trace[(size_t)blockIdx.x * 8 + i] = (long long)wall_clock64();
    };


    const double s0 = sqrt(1.0 / N), s2 = sqrt(2.0 / N);
    auto middle = [&](int lh, int R, bool inv) {
        const int gbits = bits - R;
        const int ngr = npairs << gbits;
        for (int w = tid; w < ngr; w += NT) {
            c2* zp = z + (size_t)(w >> gbits) * pstride;
            const int g = w & ((1 << gbits) - 1);
            if (!inv) {
                if (R == 3) dctc::r8_group_fwd(zp, bits, lh, g, tw);
                else if (R == 2) dctc::dit_group<2>(zp, bits, lh, g, tw);
                else dctc::dit_group<1>(zp, bits, lh, g, tw);
            } else {
                if (R == 3) dctc::r8_group_inv(zp, bits, lh, g, tw);
                else if (R == 2) dctc::dif_group_inv<2>(zp, bits, lh, g, tw);
                else dctc::dif_group_inv<1>(zp, bits, lh, g, tw);
            }
        }
        lds_barrier();
    };
    auto nold = [](int, int) { c2 r; r.x = 0.0; r.y = 0.0; return r; };
    auto nosym = [](int) { c2 r; r.x = 0.0; r.y = 0.0; return r; };
    auto nost = [](int, c2) {};

    // first-stage samples held in registers: (AX0) the 8 + 8 sample pairs of this lane's item, line a in pfa / line b in
    // pfb; (axis >= 1) the 8 samples of item tid in pfa and of item tid + NT in pfb
    c2 pfa[8], pfb[SPLIT ? 1 : 8];
    c2 qfa[FZ ? 8 : 1], qfb[FZ ? 8 : 1];     // FZ: the second stream's values at the same indices (u forward, x inverse)
    c2 dl[SLAB == 2 ? 4 : 1];                // SLAB 2: the correction's right-hand side at the four face planes, lines a / b of this item
    // (the host launches this kernel only when nfirst <= NT (AX0) / 2 NT, so the two register sets cover the tile)
    const bool act0 = tid < nfirst, act1 = !AX0 && !SPLIT && tid + NT < nfirst;
    auto issue = [&](int tile) {
        // every lane requests unconditionally (idle lanes re-read element 0 of the tile)
        const double* g = tile_in(tile);
        if (MODE == 1) {                                      // the 8 + 8 spectral pairs of this lane's merged-middle item
            const bool act = tid < nmid;
            const int pr = AX0 ? tid >> hbits : tid & (npairs - 1), t = AX0 ? tid & ((1 << hbits) - 1) : tid >> pbits;
            if (SLAB == 2) {
                // nu = (nu_u, nu_w) of the bottom and the top face for this lane's two lines; delta = -U nu on planes 0, 1, nl-2, nl-1 is
                // formed when the per-line constants are known (below)
                const size_t line = (size_t)tile_other(tile) * P.n0 + (size_t)tile_x0(tile) + (size_t)(2 * pr);
                const size_t d = line / P.face_Lr, l = line - d * P.face_Lr;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const double2 v2 = *reinterpret_cast<const double2*>(P.face_d + (d * 4 + q) * (size_t)P.face_Lr + l);
                    dl[SLAB == 2 ? q : 0].x = v2.x; dl[SLAB == 2 ? q : 0].y = v2.y;
                }
            }
            const unsigned o = AX0 ? (unsigned)(2 * pr) * lstride : 2u * pr;
            const int ga = t, gb = t == 0 ? (G >> 1) : G - t;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int ka = ga + q * G, kb = gb + q * G;
                if (AX0) {
                    const unsigned ea = act ? o + (unsigned)ka : 0u, eb = act ? o + (unsigned)kb : 0u;
                    pfa[q].x = g[ea]; pfa[q].y = g[ea + lstride];
                    pfb[q].x = g[eb]; pfb[q].y = g[eb + lstride];
                } else if (P.split == 2) {
                    pfa[q] = ldfrom(g, act ? o + P.kmap[ka] : 0u);
                    pfb[q] = ldfrom(g, act ? o + P.kmap[kb] : 0u);
                } else {
                    pfa[q] = ldfrom(g, act ? o + (unsigned)ka * estride : 0u);
                    pfb[q] = ldfrom(g, act ? o + (unsigned)kb * estride : 0u);
                }
            }
            return;
        }
        if (AX0) {
            const int gp = tid & ((1 << hbits) - 1), gq = (G - 1) - gp;
            const double* row = g + (act0 ? (size_t)(2 * (tid >> hbits)) * lstride : (size_t)0);
            const int m = act0 ? 2 : 0;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                pfa[2 * r] = ld16(row + m * (gp + G * r));
                pfb[2 * r] = ld16(row + lstride + m * (gp + G * r));
                pfa[2 * r + 1] = ld16(row + m * (gq + G * r));
                pfb[2 * r + 1] = ld16(row + lstride + m * (gq + G * r));
            }
            if (FZ) {
                const double* urow = P.fz_v + tile_base(tile) + (act0 ? (size_t)(2 * (tid >> hbits)) * lstride : (size_t)0);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    qfa[2 * r] = ld16(urow + m * (gp + G * r));
                    qfb[2 * r] = ld16(urow + lstride + m * (gp + G * r));
                    qfa[2 * r + 1] = ld16(urow + m * (gq + G * r));
                    qfb[2 * r + 1] = ld16(urow + lstride + m * (gq + G * r));
                }
            }
        } else {
            const unsigned o = 2u * (tid & (npairs - 1));
            const int g0 = tid >> pbits, g1 = (tid + NT) >> pbits;
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                pfa[r] = ldfrom(g, act0 ? o + (unsigned)dctc::first_sample(g0, r, N) * estride : 0u);
                if (!SPLIT) pfb[SPLIT ? 0 : r] = ldfrom(g, act1 ? o + (unsigned)dctc::first_sample(g1, r, N) * estride : 0u);
            }
        }
    };
    const int ntiles = P.ntiles;
    // Start offset between the two tiles of a CU.  The first 2 x CUs workgroups fill both slots of every CU at launch
    // (dispatch goes round-robin over the 8 XCDs, inside an XCD the first CUs/8 workgroups take the first slot of its CUs, the
    // next CUs/8 the second); tiles that start together run their load / LDS / store phases in lockstep and compete for the
    // same pipe instead of overlapping.  The second-slot workgroups sleep `stagger` x s_sleep(127) (~3.4 us each) first, and
    // the offset persists down the pass because every slot picks up its next tile when its own tile retires.  Measured at the
    // end of round 3 on the z round trip (experiment): 611 -> 564 us although the sleep is part of the kernel's duration.
    // (Not in the inverse kernels, MODE 1.  Measured per pass at 512^3 with the offset at 0, i.e. the branch never taken
    // (profiles/r4_dct_pass_variants.txt): with these two lines in the prologue of EVERY instantiation the y inverse pass takes
    // 483 us instead of 375 -- it then waits on memory for 64 % of its wave cycles instead of 39 % -- while its neighbours in
    // the stream, the z round trip and the y forward pass, run 632 / 347 instead of 655 / 376 us; with the lines only in the
    // forward / round-trip kernels all five are back at round 3's durations.  The sum is what counts: 2173 vs 2130 us.)
    if (MODE != 1 && P.stagger > 0 && (int)blockIdx.x < 2 * P.stag_cu && (int)(blockIdx.x >> 3) >= (P.stag_cu >> 3))
        for (int i = 0; i < P.stagger; ++i) __builtin_amdgcn_s_sleep(127);
    // workgroup slot -> tile.  xmap: the workgroups of one XCD (slot % 8: consecutive workgroups go round-robin over the
    // 8 XCDs) take a contiguous range of tiles, i.e. every XCD's L2 / TLB works on its own eighth of the array -- 10 % on
    // the x / y access pattern (profiles/r2_seg_copy_512_xmap.json), nothing for z (every tile touches every plane)
    auto slot_tile = [&](int slot) { return P.xmap ? (slot & 7) * (ntiles >> 3) + (slot >> 3) : slot; };
    {
        const int tile = slot_tile(blockIdx.x);
        // Order of the loads: the twiddle tables (L2 hits) are staged first, then the tile's samples are requested and every
        // wave consumes its own samples as they arrive.  Requesting the tile first and the tables behind it -- in either
        // form, tables stored before or after the first stage -- costs 20-30 % in every pass (measured): the in-order return
        // puts the short table loads behind 16 HBM loads per lane.
        for (int q = tid; q < twl + ewl; q += NT) tw[q] = reinterpret_cast<const c2*>(P.twid)[q];
        if (MODE == 2 || SLAB)
            for (int q = tid; q < N; q += NT) lamk[q] = (P.axis == 1 ? P.lam1 : P.lam2)[q];
        if (SLAB)
            for (int q = tid; q < 2 * N; q += NT) phik[q] = P.phi[q];
        issue(tile);
        lds_barrier();
        const int x0 = tile_x0(tile), other = tile_other(tile);
        gout = tile_out(tile);
        stamp(0);
        // MODE 2: per-line constants of the inverse symbol, requested here so that the loads are long back when the merged
        // middle needs them.  Same association as the generic kernel, ((1 + lam0) + lam1) + lam2, in one form for both
        // axes: the per-line constant first, the last-axis eigenvalue per k, then lam2 (y pass of a 3-D array) or an exact
        // + 0.0.  (One merged-middle item per lane: nmid <= NT, host-checked.)
        double ca = 0.0, cb = 0.0, lo2 = 0.0;
        if (MODE == 2 || SLAB) {
            const int i0 = x0 + 2 * ((SPLIT ? tid >> 1 : tid) & (npairs - 1));      // (SPLIT: lanes 2w, 2w + 1 share the item w)
            const double l1 = P.axis == 1 ? 0.0 : P.lam1[other];
            lo2 = P.axis == 1 ? (P.lam2 ? P.lam2[other] : 0.0) : 0.0;
            ca = 1.0 + P.lam0[i0] + l1;
            cb = 1.0 + P.lam0[i0 + 1] + l1;
        }
        if (SLAB == 2) {
            // delta_0 = (c - a) nu_u^b + nu_w^b, delta_1 = a nu_u^b (bottom face); delta_{nl-2} = -a nu_u^t, delta_{nl-1} = -((c - a) nu_u^t + nu_w^t)
            constexpr int I1 = SLAB == 2 ? 1 : 0, I2 = SLAB == 2 ? 2 : 0, I3 = SLAB == 2 ? 3 : 0;
            const double a_ = P.slab_a;
            const c2 nub_u = dl[0], nub_w = dl[I1], nut_u = dl[I2], nut_w = dl[I3];
            const double hb = P.slab_hasb ? 1.0 : 0.0, ht = P.slab_hast ? 1.0 : 0.0;
            dl[0].x = hb * ((ca - a_) * nub_u.x + nub_w.x); dl[0].y = hb * ((cb - a_) * nub_u.y + nub_w.y);
            dl[I1].x = hb * (a_ * nub_u.x); dl[I1].y = hb * (a_ * nub_u.y);
            dl[I2].x = -ht * (a_ * nut_u.x); dl[I2].y = -ht * (a_ * nut_u.y);
            dl[I3].x = -ht * ((ca - a_) * nut_u.x + nut_w.x); dl[I3].y = -ht * ((cb - a_) * nut_u.y + nut_w.y);
        }
        if (MODE != 1) {
            if (AX0) {
                if (act0)
                    dctc::fused_first2(z + (size_t)(tid >> hbits) * pstride, N, bits, tid & ((1 << hbits) - 1),
                                       [&](int s, int j, double& ea, double& oa, double& eb, double& ob) {
                                           ea = pfa[s].x; oa = pfa[s].y; eb = pfb[s].x; ob = pfb[s].y;
                                           if (FZS) {
                                               // sample + c * r, product and sum rounded separately (as v_axpbyz forms them: this file
                                               // compiles with contraction on), stored back before it enters the transform
                                               const int t = FZ ? s : 0;
                                               ea = __dadd_rn(__dmul_rn(P.fz_cx, qfa[t].x), ea); oa = __dadd_rn(__dmul_rn(P.fz_cx, qfa[t].y), oa);
                                               eb = __dadd_rn(__dmul_rn(P.fz_cx, qfb[t].x), eb); ob = __dadd_rn(__dmul_rn(P.fz_cx, qfb[t].y), ob);
                                               double* srow = P.fz_store + tile_base(tile) + (size_t)(2 * (tid >> hbits)) * lstride;
                                               st16(srow + 2 * j, ea, oa);
                                               st16(srow + lstride + 2 * j, eb, ob);
                                           } else if (FZ) {
                                               const int t = FZ ? s : 0;
                                               auto d = [&](double u) { return P.fzA + u * (P.fzB + P.fzC * u); };
                                               ea *= d(qfa[t].x); oa *= d(qfa[t].y); eb *= d(qfb[t].x); ob *= d(qfb[t].y);
                                           }
                                       });
            } else {
                c2* zp = z + (size_t)(tid & (npairs - 1)) * pstride;
                if (act0) dctc::fused_first(zp, N, bits, tid >> pbits, [&](int r, int) { return pfa[r]; });
                if (!SPLIT && act1) dctc::fused_first(zp, N, bits, (tid + NT) >> pbits, [&](int r, int) { return pfb[SPLIT ? 0 : r]; });
            }
            lds_barrier();
            stamp(1);
            for (int lh = 3; lh < bits - 3;) {
                const int R = bits - 3 - lh >= 3 ? 3 : bits - 3 - lh;
                middle(lh, R, false);
                lh += R;
            }
            stamp(2);
        }
        c2 dtot;
        dtot.x = dtot.y = 0.0;
        c2 fy[4];                                                  // SLAB 1: this item's share of y at the four face planes (lines a / b)
#pragma unroll
        for (int q = 0; q < 4; ++q) fy[q].x = fy[q].y = 0.0;
        if (MODE == 1) {
            if (tid < nmid) {
                const int pr = AX0 ? tid >> hbits : tid & (npairs - 1), t = AX0 ? tid & ((1 << hbits) - 1) : tid >> pbits;
                dctc::fused_mid<1, false>(z + (size_t)pr * pstride, N, t, tw, ew, s0, s2,
                                          [&](int slot, int k) {
                                              c2 v = slot < 8 ? pfa[slot & 7] : pfb[slot & 7];
                                              if (SLAB == 2) {
                                                  // a^_k = y^_k + sym_k (phi_k(0) d0 + phi_k(1) d1 + (-1)^k (phi_k(1) d2 + phi_k(0) d3))
                                                  const double p0 = phik[k], p1 = phik[N + k], lk = lamk[k];
                                                  const double sa = ca + lk, sb = cb + lk;
                                                  const double fa = rcp_nr(sa * sa + P.shift, 2), fb_ = rcp_nr(sb * sb + P.shift, 2);
                                                  const double sg = (k & 1) ? -1.0 : 1.0;
                                                  constexpr int I1 = SLAB == 2 ? 1 : 0, I2 = SLAB == 2 ? 2 : 0, I3 = SLAB == 2 ? 3 : 0;
                                                  v.x += fa * (p0 * dl[0].x + p1 * dl[I1].x + sg * (p1 * dl[I2].x + p0 * dl[I3].x));
                                                  v.y += fb_ * (p0 * dl[0].y + p1 * dl[I1].y + sg * (p1 * dl[I2].y + p0 * dl[I3].y));
                                              }
                                              return v;
                                          }, nost, nosym, dtot);
            }
            if (FZ) {
                // the values this lane's last stage will be added to, requested now: they travel while the inverse middle stages run
                // (same indices as the forward pass's samples: item tid owns the groups gp and G - 1 - gp of its pair of rows)
                const int gp = tid & ((1 << hbits) - 1), gq = (G - 1) - gp;
                const double* xrow = P.fz_v + tile_base(tile) + (act0 ? (size_t)(2 * (tid >> hbits)) * lstride : (size_t)0);
                const int m = act0 ? 2 : 0;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    qfa[2 * r] = ld16(xrow + m * (gp + G * r));
                    qfb[2 * r] = ld16(xrow + lstride + m * (gp + G * r));
                    qfa[2 * r + 1] = ld16(xrow + m * (gq + G * r));
                    qfb[2 * r + 1] = ld16(xrow + lstride + m * (gq + G * r));
                }
            }
        } else if (SPLIT) {
            // lane pair (2w, 2w + 1) = item w (host-checked: 2 nmid == NT, every lane is active -- the DPP exchanges need no mask)
            const int w = tid >> 1, h = tid & 1;
            const int pr = w & (npairs - 1), t = w >> pbits;
            c2* zp = z + (size_t)pr * pstride;
            auto sym = [&](int k) {
                const double lk = lamk[k];
                const double sa = ca + lk + lo2, sb = cb + lk + lo2;
                c2 r; r.x = rcp_nr(sa * sa + P.shift, 2); r.y = rcp_nr(sb * sb + P.shift, 2); return r;
            };
            c2 v[8], p[4];
            dctc::mid_half_fwd(zp, N, t, h, tw, v);
#pragma unroll
            for (int i = 0; i < 4; ++i) p[i] = lane_xor1(v[4 + i]);
            dctc::mid_half_pairs<DOT>(v, p, N, t, h, ew, s0, s2, sym, dtot);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const c2 r = lane_xor1(p[i]);
                if (t != 0) v[4 + i] = r;
            }
            dctc::mid_half_inv(zp, N, t, h, tw, v);
        } else
        for (int w = tid; w < nmid; w += NT) {
            const int pr = AX0 ? w >> hbits : w & (npairs - 1), t = AX0 ? w & ((1 << hbits) - 1) : w >> pbits;
            const unsigned o = AX0 ? (unsigned)(2 * pr) * lstride : 2u * pr;
            c2* zp = z + (size_t)pr * pstride;
            if (MODE == 2) {
                auto sym = [&](int k) {
                    const double lk = lamk[k];
                    const double sa = ca + lk + lo2, sb = cb + lk + lo2;
                    c2 r; r.x = rcp_nr(sa * sa + P.shift, 2); r.y = rcp_nr(sb * sb + P.shift, 2); return r;
                };
                dctc::fused_mid<2, DOT>(zp, N, t, tw, ew, s0, s2, nold, nost, sym, dtot);
            } else if (MODE == 0 && SLAB == 1) {
                dctc::fused_mid<0, false>(zp, N, t, tw, ew, s0, s2, nold, [&](int k, c2 v) {
                    const double p0 = phik[k], p1 = phik[N + k], lk = lamk[k];
                    const double sa = ca + lk, sb = cb + lk;
                    v.x *= rcp_nr(sa * sa + P.shift, 2); v.y *= rcp_nr(sb * sb + P.shift, 2);      // y^_k = sym_k f^_k
                    const double sg = (k & 1) ? -1.0 : 1.0;
                    fy[0].x = fma(p0, v.x, fy[0].x); fy[0].y = fma(p0, v.y, fy[0].y);            // y(plane 0)
                    fy[1].x = fma(p1, v.x, fy[1].x); fy[1].y = fma(p1, v.y, fy[1].y);            // y(plane 1)
                    fy[2].x = fma(sg * p1, v.x, fy[2].x); fy[2].y = fma(sg * p1, v.y, fy[2].y);  // y(plane N-2)
                    fy[3].x = fma(sg * p0, v.x, fy[3].x); fy[3].y = fma(sg * p0, v.y, fy[3].y);  // y(plane N-1)
                    stg(o + (unsigned)k * estride, v);
                }, nosym, dtot);
            } else if (MODE == 0) {
                if (AX0) dctc::fused_mid<0, false>(zp, N, t, tw, ew, s0, s2, nold, [&](int k, c2 v) { st1(o + (unsigned)k, v); }, nosym, dtot);
                else if (P.split == 1) dctc::fused_mid<0, false>(zp, N, t, tw, ew, s0, s2, nold, [&](int k, c2 v) { stg(o + P.kmap[k], v); }, nosym, dtot);
                else dctc::fused_mid<0, false>(zp, N, t, tw, ew, s0, s2, nold, [&](int k, c2 v) { stg(o + (unsigned)k * estride, v); }, nosym, dtot);
            }
        }
        stamp(3);
        if (MODE == 0 && SLAB == 1) {
            // the items of one line pair (same pr, t = 0 .. G/2 - 1) sit npairs lanes apart: their shares are summed through the tile's
            // LDS area, which is dead once every item has left the merged middle
            lds_barrier();
            double* sc = smem;
            if (tid < nmid) {
#pragma unroll
                for (int q = 0; q < 4; ++q) { sc[(size_t)tid * 8 + 2 * q] = fy[q].x; sc[(size_t)tid * 8 + 2 * q + 1] = fy[q].y; }
            }
            lds_barrier();
            if (tid < npairs) {
                double acc[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) acc[q] = 0.0;
                for (int t = 0; t < (G >> 1); ++t)
#pragma unroll
                    for (int q = 0; q < 8; ++q) acc[q] += sc[(size_t)(tid + npairs * t) * 8 + q];
                // acc = (y0, y1, y_{nl-2}, y_{nl-1}) of lines a / b: the face data of the capacitance system (dct_slab.hip: slab_face_gather),
                // (u'y, w'y) of the bottom face (this rank is the q side) and of the top face (the p side), into the owner's [4][Lr] block
                const size_t line = (size_t)other * P.n0 + (size_t)x0 + (size_t)(2 * tid);
                const size_t d = line / P.face_Lr, l = line - d * P.face_Lr;
                double* o = P.face_y + d * 4 * (size_t)P.face_Lr + l;
                const double a_ = P.slab_a, hb = P.slab_hasb ? 1.0 : 0.0, ht = P.slab_hast ? 1.0 : 0.0;
                *reinterpret_cast<double2*>(o) = make_double2(hb * (-(ca - a_) * acc[0] - a_ * acc[2]), hb * (-(cb - a_) * acc[1] - a_ * acc[3]));
                *reinterpret_cast<double2*>(o + P.face_Lr) = make_double2(-hb * acc[0], -hb * acc[1]);
                *reinterpret_cast<double2*>(o + 2 * (size_t)P.face_Lr) = make_double2(ht * (a_ * acc[4] + (ca - a_) * acc[6]), ht * (a_ * acc[5] + (cb - a_) * acc[7]));
                *reinterpret_cast<double2*>(o + 3 * (size_t)P.face_Lr) = make_double2(ht * acc[6], ht * acc[7]);
            }
        }
        if (MODE == 0) {
            if (P.trace) { __builtin_amdgcn_s_waitcnt(0); stamp(6); }
            return;
        }
        if (MODE == 2 && DOT) {
            double d = dtot.x + dtot.y;
            for (int off = 32; off > 0; off >>= 1) d += __shfl_down(d, off, 64);
            if ((tid & 63) == 0) dsum[tid >> 6] = d;
        }
        lds_barrier();
        if (MODE == 2 && DOT && tid == 0) {
            double d = dsum[0];
            for (int w = 1; w < NT / 64; ++w) d += dsum[w];
            P.dotp[blockIdx.x] = d;
        }
        stamp(4);
        for (int top = bits - 3; top > 3;) {
            const int R = top - 3 >= 3 ? 3 : top - 3;
            middle(top - R, R, true);
            top -= R;
        }
        stamp(5);
        for (int w = tid; w < nfirst; w += NT) {
            if (AX0) {
                const int pr = w >> hbits;
                double* row = gout + (size_t)(2 * pr) * lstride;
                int slot = 0;       // (st2 is called in the order the samples were requested: fully unrolled, a compile-time index)
                dctc::fused_last2(z + (size_t)pr * pstride, N, bits, w & ((1 << hbits) - 1),
                                  [&](int j, double ea, double oa, double eb, double ob) {
                                      if (FZ) {
                                          const int t = FZ ? slot : 0;
                                          ea = P.fz_ct * ea + P.fz_cx * qfa[t].x; oa = P.fz_ct * oa + P.fz_cx * qfa[t].y;
                                          eb = P.fz_ct * eb + P.fz_cx * qfb[t].x; ob = P.fz_ct * ob + P.fz_cx * qfb[t].y;
                                          ++slot;
                                      }
                                      st16(row + 2 * j, ea, oa);
                                      st16(row + lstride + 2 * j, eb, ob);
                                  });
            } else {
                const unsigned o = 2u * (w & (npairs - 1));
                dctc::fused_last(z + (size_t)(w & (npairs - 1)) * pstride, N, bits, w >> pbits,
                                 [&](int n, c2 v) { stg(o + (unsigned)n * estride, v); });
            }
        }
    }
    if (P.trace) { __builtin_amdgcn_s_waitcnt(0); stamp(6); }
}

// lanes per tile of the round-trip pass (option dct_rt_lanes): 512 = lane-pair split of the merged middle.  Measured in round 6
// (profiles/r6_zsplit_512_lanes_ab.txt, same box, 512^3): 708 us against 654 us for the 256-lane kernel under rocprofv3, 91.8 against
// 90.4 ms per corrector step -- 88 VGPRs, no scratch, 4 waves per SIMD, and 8 % SLOWER: the pass is bound by the VALU work of its two FFTs
// (~80 fp64 operations per grid point = 273 us at 100 % issue on 256 CUs x 64 fp64 lanes) rather than by exposed latency, and the
// split adds the DPP exchanges and selects to that work.  Kept as an option, off.
constexpr double kRoundTripLanesDefault = 256.0;

inline int choose_lt(int N, int axis, int n0, size_t rows, bool wide = false) {
    // 16 lines per tile (axis >= 1: one 128-B segment per line element), fewer only if the tile would not fit a
    // 76 KiB LDS budget (two workgroups per CU): LT/2 pairs * (N+1) complex + N/2 twiddles, 16 B each.
    // wide (fused kernel, axis >= 1, short transforms -- the z pass of a multi-GPU slab, 256^3 grids): keep the tile at
    // 16 * 512 points, i.e. LT = 16 * 512 / N lines, so that every lane has work in the outer radix-8 stages and a line
    // element is a 256-B ... 1-KiB segment
    int lt = 16;
    if (wide && axis != 0 && N < 512) {
        lt = 16 * (512 / N);
        if (lt > 128) lt = 128;
        while (lt > 16 && (n0 % lt != 0 || ((size_t)(lt / 2) * (N + 1) + (size_t)(N + 4)) * 16 > 76 * 1024)) lt /= 2;
    }
    while (lt > 2 && ((size_t)(lt / 2) * (N + 1) + (size_t)dctc::tw_len(N)) * 16 > 76 * 1024) lt -= 2;
    if (axis == 0) {
        if ((size_t)lt > rows) lt = (int)((rows + 1) & ~(size_t)1);
    } else {
        const int n0e = (n0 + 1) & ~1;
        if (lt > n0e) lt = n0e;
    }
    return lt < 2 ? 2 : lt;
}

}  // namespace

bool dct_axis_fft_supported(int n) { return n >= 4 && n <= 1024 && (n & (n - 1)) == 0; }

// fused-kernel eligibility of a pass (shared with the distributed plan, which may route one side of its y passes
// through the all-to-all block layout only when the fused kernel runs)
bool dct_axis_fused_ok(bk_ctx* ctx, int n0, int n1, int n2, int axis, const double* in, const double* out, int fuse_scale) {
    const int N = axis == 0 ? n0 : (axis == 1 ? n1 : n2);
    int bits = 0;
    while ((1 << bits) < N) ++bits;
    if ((1 << bits) != N || bits < 6 || bits > 9) return false;
    if (ctx->opt("dct_fused", 256.0) == 0.0 || ctx->opt("dct_fft", 1.0) == 0.0 || fuse_scale == 1) return false;
    if ((size_t)n0 * n1 * n2 * sizeof(double) >= ((size_t)1 << 32)) return false;
    if (n0 % 2 != 0 || (((uintptr_t)in | (uintptr_t)out) & 15) != 0) return false;
    const size_t rows = (size_t)n1 * n2;
    const int LT = choose_lt(N, axis, n0, rows, ctx->opt("dct_lt_wide", 1.0) != 0.0);
    if (LT != 16 && !(axis != 0 && (LT == 32 || LT == 64 || LT == 128))) return false;
    // first-stage work items: one (axis 0) / two (axis >= 1) per lane of the 256-lane workgroup (prefetch registers)
    if ((size_t)(LT / 2) * (N / 8) > (axis == 0 ? (size_t)512 : (size_t)512)) return false;
    if (axis == 0) return fuse_scale != 2 && rows % LT == 0 && ctx->opt("dct_fused_ax0", 1.0) != 0.0;
    return n0 % LT == 0;
}

// the slab z-solve's half passes: the fused z kernel with ONE merged-middle item per lane (per-lane symbol constants) on a slab whose
// face buffers are 16-B aligned per line pair
bool dct_slab_half_ok(bk_ctx* ctx, int n0, int n1, int nl, const double* a, const double* b) {
    if (!dct_axis_fused_ok(ctx, n0, n1, nl, 2, a, b, 0)) return false;
    const int LT = choose_lt(nl, 2, n0, (size_t)n1 * nl, ctx->opt("dct_lt_wide", 1.0) != 0.0);
    return (size_t)(LT / 2) * (nl / 16) <= 256 && (((size_t)n0 * n1) % 2 == 0) && (size_t)n0 * n1 < ((size_t)1 << 32);
}

int dct_axis_fft(bk_ctx* ctx, int n0, int n1, int n2, int axis, int inverse, const double* twid, const double* in,
                 double* out, const double* lam0, const double* lam1, const double* lam2, double shift,
                 int fuse_scale, const DctSplit* split, int* dot_blocks, const DctFuse* fz, const DctSlabHalf* sh) {
    FftK P;
    P.dotp = nullptr;
    P.face_y = nullptr; P.face_d = nullptr; P.phi = nullptr; P.face_Lr = 0; P.slab_a = 0.0; P.slab_hasb = P.slab_hast = 0;
    if (sh) {
        if (!dct_slab_half_ok(ctx, n0, n1, n2, in, out) || axis != 2 || fuse_scale != 0 || split || fz)
            return set_error(ctx, "dct_axis_fft: the slab half passes need the fused z-axis kernel (dct_slab_half_ok)");
        if (sh->Lr % 2 != 0) return set_error(ctx, "dct_axis_fft: slab half passes need an even number of lines per owner");
        P.face_y = sh->face_y; P.face_d = sh->face_d; P.phi = sh->phi; P.face_Lr = (unsigned)sh->Lr; P.slab_a = sh->a;
        P.slab_hasb = sh->has_bottom ? 1 : 0; P.slab_hast = sh->has_top ? 1 : 0;
    }
    P.fz_v = nullptr; P.fzA = 1.0; P.fzB = P.fzC = 0.0; P.fz_cx = 0.0; P.fz_ct = 1.0; P.fz_store = nullptr;
    // the pointwise work this pass is asked to take in: the factor (or the pre-axpy) on a forward pass, the axpy on an inverse one
    const bool want_fzs = fz && !inverse && fz->add != nullptr;
    const bool want_fz = fz && (inverse ? fz->xadd != nullptr : (fz->u != nullptr || want_fzs));
    if (want_fz) {
        if (axis != 0 || fuse_scale != 0 || split || !dct_axis_fused_ok(ctx, n0, n1, n2, 0, in, out, 0) ||
            (((uintptr_t)(inverse ? fz->xadd : (want_fzs ? fz->add : fz->u))) & 15) != 0 ||
            (want_fzs && (fz->u != nullptr || fz->store == nullptr || (((uintptr_t)fz->store) & 15) != 0)))
            return set_error(ctx, "dct_axis_fft: fused pointwise work needs the fused x-axis kernel (pw_fused_ok)");
        if (inverse) { P.fz_v = fz->xadd; P.fz_cx = fz->cx; P.fz_ct = fz->ct; }
        else if (want_fzs) { P.fz_v = fz->add; P.fz_cx = fz->cadd; P.fz_store = fz->store; }
        else { P.fz_v = fz->u; P.fzA = fz->A; P.fzB = fz->B; P.fzC = fz->C; }
    }
    if (dot_blocks) *dot_blocks = 0;
    P.kmap = nullptr; P.split_plane = 0; P.split = 0;
    if (split) {
        if (axis != 1 || fuse_scale != 0 || !dct_axis_fused_ok(ctx, n0, n1, n2, axis, in, out, fuse_scale))
            return set_error(ctx, "dct_axis_fft: block-layout pass needs the fused y-axis kernel");
        P.kmap = split->kmap; P.split_plane = split->plane; P.split = inverse ? 2 : 1;
    }
    P.n0 = n0; P.n1 = n1; P.n2 = n2; P.axis = axis;
    P.N = axis == 0 ? n0 : (axis == 1 ? n1 : n2);
    P.bits = 0;
    while ((1 << P.bits) < P.N) ++P.bits;
    if (!dct_axis_fft_supported(P.N)) return set_error(ctx, "dct_axis_fft: N=%d unsupported", P.N);
    P.inverse = inverse; P.in = in; P.out = out; P.twid = twid;
    P.lam0 = lam0; P.lam1 = lam1; P.lam2 = lam2; P.shift = shift;
    P.fuse_scale = fuse_scale == 1 ? 1 : 0;
    P.roundtrip = fuse_scale == 2 ? 1 : 0;
    const size_t rows = (size_t)n1 * n2;
    const bool fused_ok = dct_axis_fused_ok(ctx, n0, n1, n2, axis, in, out, fuse_scale);
    P.LT = choose_lt(P.N, axis, n0, rows, fused_ok && ctx->opt("dct_lt_wide", 1.0) != 0.0);
    {
        const int lt_opt = (int)ctx->opt("dct_lt", 0.0);      // experiment knob: lines per tile of the axis >= 1 passes
        if (lt_opt >= 2 && axis != 0 && lt_opt <= P.LT) P.LT = lt_opt & ~1;
    }
    P.ltbits = -1;
    for (int b = 1; b <= 7; ++b) if ((1 << b) == P.LT) P.ltbits = b;
    unsigned grid;
    if (axis == 0) {
        P.tiles_x = 0;
        grid = (unsigned)((rows + P.LT - 1) / P.LT);
    } else {
        P.tiles_x = (n0 + P.LT - 1) / P.LT;
        grid = (unsigned)((size_t)P.tiles_x * (axis == 1 ? n2 : n1));
    }
    P.pairvec = (axis != 0 && (n0 % 2 == 0) && (((uintptr_t)in | (uintptr_t)out) & 15) == 0) ? 1 : 0;
    P.fast = 0;            // decided after the thread count is known
    {
        const bool big = (size_t)n0 * n1 * n2 >= ((size_t)1 << 22);
        P.nt_load = big && ctx->opt("dct_nt_load", 1.0) != 0.0;
        P.nt_store = big && ctx->opt("dct_nt_store", 1.0) != 0.0;
    }
    P.trace = nullptr;
    const size_t lds = ((size_t)(P.LT / 2) * (P.N + 1) + (size_t)dctc::tw_len(P.N)) * sizeof(c2);
    // (linsolve2 drives two host threads through here concurrently: the attributes are set exactly once)
    static std::once_flag attr_once;
    static hipError_t attr_err = hipSuccess;
    std::call_once(attr_once, [] {
        const void* fns[] = {reinterpret_cast<const void*>(dct_fft_kernel<256>),
                             reinterpret_cast<const void*>(dct_fft_kernel<512>),
                             reinterpret_cast<const void*>(dct_fused_kernel<256, 0, false, false>),
                             reinterpret_cast<const void*>(dct_fused_kernel<256, 1, false, false>),
                             reinterpret_cast<const void*>(dct_fused_kernel<256, 2, false, false>),
                             reinterpret_cast<const void*>(dct_fused_kernel<256, 0, true, false>),
                             reinterpret_cast<const void*>(dct_fused_kernel<256, 1, true, false>),
                             reinterpret_cast<const void*>(dct_fused_kernel<256, 0, false, true>),
                             reinterpret_cast<const void*>(dct_fused_kernel<256, 1, false, true>),
                             reinterpret_cast<const void*>(dct_fused_kernel<256, 2, false, true>),
                             reinterpret_cast<const void*>(dct_fused_kernel<256, 2, false, false, true>),
                             reinterpret_cast<const void*>(dct_fused_kernel<256, 2, false, true, true>),
                             reinterpret_cast<const void*>(dct_fused_kernel<512, 2, false, false>),
                             reinterpret_cast<const void*>(dct_fused_kernel<512, 2, false, true>),
                             reinterpret_cast<const void*>(dct_fused_kernel<512, 2, false, false, true>),
                             reinterpret_cast<const void*>(dct_fused_kernel<512, 2, false, true, true>),
                             reinterpret_cast<const void*>(dct_fused_kernel<256, 0, true, true>),
                             reinterpret_cast<const void*>(dct_fused_kernel<256, 1, true, true>),
                             reinterpret_cast<const void*>(dct_fused_kernel<256, 0, true, false, false, true>),
                             reinterpret_cast<const void*>(dct_fused_kernel<256, 1, true, false, false, true>),
                             reinterpret_cast<const void*>(dct_fused_kernel<256, 0, true, true, false, true>),
                             reinterpret_cast<const void*>(dct_fused_kernel<256, 1, true, true, false, true>),
                             reinterpret_cast<const void*>(dct_fused_kernel<256, 0, true, false, false, true, 0, true>),
                             reinterpret_cast<const void*>(dct_fused_kernel<256, 0, true, true, false, true, 0, true>),
                             reinterpret_cast<const void*>(dct_fused_kernel<256, 0, false, false, false, false, 1>),
                             reinterpret_cast<const void*>(dct_fused_kernel<256, 1, false, false, false, false, 2>),
                             reinterpret_cast<const void*>(dct_fused_kernel<256, 0, false, true, false, false, 1>),
                             reinterpret_cast<const void*>(dct_fused_kernel<256, 1, false, true, false, false, 2>)};
        for (const void* f : fns) {
            const hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
            if (e != hipSuccess) attr_err = e;
        }
    });
    if (attr_err != hipSuccess) return set_error(ctx, "dct_axis_fft: hipFuncSetAttribute failed: %s", hipGetErrorString(attr_err));
    // 512 lanes per tile when the tile is big enough to feed them (two workgroups per CU => 16 wavefronts)
    const int nt = (int)ctx->opt("dct_threads", (size_t)P.LT * P.N >= 4096 ? 512.0 : 256.0);
    {
        const int npairs = P.LT / 2;
        const size_t nitems = (size_t)npairs * P.N;
        const bool full_tiles = axis == 0 ? (rows % P.LT == 0) : (n0 % P.LT == 0);
        const bool shapes = P.ltbits >= 1 && nitems <= (size_t)nt * 8 &&
                            (axis == 0 ? (nt % P.N == 0) : (nt % npairs == 0 && P.pairvec));
        P.fast = (ctx->opt("dct_fastio", 1.0) != 0.0 && full_tiles && shapes) ? 1 : 0;
    }
    if (fused_ok && (P.LT == 16 || (axis != 0 && (P.LT == 32 || P.LT == 64 || P.LT == 128)))) {
        const size_t ldsf = lds + ((size_t)(dctc::ew_len(P.N) + 1) + (P.roundtrip ? P.N / 2 : 0)) * sizeof(c2) +
                            (sh ? (size_t)3 * P.N * sizeof(double) : 0);     // SLAB: eigenvalues + the two basis rows
        const bool trace = ctx->opt("dct_trace", 0.0) != 0.0;
        P.trace = nullptr;
        if (trace) {
            BK_HIP(ctx, hipMalloc(&P.trace, (size_t)grid * 8 * sizeof(long long)));
            BK_HIP(ctx, hipMemsetAsync(P.trace, 0, (size_t)grid * 8 * sizeof(long long), ctx->stream));
        }
        P.ntiles = (int)grid;
        {
            // per-pass start offset: option dct_stagger >= 0 overrides every pass; else dct_stagger_<axis><mode>
            // (mode 0 forward, 1 inverse, 2 round trip); defaults swept at 512^3 (profiles/r4_dct_stagger_sweep.jsonl)
            const int mode_ = P.roundtrip ? 2 : (P.inverse ? 1 : 0);
            char key[32];
            snprintf(key, sizeof(key), "dct_stagger_%d%d", axis, mode_);
            static const int kDefault[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
            const double all = ctx->opt("dct_stagger", -1.0);
            P.stag_cu = ctx->num_cu;
            P.stagger = (int)(all >= 0.0 ? all : ctx->opt(key, (double)kDefault[axis][mode_]));
            if ((int)grid < 2 * ctx->num_cu || ctx->num_cu % 8 != 0) P.stagger = 0;
        }
        P.xmap = (grid % 8 == 0 && ((int)ctx->opt("dct_xcd_map", 3.0) >> axis & 1)) ? 1 : 0;   // bit per axis; default x, y
        const bool ntm = P.nt_load && P.nt_store;
        const int mode = P.roundtrip ? 2 : (P.inverse ? 1 : 0);
        // the round trip with 512 lanes per tile and the merged middle split over lane pairs (option dct_rt_lanes: 512 / 256):
        // tiles of exactly 512 first-stage items = 256 merged-middle items
        const bool split512 = mode == 2 && axis != 0 && (size_t)(P.LT / 2) * (P.N / 8) == 512 &&
                              ctx->opt("dct_rt_lanes", kRoundTripLanesDefault) == 512.0;
#define BK_DCT_LAUNCH(M, A, T) hipLaunchKernelGGL((dct_fused_kernel<256, M, A, T>), dim3(grid), dim3(256), ldsf, ctx->stream, P)
#define BK_DCT_LAUNCH_FZ(M, T) hipLaunchKernelGGL((dct_fused_kernel<256, M, true, T, false, true>), dim3(grid), dim3(256), ldsf, ctx->stream, P)
#define BK_DCT_LAUNCH_SLAB(M, T, S) hipLaunchKernelGGL((dct_fused_kernel<256, M, false, T, false, false, S>), dim3(grid), dim3(256), ldsf, ctx->stream, P)
        if (sh) {
            if (mode == 1) { if (ntm) BK_DCT_LAUNCH_SLAB(1, true, 2); else BK_DCT_LAUNCH_SLAB(1, false, 2); }
            else { if (ntm) BK_DCT_LAUNCH_SLAB(0, true, 1); else BK_DCT_LAUNCH_SLAB(0, false, 1); }
        } else if (axis == 0 && want_fzs) {
            if (ntm) hipLaunchKernelGGL((dct_fused_kernel<256, 0, true, true, false, true, 0, true>), dim3(grid), dim3(256), ldsf, ctx->stream, P);
            else hipLaunchKernelGGL((dct_fused_kernel<256, 0, true, false, false, true, 0, true>), dim3(grid), dim3(256), ldsf, ctx->stream, P);
        } else if (axis == 0 && want_fz) {
            if (mode == 1) { if (ntm) BK_DCT_LAUNCH_FZ(1, true); else BK_DCT_LAUNCH_FZ(1, false); }
            else { if (ntm) BK_DCT_LAUNCH_FZ(0, true); else BK_DCT_LAUNCH_FZ(0, false); }
        } else if (axis == 0) {
            if (mode == 1) { if (ntm) BK_DCT_LAUNCH(1, true, true); else BK_DCT_LAUNCH(1, true, false); }
            else { if (ntm) BK_DCT_LAUNCH(0, true, true); else BK_DCT_LAUNCH(0, true, false); }
        } else if (mode == 2 && dot_blocks && (size_t)grid <= kPartialDoubles && ctx->opt("dct_fused_dot", 1.0) != 0.0) {
            P.dotp = ctx->d_partials;
            if (split512) {
                if (ntm) hipLaunchKernelGGL((dct_fused_kernel<512, 2, false, true, true>), dim3(grid), dim3(512), ldsf, ctx->stream, P);
                else hipLaunchKernelGGL((dct_fused_kernel<512, 2, false, false, true>), dim3(grid), dim3(512), ldsf, ctx->stream, P);
            } else {
                if (ntm) hipLaunchKernelGGL((dct_fused_kernel<256, 2, false, true, true>), dim3(grid), dim3(256), ldsf, ctx->stream, P);
                else hipLaunchKernelGGL((dct_fused_kernel<256, 2, false, false, true>), dim3(grid), dim3(256), ldsf, ctx->stream, P);
            }
            *dot_blocks = (int)grid;
        } else if (mode == 2 && split512) {
            if (ntm) hipLaunchKernelGGL((dct_fused_kernel<512, 2, false, true>), dim3(grid), dim3(512), ldsf, ctx->stream, P);
            else hipLaunchKernelGGL((dct_fused_kernel<512, 2, false, false>), dim3(grid), dim3(512), ldsf, ctx->stream, P);
        } else if (mode == 2) { if (ntm) BK_DCT_LAUNCH(2, false, true); else BK_DCT_LAUNCH(2, false, false); }
        else if (mode == 1) { if (ntm) BK_DCT_LAUNCH(1, false, true); else BK_DCT_LAUNCH(1, false, false); }
        else { if (ntm) BK_DCT_LAUNCH(0, false, true); else BK_DCT_LAUNCH(0, false, false); }
#undef BK_DCT_LAUNCH
#undef BK_DCT_LAUNCH_FZ
#undef BK_DCT_LAUNCH_SLAB
        BK_HIP(ctx, hipGetLastError());
        if (trace) {
            // phase durations (wall_clock64 ticks of 10 ns) averaged over the tiles: stamps 0 start, 1 first stage done,
            // 2 forward middle, 3 merged middle, 4 barrier, 5 inverse middle, 6 all stores retired
            std::vector<long long> h((size_t)grid * 8);
            BK_HIP(ctx, hipStreamSynchronize(ctx->stream));
            BK_HIP(ctx, hipMemcpy(h.data(), P.trace, h.size() * sizeof(long long), hipMemcpyDeviceToHost));
            BK_HIP(ctx, hipFree(P.trace));
            double acc[8] = {0};
            long long tmin = h[0], tmax = 0;
            for (unsigned b = 0; b < grid; ++b) {
                long long prev = h[(size_t)b * 8];
                tmin = std::min(tmin, prev);
                for (int i = 1; i < 7; ++i) {
                    const long long t = h[(size_t)b * 8 + i];
                    if (t == 0) continue;
                    acc[i] += (double)(t - prev);
                    prev = t;
                    tmax = std::max(tmax, t);
                }
            }
            fprintf(stderr, "dct_trace axis=%d mode=%d tiles=%u span=%.1fus  phases[us]:", axis, P.roundtrip ? 2 : P.inverse, grid,
                    (tmax - tmin) * 0.01);
            for (int i = 1; i < 7; ++i) fprintf(stderr, " %d:%.2f", i, acc[i] / grid * 0.01);
            fprintf(stderr, "\n");
        }
        return 0;
    }
    if (nt == 512) hipLaunchKernelGGL(dct_fft_kernel<512>, dim3(grid), dim3(512), lds, ctx->stream, P);
    else hipLaunchKernelGGL(dct_fft_kernel<256>, dim3(grid), dim3(256), lds, ctx->stream, P);
    BK_HIP(ctx, hipGetLastError());
    return 0;
}

}  // namespace bk
