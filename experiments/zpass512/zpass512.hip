// EXPERIMENT, staged for the next round (NOT linked into libbkhip.so): the z round trip of the spectral preconditioner
// (bifurcationkit.jl_amd/csrc/dct_fast.hip: dct_fused_kernel<256, 2, false, NTM, DOT>) on 512 lanes per 16-line tile.
//
// Why: the product kernel runs 256 lanes per tile, each lane carrying two radix-8 groups through the merged middle
// (post-twiddle, inverse symbol, pre-twiddle): 193-203 VGPRs, two waves per SIMD, 0.44 of the HBM peak where the x / y
// passes reach 0.73-0.80.  A first 512-lane attempt in round 3 kept the two-group middle and spilled
// (128 VGPRs + 264 B scratch, 2x slower).  Here the middle is split across lane PAIRS (dct_split.h; host-replayed bitwise
// against the product's fused_mid<2> by host_check.cpp): one group per lane, the pairing partners exchanged with DPP
// quad-permutes.  Every phase then has one item per lane; the goal is <= 128 VGPRs without scratch = four waves per SIMD.
//
// Build + resource check (no GPU needed):   make            (prints VGPRs / scratch / occupancy of the kernel)
// Run on an MI355X:                          ./zpass512 [n0 n1 reps]   (N = 512 along z; checks a sample of lines against an
//                                            O(N^2) CPU transform, then times the pass with HIP events)
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#pragma clang fp contract(fast)
#include "dct_split.h"

using bk::dctc::c2;
namespace dc = bk::dctc;

struct ZK {
    int n0, n1, N, bits;          // array [N][n1][n0], transform along the slowest index
    const double* in;
    double* out;
    const double* twid;           // [N/2] c2 exp(-2 pi i q / N), then [N/2 + 1] c2 exp(-i pi k / 2N)
    const double* lam0;
    const double* lam1;
    const double* lam2;
    double shift;
    double* dotp;                 // per-workgroup partial sums of sum_k symbol |v^_k|^2, or NULL
    int tiles_x, ntiles, xmap;
    int stagger;                  // > 0: the workgroups that fill the SECOND slot of every CU at launch start that many s_sleep(127) later
};


__device__ __forceinline__ double rcp_nr(double d) {
    double y = __builtin_amdgcn_rcp(d);
    double e = __builtin_fma(-d, y, 1.0);
    y = __builtin_fma(y, e, y);
    e = __builtin_fma(-d, y, 1.0);
    return __builtin_fma(y, e, y);
}

__device__ __forceinline__ void lds_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// value of the neighbouring lane (lane ^ 1): DPP quad_perm [1, 0, 3, 2] on the two halves of the double
__device__ __forceinline__ double xchg1(double v) {
    const long long b = __double_as_longlong(v);
    int lo = (int)(b & 0xffffffffLL), hi = (int)(b >> 32);
    lo = __builtin_amdgcn_update_dpp(0, lo, 0xB1, 0xF, 0xF, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, 0xB1, 0xF, 0xF, true);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
__device__ __forceinline__ c2 xchg(c2 v) { c2 r; r.x = xchg1(v.x); r.y = xchg1(v.y); return r; }

// LT lines per tile on 32 LT lanes: 16 lines = 512 lanes, two tiles (16 waves) per CU; 8 lines = 256 lanes and 45 KiB of LDS,
// three tiles per CU (the x extent of a tile is then a 64-byte segment per plane instead of 128 bytes)
template <int LT, bool DOT>
__global__ void __launch_bounds__(32 * LT, LT == 16 ? 4 : 3) zpass512_kernel(ZK P) {
    constexpr int NPAIRS = LT / 2, NT = 32 * LT, PB = (LT == 16 ? 3 : 2);      // PB = log2(NPAIRS)
    extern __shared__ __attribute__((aligned(16))) double smem[];
    __shared__ double dsum[NT / 64];
    __shared__ double lamx[LT];
    const int N = P.N, bits = P.bits, G = N >> 3;
    const int pstride = N + 1;
    c2* z = reinterpret_cast<c2*>(smem);
    c2* tw = z + (size_t)NPAIRS * pstride;
    c2* ew = tw + (N >> 1);
    double* lamk = reinterpret_cast<double*>(ew + (N >> 1) + 2);
    const int tid = threadIdx.x;
    const unsigned estride = (unsigned)P.n0 * (unsigned)P.n1;
    typedef double nt_d2 __attribute__((ext_vector_type(2)));

    if (P.stagger > 0 && blockIdx.x < 512 && ((blockIdx.x >> 3) & 32))      // 8 XCDs x 32 CUs: slots 256..511 are each CU's second tile
        for (int i = 0; i < P.stagger; ++i) __builtin_amdgcn_s_sleep(127);
    const int slot = blockIdx.x;
    const int tile = P.xmap ? (slot & 7) * (P.ntiles >> 3) + (slot >> 3) : slot;
    const int x0 = (tile % P.tiles_x) * LT, other = tile / P.tiles_x;           // other = y index
    const size_t base = (size_t)x0 + (size_t)P.n0 * other;
    const double* gin = P.in + base;
    double* gout = P.out + base;

    for (int q = tid; q < N + 1; q += NT) tw[q] = reinterpret_cast<const c2*>(P.twid)[q];
    for (int q = tid; q < N; q += NT) lamk[q] = P.lam2[q];
    if (tid < LT) lamx[tid] = 1.0 + P.lam0[x0 + tid] + P.lam1[other];       // per-line constant of the inverse symbol
    // first stage: one radix-8 group per lane, samples straight from global memory
    const int fpair = tid & (NPAIRS - 1), fg = tid >> PB;                      // pair line, first-stage group (0 .. N/8)
    c2 pf[8];
    {
        const unsigned o = 2u * fpair;
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const unsigned el = o + (unsigned)dc::first_sample(fg, r, N) * estride;
            const nt_d2 t = __builtin_nontemporal_load(reinterpret_cast<const nt_d2*>(reinterpret_cast<const char*>(gin) + (size_t)(el * 8u)));
            pf[r].x = t.x; pf[r].y = t.y;
        }
    }
    lds_barrier();
    // merged-middle item of this lane: pair line mp, item t, half h
    const int h = tid & 1, mp = (tid >> 1) & (NPAIRS - 1), t = tid >> (PB + 1);
    dc::fused_first(z + (size_t)fpair * pstride, N, bits, fg, [&](int r, int) { return pf[r]; });
    lds_barrier();
    auto middle = [&](int lh, int R, bool inv) {
        const int gbits = bits - R;
        const int ngr = NPAIRS << gbits;
        for (int w = tid; w < ngr; w += NT) {
            c2* zp = z + (size_t)(w >> gbits) * pstride;
            const int g = w & ((1 << gbits) - 1);
            if (!inv) {
                if (R == 3) dc::r8_group_fwd(zp, bits, lh, g, tw);
                else if (R == 2) dc::dit_group<2>(zp, bits, lh, g, tw);
                else dc::dit_group<1>(zp, bits, lh, g, tw);
            } else {
                if (R == 3) dc::r8_group_inv(zp, bits, lh, g, tw);
                else if (R == 2) dc::dif_group_inv<2>(zp, bits, lh, g, tw);
                else dc::dif_group_inv<1>(zp, bits, lh, g, tw);
            }
        }
        lds_barrier();
    };
    for (int lh = 3; lh < bits - 3;) {
        const int R = bits - 3 - lh >= 3 ? 3 : bits - 3 - lh;
        middle(lh, R, false);
        lh += R;
    }
    // ---- merged middle, one top group per lane
    const double s2 = sqrt(2.0 / N);
    const double hs2 = 0.5 * s2, cc = hs2 * ((1.0 / N) / s2);
    const double ca = lamx[2 * mp], cb = lamx[2 * mp + 1];
    auto sym = [&](int k) {
        const double lk = lamk[k];
        const double sa = ca + lk, sb = cb + lk;
        c2 r; r.x = rcp_nr(sa * sa + P.shift); r.y = rcp_nr(sb * sb + P.shift); return r;
    };
    c2 pacc;
    pacc.x = pacc.y = 0.0;
    {
        c2* zp = z + (size_t)mp * pstride;
        dc::SplitRole role;
        role.self = t == 0; role.sp = t == 0 && h == 0;
        role.g = h == 0 ? t : (role.self ? (G >> 1) : G - t); role.G = G; role.N = N;
        c2 v[8];
        {
            c2 w[7];
            dc::split_load_fwd(zp, N, role.g, tw, v, w);
        }
        dc::split_rotate_in(role.sp, v);
        // one code path for every lane (dct_split.h: SplitRole); the scheduling barriers keep the four steps apart
#define BK_STEP(I)                                                                                            \
        {                                                                                                    \
            const c2 a = v[I], b = v[7 - I];                                                                 \
            const c2 Xa = dc::split_phase1a<DOT, I>(role, a, b, xchg(b), ew, cc, sym, pacc);                 \
            __builtin_amdgcn_sched_barrier(0);                                                               \
            const c2 Xb = dc::split_phase1b<DOT, I>(role, a, b, xchg(a), ew, cc, sym, pacc);                 \
            __builtin_amdgcn_sched_barrier(0);                                                               \
            const c2 qa = xchg(Xa), qb = xchg(Xb);                                                           \
            dc::split_phase2<I>(role, Xa, Xb, qa, qb, ew, v[I], v[7 - I]);                                   \
            __builtin_amdgcn_sched_barrier(0);                                                               \
        }
        BK_STEP(0) BK_STEP(1) BK_STEP(2) BK_STEP(3)
#undef BK_STEP
        dc::split_rotate_out(role.sp, v);
        {
            c2 w[7];
            dc::r8_twiddles(w, role.g, 0, tw);
            dc::split_inv_store(zp, N, role.g, v, w);
        }
    }
    if (DOT) {
        double d = hs2 * hs2 * (pacc.x + pacc.y);
        for (int off = 32; off > 0; off >>= 1) d += __shfl_down(d, off, 64);
        if ((tid & 63) == 0) dsum[tid >> 6] = d;
    }
    lds_barrier();
    if (DOT && tid == 0) {
        double d = dsum[0];
        for (int w = 1; w < NT / 64; ++w) d += dsum[w];
        P.dotp[blockIdx.x] = d;
    }
    for (int top = bits - 3; top > 3;) {
        const int R = top - 3 >= 3 ? 3 : top - 3;
        middle(top - R, R, true);
        top -= R;
    }
    {
        const unsigned o = 2u * fpair;
        dc::fused_last(z + (size_t)fpair * pstride, N, bits, fg, [&](int n, c2 v) {
            nt_d2 t; t.x = v.x; t.y = v.y;
            __builtin_nontemporal_store(t, reinterpret_cast<nt_d2*>(reinterpret_cast<char*>(gout) + (size_t)((o + (unsigned)n * estride) * 8u)));
        });
    }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)


static bool g_stagger_only = false;

template <int LT>
static int run(int n0, int n1, int N, int bits, int reps, size_t total, double* din, double* dout, double* dtab, double* d0, double* d1,
               double* d2, double shift, const std::vector<double>& hin, const std::vector<double>& l0, const std::vector<double>& l1,
               const std::vector<double>& l2) {
    constexpr int NPAIRS = LT / 2, NT = 32 * LT;
    double* ddot;
    ZK P;
    P.n0 = n0; P.n1 = n1; P.N = N; P.bits = bits; P.in = din; P.out = dout; P.twid = dtab; P.lam0 = d0; P.lam1 = d1; P.lam2 = d2;
    P.shift = shift; P.tiles_x = n0 / LT; P.ntiles = P.tiles_x * n1; P.xmap = 0; P.stagger = 0;
    CK(hipMalloc(&ddot, (size_t)P.ntiles * 8));
    P.dotp = ddot;
    CK(hipMemset(dout, 0, total * 8));
    const size_t lds = ((size_t)NPAIRS * (N + 1) + (size_t)(N / 2) + (size_t)(N / 2 + 2)) * sizeof(c2) + (size_t)N * sizeof(double);
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(zpass512_kernel<LT, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(zpass512_kernel<LT, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
    hipLaunchKernelGGL((zpass512_kernel<LT, true>), dim3(P.ntiles), dim3(NT), lds, 0, P);
    CK(hipDeviceSynchronize());
    // ---- check a sample of lines against the O(N^2) orthonormal DCT-II / symbol / DCT-III on the CPU, and the dot
    std::vector<double> hdot(P.ntiles);
    CK(hipMemcpy(hdot.data(), ddot, (size_t)P.ntiles * 8, hipMemcpyDeviceToHost));
    std::vector<double> C((size_t)N * N);
    for (int k = 0; k < N; ++k)
        for (int j = 0; j < N; ++j) C[(size_t)k * N + j] = (k == 0 ? std::sqrt(1.0 / N) : std::sqrt(2.0 / N)) * std::cos(M_PI * (2 * j + 1) * k / (2.0 * N));
    double worst = 0.0, scale = 0.0, dot_cpu = 0.0;
    const size_t plane = (size_t)n0 * n1;
    const int ycheck = 3 % n1;
    std::vector<double> col((size_t)N * LT);
    for (int j = 0; j < N; ++j)                                         // the tile (tile_x 0, y = ycheck): LT lines
        CK(hipMemcpy(col.data() + (size_t)j * LT, dout + (size_t)j * plane + (size_t)ycheck * n0, LT * 8, hipMemcpyDeviceToHost));
    for (int x = 0; x < LT; ++x) {
        std::vector<double> line(N), spec(N), back(N);
        for (int j = 0; j < N; ++j) line[j] = hin[(size_t)j * plane + (size_t)ycheck * n0 + x];
        for (int k = 0; k < N; ++k) { double a = 0.0; for (int j = 0; j < N; ++j) a += C[(size_t)k * N + j] * line[j]; spec[k] = a; }
        for (int k = 0; k < N; ++k) {
            const double sa = 1.0 + l0[x] + l1[ycheck] + l2[k];
            const double f = 1.0 / (sa * sa + shift);
            dot_cpu += f * spec[k] * spec[k];
            spec[k] *= f;
        }
        for (int j = 0; j < N; ++j) { double a = 0.0; for (int k = 0; k < N; ++k) a += C[(size_t)k * N + j] * spec[k]; back[j] = a; }
        for (int j = 0; j < N; ++j) {
            const double got = col[(size_t)j * LT + x];
            worst = std::fmax(worst, std::fabs(got - back[j])); scale = std::fmax(scale, std::fabs(back[j]));
        }
    }
    const double dot_gpu = hdot[(size_t)ycheck * P.tiles_x + 0];
    printf("LT %d check: max|gpu - cpu| = %.3e (scale %.3e), tile dot gpu %.15e cpu %.15e\n", LT, worst, scale, dot_gpu, dot_cpu);
    const bool ok = worst <= 1e-12 * scale && std::fabs(dot_gpu - dot_cpu) <= 1e-12 * std::fabs(dot_cpu);
    // ---- timing
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int nvar = g_stagger_only ? 6 : 4;
    for (int variant = 0; variant < nvar; ++variant) {
        P.xmap = ((variant & 1) || g_stagger_only) && (P.ntiles % 8 == 0);
        const bool dot = !g_stagger_only && (variant & 2);
        static const int kStag[6] = {0, 1, 2, 3, 4, 6};
        P.stagger = g_stagger_only ? kStag[variant] : 0;
        for (int w = 0; w < 3; ++w) {
            if (dot) hipLaunchKernelGGL((zpass512_kernel<LT, true>), dim3(P.ntiles), dim3(NT), lds, 0, P);
            else hipLaunchKernelGGL((zpass512_kernel<LT, false>), dim3(P.ntiles), dim3(NT), lds, 0, P);
        }
        CK(hipEventRecord(e0, 0));
        for (int r = 0; r < reps; ++r) {
            if (dot) hipLaunchKernelGGL((zpass512_kernel<LT, true>), dim3(P.ntiles), dim3(NT), lds, 0, P);
            else hipLaunchKernelGGL((zpass512_kernel<LT, false>), dim3(P.ntiles), dim3(NT), lds, 0, P);
        }
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms = 0.f;
        CK(hipEventElapsedTime(&ms, e0, e1));
        const double us = 1e3 * ms / reps;
        printf("LT %2d (%d lanes) stagger %d xmap %d dot %d: %.1f us per pass, %.2f TB/s (%.3f of 8 TB/s)   [product kernel, 16 lines on 256 lanes: 608-620 us]\n", LT,
               NT, P.stagger, P.xmap, (int)dot, us, 16.0 * total / (us * 1e-6) / 1e12, 16.0 * total / (us * 1e-6) / 8e12);
    }
    CK(hipFree(ddot));
    return ok ? 0 : 1;
}

int main(int argc, char** argv) {
    const int n0 = argc > 1 ? atoi(argv[1]) : 512, n1 = argc > 2 ? atoi(argv[2]) : 512, reps = argc > 3 ? atoi(argv[3]) : 20;
    const int N = 512, bits = 9;
    if (n0 % 16 != 0) { fprintf(stderr, "n0 must be a multiple of 16\n"); return 2; }
    const size_t total = (size_t)n0 * n1 * N;
    if (total * 8 >= ((size_t)1 << 32)) { fprintf(stderr, "array must stay below 4 GiB\n"); return 2; }
    std::vector<double> hin(total), tab(2 * (N / 2) + 2 * (N / 2 + 1)), l0(n0), l1(n1), l2(N);
    unsigned s = 2024u;
    for (auto& x : hin) { s = s * 1664525u + 1013904223u; x = (double)(s >> 8) / (1 << 24) - 0.5; }
    for (int j = 0; j < N / 2; ++j) { tab[2 * j] = std::cos(2.0 * M_PI * j / N); tab[2 * j + 1] = -std::sin(2.0 * M_PI * j / N); }
    for (int k = 0; k <= N / 2; ++k) { tab[N + 2 * k] = std::cos(M_PI * k / (2.0 * N)); tab[N + 2 * k + 1] = -std::sin(M_PI * k / (2.0 * N)); }
    auto lam = [](int k, int n, double ainv) { const double sn = std::sin(M_PI * k / (2.0 * n)); return -4.0 * ainv * sn * sn; };
    for (int i = 0; i < n0; ++i) l0[i] = lam(i, n0, 6.5);
    for (int i = 0; i < n1; ++i) l1[i] = lam(i, n1, 4.9);
    for (int i = 0; i < N; ++i) l2[i] = lam(i, N, 6.5);
    const double shift = 1.0;
    double *din, *dout, *dtab, *d0, *d1, *d2;
    CK(hipMalloc(&din, total * 8)); CK(hipMalloc(&dout, total * 8)); CK(hipMalloc(&dtab, tab.size() * 8));
    CK(hipMalloc(&d0, n0 * 8)); CK(hipMalloc(&d1, n1 * 8)); CK(hipMalloc(&d2, N * 8));
    CK(hipMemcpy(din, hin.data(), total * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(dtab, tab.data(), tab.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(d0, l0.data(), n0 * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(d1, l1.data(), n1 * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(d2, l2.data(), N * 8, hipMemcpyHostToDevice));
    g_stagger_only = argc > 4;                    // 4th argument: only the start-offset experiment (16-line tiles)
    int rc = 0;
    rc |= run<16>(n0, n1, N, bits, reps, total, din, dout, dtab, d0, d1, d2, shift, hin, l0, l1, l2);
    if (!g_stagger_only) rc |= run<8>(n0, n1, N, bits, reps, total, din, dout, dtab, d0, d1, d2, shift, hin, l0, l1, l2);
    return rc;
}
