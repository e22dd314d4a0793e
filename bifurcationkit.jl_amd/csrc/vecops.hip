// BLAS-1 and the fused Krylov-basis kernels (multi-dot V'w, multi-axpy w - V h with fused scale + norm).
// All HBM-bound streaming kernels: 16-byte loads per lane, grid-stride over <= kRedBlocks*2 blocks of
// 256 threads (4 wavefronts of 64), wave64 shuffle reductions -> LDS -> per-block partial -> a
// second tiny kernel (fixed order => bitwise reproducible run to run).
//
// They replace the VectorInterface calls of the reference's Krylov loops (src/BorderedArrays.jl:86-217
// and the orthogonalisation inside KrylovKit / IterativeSolvers).
#include <algorithm>
#include <type_traits>

#include "common.h"
#include "sstep.h"

namespace bk {

namespace {

constexpr int kThreads = 256;

inline int grid_for(size_t n, int per_thread, int max_blocks) {
    size_t b = (n + (size_t)kThreads * per_thread - 1) / ((size_t)kThreads * per_thread);
    if (b < 1) b = 1;
    if (b > (size_t)max_blocks) b = max_blocks;
    return (int)b;
}

inline bool aligned16(const void* p) { return ((uintptr_t)p & 15) == 0; }

// Index range of a streaming kernel's workgroup over n2 16-byte items.  xcd != 0: consecutive workgroups are dispatched
// round-robin over the 8 XCDs, so workgroup b sits on XCD b % 8; each XCD then streams through its own contiguous eighth
// of the vector (its 4 MiB L2 and its TLB see one region instead of 4-KiB pieces of the whole array) -- the grid is a
// multiple of 8 and the eighths are multiples of kThreads items (4 KiB).  Otherwise: the plain grid-stride walk.
struct StreamRange { size_t lo, hi, step; };
__device__ __forceinline__ StreamRange stream_range(size_t n2, unsigned xcd) {
    StreamRange r;
    if (xcd) {
        const size_t x = blockIdx.x & 7u, lb = blockIdx.x >> 3, nb = gridDim.x >> 3;
        const size_t per = ((n2 + 7) / 8 + kThreads - 1) / kThreads * kThreads;
        const size_t b0 = x * per < n2 ? x * per : n2, b1 = (x + 1) * per < n2 ? (x + 1) * per : n2;
        r.lo = b0 + lb * kThreads + threadIdx.x; r.hi = b1; r.step = nb * kThreads;
    } else {
        r.lo = (size_t)blockIdx.x * kThreads + threadIdx.x; r.hi = n2; r.step = (size_t)gridDim.x * kThreads;
    }
    return r;
}

// non-temporal hint only for vectors that cannot live in the caches anyway (>= 32 MiB): the cache-resident 2-D configs keep
// their operands in L2 / Infinity Cache between kernels
inline bool nt_hint(bk_ctx* ctx, size_t n) { return n >= ((size_t)1 << 22) && ctx->opt("nt_hint", 1.0) != 0.0; }

// XCD-blocked streaming (stream_range) for the same big vectors.  Measured at 512^3 (profiles/r2_xcd_streaming_512.jsonl):
// multiaxpy gains 3-4 % from k = 12 streams on and loses 6 % at k = 8, multidot loses 10 % at every k >= 8 -- so the default
// (option vec_xcd_map = 1) blocks only the multiaxpy with k >= 12; 2 = every launch, 0 = never.
inline unsigned xcd_map(bk_ctx* ctx, size_t n, int grid, bool axpy, int k) {
    const int mode = (int)ctx->opt("vec_xcd_map", 1.0);
    if (!(nt_hint(ctx, n) && grid >= 8 && grid % 8 == 0) || mode == 0) return 0u;
    return (mode >= 2 || (axpy && k >= 12)) ? 1u : 0u;
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_down(v, off, 64));
    return v;
}

// Streams that are read exactly once per kernel (Krylov basis vectors, BLAS-1 operands) are loaded with the non-temporal
// hint: measured at 512^3 (profiles/r2_kernel_variants_512_nt_loads.jsonl) multidot 6.3 -> 6.6-6.9 TB/s, multiaxpy
// 5.1-5.3 -> 5.6 TB/s (with two elements per lane in flight).
typedef double nt_d2 __attribute__((ext_vector_type(2)));
template <bool LDNT>
__device__ __forceinline__ double2 ld2(const double* p, size_t i) {
    if (LDNT) {
        const nt_d2 t = __builtin_nontemporal_load(reinterpret_cast<const nt_d2*>(p) + i);
        return make_double2(t.x, t.y);
    }
    return reinterpret_cast<const double2*>(p)[i];
}
__device__ __forceinline__ void st2nt(double* p, size_t i, double2 v) {
    nt_d2 r; r.x = v.x; r.y = v.y;
    __builtin_nontemporal_store(r, reinterpret_cast<nt_d2*>(p) + i);
}

// Grid-stride walk over n2 16-byte items with U items per lane in flight: full iterations carry no bounds guard (the
// body sees a compile-time item count, so all its loads are issued back to back -- a runtime guard per operand or per item
// makes the compiler emit load / s_waitcnt vmcnt(0) pairs, i.e. one exposed memory latency per operand), the ragged end
// runs item by item.  body(integral_constant<int, UU>, first_item, step).
template <int U, class Body>
__device__ __forceinline__ void stream_loop(size_t n2, Body&& body) {
    // a workgroup owns U ADJACENT 4-KiB chunks per iteration (one 16-KiB contiguous burst per stream), not U chunks a whole
    // grid stride apart: measured at 512^3, the strided form costs 25 % (axpby 0.55 vs 0.73 of peak) -- it multiplies the
    // number of DRAM pages the chip has open per stream
    const size_t chunk = (size_t)kThreads * U;
    const size_t gstep = (size_t)gridDim.x * chunk;
    size_t base = (size_t)blockIdx.x * chunk;
    for (; base + chunk <= n2; base += gstep) body(std::integral_constant<int, U>{}, base + threadIdx.x, (size_t)kThreads);
    if (base < n2)
        for (size_t i = base + threadIdx.x; i < n2 && i < base + chunk; i += kThreads) body(std::integral_constant<int, 1>{}, i, (size_t)kThreads);
}

// ------------------------------------------------------------------ elementwise
template <int VEC, bool NTH = false>
__global__ void __launch_bounds__(kThreads) axpbyz_kernel(size_t n, double a, const double* __restrict__ x, double b,
                                                          const double* y, double* z, int has_x, int has_y) {
    const size_t stride = (size_t)gridDim.x * kThreads;
    if (VEC == 2) {
        const size_t n2 = n >> 1;
        // operand presence is wave-uniform: hoist it out of the loop so that each variant's loads issue back to back
        auto run = [&](auto hx, auto hy) {
            constexpr bool HX = decltype(hx)::value, HY = decltype(hy)::value;
            stream_loop<4>(n2, [&](auto uc, size_t i0, size_t st) {
                constexpr int UU = decltype(uc)::value;
                double2 xv[UU], yv[UU];
#pragma unroll
                for (int u = 0; u < UU; ++u) {
                    if (HX) xv[u] = ld2<NTH>(x, i0 + u * st);
                    if (HY) yv[u] = ld2<NTH>(y, i0 + u * st);
                }
#pragma unroll
                for (int u = 0; u < UU; ++u) {
                    double2 r = make_double2(0.0, 0.0);
                    if (HX) { r.x = a * xv[u].x; r.y = a * xv[u].y; }
                    if (HY) { r.x += b * yv[u].x; r.y += b * yv[u].y; }
                    if (NTH) st2nt(z, i0 + u * st, r);
                    else reinterpret_cast<double2*>(z)[i0 + u * st] = r;
                }
            });
        };
        if (has_x && has_y) run(std::true_type{}, std::true_type{});
        else if (has_x) run(std::true_type{}, std::false_type{});
        else if (has_y) run(std::false_type{}, std::true_type{});
        else run(std::false_type{}, std::false_type{});
        if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
            const size_t i = n - 1;
            double r = 0.0;
            if (has_x) r = a * x[i];
            if (has_y) r += b * y[i];
            z[i] = r;
        }
    } else {
        for (size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += stride) {
            double r = 0.0;
            if (has_x) r = a * x[i];
            if (has_y) r += b * y[i];
            z[i] = r;
        }
    }
}

// z = x .* (A + u (B + C u)): the pointwise factor of the stencil-free preconditioned operator where the transform pass cannot
// take it in (dense transforms, tiny grids; solver.hip: ShiftPrecOp, bk_precond::apply_pw)
__global__ void __launch_bounds__(kThreads) pw_scale_kernel(size_t n, const double* __restrict__ x, const double* __restrict__ u,
                                                            double A, double B, double C, double* z) {
    const size_t stride = (size_t)gridDim.x * kThreads;
    for (size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += stride) {
        const double ui = u[i];
        z[i] = x[i] * (A + ui * (B + C * ui));
    }
}

// splitmix64-based uniform [0,1): deterministic in (seed, global index)
__global__ void __launch_bounds__(kThreads) fill_random_kernel(size_t n, size_t goff, unsigned long long seed,
                                                               double* __restrict__ x) {
    const size_t stride = (size_t)gridDim.x * kThreads;
    for (size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += stride) {
        unsigned long long z = seed + 0x9E3779B97F4A7C15ull * (unsigned long long)(goff + i + 1);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        z = z ^ (z >> 31);
        x[i] = (double)(z >> 11) * (1.0 / 9007199254740992.0);
    }
}

// ------------------------------------------------------------------ reductions, stage 1
// NV outputs per block: out[block*NV + j].
template <int VEC, int NY, bool NTH = false>   // dot of x with NY vectors y[0..NY) ; if y==x it is a squared norm
__global__ void __launch_bounds__(kThreads) dot_kernel(size_t n, const double* __restrict__ x,
                                                       const double* __restrict__ y0, const double* __restrict__ y1,
                                                       double* __restrict__ partials) {
    const size_t stride = (size_t)gridDim.x * kThreads;
    double s0 = 0.0, s1 = 0.0;
    if (VEC == 2) {
        const size_t n2 = n >> 1;
        const bool same = (x == y0);                      // squared norm: one stream
        auto run = [&](auto sm_) {
            constexpr bool SAME = decltype(sm_)::value;
            stream_loop<4>(n2, [&](auto uc, size_t i0, size_t st) {
                constexpr int UU = decltype(uc)::value;
                double2 xv[UU], av[UU], bv[UU];
#pragma unroll
                for (int u = 0; u < UU; ++u) {
                    xv[u] = ld2<NTH>(x, i0 + u * st);
                    if (!SAME) av[u] = ld2<NTH>(y0, i0 + u * st);
                    if (NY == 2) bv[u] = ld2<NTH>(y1, i0 + u * st);
                }
#pragma unroll
                for (int u = 0; u < UU; ++u) {
                    const double2 a = SAME ? xv[u] : av[u];
                    s0 = fma(xv[u].x, a.x, s0); s0 = fma(xv[u].y, a.y, s0);
                    if (NY == 2) { s1 = fma(xv[u].x, bv[u].x, s1); s1 = fma(xv[u].y, bv[u].y, s1); }
                }
            });
        };
        if (same) run(std::true_type{}); else run(std::false_type{});
        if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
            s0 = fma(x[n - 1], y0[n - 1], s0);
            if (NY == 2) s1 = fma(x[n - 1], y1[n - 1], s1);
        }
    } else {
        for (size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += stride) {
            s0 = fma(x[i], y0[i], s0);
            if (NY == 2) s1 = fma(x[i], y1[i], s1);
        }
    }
    __shared__ double sm[2][4];
    s0 = wave_sum(s0);
    if (NY == 2) s1 = wave_sum(s1);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) { sm[0][w] = s0; sm[1][w] = s1; }
    __syncthreads();
    if (threadIdx.x == 0) {
        partials[(size_t)blockIdx.x * NY + 0] = (sm[0][0] + sm[0][1]) + (sm[0][2] + sm[0][3]);
        if (NY == 2) partials[(size_t)blockIdx.x * NY + 1] = (sm[1][0] + sm[1][1]) + (sm[1][2] + sm[1][3]);
    }
}

// |x - y|^2 without materialising the difference: the explicit residual check of a converged solve (solver.hip: gmres_core) only
// needs the NORM of b - (a0 + a1 A) x -- two read streams instead of a write plus a read-back (the vector is formed only when the
// check fails and a new cycle starts from it)
template <int VEC, bool NTH = false>
__global__ void __launch_bounds__(kThreads) diff_nrm2_kernel(size_t n, const double* __restrict__ x, const double* __restrict__ y,
                                                             double* __restrict__ partials) {
    double s0 = 0.0;
    if (VEC == 2) {
        stream_loop<4>(n >> 1, [&](auto uc, size_t i0, size_t st) {
            constexpr int UU = decltype(uc)::value;
            double2 xv[UU], yv[UU];
#pragma unroll
            for (int u = 0; u < UU; ++u) { xv[u] = ld2<NTH>(x, i0 + u * st); yv[u] = ld2<NTH>(y, i0 + u * st); }
#pragma unroll
            for (int u = 0; u < UU; ++u) {
                const double dx = xv[u].x - yv[u].x, dy = xv[u].y - yv[u].y;
                s0 = fma(dx, dx, s0); s0 = fma(dy, dy, s0);
            }
        });
        if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) { const double d = x[n - 1] - y[n - 1]; s0 = fma(d, d, s0); }
    } else {
        const size_t stride = (size_t)gridDim.x * kThreads;
        for (size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += stride) { const double d = x[i] - y[i]; s0 = fma(d, d, s0); }
    }
    __shared__ double sm[4];
    s0 = wave_sum(s0);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = s0;
    __syncthreads();
    if (threadIdx.x == 0) partials[blockIdx.x] = (sm[0] + sm[1]) + (sm[2] + sm[3]);
}

// ------------------------------------------------------------------ fused MINRES passes (solver.hip: minres_core)
// y <- y + c r (has_r), partial of z . y (after the update): the Lanczos three-term update and its alpha in one pass.
template <int VEC, bool NTH = false>
__global__ void __launch_bounds__(kThreads) axpy_dot_kernel(size_t n, double c, const double* __restrict__ r, int has_r,
                                                            double* y, const double* __restrict__ z,
                                                            double* __restrict__ partials) {
    const size_t stride = (size_t)gridDim.x * kThreads;
    double s = 0.0;
    if (VEC == 2) {
        const size_t n2 = n >> 1;
        auto run = [&](auto hr) {
            constexpr bool HR = decltype(hr)::value;
            stream_loop<2>(n2, [&](auto uc, size_t i0, size_t st) {
                constexpr int UU = decltype(uc)::value;
                double2 yv[UU], rv[UU], zv[UU];
#pragma unroll
                for (int u = 0; u < UU; ++u) {
                    yv[u] = ld2<NTH>(y, i0 + u * st);
                    if (HR) rv[u] = ld2<NTH>(r, i0 + u * st);
                    zv[u] = ld2<NTH>(z, i0 + u * st);
                }
#pragma unroll
                for (int u = 0; u < UU; ++u) {
                    if (HR) {
                        yv[u].x = fma(c, rv[u].x, yv[u].x); yv[u].y = fma(c, rv[u].y, yv[u].y);
                        if (NTH) st2nt(y, i0 + u * st, yv[u]);
                        else reinterpret_cast<double2*>(y)[i0 + u * st] = yv[u];
                    }
                    s = fma(zv[u].x, yv[u].x, s); s = fma(zv[u].y, yv[u].y, s);
                }
            });
        };
        if (has_r) run(std::true_type{}); else run(std::false_type{});
        if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
            double yv = y[n - 1];
            if (has_r) { yv = fma(c, r[n - 1], yv); y[n - 1] = yv; }
            s = fma(z[n - 1], yv, s);
        }
    } else {
        for (size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += stride) {
            double yv = y[i];
            if (has_r) { yv = fma(c, r[i], yv); y[i] = yv; }
            s = fma(z[i], yv, s);
        }
    }
    __shared__ double sm[4];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partials[blockIdx.x] = (sm[0] + sm[1]) + (sm[2] + sm[3]);
}

// w <- cz z + c1 w1 + c2 w2 ;  x <- x + phi w : the MINRES direction and solution updates in one pass
template <int VEC, bool NTH = false>
__global__ void __launch_bounds__(kThreads) minres_update_kernel(size_t n, double cz, const double* __restrict__ z, double c1,
                                                                 const double* __restrict__ w1, double c2,
                                                                 const double* __restrict__ w2, double* __restrict__ w,
                                                                 double phi, double* __restrict__ x) {
    const size_t stride = (size_t)gridDim.x * kThreads;
    if (VEC == 2) {
        const size_t n2 = n >> 1;
        stream_loop<2>(n2, [&](auto uc, size_t i0, size_t st) {
            constexpr int UU = decltype(uc)::value;
            double2 zv[UU], a[UU], b[UU], xv[UU];
#pragma unroll
            for (int u = 0; u < UU; ++u) {
                zv[u] = ld2<NTH>(z, i0 + u * st);
                a[u] = ld2<NTH>(w1, i0 + u * st);
                b[u] = ld2<NTH>(w2, i0 + u * st);
                xv[u] = ld2<NTH>(x, i0 + u * st);
            }
#pragma unroll
            for (int u = 0; u < UU; ++u) {
                double2 wv;
                wv.x = fma(c2, b[u].x, fma(c1, a[u].x, cz * zv[u].x));
                wv.y = fma(c2, b[u].y, fma(c1, a[u].y, cz * zv[u].y));
                xv[u].x = fma(phi, wv.x, xv[u].x); xv[u].y = fma(phi, wv.y, xv[u].y);
                if (NTH) { st2nt(w, i0 + u * st, wv); st2nt(x, i0 + u * st, xv[u]); }
                else { reinterpret_cast<double2*>(w)[i0 + u * st] = wv; reinterpret_cast<double2*>(x)[i0 + u * st] = xv[u]; }
            }
        });
        if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
            const size_t i = n - 1;
            const double wv = fma(c2, w2[i], fma(c1, w1[i], cz * z[i]));
            w[i] = wv;
            x[i] = fma(phi, wv, x[i]);
        }
    } else {
        for (size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += stride) {
            const double wv = fma(c2, w2[i], fma(c1, w1[i], cz * z[i]));
            w[i] = wv;
            x[i] = fma(phi, wv, x[i]);
        }
    }
}

__global__ void __launch_bounds__(kThreads) absmax_kernel(size_t n, const double* __restrict__ x,
                                                          double* __restrict__ partials) {
    const size_t stride = (size_t)gridDim.x * kThreads;
    double m = 0.0;
    for (size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += stride) m = fmax(m, fabs(x[i]));
    __shared__ double sm[4];
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) partials[blockIdx.x] = fmax(fmax(sm[0], sm[1]), fmax(sm[2], sm[3]));
}

// ------------------------------------------------------------------ fused multi-dot
// partials[block][j] = sum_chunk V_j . w  (j < k), partials[block][k] = sum_chunk w . w.
// KB = compile-time bucket >= k: accumulators stay in registers, loads of absent vectors are skipped
// by a wave-uniform predicate.  Every thread keeps KB+1 independent load streams in flight.
template <int KB, int VEC, bool LDNT = false>
__global__ void __launch_bounds__(kThreads) multidot_kernel(size_t n, const double* __restrict__ V, size_t ldv, int k,
                                                            const double* __restrict__ w,
                                                            double* __restrict__ partials, const double* gate = nullptr,
                                                            unsigned xcd = 0) {
    if (gate && gate[0] == 0.0) return;        // device-resident Arnoldi: the DGKS second pass is not needed
    double acc[KB];
#pragma unroll
    for (int j = 0; j < KB; ++j) acc[j] = 0.0;
    double ww = 0.0;
    const size_t stride = (size_t)gridDim.x * kThreads;
    if (VEC == 2) {
        const StreamRange rg = stream_range(n >> 1, xcd);
        for (size_t i = rg.lo; i < rg.hi; i += rg.step) {
            const double2 wv = ld2<LDNT>(w, i);
            ww = fma(wv.x, wv.x, ww); ww = fma(wv.y, wv.y, ww);
#pragma unroll
            for (int j = 0; j < KB; ++j) {
                if (j < k) {
                    const double2 vv = ld2<LDNT>(V + (size_t)j * ldv, i);
                    acc[j] = fma(vv.x, wv.x, acc[j]);
                    acc[j] = fma(vv.y, wv.y, acc[j]);
                }
            }
        }
        if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
            const double wv = w[n - 1];
            ww = fma(wv, wv, ww);
#pragma unroll
            for (int j = 0; j < KB; ++j)
                if (j < k) acc[j] = fma(V[(size_t)j * ldv + n - 1], wv, acc[j]);
        }
    } else {
        for (size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += stride) {
            const double wv = w[i];
            ww = fma(wv, wv, ww);
#pragma unroll
            for (int j = 0; j < KB; ++j)
                if (j < k) acc[j] = fma(V[(size_t)j * ldv + i], wv, acc[j]);
        }
    }
    __shared__ double sm[4][KB + 1];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#pragma unroll
    for (int j = 0; j < KB; ++j) {
        const double s = wave_sum(acc[j]);
        if (lane == 0) sm[wid][j] = s;
    }
    {
        const double s = wave_sum(ww);
        if (lane == 0) sm[wid][KB] = s;
    }
    __syncthreads();
    for (int j = threadIdx.x; j <= k; j += kThreads) {
        const int src = (j == k) ? KB : j;
        partials[(size_t)blockIdx.x * (k + 1) + j] = (sm[0][src] + sm[1][src]) + (sm[2][src] + sm[3][src]);
    }
}

// ------------------------------------------------------------------ fused multi-axpy (+scale, +norm)
// dst = scale * (src + sum_{j<k} c[j] V_j);  partials[block] = sum_chunk dst^2 (if want_norm).
// DEV: coefficients, scale and the gate of the DGKS second pass come from device memory (dcoef: arnoldi_coef_kernel's
// layout) instead of the by-value argument -- the device-resident Arnoldi step; same body, same access pattern.
template <int KB, int VEC, bool NT = false, bool LDNT = false, int U = 1, bool DEV = false>
__global__ void __launch_bounds__(kThreads) multiaxpy_kernel(size_t n, const double* __restrict__ V, size_t ldv, int k,
                                                             Coefs cf, const double* src, double scale, double* dst,
                                                             int want_norm, double* __restrict__ partials, unsigned xcd = 0,
                                                             const double* __restrict__ dcoef = nullptr, int gated = 0) {
    if (DEV) {
        if (gated && dcoef[kMaxBasis + 1] == 0.0) return;
        scale = dcoef[kMaxBasis];
#pragma unroll
        for (int j = 0; j < KB; ++j) cf.c[j] = j < k ? dcoef[j] : 0.0;
    }
    double nn = 0.0;
    if (VEC == 2) {
        const StreamRange rg = stream_range(n >> 1, xcd);
        const size_t n2 = rg.hi, stride = rg.step;
        for (size_t i0 = rg.lo; i0 < n2; i0 += stride * U) {
            double2 r[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const size_t i = i0 + u * stride;
                r[u] = (src && i < n2) ? ld2<LDNT>(src, i) : make_double2(0.0, 0.0);
            }
#pragma unroll
            for (int j = 0; j < KB; ++j) {
                if (j < k) {
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const size_t i = i0 + u * stride;
                        if (U == 1 || i < n2) {
                            const double2 vv = ld2<LDNT>(V + (size_t)j * ldv, i);
                            r[u].x = fma(cf.c[j], vv.x, r[u].x);
                            r[u].y = fma(cf.c[j], vv.y, r[u].y);
                        }
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const size_t i = i0 + u * stride;
                if (U > 1 && i >= n2) break;
                r[u].x *= scale; r[u].y *= scale;
                if (NT) {
                    nt_d2 rr; rr.x = r[u].x; rr.y = r[u].y;
                    __builtin_nontemporal_store(rr, reinterpret_cast<nt_d2*>(dst) + i);
                }
                else reinterpret_cast<double2*>(dst)[i] = r[u];
                nn = fma(r[u].x, r[u].x, nn); nn = fma(r[u].y, r[u].y, nn);
            }
        }
        if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
            const size_t i = n - 1;
            double r = src ? src[i] : 0.0;
#pragma unroll
            for (int j = 0; j < KB; ++j)
                if (j < k) r = fma(cf.c[j], V[(size_t)j * ldv + i], r);
            r *= scale;
            dst[i] = r;
            nn = fma(r, r, nn);
        }
    } else {
        const size_t stride = (size_t)gridDim.x * kThreads;
        for (size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += stride) {
            double r = src ? src[i] : 0.0;
#pragma unroll
            for (int j = 0; j < KB; ++j)
                if (j < k) r = fma(cf.c[j], V[(size_t)j * ldv + i], r);
            r *= scale;
            dst[i] = r;
            nn = fma(r, r, nn);
        }
    }
    if (want_norm) {
        __shared__ double sm[4];
        nn = wave_sum(nn);
        if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = nn;
        __syncthreads();
        if (threadIdx.x == 0) partials[blockIdx.x] = (sm[0] + sm[1]) + (sm[2] + sm[3]);
    }
}

// ------------------------------------------------------------------ bucketed kernels, contiguous bursts per stream
// Same one-stream-at-a-time structure as the bucketed kernels above (the guard per vector keeps the streams apart in time,
// which is what the DRAM likes: measured), but a workgroup takes U ADJACENT 4-KiB chunks of the stream per visit instead of
// one (multidot) or two a whole grid stride apart (multiaxpy): U x 4 KiB contiguous per stream and workgroup.
// GRAM: the same pass also returns g = V' V_{k-1}, the Gram column of the NEWEST basis vector (V_{k-1} is one of the streams
// anyway: it is loaded first and every other V_j is dotted with both w and it) -- what the Gram-corrected single-pass
// Arnoldi step of solver.hip needs, at no extra memory traffic.  Per block: a_0..a_{k-1}, w'w, then g_0..g_{k-1}.
// vn != NULL (GRAM only): the newest vector lives OUTSIDE the k vectors of this launch (a basis of more than 32 vectors is
// visited in two launches); every V_j of the launch is then dotted with both w and it, and g has no diagonal entry.
template <int KB, int U, bool LDNT, bool GRAM = false>
__global__ void __launch_bounds__(kThreads) multidot_c_kernel(size_t n, const double* __restrict__ V, size_t ldv, int k,
                                                              const double* __restrict__ w, double* __restrict__ partials,
                                                              const double* gate, const double* __restrict__ vn = nullptr) {
    if (gate && gate[0] == 0.0) return;
    double acc[KB], accg[GRAM ? KB : 1];
#pragma unroll
    for (int j = 0; j < KB; ++j) acc[j] = 0.0;
#pragma unroll
    for (int j = 0; j < (GRAM ? KB : 1); ++j) accg[j] = 0.0;
    double ww = 0.0;
    const bool ext = GRAM && vn != nullptr;
    const int kl = (GRAM && !ext) ? k - 1 : k;       // vectors visited by the loop (GRAM: all but the newest, if it is one of them)
    const double* __restrict__ Vn = ext ? vn : V + (size_t)(k - 1) * ldv;
    stream_loop<U>(n >> 1, [&](auto uc, size_t i0, size_t st) {
        constexpr int UU = decltype(uc)::value;
        double2 wv[UU], gv[GRAM ? UU : 1];
#pragma unroll
        for (int u = 0; u < UU; ++u) wv[u] = ld2<LDNT>(w, i0 + u * st);
        if (GRAM) {
#pragma unroll
            for (int u = 0; u < UU; ++u) gv[u] = ld2<LDNT>(Vn, i0 + u * st);
        }
#pragma unroll
        for (int u = 0; u < UU; ++u) { ww = fma(wv[u].x, wv[u].x, ww); ww = fma(wv[u].y, wv[u].y, ww); }
#pragma unroll
        for (int j = 0; j < KB; ++j) {
            if (j < kl) {
                double2 vv[UU];
#pragma unroll
                for (int u = 0; u < UU; ++u) vv[u] = ld2<LDNT>(V + (size_t)j * ldv, i0 + u * st);
#pragma unroll
                for (int u = 0; u < UU; ++u) {
                    acc[j] = fma(vv[u].x, wv[u].x, acc[j]); acc[j] = fma(vv[u].y, wv[u].y, acc[j]);
                    if (GRAM) { accg[j] = fma(vv[u].x, gv[u].x, accg[j]); accg[j] = fma(vv[u].y, gv[u].y, accg[j]); }
                }
            } else if (GRAM && !ext && j == kl) {
#pragma unroll
                for (int u = 0; u < UU; ++u) {
                    acc[j] = fma(gv[u].x, wv[u].x, acc[j]); acc[j] = fma(gv[u].y, wv[u].y, acc[j]);
                    accg[j] = fma(gv[u].x, gv[u].x, accg[j]); accg[j] = fma(gv[u].y, gv[u].y, accg[j]);
                }
            }
        }
    });
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
        const double wv = w[n - 1], gv = GRAM ? Vn[n - 1] : 0.0;
        ww = fma(wv, wv, ww);
#pragma unroll
        for (int j = 0; j < KB; ++j)
            if (j < k) {
                const double vj = V[(size_t)j * ldv + n - 1];
                acc[j] = fma(vj, wv, acc[j]);
                if (GRAM) accg[j] = fma(vj, gv, accg[j]);
            }
    }
    __shared__ double sm[4][(GRAM ? 2 : 1) * KB + 1];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#pragma unroll
    for (int j = 0; j < KB; ++j) {
        const double s_ = wave_sum(acc[j]);
        if (lane == 0) sm[wid][j] = s_;
    }
    {
        const double s_ = wave_sum(ww);
        if (lane == 0) sm[wid][KB] = s_;
    }
    if (GRAM) {
#pragma unroll
        for (int j = 0; j < KB; ++j) {
            const double s_ = wave_sum(accg[j]);
            if (lane == 0) sm[wid][KB + 1 + j] = s_;
        }
    }
    __syncthreads();
    const int nv = GRAM ? 2 * k + 1 : k + 1;
    for (int j = threadIdx.x; j < nv; j += kThreads) {
        const int src = j < k ? j : (j == k ? KB : KB + 1 + (j - k - 1));
        partials[(size_t)blockIdx.x * nv + j] = (sm[0][src] + sm[1][src]) + (sm[2][src] + sm[3][src]);
    }
}

template <int KB, int U, bool NT, bool LDNT, bool DEV>
__global__ void __launch_bounds__(kThreads) multiaxpy_c_kernel(size_t n, const double* __restrict__ V, size_t ldv, int k, Coefs cf,
                                                               const double* src, double scale, double* dst, int want_norm,
                                                               double* __restrict__ partials,
                                                               const double* __restrict__ dcoef, int gated, int stag) {
    if (DEV) {
        if (gated && dcoef[kMaxBasis + 1] == 0.0) return;
        scale = dcoef[kMaxBasis];
#pragma unroll
        for (int j = 0; j < KB; ++j) cf.c[j] = j < k ? dcoef[j] : 0.0;
    }
    // Start offsets (option axpy_stagger; low byte = phases P, next byte = map, rest = sleep units per burst).  Every
    // workgroup alternates k + 1 read bursts with ONE write burst per iteration; launched together they stay in step, and the
    // DRAM sees the read and the write phases of the whole chip alternate instead of a steady mix.  Workgroup b starts
    // (phase / P) of one iteration late: phase = (b / 8) % P (alternating CUs of an XCD; map 0) or (b / 256) % P (the two
    // workgroups of a CU; map 1); one burst of the chip takes `units` x s_sleep(1).
    if (stag > 0) {
        const int P = stag & 255, map = (stag >> 8) & 255, units = stag >> 16;
        const int ph = (map == 0 ? (int)(blockIdx.x >> 3) : (int)(blockIdx.x >> 8)) % P;
        const int nburst = ((k + 2) * ph) / P;
        for (int i = 0; i < nburst * units; ++i) __builtin_amdgcn_s_sleep(1);
    }
    double nn = 0.0;
    stream_loop<U>(n >> 1, [&](auto uc, size_t i0, size_t st) {
        constexpr int UU = decltype(uc)::value;
        double2 r[UU];
#pragma unroll
        for (int u = 0; u < UU; ++u) r[u] = src ? ld2<LDNT>(src, i0 + u * st) : make_double2(0.0, 0.0);
#pragma unroll
        for (int j = 0; j < KB; ++j) {
            if (j < k) {
                double2 vv[UU];
#pragma unroll
                for (int u = 0; u < UU; ++u) vv[u] = ld2<LDNT>(V + (size_t)j * ldv, i0 + u * st);
#pragma unroll
                for (int u = 0; u < UU; ++u) { r[u].x = fma(cf.c[j], vv[u].x, r[u].x); r[u].y = fma(cf.c[j], vv[u].y, r[u].y); }
            }
        }
#pragma unroll
        for (int u = 0; u < UU; ++u) {
            r[u].x *= scale; r[u].y *= scale;
            if (NT) st2nt(dst, i0 + u * st, r[u]);
            else reinterpret_cast<double2*>(dst)[i0 + u * st] = r[u];
            nn = fma(r[u].x, r[u].x, nn); nn = fma(r[u].y, r[u].y, nn);
        }
    });
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
        const size_t i = n - 1;
        double r = src ? src[i] : 0.0;
#pragma unroll
        for (int j = 0; j < KB; ++j)
            if (j < k) r = fma(cf.c[j], V[(size_t)j * ldv + i], r);
        r *= scale;
        dst[i] = r;
        nn = fma(r, r, nn);
    }
    if (want_norm) {
        __shared__ double sm[4];
        nn = wave_sum(nn);
        if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = nn;
        __syncthreads();
        if (threadIdx.x == 0) partials[blockIdx.x] = (sm[0] + sm[1]) + (sm[2] + sm[3]);
    }
}

// ------------------------------------------------------------------ block Arnoldi step (sstep.h; solver.hip: arnoldi_block)
// Pass 1 of a block: the dots of up to KB basis vectors V_i with up to 8 RIGHT-HAND vectors held in registers -- the u basis
// vectors the previous block created (their Gram columns are still unmeasured) followed by the s new block vectors; all of them
// are consecutive basis slots starting at Rv -- and (TRI) the upper triangle of the right-hand vectors' own dots.  One visit of
// every stream for what the single-vector step read s times.  Per workgroup: KB x 8 values [i * 8 + r], then the 36 triangle
// values (sstep::tri order).  Unused right-hand slots (r >= nr) are never loaded and contribute exact zeros.
// NRT = right-hand slots of this instantiation (8, or 5 for the usual LAST block of a solve -- one step plus the previous block's four
// unmeasured vectors -- which then fits 8 basis vectors AND the triangle into one launch: 55 accumulators); per workgroup KB x NRT
// values [i * NRT + r], then the NRT (NRT + 1) / 2 triangle values in row order.
template <int KB, bool TRI, int U, bool LDNT, int NRT = sstep::kR>
__global__ void __launch_bounds__(kThreads) block_dots_kernel(size_t n, const double* __restrict__ V, size_t ldv, int kb,
                                                              const double* __restrict__ Rv, int nr, double* __restrict__ partials) {
    constexpr int NR = NRT, NT3 = TRI ? NRT * (NRT + 1) / 2 : 1;
    double acc[KB > 0 ? KB : 1][NR], tri[NT3];
#pragma unroll
    for (int i = 0; i < (KB > 0 ? KB : 1); ++i)
#pragma unroll
        for (int r = 0; r < NR; ++r) acc[i][r] = 0.0;
#pragma unroll
    for (int t = 0; t < NT3; ++t) tri[t] = 0.0;
    stream_loop<U>(n >> 1, [&](auto uc, size_t i0, size_t st) {
        constexpr int UU = decltype(uc)::value;
        double2 rv[NR][UU];
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            if (r < nr) {
#pragma unroll
                for (int u = 0; u < UU; ++u) rv[r][u] = ld2<LDNT>(Rv + (size_t)r * ldv, i0 + u * st);
            } else {
#pragma unroll
                for (int u = 0; u < UU; ++u) rv[r][u] = make_double2(0.0, 0.0);
            }
        }
        if (TRI) {
#pragma unroll
            for (int r = 0; r < NR; ++r)
#pragma unroll
                for (int c = r; c < NR; ++c) {
                    double a = tri[TRI ? r * NR - r * (r - 1) / 2 + (c - r) : 0];
#pragma unroll
                    for (int u = 0; u < UU; ++u) { a = fma(rv[r][u].x, rv[c][u].x, a); a = fma(rv[r][u].y, rv[c][u].y, a); }
                    tri[TRI ? r * NR - r * (r - 1) / 2 + (c - r) : 0] = a;
                }
        }
#pragma unroll
        for (int i = 0; i < KB; ++i) {
            if (i < kb) {
                double2 vv[UU];
#pragma unroll
                for (int u = 0; u < UU; ++u) vv[u] = ld2<LDNT>(V + (size_t)i * ldv, i0 + u * st);
#pragma unroll
                for (int r = 0; r < NR; ++r)
#pragma unroll
                    for (int u = 0; u < UU; ++u) { acc[i][r] = fma(vv[u].x, rv[r][u].x, acc[i][r]); acc[i][r] = fma(vv[u].y, rv[r][u].y, acc[i][r]); }
            }
        }
    });
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
        double rl[NR];
#pragma unroll
        for (int r = 0; r < NR; ++r) rl[r] = r < nr ? Rv[(size_t)r * ldv + n - 1] : 0.0;
        if (TRI) {
#pragma unroll
            for (int r = 0; r < NR; ++r)
#pragma unroll
                for (int c = r; c < NR; ++c) tri[TRI ? r * NR - r * (r - 1) / 2 + (c - r) : 0] = fma(rl[r], rl[c], tri[TRI ? r * NR - r * (r - 1) / 2 + (c - r) : 0]);
        }
#pragma unroll
        for (int i = 0; i < KB; ++i)
            if (i < kb) {
                const double vl = V[(size_t)i * ldv + n - 1];
#pragma unroll
                for (int r = 0; r < NR; ++r) acc[i][r] = fma(vl, rl[r], acc[i][r]);
            }
    }
    constexpr int NV = KB * NR + (TRI ? NR * (NR + 1) / 2 : 0);
    __shared__ double sm[4][NV];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < KB; ++i)
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            const double s_ = wave_sum(acc[i][r]);
            if (lane == 0) sm[wid][i * NR + r] = s_;
        }
    if (TRI) {
#pragma unroll
        for (int t = 0; t < NT3; ++t) {
            const double s_ = wave_sum(tri[t]);
            if (lane == 0) sm[wid][KB * NR + t] = s_;
        }
    }
    __syncthreads();
    for (int j = threadIdx.x; j < NV; j += kThreads)
        partials[(size_t)blockIdx.x * NV + j] = (sm[0][j] + sm[1][j]) + (sm[2][j] + sm[3][j]);
}

// Pass 2 of a block: the s new basis vectors in place over the block vectors,
//   out_q = sum_{r <= q} T(r, q) P_r + sum_{i < k} C(i, q) V_i,   q < s
// (T = R^-1 upper triangular, C = -(G^-1 Q'P) R^-1: sstep.h).  Every lane reads the s block values of its elements before it
// writes them, so the update is safe in place; reads k + s streams, writes s.
struct BlockCoefs {
    double c[32][sstep::kS];
    double t[sstep::kS][sstep::kS];
};
template <int KB, int U, bool LDNT>
__global__ void __launch_bounds__(kThreads) block_axpy_kernel(size_t n, const double* V, size_t ldv, int k, double* Pv, int s,
                                                              BlockCoefs cf) {
    constexpr int S = sstep::kS;
    stream_loop<U>(n >> 1, [&](auto uc, size_t i0, size_t st) {
        constexpr int UU = decltype(uc)::value;
        double2 acc[S][UU];
#pragma unroll
        for (int q = 0; q < S; ++q)
#pragma unroll
            for (int u = 0; u < UU; ++u) acc[q][u] = make_double2(0.0, 0.0);
#pragma unroll
        for (int r = 0; r < S; ++r) {
            if (r < s) {
                double2 pv[UU];
#pragma unroll
                for (int u = 0; u < UU; ++u) pv[u] = ld2<LDNT>(Pv + (size_t)r * ldv, i0 + u * st);
#pragma unroll
                for (int q = r; q < S; ++q)
#pragma unroll
                    for (int u = 0; u < UU; ++u) { acc[q][u].x = fma(cf.t[r][q], pv[u].x, acc[q][u].x); acc[q][u].y = fma(cf.t[r][q], pv[u].y, acc[q][u].y); }
            }
        }
#pragma unroll
        for (int i = 0; i < KB; ++i) {
            if (i < k) {
                double2 vv[UU];
#pragma unroll
                for (int u = 0; u < UU; ++u) vv[u] = ld2<LDNT>(V + (size_t)i * ldv, i0 + u * st);
#pragma unroll
                for (int q = 0; q < S; ++q)
#pragma unroll
                    for (int u = 0; u < UU; ++u) { acc[q][u].x = fma(cf.c[i][q], vv[u].x, acc[q][u].x); acc[q][u].y = fma(cf.c[i][q], vv[u].y, acc[q][u].y); }
            }
        }
#pragma unroll
        for (int q = 0; q < S; ++q) {
            if (q < s) {
#pragma unroll
                for (int u = 0; u < UU; ++u) {
                    if (LDNT) st2nt(Pv + (size_t)q * ldv, i0 + u * st, acc[q][u]);
                    else reinterpret_cast<double2*>(Pv + (size_t)q * ldv)[i0 + u * st] = acc[q][u];
                }
            }
        }
    });
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
        const size_t e = n - 1;
        double pl[S], o[S];
        for (int r = 0; r < S; ++r) pl[r] = r < s ? Pv[(size_t)r * ldv + e] : 0.0;
        for (int q = 0; q < S; ++q) {
            double v = 0.0;
            for (int r = 0; r <= q; ++r) v = fma(cf.t[r][q], pl[r], v);
            for (int i = 0; i < k && i < KB; ++i) v = fma(cf.c[i][q], V[(size_t)i * ldv + e], v);
            o[q] = v;
        }
        for (int q = 0; q < s; ++q) Pv[(size_t)q * ldv + e] = o[q];
    }
}

// second reduction stage into a DEVICE buffer: the same fixed summation order as reduce_stage2_kernel (context.hip), so the
// device-resident and the host-driven Arnoldi steps produce bitwise identical projections
__global__ void __launch_bounds__(256) reduce_stage2_dev(const double* __restrict__ partials, int nblocks, int nvals,
                                                         double* __restrict__ out, const double* gate) {
    if (gate && gate[0] == 0.0) return;
    const int v = blockIdx.x;
    double acc = 0.0;
    for (int b = threadIdx.x; b < nblocks; b += 256) acc += partials[(size_t)b * nvals + v];
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    __shared__ double sm[4];
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double r = sm[0];
        for (int w = 1; w < 4; ++w) r += sm[w];
        out[v] = r;
    }
}

// ------------------------------------------------------------------ device-resident Arnoldi step (no host round trip)
// One record per speculative Arnoldi step: rec[0..k) = h = V'w, rec[kMaxBasis] = beta, rec[kMaxBasis + 1] = flag.
// coef[0..k) = -h, coef[kMaxBasis] = 1 / beta feed the multiaxpy that follows in the stream.
// (0: trustworthy; 1: severe cancellation or breakdown -- the host repeats that step on its own path).
// coef[kMaxBasis + 1] is the gate of the DGKS second pass: 1 when the remainder kept less than eta of ||w||.
// coef[kMaxBasis + 2] carries the running defect estimate of the cycle (solver.hip: arnoldi_step; zeroed by the host at the
// start of a cycle).
__global__ void arnoldi_coef_kernel(const double* __restrict__ hw, int k, double eta2, double orth_tol, double* __restrict__ rec,
                                    double* __restrict__ coef) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const double ww = hw[k];
    double hsq = 0.0;
    for (int i = 0; i < k; ++i) { const double h = hw[i]; hsq += h * h; rec[i] = h; coef[i] = -h; }
    const double b2 = ww - hsq;
    const bool ok = ww > 0.0 && b2 > kCancelTol * ww;
    const double be = ok ? sqrt(b2) : 1.0;
    rec[kMaxBasis] = ok ? be : 0.0;
    rec[kMaxBasis + 1] = ok ? 0.0 : 1.0;
    coef[kMaxBasis] = 1.0 / be;
    double gate = 0.0;
    if (ok) {
        const double dlt = coef[kMaxBasis + 2];
        const double grown = (dlt + 4.440892098500626e-16) * sqrt(ww / b2);
        if (b2 < eta2 * ww || grown > orth_tol) gate = 1.0;
        else if (grown > dlt) coef[kMaxBasis + 2] = grown;
    }
    coef[kMaxBasis + 1] = gate;
}

// second ("twice is enough") pass: s = V'v_k, vv = v_k'v_k from hw; v_k <- (v_k - V s) / cn, h += beta s, beta *= cn with the
// Pythagorean cn^2 = vv - |s|^2 (no cancellation after a first pass).  Runs only when the gate is set.
__global__ void arnoldi_coef2_kernel(const double* __restrict__ hw, int k, double* __restrict__ rec, double* __restrict__ coef) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    if (coef[kMaxBasis + 1] == 0.0) return;
    const double vv = hw[k], be = rec[kMaxBasis];
    double ssq = 0.0;
    for (int i = 0; i < k; ++i) { const double s_ = hw[i]; ssq += s_ * s_; coef[i] = -s_; rec[i] += be * s_; }
    const double n2 = vv - ssq;
    const bool ok = n2 > kCancelTol * vv;
    const double cn = ok ? sqrt(n2) : 1.0;
    coef[kMaxBasis] = 1.0 / cn;
    rec[kMaxBasis] = ok ? be * cn : 0.0;
    if (!ok) rec[kMaxBasis + 1] = 1.0;
}

// Gram-corrected single pass on the device (solver.hip: arnoldi_step's gram branch, same arithmetic up to the order of the
// k-term sums): hw = [a = V'w (k) ; w'w ; g = V'v_{k-1} (k)], G (ldg = kMaxBasis + 1, column-major, device) holds the
// measured Gram matrix of the cycle.  One wavefront; lane i owns row i (k <= 32).
__global__ void __launch_bounds__(64) arnoldi_gram_coef_kernel(const double* __restrict__ hw, int k, double* __restrict__ G,
                                                               double* __restrict__ rec, double* __restrict__ coef) {
    __shared__ double sa[64], se[64];
    const int i = threadIdx.x, ldg = kMaxBasis + 1;
    const bool on = i < k;
    const double ai = on ? hw[i] : 0.0;
    if (on) {
        const double gi = hw[k + 1 + i];
        G[(size_t)i + (size_t)(k - 1) * ldg] = gi;
        G[(size_t)(k - 1) + (size_t)i * ldg] = gi;
    }
    sa[i] = ai;
    __syncthreads();
    double e1 = 0.0;
    if (on)
        for (int l = 0; l < k; ++l) e1 += (G[(size_t)i + (size_t)l * ldg] - (i == l ? 1.0 : 0.0)) * sa[l];
    se[i] = e1;
    __syncthreads();
    double e2 = 0.0;
    if (on)
        for (int l = 0; l < k; ++l) e2 += (G[(size_t)i + (size_t)l * ldg] - (i == l ? 1.0 : 0.0)) * se[l];
    const double ci = ai - e1 + e2;
    const double proj = wave_sum(ci * ai);           // lanes >= k contribute 0
    const double pr = __shfl(proj, 0, 64);
    const double ww = hw[k];
    const double b2 = ww - pr;
    const bool ok = ww > 0.0 && b2 > kCancelTol * ww;
    const double be = ok ? sqrt(b2) : 1.0;
    if (on) { rec[i] = ci; coef[i] = -ci; }
    if (i == 0) {
        rec[kMaxBasis] = ok ? be : 0.0;
        rec[kMaxBasis + 1] = ok ? 0.0 : 1.0;
        coef[kMaxBasis] = 1.0 / be;
        coef[kMaxBasis + 1] = 0.0;                   // no second pass
    }
}

// ------------------------------------------------------------------ basis rotation  dst_j = sum_i Q(i,j) V_i
// One thread owns element(s) e: it loads V_i[e] for all i < m into registers, then forms the kout outputs
// one at a time -- so dst may alias V (Krylov-Schur restart rotates the basis in place).  Q (m x kout,
// column-major) is wave-uniform and comes through the scalar cache.
template <int KB>
__global__ void __launch_bounds__(kThreads) combine_kernel(size_t n, const double* V, size_t ldv, int m,
                                                           const double* __restrict__ Q, int kout, double* dst,
                                                           size_t lddst) {
    const size_t stride = (size_t)gridDim.x * kThreads;
    for (size_t e = (size_t)blockIdx.x * kThreads + threadIdx.x; e < n; e += stride) {
        double in[KB];
#pragma unroll
        for (int i = 0; i < KB; ++i) in[i] = (i < m) ? V[(size_t)i * ldv + e] : 0.0;
        for (int j = 0; j < kout; ++j) {
            const double* q = Q + (size_t)j * m;
            double acc = 0.0;
#pragma unroll
            for (int i = 0; i < KB; ++i)
                if (i < m) acc = fma(q[i], in[i], acc);
            dst[(size_t)j * lddst + e] = acc;
        }
    }
}

}  // namespace

// ================================================================== launchers
int v_copy(bk_ctx* ctx, size_t n, const double* x, double* y) {
    if (n == 0 || x == y) return 0;
    ProfScope ps(ctx, "blas1", 16.0 * n);
    BK_HIP(ctx, hipMemcpyAsync(y, x, n * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
    return 0;
}

int v_zero(bk_ctx* ctx, size_t n, double* x) {
    if (n == 0) return 0;
    ProfScope ps(ctx, "blas1", 8.0 * n);
    BK_HIP(ctx, hipMemsetAsync(x, 0, n * sizeof(double), ctx->stream));
    return 0;
}

int v_axpbyz(bk_ctx* ctx, size_t n, double a, const double* x, double b, const double* y, double* z) {
    if (n == 0) return 0;
    const int has_y = (y != nullptr && b != 0.0) ? 1 : 0;
    const int hx = (x != nullptr && a != 0.0) ? 1 : 0;
    ProfScope ps(ctx, "blas1", 8.0 * n * (1 + hx + has_y));
    const bool vec = aligned16(z) && (!hx || aligned16(x)) && (!has_y || aligned16(y));
    const int grid = grid_for(n, vec ? 2 : 1, 4096);
    if (vec && nt_hint(ctx, n))
        hipLaunchKernelGGL((axpbyz_kernel<2, true>), dim3(grid), dim3(kThreads), 0, ctx->stream, n, a, x, b, y, z, hx, has_y);
    else if (vec)
        hipLaunchKernelGGL((axpbyz_kernel<2>), dim3(grid), dim3(kThreads), 0, ctx->stream, n, a, x, b, y, z, hx, has_y);
    else
        hipLaunchKernelGGL((axpbyz_kernel<1>), dim3(grid), dim3(kThreads), 0, ctx->stream, n, a, x, b, y, z, hx, has_y);
    BK_HIP(ctx, hipGetLastError());
    return 0;
}

int v_pw_scale(bk_ctx* ctx, size_t n, const double* x, const double* u, double A, double B, double C, double* z) {
    if (n == 0) return 0;
    ProfScope ps(ctx, "blas1", 24.0 * n);
    hipLaunchKernelGGL(pw_scale_kernel, dim3(grid_for(n, 1, 4096)), dim3(kThreads), 0, ctx->stream, n, x, u, A, B, C, z);
    BK_HIP(ctx, hipGetLastError());
    return 0;
}

int v_axpby(bk_ctx* ctx, size_t n, double a, const double* x, double b, double* y) {
    return v_axpbyz(ctx, n, a, x, b, y, y);
}

int v_scale(bk_ctx* ctx, size_t n, double a, double* x) { return v_axpbyz(ctx, n, a, x, 0.0, nullptr, x); }

int v_fill_random(bk_ctx* ctx, size_t n, size_t goff, unsigned long long seed, double* x) {
    if (n == 0) return 0;
    hipLaunchKernelGGL(fill_random_kernel, dim3(grid_for(n, 1, 4096)), dim3(kThreads), 0, ctx->stream, n, goff, seed, x);
    BK_HIP(ctx, hipGetLastError());
    return 0;
}

static int dot_launch(bk_ctx* ctx, size_t n, const double* x, const double* y0, const double* y1, int ny, double* out) {
    const bool vec = aligned16(x) && aligned16(y0) && (ny == 1 || aligned16(y1));
    const int grid = grid_for(n, vec ? 2 : 1, kRedBlocks);
    {
        ProfScope ps(ctx, "blas1", 8.0 * n * ((x == y0 ? 1 : 2) + (ny - 1)));
        if (vec && nt_hint(ctx, n)) {
            if (ny == 1) hipLaunchKernelGGL((dot_kernel<2, 1, true>), dim3(grid), dim3(kThreads), 0, ctx->stream, n, x, y0, y1, ctx->d_partials);
            else hipLaunchKernelGGL((dot_kernel<2, 2, true>), dim3(grid), dim3(kThreads), 0, ctx->stream, n, x, y0, y1, ctx->d_partials);
        } else if (ny == 1) {
            if (vec) hipLaunchKernelGGL((dot_kernel<2, 1>), dim3(grid), dim3(kThreads), 0, ctx->stream, n, x, y0, y1, ctx->d_partials);
            else hipLaunchKernelGGL((dot_kernel<1, 1>), dim3(grid), dim3(kThreads), 0, ctx->stream, n, x, y0, y1, ctx->d_partials);
        } else {
            if (vec) hipLaunchKernelGGL((dot_kernel<2, 2>), dim3(grid), dim3(kThreads), 0, ctx->stream, n, x, y0, y1, ctx->d_partials);
            else hipLaunchKernelGGL((dot_kernel<1, 2>), dim3(grid), dim3(kThreads), 0, ctx->stream, n, x, y0, y1, ctx->d_partials);
        }
        BK_HIP(ctx, hipGetLastError());
    }
    BK_TRY(reduce_finish(ctx, grid, ny, 0));
    for (int j = 0; j < ny; ++j) out[j] = ctx->h_red[j];
    return 0;
}

int v_dot(bk_ctx* ctx, size_t n, const double* x, const double* y, double* out) {
    return dot_launch(ctx, n, x, y, nullptr, 1, out);
}

int v_dot2(bk_ctx* ctx, size_t n, const double* x, const double* y1, const double* y2, double* out2) {
    return dot_launch(ctx, n, x, y1, y2, 2, out2);
}

int v_diff_nrm2(bk_ctx* ctx, size_t n, const double* x, const double* y, double* out) {
    const bool vec = aligned16(x) && aligned16(y);
    const int grid = grid_for(n, vec ? 2 : 1, kRedBlocks);
    {
        ProfScope ps(ctx, "blas1", 16.0 * n);
        if (vec && nt_hint(ctx, n)) hipLaunchKernelGGL((diff_nrm2_kernel<2, true>), dim3(grid), dim3(kThreads), 0, ctx->stream, n, x, y, ctx->d_partials);
        else if (vec) hipLaunchKernelGGL((diff_nrm2_kernel<2>), dim3(grid), dim3(kThreads), 0, ctx->stream, n, x, y, ctx->d_partials);
        else hipLaunchKernelGGL((diff_nrm2_kernel<1>), dim3(grid), dim3(kThreads), 0, ctx->stream, n, x, y, ctx->d_partials);
        BK_HIP(ctx, hipGetLastError());
    }
    BK_TRY(reduce_finish(ctx, grid, 1, 0));
    *out = sqrt(ctx->h_red[0]);
    return 0;
}

int v_nrm2(bk_ctx* ctx, size_t n, const double* x, double* out) {
    double s = 0.0;
    BK_TRY(dot_launch(ctx, n, x, x, nullptr, 1, &s));
    *out = sqrt(s);
    return 0;
}

// y <- y + c r (r may be NULL);  *out = z . y
int v_axpy_dot(bk_ctx* ctx, size_t n, double c, const double* r, double* y, const double* z, double* out) {
    const int has_r = (r != nullptr && c != 0.0) ? 1 : 0;
    const bool vec = aligned16(y) && aligned16(z) && (!has_r || aligned16(r));
    const int grid = grid_for(n, vec ? 2 : 1, kRedBlocks);
    {
        ProfScope ps(ctx, "blas1", 8.0 * n * (2 + 2 * has_r));
        if (vec && nt_hint(ctx, n)) hipLaunchKernelGGL((axpy_dot_kernel<2, true>), dim3(grid), dim3(kThreads), 0, ctx->stream, n, c, r, has_r, y, z, ctx->d_partials);
        else if (vec) hipLaunchKernelGGL((axpy_dot_kernel<2>), dim3(grid), dim3(kThreads), 0, ctx->stream, n, c, r, has_r, y, z, ctx->d_partials);
        else hipLaunchKernelGGL((axpy_dot_kernel<1>), dim3(grid), dim3(kThreads), 0, ctx->stream, n, c, r, has_r, y, z, ctx->d_partials);
        BK_HIP(ctx, hipGetLastError());
    }
    BK_TRY(reduce_finish(ctx, grid, 1, 0));
    *out = ctx->h_red[0];
    return 0;
}

// w <- cz z + c1 w1 + c2 w2 ;  x <- x + phi w   (w must not alias z, w1, w2)
int v_minres_update(bk_ctx* ctx, size_t n, double cz, const double* z, double c1, const double* w1, double c2, const double* w2,
                    double* w, double phi, double* x) {
    const bool vec = aligned16(z) && aligned16(w1) && aligned16(w2) && aligned16(w) && aligned16(x);
    const int grid = grid_for(n, vec ? 2 : 1, 4096);
    ProfScope ps(ctx, "blas1", 8.0 * n * 6);
    if (vec && nt_hint(ctx, n)) hipLaunchKernelGGL((minres_update_kernel<2, true>), dim3(grid), dim3(kThreads), 0, ctx->stream, n, cz, z, c1, w1, c2, w2, w, phi, x);
    else if (vec) hipLaunchKernelGGL((minres_update_kernel<2>), dim3(grid), dim3(kThreads), 0, ctx->stream, n, cz, z, c1, w1, c2, w2, w, phi, x);
    else hipLaunchKernelGGL((minres_update_kernel<1>), dim3(grid), dim3(kThreads), 0, ctx->stream, n, cz, z, c1, w1, c2, w2, w, phi, x);
    BK_HIP(ctx, hipGetLastError());
    return 0;
}

// Two consecutive MINRES direction / solution updates in ONE pass (round 6): with (m2, m1) = (w_{k-2}, w_{k-1}),
//   wa = cza za + c1a m2 + c2a m1,   wb = czb zb + c1b m1 + c2b wa,   x <- (x + phia wa) + phib wb
// -- the arithmetic of two v_minres_update calls, element for element, in 8 array streams instead of 12.  wb may alias m2 (an element is
// read before it is written by its own lane); wa must not alias any input.
template <bool NTH>
__global__ void __launch_bounds__(kThreads) minres_update2_kernel(size_t n2, double cza, const double* __restrict__ za, double c1a, double c2a,
                                                                  double czb, const double* __restrict__ zb, double c1b, double c2b,
                                                                  const double* m2, const double* __restrict__ m1, double* __restrict__ wa,
                                                                  double* wb, double phia, double phib, double* __restrict__ x) {
    stream_loop<2>(n2, [&](auto uc, size_t i0, size_t st) {
        constexpr int UU = decltype(uc)::value;
        double2 av[UU], bv[UU], p2[UU], p1[UU], xv[UU];
#pragma unroll
        for (int u = 0; u < UU; ++u) {
            av[u] = ld2<NTH>(za, i0 + u * st);
            bv[u] = ld2<NTH>(zb, i0 + u * st);
            p2[u] = ld2<NTH>(m2, i0 + u * st);
            p1[u] = ld2<NTH>(m1, i0 + u * st);
            xv[u] = ld2<NTH>(x, i0 + u * st);
        }
#pragma unroll
        for (int u = 0; u < UU; ++u) {
            double2 a_, b_;
            a_.x = fma(c2a, p1[u].x, fma(c1a, p2[u].x, cza * av[u].x));
            a_.y = fma(c2a, p1[u].y, fma(c1a, p2[u].y, cza * av[u].y));
            b_.x = fma(c2b, a_.x, fma(c1b, p1[u].x, czb * bv[u].x));
            b_.y = fma(c2b, a_.y, fma(c1b, p1[u].y, czb * bv[u].y));
            xv[u].x = fma(phib, b_.x, fma(phia, a_.x, xv[u].x));
            xv[u].y = fma(phib, b_.y, fma(phia, a_.y, xv[u].y));
            if (NTH) { st2nt(wa, i0 + u * st, a_); st2nt(wb, i0 + u * st, b_); st2nt(x, i0 + u * st, xv[u]); }
            else {
                reinterpret_cast<double2*>(wa)[i0 + u * st] = a_; reinterpret_cast<double2*>(wb)[i0 + u * st] = b_;
                reinterpret_cast<double2*>(x)[i0 + u * st] = xv[u];
            }
        }
    });
}

// 0: done; 1: this shape is not covered (odd length / unaligned buffers) -- the caller takes two v_minres_update calls
int v_minres_update2(bk_ctx* ctx, size_t n, double cza, const double* za, double c1a, double c2a, double czb, const double* zb, double c1b,
                     double c2b, const double* m2, const double* m1, double* wa, double* wb, double phia, double phib, double* x) {
    if ((n & 1) || !(aligned16(za) && aligned16(zb) && aligned16(m2) && aligned16(m1) && aligned16(wa) && aligned16(wb) && aligned16(x))) return 1;
    const int grid = grid_for(n, 2, 4096);
    ProfScope ps(ctx, "blas1", 8.0 * n * 8);
    if (nt_hint(ctx, n)) hipLaunchKernelGGL((minres_update2_kernel<true>), dim3(grid), dim3(kThreads), 0, ctx->stream, n >> 1, cza, za, c1a, c2a, czb, zb, c1b, c2b, m2, m1, wa, wb, phia, phib, x);
    else hipLaunchKernelGGL((minres_update2_kernel<false>), dim3(grid), dim3(kThreads), 0, ctx->stream, n >> 1, cza, za, c1a, c2a, czb, zb, c1b, c2b, m2, m1, wa, wb, phia, phib, x);
    BK_HIP(ctx, hipGetLastError());
    return 0;
}

int v_nrminf(bk_ctx* ctx, size_t n, const double* x, double* out) {
    const int grid = grid_for(n, 1, kRedBlocks);
    {
        ProfScope ps(ctx, "blas1", 8.0 * n);
        hipLaunchKernelGGL(absmax_kernel, dim3(grid), dim3(kThreads), 0, ctx->stream, n, x, ctx->d_partials);
        BK_HIP(ctx, hipGetLastError());
    }
    BK_TRY(reduce_finish(ctx, grid, 1, 1));
    *out = ctx->h_red[0];
    return 0;
}

// Big (HBM-streaming) vectors, k <= 32: the contiguous-burst kernels.  Measured at 512^3 with the GMRES layout (dst = V[k];
// profiles/r3_krylov_burst_sweep.txt): multidot 0.82-0.85 -> 0.88-0.91 of peak with 8 adjacent chunks and 512 workgroups
// (4 chunks below 6 vectors), multiaxpy 0.67-0.69 -> 0.71-0.75 with 4 chunks and 512 workgroups.
constexpr int kBurstMax = 32;
static bool burst_ok(bk_ctx* ctx, bool vec, size_t n, int k) {
    return vec && k >= 1 && k <= kBurstMax && nt_hint(ctx, n) && ctx->opt("krylov_burst", 1.0) != 0.0;
}
static int burst_grid(bk_ctx* ctx, size_t n, const char* key) {
    return grid_for(n, 2, std::min(kRedBlocks, std::max(8, (int)ctx->opt(key, 512.0))));
}
static void launch_multidot_burst(bk_ctx* ctx, int grid, size_t n, const double* V, size_t ldv, int k, const double* w, const double* gate) {
    const bool u8 = k >= 6 && ctx->opt("dot_burst", 0.0) != 4.0;
#define BK_MDC(KB)                                                                                                                                  \
    do {                                                                                                                                           \
        if (u8) hipLaunchKernelGGL((multidot_c_kernel<KB, 8, true>), dim3(grid), dim3(kThreads), 0, ctx->stream, n, V, ldv, k, w, ctx->d_partials, gate); \
        else hipLaunchKernelGGL((multidot_c_kernel<KB, 4, true>), dim3(grid), dim3(kThreads), 0, ctx->stream, n, V, ldv, k, w, ctx->d_partials, gate);    \
    } while (0)
    if (k <= 4) BK_MDC(4);
    else if (k <= 8) BK_MDC(8);
    else if (k <= 16) BK_MDC(16);
    else if (k <= 24) BK_MDC(24);
    else BK_MDC(32);
#undef BK_MDC
}
// h = V'w, w'w and the Gram column g = V'V_{k-1} in one pass (any vector size; non-temporal loads for HBM-sized ones)
static void launch_multidot_gram(bk_ctx* ctx, int grid, size_t n, const double* V, size_t ldv, int k, const double* w, const double* gate,
                                 const double* vn = nullptr) {
    const bool nt = nt_hint(ctx, n);
    // bursts of 8 adjacent chunks from 6 vectors on, as the plain multidot (4 for the two largest buckets: registers)
#define BK_MDG(KB, UU)                                                                                                                              \
    do {                                                                                                                                           \
        if (nt) hipLaunchKernelGGL((multidot_c_kernel<KB, UU, true, true>), dim3(grid), dim3(kThreads), 0, ctx->stream, n, V, ldv, k, w, ctx->d_partials, gate, vn); \
        else hipLaunchKernelGGL((multidot_c_kernel<KB, 4, false, true>), dim3(grid), dim3(kThreads), 0, ctx->stream, n, V, ldv, k, w, ctx->d_partials, gate, vn);    \
    } while (0)
    if (k <= 4) BK_MDG(4, 4);
    else if (k < 6) BK_MDG(8, 4);
    else if (k <= 8) BK_MDG(8, 8);
    else if (k <= 16) BK_MDG(16, 8);
    else if (k <= 24) BK_MDG(24, 4);
    else BK_MDG(32, 4);
#undef BK_MDG
}
static void launch_multiaxpy_burst(bk_ctx* ctx, int grid, size_t n, const double* V, size_t ldv, int k, const Coefs& cf, const double* src,
                                   double scale, double* dst, int want_norm, const double* dcoef, int gated) {
    // start offsets: P phases (option axpy_stagger), map (axpy_stagger_map), s_sleep(1) units per chip-wide burst
    // (axpy_stagger_units: 512 workgroups x 16 KiB at ~5.7 TB/s = 1.5 us = 55 x 64 cycles)
    int stag = 0;
    {
        const int P = (int)ctx->opt("axpy_stagger", 0.0);
        if (P > 1 && P < 256 && grid >= 2 * P)
            stag = P | (((int)ctx->opt("axpy_stagger_map", 0.0) & 255) << 8) | (((int)ctx->opt("axpy_stagger_units", 55.0) & 0x7fff) << 16);
    }
#define BK_MAC(KB)                                                                                                                                  \
    do {                                                                                                                                           \
        if (dcoef) hipLaunchKernelGGL((multiaxpy_c_kernel<KB, 4, true, true, true>), dim3(grid), dim3(kThreads), 0, ctx->stream, n, V, ldv, k, cf, src, scale, dst, want_norm, ctx->d_partials, dcoef, gated, stag); \
        else hipLaunchKernelGGL((multiaxpy_c_kernel<KB, 4, true, true, false>), dim3(grid), dim3(kThreads), 0, ctx->stream, n, V, ldv, k, cf, src, scale, dst, want_norm, ctx->d_partials, dcoef, gated, stag);     \
    } while (0)
    if (k <= 4) BK_MAC(4);
    else if (k <= 8) BK_MAC(8);
    else if (k <= 16) BK_MAC(16);
    else if (k <= 24) BK_MAC(24);
    else BK_MAC(32);
#undef BK_MAC
}

template <int KB>
static void launch_multidot(bk_ctx* ctx, bool vec, int grid, size_t n, const double* V, size_t ldv, int k, const double* w) {
    const unsigned xcd = xcd_map(ctx, n, grid, false, k);
    if (vec && nt_hint(ctx, n) && ctx->opt("dot_variant", 1.0) == 1.0) hipLaunchKernelGGL((multidot_kernel<KB, 2, true>), dim3(grid), dim3(kThreads), 0, ctx->stream, n, V, ldv, k, w, ctx->d_partials, (const double*)nullptr, xcd);
    else if (vec) hipLaunchKernelGGL((multidot_kernel<KB, 2>), dim3(grid), dim3(kThreads), 0, ctx->stream, n, V, ldv, k, w, ctx->d_partials);
    else hipLaunchKernelGGL((multidot_kernel<KB, 1>), dim3(grid), dim3(kThreads), 0, ctx->stream, n, V, ldv, k, w, ctx->d_partials);
}

int v_multidot(bk_ctx* ctx, size_t n, const double* V, size_t ldv, int k, const double* w, double* out) {
    if (k < 0 || k > kMaxBasis) return set_error(ctx, "v_multidot: k=%d out of range", k);
    const bool vec = aligned16(V) && aligned16(w) && (ldv % 2 == 0);
    const bool burst = burst_ok(ctx, vec, n, k);
    const int grid = burst ? burst_grid(ctx, n, "dot_blocks") : grid_for(n, vec ? 2 : 1, kRedBlocks);
    {
        ProfScope ps(ctx, "multidot", 8.0 * n * (k + 1));
        if (burst) launch_multidot_burst(ctx, grid, n, V, ldv, k, w, nullptr);
        else if (k <= 4) launch_multidot<4>(ctx, vec, grid, n, V, ldv, k, w);
        else if (k <= 8) launch_multidot<8>(ctx, vec, grid, n, V, ldv, k, w);
        else if (k <= 16) launch_multidot<16>(ctx, vec, grid, n, V, ldv, k, w);
        else if (k <= 24) launch_multidot<24>(ctx, vec, grid, n, V, ldv, k, w);
        else if (k <= 32) launch_multidot<32>(ctx, vec, grid, n, V, ldv, k, w);
        else if (k <= 48) launch_multidot<48>(ctx, vec, grid, n, V, ldv, k, w);
        else launch_multidot<64>(ctx, vec, grid, n, V, ldv, k, w);
        BK_HIP(ctx, hipGetLastError());
    }
    BK_TRY(reduce_finish(ctx, grid, k + 1, 0));
    for (int j = 0; j <= k; ++j) out[j] = ctx->h_red[j];
    return 0;
}

// out[0..k) = V'w, out[k] = w'w, gram[0..k) = V'V_{k-1} (1 <= k <= 32, 16-byte aligned operands, even ldv)
bool v_multidot_gram_ok(bk_ctx* ctx, size_t n, const double* V, size_t ldv, int k, const double* w) {
    return k >= 1 && k <= 2 * kBurstMax && n >= 2 && aligned16(V) && aligned16(w) && (ldv % 2 == 0) && ctx->opt("gmres_gram", 1.0) != 0.0;
}
int v_multidot_gram(bk_ctx* ctx, size_t n, const double* V, size_t ldv, int k, const double* w, double* out, double* gram) {
    const int grid = nt_hint(ctx, n) ? burst_grid(ctx, n, "dot_blocks") : grid_for(n, 2 * 4, 512);
    if (k <= kBurstMax) {
        {
            ProfScope ps(ctx, "multidot", 8.0 * n * (k + 1));
            launch_multidot_gram(ctx, grid, n, V, ldv, k, w, nullptr);
            BK_HIP(ctx, hipGetLastError());
        }
        BK_TRY(reduce_finish(ctx, grid, 2 * k + 1, 0));
        for (int j = 0; j <= k; ++j) out[j] = ctx->h_red[j];
        for (int j = 0; j < k; ++j) gram[j] = ctx->h_red[k + 1 + j];
        return 0;
    }
    // more than 32 vectors (the eigensolver's basis): two launches -- V[0..32) against w and the newest vector V[k-1] from
    // outside, then V[32..k) which contains it.  (w and V[k-1] are read twice: 2 of k + 3 streams.)
    const int k0 = kBurstMax, k1 = k - kBurstMax;
    {
        ProfScope ps(ctx, "multidot", 8.0 * n * (k0 + 2));
        launch_multidot_gram(ctx, grid, n, V, ldv, k0, w, nullptr, V + (size_t)(k - 1) * ldv);
        BK_HIP(ctx, hipGetLastError());
    }
    BK_TRY(reduce_finish(ctx, grid, 2 * k0 + 1, 0));
    for (int j = 0; j < k0; ++j) { out[j] = ctx->h_red[j]; gram[j] = ctx->h_red[k0 + 1 + j]; }
    {
        ProfScope ps(ctx, "multidot", 8.0 * n * (k1 + 1));
        launch_multidot_gram(ctx, grid, n, V + (size_t)k0 * ldv, ldv, k1, w, nullptr);
        BK_HIP(ctx, hipGetLastError());
    }
    BK_TRY(reduce_finish(ctx, grid, 2 * k1 + 1, 0));
    for (int j = 0; j < k1; ++j) { out[k0 + j] = ctx->h_red[j]; gram[k0 + j] = ctx->h_red[k1 + 1 + j]; }
    out[k] = ctx->h_red[k1];
    return 0;
}

// ---- block Arnoldi step (sstep.h): pass 1 -- D[i * 8 + r] = <V_i, rhs_r> for the k_old measured basis vectors, T = the packed
// upper triangle of the right-hand vectors' own dots; rhs_r = basis slot r0 + r, r < nr.  The first launch carries the triangle
// and up to 4 basis vectors, every further one 8 basis vectors (register budget: 68 / 64 accumulators); the right-hand vectors are
// re-read by every launch.  One global reduction (and host synchronisation) per launch.
bool v_block_ok(bk_ctx* ctx, size_t n, const double* V, size_t ldv) { return n >= 2 && aligned16(V) && (ldv % 2 == 0); }
int v_block_dots(bk_ctx* ctx, size_t n, const double* V, size_t ldv, int kold, int r0, int nr, double* D, double* T) {
    if (kold < 0 || kold > kMaxBasis || nr < 1 || nr > sstep::kR) return set_error(ctx, "v_block_dots: bad sizes");
    const bool nt = nt_hint(ctx, n);
    const int grid = nt ? burst_grid(ctx, n, "dot_blocks") : grid_for(n, 2 * 2, 512);
    const double* Rv = V + (size_t)r0 * ldv;
    int done = 0;                        // measured basis vectors covered by the first launch (the one that carries the triangle)
    constexpr int NR5 = 5;
    if (nr <= NR5 && kold > 4 && ctx->opt("block_dots_nr5", 1.0) != 0.0) {
        // few right-hand vectors (a solve's last block is usually ONE step: s = 1, u <= 4): 8 basis vectors and the triangle in one launch
        const int kb = std::min(kold, 8);
        constexpr int NV = 8 * NR5 + NR5 * (NR5 + 1) / 2;
        static_assert(NV <= kPartialVals, "d_partials too small");
        {
            ProfScope ps(ctx, "multidot", 8.0 * n * (kb + nr));
            if (nt) hipLaunchKernelGGL((block_dots_kernel<8, true, 2, true, NR5>), dim3(grid), dim3(kThreads), 0, ctx->stream, n, V, ldv, kb, Rv, nr, ctx->d_partials);
            else hipLaunchKernelGGL((block_dots_kernel<8, true, 2, false, NR5>), dim3(grid), dim3(kThreads), 0, ctx->stream, n, V, ldv, kb, Rv, nr, ctx->d_partials);
            BK_HIP(ctx, hipGetLastError());
        }
        BK_TRY(reduce_finish(ctx, grid, NV, 0));
        for (int i = 0; i < kb; ++i)
            for (int r = 0; r < sstep::kR; ++r) D[i * sstep::kR + r] = r < NR5 ? ctx->h_red[i * NR5 + r] : 0.0;
        for (int t = 0; t < sstep::kTri; ++t) T[t] = 0.0;
        for (int r = 0, t = 0; r < NR5; ++r)
            for (int c = r; c < NR5; ++c, ++t) T[sstep::tri(r, c)] = ctx->h_red[8 * NR5 + t];
        done = kb;
    } else {
        const int kb = std::min(kold, 4);
        {
            ProfScope ps(ctx, "multidot", 8.0 * n * (kb + nr));
            if (nt) hipLaunchKernelGGL((block_dots_kernel<4, true, 2, true>), dim3(grid), dim3(kThreads), 0, ctx->stream, n, V, ldv, kb, Rv, nr, ctx->d_partials);
            else hipLaunchKernelGGL((block_dots_kernel<4, true, 2, false>), dim3(grid), dim3(kThreads), 0, ctx->stream, n, V, ldv, kb, Rv, nr, ctx->d_partials);
            BK_HIP(ctx, hipGetLastError());
        }
        constexpr int NV = 4 * sstep::kR + sstep::kTri;
        static_assert(NV <= kPartialVals && 8 * sstep::kR <= kPartialVals && kMaxBasis + 2 <= kPartialVals, "d_partials too small");
        BK_TRY(reduce_finish(ctx, grid, NV, 0));
        for (int i = 0; i < kb; ++i)
            for (int r = 0; r < sstep::kR; ++r) D[i * sstep::kR + r] = ctx->h_red[i * sstep::kR + r];
        for (int t = 0; t < sstep::kTri; ++t) T[t] = ctx->h_red[4 * sstep::kR + t];
        done = kb;
    }
    for (int i0 = done; i0 < kold; i0 += 8) {
        const int kb = std::min(kold - i0, 8);
        {
            ProfScope ps(ctx, "multidot", 8.0 * n * (kb + nr));
            if (nt) hipLaunchKernelGGL((block_dots_kernel<8, false, 2, true>), dim3(grid), dim3(kThreads), 0, ctx->stream, n, V + (size_t)i0 * ldv, ldv, kb, Rv, nr, ctx->d_partials);
            else hipLaunchKernelGGL((block_dots_kernel<8, false, 2, false>), dim3(grid), dim3(kThreads), 0, ctx->stream, n, V + (size_t)i0 * ldv, ldv, kb, Rv, nr, ctx->d_partials);
            BK_HIP(ctx, hipGetLastError());
        }
        BK_TRY(reduce_finish(ctx, grid, 8 * sstep::kR, 0));
        for (int i = 0; i < kb; ++i)
            for (int r = 0; r < sstep::kR; ++r) D[(i0 + i) * sstep::kR + r] = ctx->h_red[i * sstep::kR + r];
    }
    return 0;
}

// pass 2: basis slots k .. k + s - 1 <- the s new basis vectors (Cm: k x 4 row-major, Tm: 4 x 4 row-major upper; sstep.h)
int v_block_axpy(bk_ctx* ctx, size_t n, double* V, size_t ldv, int k, int s, const double* Cm, const double* Tm) {
    if (k < 1 || k > 32 || s < 1 || s > sstep::kS) return set_error(ctx, "v_block_axpy: bad sizes");
    BlockCoefs cf;
    for (int i = 0; i < 32; ++i)
        for (int q = 0; q < sstep::kS; ++q) cf.c[i][q] = i < k ? Cm[i * sstep::kS + q] : 0.0;
    for (int r = 0; r < sstep::kS; ++r)
        for (int q = 0; q < sstep::kS; ++q) cf.t[r][q] = Tm[r * sstep::kS + q];
    const bool nt = nt_hint(ctx, n);
    const int grid = nt ? burst_grid(ctx, n, "axpy_burst_blocks") : grid_for(n, 2 * 2, 1024);
    double* Pv = V + (size_t)k * ldv;
    ProfScope ps(ctx, "multiaxpy", 8.0 * n * (k + 2 * s));
#define BK_BAX(KB)                                                                                                                          \
    do {                                                                                                                                    \
        if (nt) hipLaunchKernelGGL((block_axpy_kernel<KB, 4, true>), dim3(grid), dim3(kThreads), 0, ctx->stream, n, V, ldv, k, Pv, s, cf);   \
        else hipLaunchKernelGGL((block_axpy_kernel<KB, 4, false>), dim3(grid), dim3(kThreads), 0, ctx->stream, n, V, ldv, k, Pv, s, cf);     \
    } while (0)
    if (k <= 8) BK_BAX(8);
    else if (k <= 16) BK_BAX(16);
    else BK_BAX(32);
#undef BK_BAX
    BK_HIP(ctx, hipGetLastError());
    return 0;
}

template <int KB>
static void launch_multiaxpy(bk_ctx* ctx, bool vec, int grid, size_t n, const double* V, size_t ldv, int k, const Coefs& cf,
                             const double* src, double scale, double* dst, int want_norm) {
    const bool nt = ctx->opt("axpy_nt", 1.0) != 0.0;       // non-temporal store of the one output stream: +1.5 % at 512^3
    const int variant = nt_hint(ctx, n) ? (int)ctx->opt("axpy_variant", 3.0) : 0; // 0 plain loads, 1 non-temporal loads of the basis, 2 two elements per lane, 3 both (default)
    const unsigned xcd = xcd_map(ctx, n, grid, true, k);
    if (vec && variant == 1) hipLaunchKernelGGL((multiaxpy_kernel<KB, 2, true, true, 1>), dim3(grid), dim3(kThreads), 0, ctx->stream, n, V, ldv, k, cf, src, scale, dst, want_norm, ctx->d_partials, xcd);
    else if (vec && variant == 2) hipLaunchKernelGGL((multiaxpy_kernel<KB, 2, true, false, 2>), dim3(grid), dim3(kThreads), 0, ctx->stream, n, V, ldv, k, cf, src, scale, dst, want_norm, ctx->d_partials, xcd);
    else if (vec && variant == 3) hipLaunchKernelGGL((multiaxpy_kernel<KB, 2, true, true, 2>), dim3(grid), dim3(kThreads), 0, ctx->stream, n, V, ldv, k, cf, src, scale, dst, want_norm, ctx->d_partials, xcd);
    else if (vec && nt) hipLaunchKernelGGL((multiaxpy_kernel<KB, 2, true>), dim3(grid), dim3(kThreads), 0, ctx->stream, n, V, ldv, k, cf, src, scale, dst, want_norm, ctx->d_partials);
    else if (vec) hipLaunchKernelGGL((multiaxpy_kernel<KB, 2>), dim3(grid), dim3(kThreads), 0, ctx->stream, n, V, ldv, k, cf, src, scale, dst, want_norm, ctx->d_partials);
    else hipLaunchKernelGGL((multiaxpy_kernel<KB, 1>), dim3(grid), dim3(kThreads), 0, ctx->stream, n, V, ldv, k, cf, src, scale, dst, want_norm, ctx->d_partials);
}

int v_multiaxpy(bk_ctx* ctx, size_t n, const double* V, size_t ldv, int k, const double* c, const double* src,
                double scale, double* dst, double* nrm2sq) {
    if (k < 0 || k > kMaxBasis) return set_error(ctx, "v_multiaxpy: k=%d out of range", k);
    Coefs cf;
    for (int j = 0; j < kMaxBasis; ++j) cf.c[j] = (j < k) ? c[j] : 0.0;
    const bool vec = aligned16(V) && aligned16(dst) && (!src || aligned16(src)) && (ldv % 2 == 0);
    const int want = nrm2sq ? 1 : 0;
    // 1024 blocks = one resident wave of workgroups walking the k+2 streams in lock step: 5.9 TB/s vs 5.7 TB/s with
    // 4096 blocks at 512^3 (DRAM page locality)
    int cap = (int)ctx->opt("axpy_blocks", kRedBlocks);
    if (want && cap > kRedBlocks) cap = kRedBlocks;
    const bool burst = burst_ok(ctx, vec, n, k);
    const int grid = burst ? burst_grid(ctx, n, "axpy_burst_blocks") : grid_for(n, vec ? 2 : 1, cap);
    {
        ProfScope ps(ctx, "multiaxpy", 8.0 * n * (k + 1 + (src ? 1 : 0)));
        if (burst) launch_multiaxpy_burst(ctx, grid, n, V, ldv, k, cf, src, scale, dst, want, nullptr, 0);
        else if (k <= 4) launch_multiaxpy<4>(ctx, vec, grid, n, V, ldv, k, cf, src, scale, dst, want);
        else if (k <= 8) launch_multiaxpy<8>(ctx, vec, grid, n, V, ldv, k, cf, src, scale, dst, want);
        else if (k <= 16) launch_multiaxpy<16>(ctx, vec, grid, n, V, ldv, k, cf, src, scale, dst, want);
        else if (k <= 24) launch_multiaxpy<24>(ctx, vec, grid, n, V, ldv, k, cf, src, scale, dst, want);
        else if (k <= 32) launch_multiaxpy<32>(ctx, vec, grid, n, V, ldv, k, cf, src, scale, dst, want);
        else if (k <= 48) launch_multiaxpy<48>(ctx, vec, grid, n, V, ldv, k, cf, src, scale, dst, want);
        else launch_multiaxpy<64>(ctx, vec, grid, n, V, ldv, k, cf, src, scale, dst, want);
        BK_HIP(ctx, hipGetLastError());
    }
    if (want) {
        BK_TRY(reduce_finish(ctx, grid, 1, 0));
        *nrm2sq = ctx->h_red[0];
    }
    return 0;
}


// One speculative Arnoldi orthogonalisation step entirely in the stream: hw = [V'w ; w'w] (all-reduced over RCCL ranks),
// coefficients, V_k = (w - V h) / beta.  Nothing is copied to the host; `rec` (kRecLen doubles, device) receives h, beta and
// the trust flag for the host to pick up after a later synchronisation.
int v_arnoldi_step_dev(bk_ctx* ctx, size_t n, double* V, size_t ldv, int k, const double* w, double eta, double orth_tol,
                       double* rec, double* coef, double* gram) {
    if (k < 1 || k > kMaxBasis - 1) return set_error(ctx, "v_arnoldi_step_dev: k=%d out of range", k);
    const bool vec = aligned16(V) && aligned16(w) && (ldv % 2 == 0);
    const int grid = grid_for(n, vec ? 2 : 1, kRedBlocks);
    double* dst = V + (size_t)k * ldv;
    const double* gate = coef + kMaxBasis + 1;
    // the same tuned instantiations as the host-driven step (non-temporal loads, two elements per lane) for vectors that
    // stream from HBM; plain accesses for the cache-resident sizes
    const bool big = vec && nt_hint(ctx, n);
    const Coefs cf0{};
    const bool burst = burst_ok(ctx, vec, n, k);
    const int grid_d = burst ? burst_grid(ctx, n, "dot_blocks") : grid, grid_a = burst ? burst_grid(ctx, n, "axpy_burst_blocks") : grid;
    auto dots = [&](const double* x, const double* g) {
        if (burst) {
            launch_multidot_burst(ctx, grid_d, n, V, ldv, k, x, g);
            hipLaunchKernelGGL(reduce_stage2_dev, dim3(k + 1), dim3(256), 0, ctx->stream, ctx->d_partials, grid_d, k + 1, ctx->d_red, g);
            return;
        }
#define BK_MD_DEV(KB)                                                                                                           \
    do {                                                                                                                        \
        if (big) hipLaunchKernelGGL((multidot_kernel<KB, 2, true>), dim3(grid), dim3(kThreads), 0, ctx->stream, n, V, ldv, k, x, ctx->d_partials, g, 0u); \
        else if (vec) hipLaunchKernelGGL((multidot_kernel<KB, 2>), dim3(grid), dim3(kThreads), 0, ctx->stream, n, V, ldv, k, x, ctx->d_partials, g); \
        else hipLaunchKernelGGL((multidot_kernel<KB, 1>), dim3(grid), dim3(kThreads), 0, ctx->stream, n, V, ldv, k, x, ctx->d_partials, g);     \
    } while (0)
        if (k <= 4) BK_MD_DEV(4);
        else if (k <= 8) BK_MD_DEV(8);
        else if (k <= 16) BK_MD_DEV(16);
        else if (k <= 24) BK_MD_DEV(24);
        else if (k <= 32) BK_MD_DEV(32);
        else if (k <= 48) BK_MD_DEV(48);
        else BK_MD_DEV(64);
#undef BK_MD_DEV
        hipLaunchKernelGGL(reduce_stage2_dev, dim3(k + 1), dim3(256), 0, ctx->stream, ctx->d_partials, grid, k + 1, ctx->d_red, g);
    };
    auto axpys = [&](const double* src, int gated) {
        if (burst) { launch_multiaxpy_burst(ctx, grid_a, n, V, ldv, k, cf0, src, 1.0, dst, 0, coef, gated); return; }
        const unsigned xcd = xcd_map(ctx, n, grid, true, k);
#define BK_MA_DEV(KB)                                                                                                         \
    do {                                                                                                                      \
        if (big) hipLaunchKernelGGL((multiaxpy_kernel<KB, 2, true, true, 2, true>), dim3(grid), dim3(kThreads), 0, ctx->stream, n, V, ldv, k, cf0, src, 1.0, dst, 0, ctx->d_partials, xcd, coef, gated); \
        else if (vec) hipLaunchKernelGGL((multiaxpy_kernel<KB, 2, false, false, 1, true>), dim3(grid), dim3(kThreads), 0, ctx->stream, n, V, ldv, k, cf0, src, 1.0, dst, 0, ctx->d_partials, 0u, coef, gated); \
        else hipLaunchKernelGGL((multiaxpy_kernel<KB, 1, false, false, 1, true>), dim3(grid), dim3(kThreads), 0, ctx->stream, n, V, ldv, k, cf0, src, 1.0, dst, 0, ctx->d_partials, 0u, coef, gated);     \
    } while (0)
        if (k <= 4) BK_MA_DEV(4);
        else if (k <= 8) BK_MA_DEV(8);
        else if (k <= 16) BK_MA_DEV(16);
        else if (k <= 24) BK_MA_DEV(24);
        else if (k <= 32) BK_MA_DEV(32);
        else if (k <= 48) BK_MA_DEV(48);
        else BK_MA_DEV(64);
#undef BK_MA_DEV
    };
    // ranks: the small all-reduce is ENQUEUED (ncclAllReduce, or the host-staged communicator's proxy hand-over)
    if (gram) {
        // Gram-corrected single pass: multidot with the Gram column, coefficients, ONE multiaxpy -- no second pass
        // (the caller, gmres_core, only passes `gram` for k <= 32 and operands v_multidot_gram_ok accepts; anything else is a
        // programming error of the library, not a property of the data)
        if (k > kBurstMax || !v_multidot_gram_ok(ctx, n, V, ldv, k, w)) return set_error(ctx, "v_arnoldi_step_dev: gram step out of range");
        const int gg = nt_hint(ctx, n) ? burst_grid(ctx, n, "dot_blocks") : grid_for(n, 2 * 4, 512);
        {
            ProfScope ps(ctx, "multidot", 8.0 * n * (k + 1));
            launch_multidot_gram(ctx, gg, n, V, ldv, k, w, nullptr);
            hipLaunchKernelGGL(reduce_stage2_dev, dim3(2 * k + 1), dim3(256), 0, ctx->stream, ctx->d_partials, gg, 2 * k + 1, ctx->d_red,
                               (const double*)nullptr);
            BK_HIP(ctx, hipGetLastError());
        }
        BK_TRY(comm_allreduce_dev(ctx, ctx->stream, ctx->d_red, 2 * k + 1, 0));
        hipLaunchKernelGGL(arnoldi_gram_coef_kernel, dim3(1), dim3(64), 0, ctx->stream, ctx->d_red, k, gram, rec, coef);
        {
            ProfScope ps(ctx, "multiaxpy", 8.0 * n * (k + 2));
            axpys(w, 0);
            BK_HIP(ctx, hipGetLastError());
        }
        return 0;
    }
    {
        ProfScope ps(ctx, "multidot", 8.0 * n * (k + 1));
        dots(w, nullptr);
        BK_HIP(ctx, hipGetLastError());
    }
    BK_TRY(comm_allreduce_dev(ctx, ctx->stream, ctx->d_red, k + 1, 0));
    hipLaunchKernelGGL(arnoldi_coef_kernel, dim3(1), dim3(64), 0, ctx->stream, ctx->d_red, k, eta * eta, orth_tol, rec, coef);
    {
        ProfScope ps(ctx, "multiaxpy", 8.0 * n * (k + 2));
        axpys(w, 0);
        BK_HIP(ctx, hipGetLastError());
    }
    // DGKS second pass, gated on the device (kernels return at once when the first pass kept >= eta of ||w||).  On ranks
    // the small all-reduce runs unconditionally (every rank takes the same decision from the same reduced numbers).
    dots(dst, gate);
    BK_TRY(comm_allreduce_dev(ctx, ctx->stream, ctx->d_red, k + 1, 0));
    hipLaunchKernelGGL(arnoldi_coef2_kernel, dim3(1), dim3(64), 0, ctx->stream, ctx->d_red, k, rec, coef);
    axpys(dst, 1);
    BK_HIP(ctx, hipGetLastError());
    return 0;
}

int v_basis_combine(bk_ctx* ctx, size_t n, const double* V, size_t ldv, int m, const double* Qhost, int kout,
                    double* dst, size_t lddst) {
    if (m < 1 || m > kMaxBasis || kout < 1 || kout > kMaxBasis) return set_error(ctx, "v_basis_combine: bad sizes");
    double* Qd = nullptr;
    BK_TRY(ws_get(ctx, (size_t)kMaxBasis * kMaxBasis, &Qd));
    // pageable host -> device copy of a few KB; synchronous w.r.t. the host buffer
    hipError_t e = hipMemcpyAsync(Qd, Qhost, sizeof(double) * (size_t)m * kout, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) { ws_put(ctx, Qd); return set_error(ctx, "v_basis_combine: upload failed: %s", hipGetErrorString(e)); }
    const int grid = grid_for(n, 1, 4096);
    {
        ProfScope ps(ctx, "combine", 8.0 * n * (m + kout));
        if (m <= 16) hipLaunchKernelGGL((combine_kernel<16>), dim3(grid), dim3(kThreads), 0, ctx->stream, n, V, ldv, m, Qd, kout, dst, lddst);
        else if (m <= 32) hipLaunchKernelGGL((combine_kernel<32>), dim3(grid), dim3(kThreads), 0, ctx->stream, n, V, ldv, m, Qd, kout, dst, lddst);
        else if (m <= 48) hipLaunchKernelGGL((combine_kernel<48>), dim3(grid), dim3(kThreads), 0, ctx->stream, n, V, ldv, m, Qd, kout, dst, lddst);
        else hipLaunchKernelGGL((combine_kernel<64>), dim3(grid), dim3(kThreads), 0, ctx->stream, n, V, ldv, m, Qd, kout, dst, lddst);
    }
    e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);   // Qd is recycled through the pool
    ws_put(ctx, Qd);
    if (e != hipSuccess) return set_error(ctx, "v_basis_combine: %s", hipGetErrorString(e));
    return 0;
}

}  // namespace bk

using namespace bk;

extern "C" {

int bk_vec_copy(bk_ctx* ctx, size_t n, const double* x, double* y) { return v_copy(ctx, n, x, y); }
int bk_vec_zero(bk_ctx* ctx, size_t n, double* x) { return v_zero(ctx, n, x); }
int bk_vec_scale(bk_ctx* ctx, size_t n, double a, double* x) { return v_scale(ctx, n, a, x); }
int bk_vec_axpby(bk_ctx* ctx, size_t n, double a, const double* x, double b, double* y) {
    return v_axpby(ctx, n, a, x, b, y);
}
int bk_vec_dot(bk_ctx* ctx, size_t n, const double* x, const double* y, double* out) {
    return v_dot(ctx, n, x, y, out);
}
int bk_vec_nrm2(bk_ctx* ctx, size_t n, const double* x, double* out) { return v_nrm2(ctx, n, x, out); }
int bk_vec_nrminf(bk_ctx* ctx, size_t n, const double* x, double* out) { return v_nrminf(ctx, n, x, out); }
int bk_krylov_multidot(bk_ctx* ctx, size_t n, const double* V, size_t ldv, int k, const double* w, double* out) {
    if (!ctx || !V || !w || !out) return -1;
    return v_multidot(ctx, n, V, ldv, k, w, out);
}
int bk_krylov_multiaxpy(bk_ctx* ctx, size_t n, const double* V, size_t ldv, int k, const double* c, const double* src,
                        double scale, double* dst, double* nrm2sq) {
    if (!ctx || !V || !c || !dst) return -1;
    return v_multiaxpy(ctx, n, V, ldv, k, c, src, scale, dst, nrm2sq);
}

}  // extern "C"
