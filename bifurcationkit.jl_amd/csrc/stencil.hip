// Matrix-free stencil operators: Swift-Hohenberg 1-D/2-D/3-D and 2-D complex Ginzburg-Landau.
//
// What they replace in the reference: the sparse products `L1 * u` / `mul!(f, Delta, u)` with the
// assembled `L1 = (I + Lap)^2` (25 nnz/row in 3-D, ~416 B/point of CSC traffic) plus the broadcast
// nonlinear terms -- examples/SH3d.jl:44-53, examples/SH2d-fronts.jl:31-34,124-127,
// examples/SHpde_snaking.jl:19-25, examples/cGL2d.jl:24-54,281-318.
//
// Boundary rule.  The reference builds 1-D second differences D with modified corner entries and
// forms L1 = A*A as a matrix product ("apply A twice with the same boundary rule").  Equivalently:
//   Neumann-ghost (corner diagonal -1/h^2):  apply the interior stencil to the EVEN reflection of v
//       about the half-point:  v[-1] = v[0], v[-2] = v[1], v[N] = v[N-1], v[N+1] = v[N-2]
//   Dirichlet (corner diagonal -2/h^2):      ODD reflection about the ghost node:
//       v[-1] = 0, v[-2] = -v[0], v[N] = 0, v[N+1] = -v[N-1]
// (D w)[-1] evaluated on that extension reproduces the matrix's truncated row exactly, so the
// composite (I + D)^2 is the plain 5-/13-/25-point stencil on index-reflected data.
//
// Kernels (fp64, HBM-bound, no MFMA):
//   sh_gather_kernel   one thread per point, 25 (13) reflected taps read through L1/L2 -- the
//                      simple, obviously-correct variant (ctx option sh_kernel = 0).
//   sh_stream_kernel   2.5-D streaming: a block owns a 64x16 (x,y) tile and marches along z; each
//                      incoming plane is staged ONCE in LDS with a 2-cell reflected halo, its in-plane
//                      parts b = B v and bb = B^2 v (B = c0 + ax Sx + ay Sy) are formed from LDS and
//                      scattered into five register accumulators (out[p-2..p+2]), using
//                      (B + az Sz)^2 = B^2 + 2 az B Sz + az^2 Sz^2.  HBM traffic = read v + read u +
//                      write out = 24 B/point (16 B/point for the residual) + halo re-reads that hit
//                      L2 because the blockIdx -> tile map keeps neighbouring tiles on one XCD.
#include "launch_plan.h"
#include "ops.h"

namespace bk {

namespace {

__device__ __forceinline__ int mirror_idx(int i, int n) {
    // even reflection about the half-points, then clamp (tile overhang only)
    if (i < 0) i = -1 - i;
    if (i >= n) i = 2 * n - 1 - i;
    return i < 0 ? 0 : (i >= n ? n - 1 : i);
}

struct ShK {                // kernel-side copy of ShArgs (+ derived constants)
    int nx, ny, nz, nzg, zoff;
    double ax, ay, az, c0;
    double l, nu, a0, a1;
    double ag;              // coefficient of the pointwise term g(u) v (= a1 unless the caller folds a shift: ShArgs::ag)
    int mode;
    const double* v;
    const double* u;
    double* out;
    const double* halo_lo;
    const double* halo_hi;
    int zchunk, ntx, nty, nzc, nblocks, grid8;
    int zc0, zcstep;        // z-chunks zc0, zc0 + zcstep, ... of this launch (multi-GPU: interior and face chunks go in separate launches)
    int vec_ok;
    int vload;              // plane staging with 16-B loads (nx a multiple of the tile width, 16-B aligned planes)
    int nt;                 // non-temporal hint on the u loads and the output stores (touched once)
    const double* addv;     // FD kernels: out += addc * addv, and the block's share of v . out goes to dotp[tile]
    double addc;
    double* dotp;
    int stagger, stag_cu;   // 3-D: start offset of the workgroups in each CU's second / third slot, in s_sleep(1) units per slot
};

// pointer to local plane lp in [-2, nz+2): halo buffers hold the two planes beyond each interior slab face
__device__ __forceinline__ const double* plane_ptr(const ShK& P, int lp) {
    const size_t plane = (size_t)P.nx * P.ny;
    if (lp < 0) return P.halo_lo + (size_t)(lp + 2) * plane;
    if (lp >= P.nz) return P.halo_hi + (size_t)(lp - P.nz) * plane;
    return P.v + (size_t)lp * plane;
}

__device__ __forceinline__ double g_of_u(int mode, double l, double nu, double u) {
    // JVP: l + 2 nu u - 3 u^2 (SH3d.jl:52) ; residual: (l + nu u - u^2) * u = l u + nu u^2 - u^3 (SH3d.jl:46)
    return mode == 0 ? fma(u, fma(-3.0, u, 2.0 * nu), l) : fma(u, nu - u, l);
}

// ------------------------------------------------------------------ gather variant
__global__ void __launch_bounds__(256) sh_gather_kernel(ShK P) {
    const size_t plane = (size_t)P.nx * P.ny;
    const size_t total = plane * (size_t)P.nz;
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int i = (int)(idx % P.nx);
    const int j = (int)((idx / P.nx) % P.ny);
    const int k = (int)(idx / plane);
    const int im1 = mirror_idx(i - 1, P.nx), ip1 = mirror_idx(i + 1, P.nx);
    const int im2 = mirror_idx(i - 2, P.nx), ip2 = mirror_idx(i + 2, P.nx);
    const int jm1 = mirror_idx(j - 1, P.ny), jp1 = mirror_idx(j + 1, P.ny);
    const int jm2 = mirror_idx(j - 2, P.ny), jp2 = mirror_idx(j + 2, P.ny);
    const double ax = P.ax, ay = P.ay, az = P.az, c0 = P.c0;
    const double* p0 = plane_ptr(P, k);
    const size_t r0 = (size_t)j * P.nx, rm1 = (size_t)jm1 * P.nx, rp1 = (size_t)jp1 * P.nx;
    const size_t rm2 = (size_t)jm2 * P.nx, rp2 = (size_t)jp2 * P.nx;
    const double vc = p0[r0 + i];
    double s = (c0 * c0 + 2.0 * (ax * ax + ay * ay + az * az)) * vc;
    s += 2.0 * c0 * ax * (p0[r0 + im1] + p0[r0 + ip1]);
    s += 2.0 * c0 * ay * (p0[rm1 + i] + p0[rp1 + i]);
    s += ax * ax * (p0[r0 + im2] + p0[r0 + ip2]);
    s += ay * ay * (p0[rm2 + i] + p0[rp2 + i]);
    s += 2.0 * ax * ay * ((p0[rm1 + im1] + p0[rm1 + ip1]) + (p0[rp1 + im1] + p0[rp1 + ip1]));
    if (az != 0.0) {
        const int gk = k + P.zoff;
        const double* pm1 = plane_ptr(P, mirror_idx(gk - 1, P.nzg) - P.zoff);
        const double* pp1 = plane_ptr(P, mirror_idx(gk + 1, P.nzg) - P.zoff);
        const double* pm2 = plane_ptr(P, mirror_idx(gk - 2, P.nzg) - P.zoff);
        const double* pp2 = plane_ptr(P, mirror_idx(gk + 2, P.nzg) - P.zoff);
        s += 2.0 * c0 * az * (pm1[r0 + i] + pp1[r0 + i]);
        s += az * az * (pm2[r0 + i] + pp2[r0 + i]);
        s += 2.0 * ax * az * ((pm1[r0 + im1] + pm1[r0 + ip1]) + (pp1[r0 + im1] + pp1[r0 + ip1]));
        s += 2.0 * ay * az * ((pm1[rm1 + i] + pm1[rp1 + i]) + (pp1[rm1 + i] + pp1[rp1 + i]));
    }
    const double uc = P.mode == 0 ? P.u[idx] : vc;
    const double g = g_of_u(P.mode, P.l, P.nu, uc);
    P.out[idx] = P.ag == P.a1 ? P.a0 * vc + P.a1 * (g * vc - s) : P.a0 * vc + P.ag * (g * vc) - P.a1 * s;
}

// ------------------------------------------------------------------ streaming variant
constexpr int TX = 64, TY = 16;            // tile
constexpr int NTX = TX / 2, NTY = 8;       // 32 x 8 threads, each owns 2 (x) x 2 (y: rows ty, ty+8) points
constexpr int LW = TX + 4, LH = TY + 4;    // LDS plane with a 2-cell halo
constexpr int LWP = LW;                    // row stride (68 doubles = 544 B: 16-B aligned rows)
constexpr int NLOAD = (LW * LH + 255) / 256;
constexpr int NPAIR = (LW / 2) * LH;       // plane staging as (x, x+1) pairs: 34 x 20 = 680 16-byte loads per plane
constexpr int NLOADV = (NPAIR + 255) / 256;
typedef double sh_nt_d2 __attribute__((ext_vector_type(2)));

// VL: plane staging with 16-byte loads (host-checked shapes), else 8-byte gathers.  162 VGPRs -> 3 waves / SIMD; capping the
// registers for a 4th wave (__launch_bounds__(256, 4)) spills and halves the rate (measured)
// FD (3-D only): the Lanczos step of MINRES / CG in the same pass -- out = a0 v + a1 J v + addc * addv and the dot product
// v . out (per-tile partial sums, reduced by reduce_finish).  The output plane p-2 becomes final while plane p is being
// processed: the FD kernels stage the planes in a ring of FOUR LDS buffers instead of two, so that the input's own points of
// plane p-2 are still there while plane p+1 is being staged (a register delay line costs 24 VGPRs and with them the third
// resident workgroup per CU; a ring of three would need one more barrier per plane).
// WPE: waves per SIMD the register allocation aims at -- the FD kernel needs 174 VGPRs, 6 above the 3-wave limit.
template <bool DIM3, bool VL, bool FD = false, int WPE = 1>
__global__ void __launch_bounds__(256, WPE) sh_stream_kernel(ShK P) {
    constexpr int NBUF = FD ? 4 : 2;
    __shared__ __attribute__((aligned(16))) double lds[NBUF][LH * LWP];

    // XCD-aware block -> tile map: hardware places block b on XCD b % 8; give each XCD a contiguous range
    // of tiles so that neighbouring tiles (which share halo cells) hit the same L2.
    const int b = blockIdx.x;
    const int L = (b & 7) * (P.grid8 >> 3) + (b >> 3);
    if (L >= P.nblocks) return;
    // Start offset (option sh_stagger): the 3 workgroups of a CU march along z in lockstep -- stage plane, barrier, 25-point
    // update, store -- and so does the whole chip; slot s of each CU (dispatch order: the first CUs workgroups fill slot 0, ...)
    // starts s x stagger x 64 cycles late, so that the load and the store phases of the three interleave.
    if (DIM3 && P.stagger > 0 && b < 3 * P.stag_cu) {
        const int slot = (b >> 3) / (P.stag_cu >> 3);
        for (int i = 0; i < slot * P.stagger; ++i) __builtin_amdgcn_s_sleep(1);
    }
    const int tix = L % P.ntx;
    const int tiy = (L / P.ntx) % P.nty;
    const int zc = P.zc0 + (L / (P.ntx * P.nty)) * P.zcstep;
    const int x0 = tix * TX, y0 = tiy * TY;
    const int zs = zc * P.zchunk;
    const int ze = min(zs + P.zchunk, P.nz);

    const int tid = threadIdx.x;
    const int tx = tid & (NTX - 1), ty = tid >> 5;
    const size_t plane = (size_t)P.nx * P.ny;

    // plane-independent source offsets of the LDS cells this thread stages
    int off[VL ? 1 : NLOAD];
    if (!VL) {
#pragma unroll
        for (int r = 0; r < NLOAD; ++r) {
            const int c = tid + r * 256;
            const int ly = c / LW, lx = c - ly * LW;
            const int gy = mirror_idx(y0 - 2 + ly, P.ny), gx = mirror_idx(x0 - 2 + lx, P.nx);
            off[r] = (c < LW * LH) ? gy * P.nx + gx : -1;
        }
    }
    // own points
    const int ox = x0 + 2 * tx;
    const int oy[2] = {y0 + ty, y0 + ty + NTY};
    const bool okx0 = ox < P.nx, okx1 = ox + 1 < P.nx;
    const bool oky[2] = {oy[0] < P.ny, oy[1] < P.ny};

    const double ax = P.ax, ay = P.ay, az = P.az, c0 = P.c0;
    const double kc = c0 * c0 + 2.0 * (ax * ax + ay * ay);     // B^2 centre
    const double k1x = 2.0 * c0 * ax, k1y = 2.0 * c0 * ay;
    const double k2x = ax * ax, k2y = ay * ay, kxy = 2.0 * ax * ay;
    const double s = -P.a1;

    double acc[2][2][5];
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int q = 0; q < 5; ++q) acc[r][c][q] = 0.0;

    // first / last plane this block touches (local indices), restricted to existing global planes
    int p_first = DIM3 ? zs - 2 : 0, p_last = DIM3 ? ze + 1 : 0;
    if (DIM3) {
        if (p_first + P.zoff < 0) p_first = -P.zoff;
        if (p_last + P.zoff > P.nzg - 1) p_last = P.nzg - 1 - P.zoff;
    }

    // 16-byte staging: the tile starts at an even x and the row has an even length, so the LDS cells (lx, lx+1) with lx
    // even are the aligned pair (gx, gx+1) of the source row -- except at the domain faces, where the reflected cells
    // (-2, -1) -> (1, 0) and (nx, nx+1) -> (nx-1, nx-2) are the SWAPPED pair at 0 / nx-2
    int offv[VL ? NLOADV : 1];
    unsigned swapmask = 0;
    if (VL) {
#pragma unroll
        for (int r = 0; r < NLOADV; ++r) {
            const int c = tid + r * 256;
            const int ly = c / (LW / 2), lx = 2 * (c - ly * (LW / 2));
            const int gy = mirror_idx(y0 - 2 + ly, P.ny);
            int gx = x0 - 2 + lx;
            if (gx < 0) { gx = 0; swapmask |= 1u << r; }
            else if (gx >= P.nx) { gx = P.nx - 2; swapmask |= 1u << r; }
            offv[r] = (c < NPAIR) ? gy * P.nx + gx : -1;
        }
    }
    double rv[VL ? 1 : NLOAD];
    double2 rvv[VL ? NLOADV : 1];
    double ru[2][2];
    auto load_plane = [&](int p) {
        const double* src = plane_ptr(P, p);
        if (VL) {
#pragma unroll
            for (int r = 0; r < NLOADV; ++r)
                rvv[r] = (offv[r] >= 0) ? *reinterpret_cast<const double2*>(src + offv[r]) : make_double2(0.0, 0.0);
        } else {
#pragma unroll
            for (int r = 0; r < NLOAD; ++r) rv[r] = (off[r] >= 0) ? src[off[r]] : 0.0;
        }
        if (P.mode == 0 && p >= 0 && p < P.nz) {
            const double* us = P.u + (size_t)p * plane;
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const size_t o = (size_t)oy[r] * P.nx + ox;
                if (oky[r] && okx1 && P.vec_ok) {
                    if (P.nt) {
                        const sh_nt_d2 t = __builtin_nontemporal_load(reinterpret_cast<const sh_nt_d2*>(us + o));
                        ru[r][0] = t.x; ru[r][1] = t.y;
                    } else {
                        const double2 t = *reinterpret_cast<const double2*>(us + o);
                        ru[r][0] = t.x; ru[r][1] = t.y;
                    }
                } else {
                    ru[r][0] = (oky[r] && okx0) ? us[o] : 0.0;
                    ru[r][1] = (oky[r] && okx1) ? us[o + 1] : 0.0;
                }
            }
        }
    };
    auto stage_plane = [&](int buf) {
        if (VL) {
#pragma unroll
            for (int r = 0; r < NLOADV; ++r) {
                const int c = tid + r * 256;
                if (c < NPAIR) {
                    const double2 t = rvv[r];
                    reinterpret_cast<double2*>(lds[buf])[c] = (swapmask >> r) & 1u ? make_double2(t.y, t.x) : t;
                }
            }
        } else {
#pragma unroll
            for (int r = 0; r < NLOAD; ++r) {
                const int c = tid + r * 256;
                if (c < LW * LH) lds[buf][c] = rv[r];
            }
        }
    };

    int cur = 0;
    load_plane(p_first);
    stage_plane(cur);
    double uc[2][2] = {{ru[0][0], ru[0][1]}, {ru[1][0], ru[1][1]}};
    __syncthreads();

    // FD state: the addend of the plane being emitted, the dot accumulator
    double ad[2][2] = {{0.0, 0.0}, {0.0, 0.0}};
    double dacc = 0.0;
    auto load_addend = [&](int k) {
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const double* as = P.addv + (size_t)k * plane + (size_t)oy[r] * P.nx + ox;
            if (oky[r] && okx1 && P.vec_ok) {
                const sh_nt_d2 t = __builtin_nontemporal_load(reinterpret_cast<const sh_nt_d2*>(as));
                ad[r][0] = t.x; ad[r][1] = t.y;
            } else {
                ad[r][0] = (oky[r] && okx0) ? as[0] : 0.0;
                ad[r][1] = (oky[r] && okx1) ? as[1] : 0.0;
            }
        }
    };

    for (int p = p_first; p <= p_last; ++p) {
        // (requested before the next plane's loads: they return in order, and the addend is needed first)
        if (FD && P.addv && p - 2 >= zs && p - 2 < ze) load_addend(p - 2);
        if (p < p_last) load_plane(p + 1);          // global loads for the next plane in flight during compute
        const double* Lp = lds[cur];
        const int gp = p + P.zoff;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int cy = ty + r * NTY + 2;
            const int cx = 2 * tx + 2;
            const double* row0 = Lp + cy * LWP + cx;
            const double* rowm1 = row0 - LWP;
            const double* rowp1 = row0 + LWP;
            const double* rowm2 = row0 - 2 * LWP;
            const double* rowp2 = row0 + 2 * LWP;
            const double m2 = row0[-2], m1 = row0[-1], v0 = row0[0], v1 = row0[1], q2 = row0[2], q3 = row0[3];
            const double am1 = rowm1[-1], a0 = rowm1[0], a1 = rowm1[1], a2 = rowm1[2];
            const double bm1 = rowp1[-1], b0 = rowp1[0], b1 = rowp1[1], b2 = rowp1[2];
            const double t0 = rowm2[0], t1 = rowm2[1], w0 = rowp2[0], w1 = rowp2[1];
            const double vv[2] = {v0, v1};
            const double sx1[2] = {m1 + v1, v0 + q2};                 // E + W
            const double sy1[2] = {a0 + b0, a1 + b1};                 // N + S
            const double sx2[2] = {m2 + q2, m1 + q3};                 // EE + WW
            const double sy2[2] = {t0 + w0, t1 + w1};                 // NN + SS
            const double sd[2] = {(am1 + a1) + (bm1 + b1), (a0 + a2) + (b0 + b2)};   // 4 diagonals
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const double v = vv[c];
                const double bb = kc * v + k1x * sx1[c] + k1y * sy1[c] + k2x * sx2[c] + k2y * sy2[c] + kxy * sd[c];
                double* A = acc[r][c];
                if (DIM3) {
                    const double bpl = c0 * v + ax * sx1[c] + ay * sy1[c];       // B v
                    const double e2 = s * (az * az) * v;
                    const double e1 = s * (2.0 * az) * bpl;
                    const bool own = (p >= 0 && p < P.nz);
                    const double e0 = s * (bb + 2.0 * az * az * v) +
                                      (own ? (P.a0 + P.ag * g_of_u(P.mode, P.l, P.nu, P.mode == 0 ? uc[r][c] : v)) * v : 0.0);
                    A[0] += e2; A[1] += e1; A[2] += e0; A[3] += e1; A[4] += e2;
                    // reflected ghost planes of the global z-boundaries
                    if (gp == 0) { A[2] += e1; A[3] += e2; }                    // ghost -1 carries plane 0
                    if (gp == 1) { A[1] += e2; }                                // ghost -2 carries plane 1
                    if (gp == P.nzg - 1) { A[2] += e1; A[1] += e2; }            // ghost N carries plane N-1
                    if (gp == P.nzg - 2) { A[3] += e2; }                        // ghost N+1 carries plane N-2
                } else {
                    A[2] = s * bb + (P.a0 + P.ag * g_of_u(P.mode, P.l, P.nu, P.mode == 0 ? uc[r][c] : v)) * v;
                }
            }
        }
        // out plane k = p - 2 (3-D) / p (2-D) is final now
        const int k = DIM3 ? p - 2 : p;
        if (k >= zs && k < ze) {
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                if (!oky[r]) continue;
                double* dst = P.out + (size_t)k * plane + (size_t)oy[r] * P.nx + ox;
                double o0 = acc[r][0][DIM3 ? 0 : 2], o1 = acc[r][1][DIM3 ? 0 : 2];
                if (FD) {
                    if (P.addv) { o0 = fma(P.addc, ad[r][0], o0); o1 = fma(P.addc, ad[r][1], o1); }
                    const double* vo = lds[(cur + 2) & (NBUF - 1)] + (ty + r * NTY + 2) * LWP + 2 * tx + 2;      // plane p-2
                    if (okx0) dacc = fma(vo[0], o0, dacc);
                    if (okx1) dacc = fma(vo[1], o1, dacc);
                }
                if (okx1 && P.vec_ok) {
                    if (P.nt) {
                        sh_nt_d2 t; t.x = o0; t.y = o1;
                        __builtin_nontemporal_store(t, reinterpret_cast<sh_nt_d2*>(dst));
                    } else *reinterpret_cast<double2*>(dst) = make_double2(o0, o1);
                } else {
                    if (okx0) dst[0] = o0;
                    if (okx1) dst[1] = o1;
                }
            }
        }
        if (DIM3) {
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    double* A = acc[r][c];
                    A[0] = A[1]; A[1] = A[2]; A[2] = A[3]; A[3] = A[4]; A[4] = 0.0;
                }
        }
        const int nxt = (cur + 1) & (NBUF - 1);
        if (p < p_last) {
            stage_plane(nxt);
#pragma unroll
            for (int r = 0; r < 2; ++r) { uc[r][0] = ru[r][0]; uc[r][1] = ru[r][1]; }
        }
        __syncthreads();
        cur = nxt;
    }
    if (DIM3) {
        // flush: outputs k = p_last-1, p_last (only reached when the block's range ends at the global top)
        for (int k = p_last - 1; k <= p_last; ++k) {
            if (k >= zs && k < ze) {
                const int q = k - (p_last - 1);
                if (FD && P.addv) load_addend(k);
#pragma unroll
                for (int r = 0; r < 2; ++r) {
                    if (!oky[r]) continue;
                    double* dst = P.out + (size_t)k * plane + (size_t)oy[r] * P.nx + ox;
                    double o0 = q == 0 ? acc[r][0][0] : acc[r][0][1];
                    double o1 = q == 0 ? acc[r][1][0] : acc[r][1][1];
                    if (FD) {
                        // `cur` has moved one past plane p_last: p_last-1 / p_last are two / three slots further on
                        if (P.addv) { o0 = fma(P.addc, ad[r][0], o0); o1 = fma(P.addc, ad[r][1], o1); }
                        const double* vo = lds[(cur + 2 + q) & (NBUF - 1)] + (ty + r * NTY + 2) * LWP + 2 * tx + 2;
                        if (okx0) dacc = fma(vo[0], o0, dacc);
                        if (okx1) dacc = fma(vo[1], o1, dacc);
                    }
                    if (okx0) dst[0] = o0;
                    if (okx1) dst[1] = o1;
                }
            }
        }
    }
    if (FD) {
        // the tile's share of v . out
        __syncthreads();                                      // the flush above may still be reading the staging planes
        for (int off = 32; off > 0; off >>= 1) dacc += __shfl_down(dacc, off, 64);
        if ((tid & 63) == 0) lds[0][tid >> 6] = dacc;
        __syncthreads();
        if (tid == 0) P.dotp[L] = (lds[0][0] + lds[0][1]) + (lds[0][2] + lds[0][3]);
    }
}

// ------------------------------------------------------------------ cGL 2-D (Dirichlet, 5-point, two fields)
struct CglK {
    int nx, ny;
    double ax, ay, r, mu, nu, c3, c5, gamma, a0, a1;
    int mode;
    const double* v;
    const double* u;
    double* out;
};

__global__ void __launch_bounds__(256) cgl_kernel(CglK P) {
    const size_t n = (size_t)P.nx * P.ny;
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= n) return;
    const int i = (int)(idx % P.nx), j = (int)(idx / P.nx);
    const double* v1 = P.v;
    const double* v2 = P.v + n;
    auto lap = [&](const double* f) {
        const double c = f[idx];
        const double w = i > 0 ? f[idx - 1] : 0.0, e = i + 1 < P.nx ? f[idx + 1] : 0.0;
        const double s = j > 0 ? f[idx - P.nx] : 0.0, nn = j + 1 < P.ny ? f[idx + P.nx] : 0.0;
        return P.ax * ((w + e) - 2.0 * c) + P.ay * ((s + nn) - 2.0 * c);
    };
    const double d1 = lap(v1), d2 = lap(v2);
    const double x1 = v1[idx], x2 = v2[idx];
    const double r = P.r, mu = P.mu, nu = P.nu, c3 = P.c3, c5 = P.c5;
    double o1, o2;
    if (P.mode == 0 || P.mode == 2) {
        // closed-form Jacobian block of the nonlinearity: Jcgl, examples/cGL2d.jl:66-69 (mode 2: its transpose, the
        // adjoint Jacobian of the Hopf machinery -- the Laplacian part is symmetric)
        const double u1 = P.u[idx], u2 = P.u[idx + n];
        const double ua = u1 * u1 + u2 * u2;
        const double f1u = r - 2 * u1 * (c3 * u1 - mu * u2) - c3 * ua - 4 * c5 * ua * u1 * u1 - c5 * ua * ua;
        const double f1v = -nu - 2 * u2 * (c3 * u1 - mu * u2) + mu * ua - 4 * c5 * ua * u1 * u2;
        const double f2u = nu - 2 * u1 * (c3 * u2 + mu * u1) - mu * ua - 4 * c5 * ua * u1 * u2;
        const double f2v = r - 2 * u2 * (c3 * u2 + mu * u1) - c3 * ua - 4 * c5 * ua * u2 * u2 - c5 * ua * ua;
        o1 = d1 + f1u * x1 + (P.mode == 0 ? f1v : f2u) * x2;
        o2 = d2 + (P.mode == 0 ? f2u : f1v) * x1 + f2v * x2;
    } else {
        // NL, examples/cGL2d.jl:24-40
        const double ua = x1 * x1 + x2 * x2;
        o1 = d1 + (r * x1 - nu * x2 - ua * (c3 * x1 - mu * x2) - c5 * ua * ua * x1 + P.gamma);
        o2 = d2 + (r * x2 + nu * x1 - ua * (c3 * x2 + mu * x1) - c5 * ua * ua * x2);
    }
    P.out[idx] = P.a0 * x1 + P.a1 * o1;
    P.out[idx + n] = P.a0 * x2 + P.a1 * o2;
}

// ------------------------------------------------------------------ SH 1-D (Dirichlet, odd reflection)
struct Sh1dK {
    int nx;
    double ax, lam, nu, a0, a1;
    int mode;
    const double* v;
    const double* u;
    double* out;
};

__global__ void __launch_bounds__(256) sh1d_kernel(Sh1dK P) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P.nx) return;
    const int n = P.nx;
    auto at = [&](int q) -> double {
        // odd reflection about the ghost nodes -1 and n: v[-1]=0, v[-2]=-v[0], v[n]=0, v[n+1]=-v[n-1]
        if (q == -1 || q == n) return 0.0;
        if (q < -1) { const int m = -2 - q; return m < n ? -P.v[m] : 0.0; }
        if (q > n) { const int m = 2 * n - q; return m >= 0 ? -P.v[m] : 0.0; }
        return P.v[q];
    };
    const double c0 = 1.0 - 2.0 * P.ax;
    const double vc = at(i);
    // (I + D)^2 v = (c0 + ax S)^2 v = (c0^2 + 2 ax^2) v + 2 c0 ax (v[i-1] + v[i+1]) + ax^2 (v[i-2] + v[i+2])
    const double sq = (c0 * c0 + 2.0 * P.ax * P.ax) * vc + 2.0 * c0 * P.ax * (at(i - 1) + at(i + 1)) +
                      P.ax * P.ax * (at(i - 2) + at(i + 2));
    const double uc = P.mode == 0 ? P.u[i] : vc;
    const double u2 = uc * uc;
    // JVP: lam + 3 nu u^2 - 5 u^4 ; residual: (lam + nu u^2 - u^4) u   (SHpde_snaking.jl:19-25); L1 = -(I+D)^2
    const double g = P.mode == 0 ? P.lam + 3.0 * P.nu * u2 - 5.0 * u2 * u2 : P.lam + P.nu * u2 - u2 * u2;
    P.out[i] = P.a0 * vc + P.a1 * (g * vc - sq);
}

// ------------------------------------------------------------------ dF/dp by finite differences, cancellation-free
// The PALC corrector forms dFdp = (F(x, p + eps) - F(x, p)) / eps (src/continuation/Palc.jl:239-240, Tangents.jl:77-82).
// Every parameter of the problems on this path multiplies a pointwise term phi_p(u) and the stencil part does not
// depend on it, so the quotient equals c * phi_p(u) with the scalar c = ((p + eps) - p) / eps -- evaluated this way the
// O(eps_mach |L1 u| / eps) ~ 1e-8 white rounding noise of the two-residual form never appears (on a large domain that
// noise excites the slow band of the Jacobian and costs GMRES 20-80 extra iterations per solve).
struct DpK {
    int pde, ipar;
    size_t n;          // grid points per field
    double c;
    const double* u;
    double* out;
};

__global__ void __launch_bounds__(256) dparam_kernel(DpK P) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < P.n; i += (size_t)gridDim.x * 256) {
        if (P.pde == BK_PDE_SH) {                       // F = -L1 u + l u + nu u^2 - u^3
            const double u = P.u[i];
            P.out[i] = P.c * (P.ipar == 0 ? u : u * u);
        } else if (P.pde == BK_PDE_SH1D) {              // R = L1 u + lam u + nu u^3 - u^5
            const double u = P.u[i];
            P.out[i] = P.c * (P.ipar == 0 ? u : u * u * u);
        } else {                                        // cGL, params (r, mu, nu, c3, c5, gamma): examples/cGL2d.jl:24-40
            const double u1 = P.u[i], u2 = P.u[i + P.n];
            const double ua = u1 * u1 + u2 * u2;
            double o1, o2;
            switch (P.ipar) {
                case 0: o1 = u1; o2 = u2; break;
                case 1: o1 = ua * u2; o2 = -ua * u1; break;
                case 2: o1 = -u2; o2 = u1; break;
                case 3: o1 = -ua * u1; o2 = -ua * u2; break;
                case 4: o1 = -ua * ua * u1; o2 = -ua * ua * u2; break;
                default: o1 = 1.0; o2 = 0.0; break;
            }
            P.out[i] = P.c * o1;
            P.out[i + P.n] = P.c * o2;
        }
    }
}

}  // namespace

int pde_dparam(bk_ctx* ctx, int pde, int ipar, size_t npts, double c, const double* u, double* out) {
    DpK P{pde, ipar, npts, c, u, out};
    const size_t nf = pde == BK_PDE_CGL2D ? 2 : 1;
    ProfScope ps(ctx, "blas1", 16.0 * npts * nf);
    size_t grid = (npts + 255) / 256;
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(dparam_kernel, dim3((unsigned)grid), dim3(256), 0, ctx->stream, P);
    BK_HIP(ctx, hipGetLastError());
    return 0;
}

// the fused Lanczos step needs the 3-D streaming kernel over the whole local array (a rank's slab with its halo planes exchanged) and one
// partial sum per tile
static int sh_zchunk_of(bk_ctx* ctx, const ShArgs& a, int tiles) {
    int zchunk = (int)ctx->opt("sh_zchunk", 0.0);
    if (zchunk <= 0) zchunk = sh_plan_zchunk(a.nz, tiles, 3L * (ctx->num_cu > 0 ? ctx->num_cu : 256), a.part != 0);
    return zchunk > a.nz ? a.nz : zchunk;
}
bool sh_fused_dot_ok(bk_ctx* ctx, const ShArgs& a) {
    if (a.az == 0.0 || a.mode != 0 || a.part != 0) return false;      // (slabs of a multi-rank run: the halo planes must be in place)
    if ((int)ctx->opt("sh_kernel", 1.0) == 0 || ctx->opt("jvp_fused_dot", 1.0) == 0.0) return false;
    const int tiles = ((a.nx + TX - 1) / TX) * ((a.ny + TY - 1) / TY);
    const int zchunk = sh_zchunk_of(ctx, a, tiles);
    const long nblocks = (long)tiles * ((a.nz + zchunk - 1) / zchunk);
    return nblocks <= (long)kPartialDoubles && (!a.addv || (((uintptr_t)a.addv & 15) == 0));
}

int sh_apply(bk_ctx* ctx, const ShArgs& a) {
    if (a.nx < 2 || a.ny < 2 || (a.az != 0.0 && a.nzg < 2))
        return set_error(ctx, "sh_apply: every grid extent must be >= 2");
    if (a.az != 0.0 && ctx->nranks > 1 && a.nz < 2) return set_error(ctx, "sh_apply: slab thinner than 2 planes");
    ShK P;
    P.nx = a.nx; P.ny = a.ny; P.nz = a.nz; P.nzg = a.nzg; P.zoff = a.zoff;
    P.ax = a.ax; P.ay = a.ay; P.az = a.az;
    P.c0 = 1.0 - 2.0 * (a.ax + a.ay + a.az);
    P.l = a.l; P.nu = a.nu; P.a0 = a.a0; P.a1 = a.a1; P.mode = a.mode;
    P.ag = a.ag_set ? a.ag : a.a1;
    P.v = a.v; P.u = a.u; P.out = a.out; P.halo_lo = a.halo_lo; P.halo_hi = a.halo_hi;
    P.addv = nullptr; P.addc = 0.0; P.dotp = nullptr;
    P.stag_cu = ctx->num_cu;
    P.stagger = ctx->num_cu % 8 == 0 ? (int)ctx->opt("sh_stagger", 0.0) : 0;
    if (a.dot_blocks) {
        if (!sh_fused_dot_ok(ctx, a)) return set_error(ctx, "sh_apply: fused dot requested on an unsupported path");
        P.addv = (a.addc != 0.0) ? a.addv : nullptr; P.addc = a.addc; P.dotp = ctx->d_partials;
    }
    const size_t n = (size_t)a.nx * a.ny * a.nz;
    const bool dim3d = a.az != 0.0;
    const int variant = (int)ctx->opt("sh_kernel", 1.0);
    P.zc0 = 0; P.zcstep = 1;
    if (variant == 0) {
        if (a.part == 1) return 0;                            // the gather cross-check runs whole once the halos are in
        ProfScope ps(ctx, a.mode == 0 ? "jvp" : "residual", (a.mode == 0 ? 24.0 : 16.0) * n);
        P.zchunk = 0; P.ntx = P.nty = P.nzc = P.nblocks = P.grid8 = 0; P.vec_ok = 0; P.vload = 0; P.nt = 0;
        const size_t grid = (n + 255) / 256;
        hipLaunchKernelGGL(sh_gather_kernel, dim3((unsigned)grid), dim3(256), 0, ctx->stream, P);
    } else {
        P.ntx = (a.nx + TX - 1) / TX;
        P.nty = (a.ny + TY - 1) / TY;
        // rounds x planes per workgroup (launch_plan.h); 3 workgroups per CU stay resident (162 VGPRs)
        const int zchunk = dim3d ? sh_zchunk_of(ctx, a, P.ntx * P.nty) : 1;
        P.zchunk = zchunk;
        P.nzc = (a.nz + zchunk - 1) / zchunk;
        P.vec_ok = ((a.nx & 1) == 0) && (((uintptr_t)a.out & 15) == 0) &&
                   (a.mode != 0 || ((uintptr_t)a.u & 15) == 0);
        // 16-byte plane staging: full tiles in x (no overhang), even plane size (every plane 16-B aligned), aligned bases
        P.vload = ctx->opt("sh_vload", 1.0) != 0.0 && a.nx % TX == 0 && (((size_t)a.nx * a.ny) & 1) == 0 &&
                  (((uintptr_t)a.v & 15) == 0) && (!a.halo_lo || ((uintptr_t)a.halo_lo & 15) == 0) &&
                  (!a.halo_hi || ((uintptr_t)a.halo_hi & 15) == 0);
        P.nt = n >= ((size_t)1 << 22) && ctx->opt("sh_nt", 1.0) != 0.0;
        // part 0: everything; part 1: the chunks that touch no halo plane (1 .. nzc-2); part 2: the two face chunks
        auto launch = [&](int zc0, int count, int step = 1) {
            if (count <= 0) return;
            P.zc0 = zc0; P.zcstep = step;
            P.nblocks = P.ntx * P.nty * count;
            P.grid8 = (P.nblocks + 7) / 8 * 8;
            ProfScope ps(ctx, a.mode == 0 ? "jvp" : "residual", ((a.mode == 0 ? 24.0 : 16.0) + (P.addv ? 8.0 : 0.0)) * n * count / P.nzc);
            if (P.dotp) {
                const bool w3 = ctx->opt("jvp_fd_waves", 3.0) == 3.0;
                if (P.vload && w3) hipLaunchKernelGGL((sh_stream_kernel<true, true, true, 3>), dim3(P.grid8), dim3(256), 0, ctx->stream, P);
                else if (P.vload) hipLaunchKernelGGL((sh_stream_kernel<true, true, true, 2>), dim3(P.grid8), dim3(256), 0, ctx->stream, P);
                else if (w3) hipLaunchKernelGGL((sh_stream_kernel<true, false, true, 3>), dim3(P.grid8), dim3(256), 0, ctx->stream, P);
                else hipLaunchKernelGGL((sh_stream_kernel<true, false, true, 2>), dim3(P.grid8), dim3(256), 0, ctx->stream, P);
                *a.dot_blocks = P.nblocks;
            } else if (dim3d && P.vload) hipLaunchKernelGGL((sh_stream_kernel<true, true>), dim3(P.grid8), dim3(256), 0, ctx->stream, P);
            else if (dim3d) hipLaunchKernelGGL((sh_stream_kernel<true, false>), dim3(P.grid8), dim3(256), 0, ctx->stream, P);
            else if (P.vload) hipLaunchKernelGGL((sh_stream_kernel<false, true>), dim3(P.grid8), dim3(256), 0, ctx->stream, P);
            else hipLaunchKernelGGL((sh_stream_kernel<false, false>), dim3(P.grid8), dim3(256), 0, ctx->stream, P);
        };
        // face chunks: the first one (reads planes -2, -1) and the last one -- the last two when the last chunk is a
        // single plane, because its lower neighbour then reads plane nz
        const int tail = (a.nz - (P.nzc - 1) * zchunk < 2) ? 2 : 1;
        const int inner = (zchunk >= 2) ? P.nzc - 1 - tail : 0;
        if (a.part == 0 || inner <= 0) {
            if (a.part != 1) launch(0, P.nzc);                // nothing can be split off: everything waits for the halos
        } else if (a.part == 1) {
            launch(1, inner);
        } else if (tail == 1) {
            launch(0, 2, P.nzc - 1);                          // both face chunks (first and last) in one launch
        } else {
            launch(0, 1);
            launch(1 + inner, tail);
        }
    }
    BK_HIP(ctx, hipGetLastError());
    return 0;
}

int cgl_apply(bk_ctx* ctx, const CglArgs& a) {
    CglK P{a.nx, a.ny, a.ax, a.ay, a.r, a.mu, a.nu, a.c3, a.c5, a.gamma, a.a0, a.a1, a.mode, a.v, a.u, a.out};
    const size_t n = (size_t)a.nx * a.ny;
    ProfScope ps(ctx, a.mode != 1 ? "jvp" : "residual", (a.mode != 1 ? 48.0 : 32.0) * n);
    hipLaunchKernelGGL(cgl_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, P);
    BK_HIP(ctx, hipGetLastError());
    return 0;
}

int sh1d_apply(bk_ctx* ctx, const Sh1dArgs& a) {
    Sh1dK P{a.nx, a.ax, a.lam, a.nu, a.a0, a.a1, a.mode, a.v, a.u, a.out};
    ProfScope ps(ctx, a.mode == 0 ? "jvp" : "residual", (a.mode == 0 ? 24.0 : 16.0) * a.nx);
    hipLaunchKernelGGL(sh1d_kernel, dim3((a.nx + 255) / 256), dim3(256), 0, ctx->stream, P);
    BK_HIP(ctx, hipGetLastError());
    return 0;
}

}  // namespace bk
