// Micro-benchmark (experiment, not part of the library): copy bandwidth of the DCT passes' tile access pattern.
// A workgroup of 256 lanes copies a tile of LT lines x N elements of a [n2][n1][n0] fp64 array with 16-byte accesses:
// element n of line l sits at base + l + n*stride (stride = n0: y pass, n0*n1: z pass), so one wave instruction touches
// 64 / (LT/2) segments of LT*8 bytes.  The workgroup holds `lds` bytes of LDS so that the occupancy matches the FFT
// kernel (76 KiB -> 2 workgroups per CU).   Build: hipcc --offload-arch=gfx950 -O3 seg_copy_bench.hip -o seg_copy_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef double d2 __attribute__((ext_vector_type(2)));

template <int NL>   // NL 16-byte loads per lane per tile
__global__ void __launch_bounds__(256, 2) tile_copy(const double* in, double* out, int n0, int n1, int n2, int axis, int LT,
                                                    int N, int tiles_x, int ntiles, int nt, int nch, int xmap) {
    extern __shared__ double smem[];
    const int tid = threadIdx.x;
    const int npairs = LT >> 1;
    const size_t stride = axis == 1 ? (size_t)n0 : (size_t)n0 * n1;
    for (int tile0 = blockIdx.x; tile0 < ntiles; tile0 += gridDim.x) {
        // xmap 1: the workgroups of one XCD (blockIdx % 8) take a contiguous range of tiles
        const int tile = xmap ? (tile0 & 7) * (ntiles >> 3) + (tile0 >> 3) : tile0;
        const int chunk = tile % nch, t2 = tile / nch;   // a tile covers N of the n elements along the axis
        const int tx = t2 % tiles_x, other = t2 / tiles_x;
        const size_t base = axis == 0 ? (size_t)t2 * LT * n0 + (size_t)chunk * N
                                      : (axis == 1 ? (size_t)tx * LT + (size_t)n0 * n1 * other : (size_t)tx * LT + (size_t)n0 * other) +
                                            (size_t)chunk * N * stride;
        d2 v[NL];
#pragma unroll
        for (int u = 0; u < NL; ++u) {
            const int q = tid + u * 256;                 // item: pair (q % npairs), element n = q / npairs
            const size_t off = axis == 0 ? (size_t)2 * (q % (N / 2)) + (size_t)(q / (N / 2)) * n0 : (size_t)2 * (q % npairs) + (size_t)(q / npairs) * stride;
            const d2* p = reinterpret_cast<const d2*>(in + base + off);
            v[u] = nt ? __builtin_nontemporal_load(p) : *p;
        }
        if (smem[tid] == 12345.678) v[0].x += 1.0;       // keep the LDS allocation alive
#pragma unroll
        for (int u = 0; u < NL; ++u) {
            const int q = tid + u * 256;
            const size_t off = axis == 0 ? (size_t)2 * (q % (N / 2)) + (size_t)(q / (N / 2)) * n0 : (size_t)2 * (q % npairs) + (size_t)(q / npairs) * stride;
            d2* p = reinterpret_cast<d2*>(out + base + off);
            if (nt) __builtin_nontemporal_store(v[u], p); else *p = v[u];
        }
    }
}

int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 512;
    const size_t tot = (size_t)n * n * n;
    double *a, *b;
    hipMalloc(&a, tot * 8); hipMalloc(&b, tot * 8);
    hipMemset(a, 0, tot * 8); hipMemset(b, 0, tot * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipFuncSetAttribute(reinterpret_cast<const void*>(tile_copy<16>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute(reinterpret_cast<const void*>(tile_copy<32>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    printf("{\"n\": %d, \"rows\": [\n", n);
    bool first = true;
    for (int axis = 0; axis < 3; ++axis)
        for (int LT : {16, 32, 64, 128})
            for (int NL : {16})
                for (int lds_kb : {76})
                    for (int persist : {0, 2}) {
                        // tile = LT lines x N elements with LT * N = NL * 512 doubles
                        const int N = NL * 512 / LT;
                        if (N > n || N < 64 || (axis == 0 && LT != 16)) continue;
                        const int nch = n / N;
                        const int tiles_x = axis == 0 ? 1 : n / LT;
                        const int ntiles = (axis == 0 ? (int)((size_t)n * n / LT) : tiles_x * n) * nch;
                        const int grid = ntiles;   // (persistent static grids measured slower: profiles/r2_seg_copy_512.json, first run)
                        const int xmap = persist ? 1 : 0;
                        float best = 1e30f;
                        for (int rep = 0; rep < 6; ++rep) {
                            hipEventRecord(e0, 0);
                            if (NL == 16) hipLaunchKernelGGL(tile_copy<16>, dim3(grid), dim3(256), lds_kb * 1024, 0, a, b, n, n, n, axis, LT, N, tiles_x, ntiles, 1, nch, xmap);
                            else hipLaunchKernelGGL(tile_copy<32>, dim3(grid), dim3(256), lds_kb * 1024, 0, a, b, n, n, n, axis, LT, N, tiles_x, ntiles, 1, nch, xmap);
                            hipEventRecord(e1, 0);
                            hipEventSynchronize(e1);
                            float ms; hipEventElapsedTime(&ms, e0, e1);
                            if (rep >= 2 && ms < best) best = ms;
                        }
                        printf("%s {\"axis\": %d, \"LT\": %d, \"seg_bytes\": %d, \"loads_per_lane\": %d, \"lds_kb\": %d, \"xcd_contiguous\": %d, \"us\": %.1f, \"TBs\": %.2f}",
                               first ? "" : ",\n", axis, LT, LT * 8, NL, lds_kb, xmap, best * 1e3, 2.0 * tot * 8 / (best * 1e-3) / 1e12);
                        first = false;
                    }
    printf("\n]}\n");
    return hipGetLastError() == hipSuccess ? 0 : 1;
}
