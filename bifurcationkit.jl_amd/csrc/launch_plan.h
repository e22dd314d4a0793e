// Host-side launch planning shared by the kernels' launchers and replayed on the CPU by tests/test_launch_plan_host.py.
#pragma once

namespace bk {

// z-chunk length of the streaming Swift-Hohenberg kernel (stencil.hip).  The kernel keeps `resident` workgroups on the device
// at a time (3 per CU: 162 VGPRs), a workgroup owns one (tile, chunk) pair and walks chunk + 4 planes (4 to prime its plane
// pipeline).  The plan minimises  rounds * planes per workgroup,  rounds = ceil(chunks * tiles / resident):
// 512^3 (256 tiles, 768 resident): 3 chunks of 171 planes = exactly one round -- 6.8 % faster than the 8 x 64 of rounds 1-2
// (2.67 rounds); the 64-plane slab of 8 ranks: 3 x 22 against 4 x 16, 17 % (profiles/r2_jvp_zchunk_sweep_512.jsonl).
// split: the halo exchange is overlapped, so the interior chunks and the two face chunks are separate launches.
// Chunks are never shorter than 8 planes unless the slab itself is.
inline int sh_plan_zchunk(int nz, int tiles, long resident, bool split) {
    if (nz < 1) return 1;
    if (tiles < 1) tiles = 1;
    if (resident < 1) resident = 1;
    long best = -1;
    int zchunk = nz;
    for (int c = 1; c <= 64; ++c) {
        const int zc = (nz + c - 1) / c;
        if (zc < 8 && c > 1) break;
        const int nzc = (nz + zc - 1) / zc;
        const long per = zc + 4;
        long cost;
        if (split && nzc >= 3) cost = (((long)(nzc - 2) * tiles + resident - 1) / resident + (2L * tiles + resident - 1) / resident) * per;
        else cost = (((long)nzc * tiles + resident - 1) / resident) * per;
        if (best < 0 || cost < best) { best = cost; zchunk = zc; }
    }
    return zchunk;
}

}  // namespace bk
