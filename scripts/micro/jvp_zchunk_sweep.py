"""JVP / residual time against the z-chunk length of the streaming kernel, at the full grid and at the z-slabs of 2 / 4 / 8 ranks.
Usage: python scripts/micro/jvp_zchunk_sweep.py [n]"""
import ctypes as C
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from bk_amd import hip  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
ctx = hip.Context(0)
for nz in (n, n // 2, n // 4, n // 8):
    N = n * n * nz
    prob = hip.SwiftHohenberg(ctx, (n, n, nz), (math.pi * n / 32, math.pi * n / 32, math.pi * nz / 32))
    g = torch.Generator(device="cuda").manual_seed(0)
    u = hip.HipVec(ctx, torch.rand(N, dtype=torch.float64, device="cuda", generator=g))
    v = hip.HipVec(ctx, torch.rand(N, dtype=torch.float64, device="cuda", generator=g))
    out = v.similar()
    J = prob.jacobian(u, 0.1)
    f = lambda: ctx.check(ctx.lib.bk_op_apply(J.h, C.c_void_p(v.t.data_ptr()), 0.0, 1.0, C.c_void_p(out.t.data_ptr())))
    cands = sorted({0} | {math.ceil(nz / c) for c in (1, 2, 3, 4, 5, 6, 8, 9, 12, 16) if math.ceil(nz / c) >= 8})
    for rep in range(2):
        for zc in cands:
            ctx.set_option("sh_zchunk", zc)
            for _ in range(3):
                f()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(20):
                f()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 20
            print(json.dumps(dict(kernel="sh3d_jvp", grid=[n, n, nz], zchunk=zc, chunks=(math.ceil(nz / zc) if zc else None), us=dt * 1e6,
                                  gbs=24.0 * N / dt / 1e9, rep=rep)), flush=True)
    ctx.set_option("sh_zchunk", 0)
    del u, v, out, J, prob
