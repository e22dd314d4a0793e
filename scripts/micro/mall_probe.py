"""Does a producer -> consumer pair through the 256-MB Infinity Cache beat HBM?  BLAS-1 kernels on vectors of 16 MiB ... 1 GiB:
effective GB/s of  y <- a x + b y  (2 reads + 1 write), of  x . y  (2 reads) and of a write-then-read pair
(z <- a x ; z . z) whose intermediate either fits the cache or not.  One JSON line per size."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from bk_amd import hip  # noqa: E402

ctx = hip.Context(0)


def timeit(fn, reps=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


for lg in range(21, 28):
    n = 1 << lg
    g = torch.Generator(device="cuda").manual_seed(0)
    x = hip.HipVec(ctx, torch.rand(n, dtype=torch.float64, device="cuda", generator=g))
    y = hip.HipVec(ctx, torch.rand(n, dtype=torch.float64, device="cuda", generator=g))
    z = x.similar()
    t_axpby = timeit(lambda: y.add_(x, 1e-9, 1.0))
    t_dot = timeit(lambda: x.inner(y))

    def pair():
        ctx.check(ctx.lib.bk_vec_copy(ctx.h, n, hip._ptr(x.t), hip._ptr(z.t)))
        z.scale_(1.0)
        return z.inner(z)
    t_pair = timeit(pair)
    print(json.dumps(dict(mib=n * 8 / 2 ** 20, axpby_gbs=24.0 * n / t_axpby / 1e9, dot_gbs=16.0 * n / t_dot / 1e9,
                          dot_us=t_dot * 1e6, axpby_us=t_axpby * 1e6,
                          copy_scale_nrm2_gbs=(16.0 + 16.0 + 8.0) * n / t_pair / 1e9, pair_us=t_pair * 1e6)), flush=True)
    del x, y, z
    torch.cuda.empty_cache()
