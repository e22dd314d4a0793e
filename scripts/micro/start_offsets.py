"""Start-offset / LDS-layout sweep of the hot kernels at full size (512^3 by default), one JSON line per measurement.

  python scripts/micro/start_offsets.py [n] [parts]      parts: comma list of dct, axpy, jvp (default: all)

* dct : one preconditioner application (5 fused passes back to back, as the solver issues them) with the start offset of
        ONE pass varied at a time (options dct_stagger_<axis><mode>), then with the best value of every pass combined.
        Run it twice -- default library and BKHIP_LIB=.../libbkhip_layout0.so -- for the LDS-layout A/B.
* axpy: the multiaxpy exactly as GMRES issues it (dst = V[k], src = w from another allocation) over phases x map x units.
* jvp : the streaming JVP over sh_stagger.
The last line ("best") carries the winning settings as bench.py --opt arguments.
"""
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
parts = sys.argv[2].split(",") if len(sys.argv) > 2 else ["dct", "axpy", "jvp"]
lib_tag = os.path.basename(os.environ.get("BKHIP_LIB", "libbkhip.so"))
ctx = hip.Context(0)
N = n ** 3
prob = hip.SwiftHohenberg(ctx, (n, n, n), (math.pi * n / 32,) * 3)
g = torch.Generator(device="cuda").manual_seed(0)
v = hip.HipVec(ctx, torch.rand(N, dtype=torch.float64, device="cuda", generator=g))
out = v.similar()
best = {}


def timeit(fn, reps=10, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def emit(**kw):
    print(json.dumps(dict(lib=lib_tag, n=n, **kw)), flush=True)


if "dct" in parts:
    P = hip.DCTPreconditioner(prob, 1.0)
    f = lambda: ctx.check(ctx.lib.bk_precond_apply(P.h, C.c_void_p(v.t.data_ptr()), C.c_void_p(out.t.data_ptr())))
    passes = ["00", "10", "22", "11", "01"]          # x forward, y forward, z round trip, y inverse, x inverse
    ctx.set_option("dct_stagger", 0)
    base = min(timeit(f, 20), timeit(f, 20))
    emit(kernel="precond_apply", variant="stagger 0 everywhere", ms=base * 1e3, frac=80.0 * N / base / 8e12)
    ctx.set_option("dct_stagger", -1)
    chosen = {}
    for ps in passes:
        res = {}
        for s in (0, 1, 2, 3, 4, 6, 8, 12):
            ctx.set_option("dct_stagger_" + ps, s)
            res[s] = timeit(f, 15)
        ctx.set_option("dct_stagger_" + ps, 0)
        sbest = min(res, key=res.get)
        chosen[ps] = sbest if res[sbest] < 0.997 * res[0] else 0
        emit(kernel="precond_apply", variant="pass " + ps, us_by_stagger={k: round(t * 1e6, 1) for k, t in res.items()},
             gain_us=round((res[0] - res[sbest]) * 1e6, 1), chosen=chosen[ps])
    for ps, s in chosen.items():
        ctx.set_option("dct_stagger_" + ps, s)
    comb = min(timeit(f, 20), timeit(f, 20))
    emit(kernel="precond_apply", variant="combined", chosen=chosen, ms=comb * 1e3, frac=80.0 * N / comb / 8e12,
         gain_pct=round(100.0 * (base - comb) / base, 2))
    # one global value for all passes, for comparison
    for s in (2, 4, 6):
        ctx.set_option("dct_stagger", s)
        emit(kernel="precond_apply", variant="all passes stagger %d" % s, ms=timeit(f, 15) * 1e3)
    ctx.set_option("dct_stagger", -1)
    best.update({"dct_stagger_" + ps: s for ps, s in chosen.items() if s})
    del P

if "jvp" in parts:
    u = hip.HipVec(ctx, torch.rand(N, dtype=torch.float64, device="cuda", generator=g))
    J = prob.jacobian(u, 0.1)
    f = lambda: ctx.check(ctx.lib.bk_op_apply(J.h, C.c_void_p(v.t.data_ptr()), 0.0, 1.0, C.c_void_p(out.t.data_ptr())))
    res = {}
    for s in (0, 5, 10, 20, 40, 80, 160):
        ctx.set_option("sh_stagger", s)
        res[s] = min(timeit(f, 15), timeit(f, 15))
    ctx.set_option("sh_stagger", 0)
    sbest = min(res, key=res.get)
    emit(kernel="sh3d_jvp", us_by_stagger={k: round(t * 1e6, 1) for k, t in res.items()}, frac0=24.0 * N / res[0] / 8e12,
         frac_best=24.0 * N / res[sbest] / 8e12, chosen=sbest)
    if res[sbest] < 0.995 * res[0]:
        best["sh_stagger"] = sbest
    del J, u

if "axpy" in parts:
    ld = (N + 31) // 32 * 32
    kmax = 25
    V = torch.rand(ld * (kmax + 1), dtype=torch.float64, device="cuda", generator=g)
    score = {}
    for k in (2, 4, 8, 12, 16, 24):
        cc = (C.c_double * k)(*([0.01] * k))
        dst = V.data_ptr() + 8 * ld * k
        f = lambda: ctx.check(ctx.lib.bk_krylov_multiaxpy(ctx.h, N, C.c_void_p(V.data_ptr()), ld, k, cc, C.c_void_p(v.t.data_ptr()),
                                                          1.0, C.c_void_p(dst), None))
        ctx.set_option("axpy_stagger", 0)
        t0 = min(timeit(f, 6), timeit(f, 6))
        row = {"off": round(8.0 * N * (k + 2) / t0 / 8e12, 4)}
        for ph in (2, 3, 4, 8):
            for mp in (0, 1):
                for units in (30, 55, 110):
                    ctx.set_option("axpy_stagger", ph)
                    ctx.set_option("axpy_stagger_map", mp)
                    ctx.set_option("axpy_stagger_units", units)
                    t = timeit(f, 5, 1)
                    key = "P%d m%d u%d" % (ph, mp, units)
                    row[key] = round(8.0 * N * (k + 2) / t / 8e12, 4)
                    score[key] = score.get(key, 0.0) + (t0 - t) / t0
        emit(kernel="multiaxpy(gmres layout)", k=k, frac_of_8TBs=row)
    ctx.set_option("axpy_stagger", 0)
    kb = max(score, key=score.get)
    emit(kernel="multiaxpy(gmres layout)", variant="mean relative gain", score={k_: round(s / 6, 4) for k_, s in sorted(score.items(), key=lambda kv: -kv[1])[:8]})
    if score[kb] / 6 > 0.01:
        ph, mp, units = [int(x[1:]) for x in kb.split()]
        best.update(axpy_stagger=ph, axpy_stagger_map=mp, axpy_stagger_units=units)
    del V

emit(best=best, bench_args=" ".join("--opt %s=%s" % kv for kv in best.items()))
