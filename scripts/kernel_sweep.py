"""Kernel micro-benchmarks on the GPU box: HBM GB/s of the hot kernels at full size (512^3 = 1 GiB vectors).
Prints one JSON line per measurement.  Usage: python scripts/kernel_sweep.py [size]"""
import ctypes as C
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from bk_amd import hip  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
what = sys.argv[2].split(",") if len(sys.argv) > 2 else ["jvp", "krylov", "blas", "precond"]
ctx = hip.Context(0)

if "slabemu" in what:
    # local cost of one preconditioner application on the z-slab one of R ranks owns (512 x 512 x 512/R), transposed z pass
    # replaced by the slab z-solve (two local round trips + the face kernels; the two small all-to-alls are local copies here)
    for R in (8, 4, 2):
        nzl = n // R
        N_ = n * n * nzl
        g_ = torch.Generator(device="cuda").manual_seed(0)
        v_ = hip.HipVec(ctx, torch.rand(N_, dtype=torch.float64, device="cuda", generator=g_))
        o_ = v_.similar()
        # (emu, split): transposed z pass / slab z-solve as two round trips (round 2) / as forward + inverse halves (round 5)
        for emu, split in ((0, 1), (R, 0), (R, 1)):
            ctx.set_option("dct_slab_emulate", emu)
            ctx.set_option("dct_slab_split", split)
            pr = hip.SwiftHohenberg(ctx, (n, n, nzl), (math.pi * n / 32, math.pi * n / 32, math.pi * nzl / 32))
            P_ = hip.DCTPreconditioner(pr, 1.0)
            f_ = lambda: ctx.check(ctx.lib.bk_precond_apply(P_.h, C.c_void_p(v_.t.data_ptr()), C.c_void_p(o_.t.data_ptr())))
            for _ in range(2):
                f_()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(10):
                f_()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 10
            print(json.dumps(dict(kernel="precond_apply_on_slab", n=n, R=R, nzl=nzl, slab_zsolve=bool(emu), halves=bool(emu and split), ms=dt * 1e3,
                                  passes=6 if emu else 5, gbs=16.0 * N_ * (6 if emu else 5) / dt / 1e9)), flush=True)
            del P_, pr
        ctx.set_option("dct_slab_emulate", 0)
        ctx.set_option("dct_slab_split", 1)
        del v_, o_
    sys.exit(0)

import os
for kv in os.environ.get("BK_OPTS", "").split(","):
    if "=" in kv:
        ctx.set_option(kv.split("=")[0], float(kv.split("=")[1]))
N = n ** 3
prob = hip.SwiftHohenberg(ctx, (n, n, n), (math.pi * n / 32,) * 3)
g = torch.Generator(device="cuda").manual_seed(0)
u = hip.HipVec(ctx, torch.rand(N, dtype=torch.float64, device="cuda", generator=g))
v = hip.HipVec(ctx, torch.rand(N, dtype=torch.float64, device="cuda", generator=g))
out = v.similar()


def timeit(fn, reps=5, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def report(name, secs, nbytes, **kw):
    print(json.dumps(dict(kernel=name, n=n, ms=secs * 1e3, gbs=nbytes / secs / 1e9, frac_of_8TBs=nbytes / secs / 8e12, **kw)),
          flush=True)


if "jvp" in what:
    J = prob.jacobian(u, 0.1)
    lib = ctx.lib
    pv = (C.c_double * 2)(0.1, 1.2)
    for variant, zchunks in ((0, [0]), (1, [0, 8, 16, 32, 64, 128, 512])):
        ctx.set_option("sh_kernel", variant)
        for zc in zchunks:
            ctx.set_option("sh_zchunk", zc)
            f = lambda: ctx.check(lib.bk_op_apply(J.h, C.c_void_p(v.t.data_ptr()), 0.0, 1.0, C.c_void_p(out.t.data_ptr())))
            report("sh3d_jvp", timeit(f), 24.0 * N, variant=variant, zchunk=zc)
            f = lambda: ctx.check(lib.bk_residual(prob.h, C.c_void_p(u.t.data_ptr()), pv, 2, C.c_void_p(out.t.data_ptr())))
            report("sh3d_residual", timeit(f), 16.0 * N, variant=variant, zchunk=zc)
    ctx.set_option("sh_kernel", 1)
    ctx.set_option("sh_zchunk", 0)

if "blas" in what:
    report("copy(torch)", timeit(lambda: out.t.copy_(v.t)), 16.0 * N)
    report("axpby", timeit(lambda: out.add_(v, 0.5, 2.0)), 24.0 * N)
    report("dot", timeit(lambda: u.inner(v)), 16.0 * N)
    report("nrm2", timeit(lambda: u.norm()), 8.0 * N)
    report("nrminf", timeit(lambda: u.norminf()), 8.0 * N)

if "krylov" in what:
    ld = (N + 31) // 32 * 32
    kmax = 30 if n >= 512 else 45
    V = torch.rand(ld * kmax, dtype=torch.float64, device="cuda", generator=g)
    hbuf = (C.c_double * 65)()
    for k in (1, 4, 8, 16, 24, 30, 45):
        if k > kmax:
            continue
        f = lambda: ctx.check(ctx.lib.bk_krylov_multidot(ctx.h, N, C.c_void_p(V.data_ptr()), ld, k, C.c_void_p(v.t.data_ptr()), hbuf))
        report("multidot", timeit(f, reps=3, warm=1), 8.0 * N * (k + 1), k=k)
        cc = (C.c_double * k)(*([0.01] * k))
        f = lambda: ctx.check(ctx.lib.bk_krylov_multiaxpy(ctx.h, N, C.c_void_p(V.data_ptr()), ld, k, cc, C.c_void_p(v.t.data_ptr()),
                                                          1.0, C.c_void_p(out.t.data_ptr()), None))
        report("multiaxpy", timeit(f, reps=3, warm=1), 8.0 * N * (k + 2), k=k)
    del V

if "gmrespad" in what:
    # the multiaxpy exactly as GMRES issues it: dst = V[k] (the next basis vector, k strides above V[0]), src = w from another
    # allocation; basis stride = N + pad.  Which pads separate the write stream from the concurrent read streams?
    kmax = 13
    hbuf = (C.c_double * 65)()
    for pad in (0, 32, 64, 128, 256, 512, 544, 1024, 2048, 2080, 4096, 8192, 8224, 32768, 66080, 131072, 262144):
        ld = (N + 31) // 32 * 32 + pad
        V = torch.rand(ld * (kmax + 1), dtype=torch.float64, device="cuda", generator=g)
        for k in (2, 4, 8, 12):
            cc = (C.c_double * k)(*([0.01] * k))
            dstp = C.c_void_p(V.data_ptr() + 8 * ld * k)
            fa = lambda: ctx.check(ctx.lib.bk_krylov_multiaxpy(ctx.h, N, C.c_void_p(V.data_ptr()), ld, k, cc, C.c_void_p(v.t.data_ptr()),
                                                               1.0, dstp, None))
            fd = lambda: ctx.check(ctx.lib.bk_krylov_multidot(ctx.h, N, C.c_void_p(V.data_ptr()), ld, k, C.c_void_p(v.t.data_ptr()), hbuf))
            report("multiaxpy", timeit(fa, reps=6, warm=1), 8.0 * N * (k + 2), k=k, pad=pad)
            report("multidot", timeit(fd, reps=6, warm=1), 8.0 * N * (k + 1), k=k, pad=pad)
        del V
    torch.cuda.empty_cache()

if "burst" in what:
    # GMRES layout (dst = V[k]): the contiguous-burst kernels (krylov_burst = 1, the default for HBM-sized vectors) against the
    # round-2 grid-stride kernels (0), interleaved
    kmax = 30 if n >= 512 else 32
    hbuf = (C.c_double * 65)()
    ld = (N + 31) // 32 * 32
    V = torch.rand(ld * (kmax + 1), dtype=torch.float64, device="cuda", generator=g)
    for k in (1, 2, 4, 6, 8, 10, 12, 16, 20, 24, 30):
        if k > kmax:
            continue
        cc = (C.c_double * k)(*([0.01] * k))
        dstp = C.c_void_p(V.data_ptr() + 8 * ld * k)
        fa = lambda: ctx.check(ctx.lib.bk_krylov_multiaxpy(ctx.h, N, C.c_void_p(V.data_ptr()), ld, k, cc, C.c_void_p(v.t.data_ptr()),
                                                           1.0, dstp, None))
        fd = lambda: ctx.check(ctx.lib.bk_krylov_multidot(ctx.h, N, C.c_void_p(V.data_ptr()), ld, k, C.c_void_p(v.t.data_ptr()), hbuf))
        for rep in range(2):
            for b in (0, 1):
                ctx.set_option("krylov_burst", b)
                report("multiaxpy", timeit(fa, reps=6, warm=1), 8.0 * N * (k + 2), k=k, burst=b, rep=rep)
                report("multidot", timeit(fd, reps=6, warm=1), 8.0 * N * (k + 1), k=k, burst=b, rep=rep)
    ctx.set_option("krylov_burst", 1)
    del V

if "axpy" in what:
    ld = (N + 31) // 32 * 32
    V = torch.rand(ld * 16, dtype=torch.float64, device="cuda", generator=g)
    for k in (4, 16):
        cc = (C.c_double * k)(*([0.01] * k))
        f = lambda: ctx.check(ctx.lib.bk_krylov_multiaxpy(ctx.h, N, C.c_void_p(V.data_ptr()), ld, k, cc, C.c_void_p(v.t.data_ptr()),
                                                          1.0, C.c_void_p(out.t.data_ptr()), None))
        for nt in (0, 1):
            for blocks in (1024, 2048, 4096, 8192, 16384, 65536):
                ctx.set_option("axpy_nt", nt)
                ctx.set_option("axpy_blocks", blocks)
                report("multiaxpy", timeit(f, reps=3, warm=1), 8.0 * N * (k + 2), k=k, nt=nt, blocks=blocks)
    ctx.set_option("axpy_nt", 0)
    ctx.set_option("axpy_blocks", 4096)
    del V

if "variants" in what:
    # round-2 A/B variants behind context options, interleaved repetitions (same buffers, same clocks)
    ld = (N + 31) // 32 * 32
    V = torch.rand(ld * 16, dtype=torch.float64, device="cuda", generator=g)
    hbuf = (C.c_double * 65)()
    for k in (4, 8, 12, 16):
        cc = (C.c_double * k)(*([0.01] * k))
        fa = lambda: ctx.check(ctx.lib.bk_krylov_multiaxpy(ctx.h, N, C.c_void_p(V.data_ptr()), ld, k, cc, C.c_void_p(v.t.data_ptr()),
                                                           1.0, C.c_void_p(out.t.data_ptr()), None))
        fd = lambda: ctx.check(ctx.lib.bk_krylov_multidot(ctx.h, N, C.c_void_p(V.data_ptr()), ld, k, C.c_void_p(v.t.data_ptr()), hbuf))
        for rep in range(2):
            for var in (0, 1, 2, 3):
                ctx.set_option("axpy_variant", var)
                report("multiaxpy", timeit(fa, reps=5, warm=1), 8.0 * N * (k + 2), k=k, variant=var, rep=rep)
            for var in (0, 1):
                ctx.set_option("dot_variant", var)
                report("multidot", timeit(fd, reps=5, warm=1), 8.0 * N * (k + 1), k=k, variant=var, rep=rep)
        ctx.set_option("axpy_variant", 0)
        ctx.set_option("dot_variant", 0)
    for blocks in (512, 768, 1024, 1536, 2048):
        ctx.set_option("axpy_blocks", blocks)
        k = 8
        cc = (C.c_double * k)(*([0.01] * k))
        fa = lambda: ctx.check(ctx.lib.bk_krylov_multiaxpy(ctx.h, N, C.c_void_p(V.data_ptr()), ld, k, cc, C.c_void_p(v.t.data_ptr()),
                                                           1.0, C.c_void_p(out.t.data_ptr()), None))
        report("multiaxpy", timeit(fa, reps=5, warm=1), 8.0 * N * (k + 2), k=k, blocks=blocks)
    ctx.set_option("axpy_blocks", 1024)
    del V
    P = hip.DCTPreconditioner(prob, 1.0)
    o2 = v.similar()
    f = lambda: ctx.check(ctx.lib.bk_precond_apply(P.h, C.c_void_p(v.t.data_ptr()), C.c_void_p(out.t.data_ptr())))
    f()
    ref = out.copy()
    for rep in range(2):
        for steps in (2, 1):
            ctx.set_option("dct_rcp_steps", steps)
            t = timeit(f, reps=5, warm=1)
            err = out.copy().add_(ref, -1.0).norminf() / ref.norminf()
            report("dct_precond", t, 80.0 * N, rcp_steps=steps, rel_diff_vs_2steps=err, rep=rep)
    ctx.set_option("dct_rcp_steps", 2)

if "pad" in what:
    # stride of the Krylov basis: N doubles = 2^30 bytes at 512^3 puts element i of every basis vector on the same HBM
    # channel / bank; pad the stride by a few KiB and compare (interleaved repetitions)
    kmax = 13
    pads = (0, 32, 512, 2048 + 32, 8192 + 32, 65536 + 512 + 32)
    hbuf = (C.c_double * 65)()
    V = torch.rand((N + max(pads)) * kmax, dtype=torch.float64, device="cuda", generator=g)
    for k in (4, 8, 12):
        cc = (C.c_double * k)(*([0.01] * k))
        for rep in range(2):
            for pad in pads:
                ld = N + pad
                dst = V[ld * k:].data_ptr()      # the next basis vector, as in the Arnoldi step
                fa = lambda: ctx.check(ctx.lib.bk_krylov_multiaxpy(ctx.h, N, C.c_void_p(V.data_ptr()), ld, k, cc, C.c_void_p(v.t.data_ptr()),
                                                                   1.0, C.c_void_p(dst), None))
                fd = lambda: ctx.check(ctx.lib.bk_krylov_multidot(ctx.h, N, C.c_void_p(V.data_ptr()), ld, k, C.c_void_p(v.t.data_ptr()), hbuf))
                report("multiaxpy", timeit(fa, reps=5, warm=1), 8.0 * N * (k + 2), k=k, pad=pad, rep=rep)
                report("multidot", timeit(fd, reps=5, warm=1), 8.0 * N * (k + 1), k=k, pad=pad, rep=rep)
    del V

if "xcd" in what:
    # XCD-blocked streaming (vec_xcd_map) A/B, interleaved repetitions
    ld = (N + 31) // 32 * 32
    V = torch.rand(ld * 16, dtype=torch.float64, device="cuda", generator=g)
    hbuf = (C.c_double * 65)()
    for k in (4, 8, 12, 16):
        cc = (C.c_double * k)(*([0.01] * k))
        fa = lambda: ctx.check(ctx.lib.bk_krylov_multiaxpy(ctx.h, N, C.c_void_p(V.data_ptr()), ld, k, cc, C.c_void_p(v.t.data_ptr()),
                                                           1.0, C.c_void_p(out.t.data_ptr()), None))
        fd = lambda: ctx.check(ctx.lib.bk_krylov_multidot(ctx.h, N, C.c_void_p(V.data_ptr()), ld, k, C.c_void_p(v.t.data_ptr()), hbuf))
        for rep in range(2):
            for m in (0, 2):
                ctx.set_option("vec_xcd_map", m)
                report("multiaxpy", timeit(fa, reps=5, warm=1), 8.0 * N * (k + 2), k=k, xcd_map=m, rep=rep)
                report("multidot", timeit(fd, reps=5, warm=1), 8.0 * N * (k + 1), k=k, xcd_map=m, rep=rep)
    ctx.set_option("vec_xcd_map", 1)
    del V

if "jvp2" in what:
    J = prob.jacobian(u, 0.1)
    lib = ctx.lib
    pv = (C.c_double * 2)(0.1, 1.2)
    fj = lambda: ctx.check(lib.bk_op_apply(J.h, C.c_void_p(v.t.data_ptr()), 0.0, 1.0, C.c_void_p(out.t.data_ptr())))
    fr = lambda: ctx.check(lib.bk_residual(prob.h, C.c_void_p(u.t.data_ptr()), pv, 2, C.c_void_p(out.t.data_ptr())))
    for rep in range(2):
        for vl, nt, mw in ((0, 0, 1), (0, 1, 1), (1, 0, 1), (1, 1, 1)):
            ctx.set_option("sh_vload", vl)
            ctx.set_option("sh_nt", nt)
            ctx.set_option("sh_minw", mw)
            report("sh3d_jvp", timeit(fj, reps=8, warm=2), 24.0 * N, vload=vl, nt=nt, minw=mw, rep=rep)
            report("sh3d_residual", timeit(fr, reps=8, warm=2), 16.0 * N, vload=vl, nt=nt, minw=mw, rep=rep)
    for zc in (16, 32, 64):
        ctx.set_option("sh_zchunk", zc)
        report("sh3d_jvp", timeit(fj, reps=8, warm=2), 24.0 * N, vload=1, nt=1, minw=mw, zchunk=zc)
    ctx.set_option("sh_zchunk", 0)
    ctx.set_option("sh_vload", 1)
    ctx.set_option("sh_nt", 1)
    ctx.set_option("sh_minw", 1)

if "nt2" in what:
    # second batch of round-2 variants: 16-byte plane staging / non-temporal hints in the JVP, DCT passes, BLAS-1
    J = prob.jacobian(u, 0.1)
    lib = ctx.lib
    pv = (C.c_double * 2)(0.1, 1.2)
    fj = lambda: ctx.check(lib.bk_op_apply(J.h, C.c_void_p(v.t.data_ptr()), 0.0, 1.0, C.c_void_p(out.t.data_ptr())))
    fr = lambda: ctx.check(lib.bk_residual(prob.h, C.c_void_p(u.t.data_ptr()), pv, 2, C.c_void_p(out.t.data_ptr())))
    for rep in range(2):
        for vl, nt in ((0, 0), (1, 0), (0, 1), (1, 1)):
            ctx.set_option("sh_vload", vl)
            ctx.set_option("sh_nt", nt)
            report("sh3d_jvp", timeit(fj, reps=8, warm=2), 24.0 * N, vload=vl, nt=nt, rep=rep)
            report("sh3d_residual", timeit(fr, reps=8, warm=2), 16.0 * N, vload=vl, nt=nt, rep=rep)
    ctx.set_option("sh_vload", 1)
    ctx.set_option("sh_nt", 1)
    P = hip.DCTPreconditioner(prob, 1.0)
    f = lambda: ctx.check(ctx.lib.bk_precond_apply(P.h, C.c_void_p(v.t.data_ptr()), C.c_void_p(out.t.data_ptr())))
    for rep in range(2):
        for nl, ns in ((0, 0), (1, 0), (0, 1), (1, 1)):
            ctx.set_option("dct_nt_load", nl)
            ctx.set_option("dct_nt_store", ns)
            report("dct_precond", timeit(f, reps=5, warm=1), 80.0 * N, nt_load=nl, nt_store=ns, rep=rep)
    ctx.set_option("dct_nt_load", 1)
    ctx.set_option("dct_nt_store", 1)
    for rep in range(2):
        for h in (0, 1):
            ctx.set_option("nt_hint", h)
            report("axpby", timeit(lambda: out.add_(v, 0.5, 2.0), reps=8, warm=2), 24.0 * N, nt_hint=h, rep=rep)
            report("dot", timeit(lambda: u.inner(v), reps=8, warm=2), 16.0 * N, nt_hint=h, rep=rep)
            report("scale", timeit(lambda: out.scale_(1.0000001), reps=8, warm=2), 16.0 * N, nt_hint=h, rep=rep)
    ctx.set_option("nt_hint", 1)

if "precond" in what:
    P = hip.DCTPreconditioner(prob, 1.0)
    f = lambda: ctx.check(ctx.lib.bk_precond_apply(P.h, C.c_void_p(v.t.data_ptr()), C.c_void_p(out.t.data_ptr())))
    if not os.environ.get("BK_SWEEP_FAST"):
        ctx.set_option("dct_fft", 0)
        report("dct_precond", timeit(f, reps=1, warm=1), 16.0 * N, fft=0)
    ctx.set_option("dct_fft", 1)
    for rt in (0, 1):
        for nt in (256, 512):
            ctx.set_option("dct_roundtrip", rt)
            ctx.set_option("dct_threads", nt)
            passes = 5 if rt else 6
            t = timeit(f, reps=3, warm=1)
            report("dct_precond", t, 16.0 * N, fft=1, roundtrip=rt, threads=nt, passes=passes,
                   hbm_gbs_all_passes=16.0 * N * passes / t / 1e9)
