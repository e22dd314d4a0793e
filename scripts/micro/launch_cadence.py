"""Launch cadence of dependent small kernels (the cache-resident configs are bound by it): back-to-back JVPs, preconditioner
applications and axpbys on a 512 x 512 grid, and a whole GMRES solve, on the legacy null stream (what
torch.cuda.current_stream() is by default) and -- with the argument `streams` -- on an explicit stream.
Measured (round 2): 3.6 us per dependent launch on either stream, 9.7 us per DCT pass (one tile pipeline), 84 us per
operator application of the preconditioned GMRES (12 launches).  Two ways of cutting launches out of the device-resident
Arnoldi step were built, measured here and removed again: (1) second reduction stage + coefficient logic finished by the last
workgroup of the multidot launch (ticket counter, agent-scope fences): 144 us -- the fences of 512 workgroups cost far more
than the two kernel boundaries they replace; (2) second stage + coefficients as ONE single-workgroup kernel: 92 us -- the
k + 1 values reduce one after the other behind memory latency instead of in k + 1 parallel workgroups."""
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from bk_amd import hip  # noqa: E402


def run(tag):
    ctx = hip.Context(0)
    dims, ls_ = (512, 512), (16 * np.pi, 8 * np.pi / np.sqrt(3))
    prob = hip.SwiftHohenberg(ctx, dims, ls_, l=-0.1, nu=1.3)
    rng = np.random.default_rng(0)
    u = prob.vec(rng.standard_normal(prob.nglobal))
    v = prob.vec(rng.standard_normal(prob.nglobal))
    J = prob.jacobian(u, -0.1)
    P = hip.DCTPreconditioner(prob, 1.0)
    out = v.similar()
    import ctypes as C
    lib = ctx.lib
    fj = lambda: ctx.check(lib.bk_op_apply(J.h, C.c_void_p(v.t.data_ptr()), 0.0, 1.0, C.c_void_p(out.t.data_ptr())))
    fp = lambda: ctx.check(lib.bk_precond_apply(P.h, C.c_void_p(v.t.data_ptr()), C.c_void_p(out.t.data_ptr())))
    fa = lambda: ctx.check(lib.bk_vec_axpby(ctx.h, prob.nglobal, 0.5, C.c_void_p(v.t.data_ptr()), 0.5, C.c_void_p(out.t.data_ptr())))
    res = {}
    for name, f, per in (("jvp", fj, 1), ("precond(3 passes)", fp, 3), ("axpby", fa, 1)):
        for _ in range(20):
            f()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        reps = 500
        for _ in range(reps):
            f()
        torch.cuda.synchronize()
        res[name] = (time.perf_counter() - t0) / reps / per * 1e6
    ls = hip.GMRESKrylovKit(dim=30, rtol=1e-9, atol=1e-12, maxiter=150, Pl=P)
    for _ in range(3):
        x, ok, it = ls(J, v)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        x, ok, it = ls(J, v)
    torch.cuda.synchronize()
    res["gmres_us_per_operator_application"] = (time.perf_counter() - t0) / 20 / it * 1e6
    res["gmres_numops"] = it
    print(json.dumps(dict(stream=tag, us_per_launch=res)), flush=True)
    ctx.close()


run("null (torch default)")
if len(sys.argv) > 1 and sys.argv[1] == "streams":          # measured: no difference (3.6 us per dependent launch on both)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        run("explicit torch.cuda.Stream")
