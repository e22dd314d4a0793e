"""Diagnostic: MINRES / CG iterates with the two fused passes switched separately (options jvp_fused_dot, dct_fused_dot,
minres_fused) on a few 3-D SH grids."""
import sys
import numpy as np
sys.path.insert(0, ".")
from bk_amd import hip

ctx = hip.Context(0)
ls3 = (np.pi, 2.5, 2.0)
sizes = ((64, 64, 64), (70, 34, 20), (256, 256, 256)) if len(sys.argv) < 2 else tuple((int(a),) * 3 for a in sys.argv[1:])
for dims in sizes:
    prob = hip.SwiftHohenberg(ctx, dims, ls3)
    rng = np.random.default_rng(sum(dims))
    n = int(np.prod(dims))
    x = np.linspace(-ls3[0], ls3[0], dims[0])
    u = (np.cos(x)[:, None, None] * np.ones((1, dims[1], dims[2]))).transpose(2, 1, 0).ravel() * 0.3 + 0.1 * rng.standard_normal(n)
    J = prob.jacobian(prob.vec(u), 0.1)
    rhs = rng.standard_normal(n)
    P = hip.DCTPreconditioner(prob, 1.0)
    for alg, (a0, a1) in (("minres", (-0.1, 1.0)), ("cg", (2.0, -1.0))):
        ref = None
        hists = []
        for mf, jf, df in ((0, 0, 0), (1, 2, 1), (1, 3, 1)):
            ctx.set_option("minres_fused", mf); ctx.set_option("jvp_fused_dot", 1 if jf else 0); ctx.set_option("jvp_fd_waves", jf if jf else 3); ctx.set_option("dct_fused_dot", df)
            ctx.set_option("solver_trace", 1); ctx.solver_history(reset=True)
            xs, ok, it = hip.KrylovLSSymmetric(alg, atol=1e-13, rtol=1e-10, Pl=P)(J, prob.vec(rhs), a0, a1)
            h = ctx.solver_history(reset=True); ctx.set_option("solver_trace", 0)
            hists.append(np.array(h[0]) if h else None)
            xs = xs.numpy()
            r = rhs - (a0 * xs + a1 * J(prob.vec(xs)).numpy())
            if ref is None:
                ref = xs
            prof = ""
            if dims[0] >= 256:
                ctx.prof_enable(True); ctx.prof_reset()
                hip.KrylovLSSymmetric(alg, atol=1e-13, rtol=1e-10, Pl=P)(J, prob.vec(rhs), a0, a1)
                prof = {k: {a: round(float(b), 3) for a, b in ctx.prof_get(k).items()} for k in ("jvp", "dct_pass", "blas1")}
                ctx.prof_enable(False)
            if prof:
                print(prof)
            print(dims, alg, "fused", mf, "jvp", jf, "dct", df, "ok", ok, "it", it, "res %.3e" % (np.linalg.norm(r) / np.linalg.norm(rhs)),
                  "dx %.3e" % (np.abs(xs - ref).max() / np.abs(ref).max()), flush=True)

        if alg == "minres" and hists[0] is not None:
            for i, h in enumerate(hists[1:], 1):
                m = min(len(h), len(hists[0]))
                print("  history rel. deviation of variant", i, " ".join("%.1e" % (abs(h[j] - hists[0][j]) / hists[0][j]) for j in range(m)))
            print("  history (unfused)", " ".join("%.2e" % v for v in hists[0]))
