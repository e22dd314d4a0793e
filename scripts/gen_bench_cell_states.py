#!/usr/bin/env python
"""Generate tests/golden/bench_cell_states.npz: the two Newton-converged hexagon cell solutions (64 x 32 x 32, the bench's cell)
the `fixed_input` record of bench.py starts from -- computed ONCE by the CPU oracle (reference formulation: assembled sparse L1,
MGS2 GMRES(30), Pl = (L1 + I)^-1; examples/SH3d.jl:127 `sol_hexa` pattern, l = 0.1, nu = 1.2) and committed, so that the headline's
operator-application count no longer depends on the last digits of each build's own cell Newton solves (Next 7).

    python scripts/gen_bench_cell_states.py        # ~10 s, single thread
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from oracle import krylov, operators, palc  # noqa: E402

ds = -0.001
shc = operators.SwiftHohenberg(bench.CELL, bench.CELL_L)
pc = palc.Problem(lambda x, p: shc.F(x, p, 1.2), lambda x, p: (lambda dx: shc.dF(x, p, 1.2, dx)))
Plc = operators.dct_preconditioner(bench.CELL, bench.CELL_L, 1.0)
ls = lambda J, r, a0=0.0, a1=1.0: krylov.gmres_krylovkit(J, r, a0, a1, krylovdim=30, maxiter=150, rtol=1e-9, atol=1e-12, Pl=Plc)[:3]
p0, p1 = 0.1, 0.1 + ds / 150.0
c0 = palc.newton(pc, bench.hex_guess_np(), p0, ls, tol=1e-10, max_iterations=40, normN=palc.norminf)
c1 = palc.newton(pc, c0["u"], p1, ls, tol=1e-10, max_iterations=20, normN=palc.norminf)
assert c0["converged"] and c1["converged"], (c0["residuals"], c1["residuals"])
out = os.path.join(ROOT, "tests", "golden", "bench_cell_states.npz")
np.savez_compressed(out, u0=c0["u"], u1=c1["u"], p0=p0, p1=p1, cell=np.array(bench.CELL), cell_l=np.array(bench.CELL_L),
                    residual_inf=np.array([c0["residuals"][-1], c1["residuals"][-1]]))
print(out, os.path.getsize(out), "bytes; residuals", c0["residuals"][-1], c1["residuals"][-1], "umax", np.abs(c0["u"]).max())
