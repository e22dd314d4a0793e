import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from bk_amd import hip
from oracle import operators
ctx = hip.Context(0)
dims, ls_ = (16, 9), (np.pi, np.pi / 2)
c = operators.CGL2d(dims, ls_)
r = 0.7828427124746191
prob = hip.CGL2d(ctx, dims, ls_, r=r)
n2 = 2 * c.n
p = c.default_params(); p['r'] = r
Jm = c.J(np.zeros(n2), **p)
dense = np.linalg.eigvals(Jm.toarray())
print("dense rightmost", np.sort(dense.real)[-4:])
P = hip.LaplacePreconditioner(prob, 1.0)
J = prob.jacobian(prob.vec(np.zeros(n2)), r)
rng = np.random.default_rng(0)
x = rng.standard_normal(n2)
import scipy.sparse as sp, scipy.sparse.linalg as spla
ref = spla.spsolve((Jm - sp.identity(n2)).tocsc(), x)
for name, ls in [("IS+P", hip.GMRESIterativeSolvers(reltol=1e-11, restart=60, maxiter=600, Pl=P)),
                 ("IS", hip.GMRESIterativeSolvers(reltol=1e-11, restart=63, maxiter=3000)),
                 ("KK", hip.GMRESKrylovKit(dim=63, rtol=1e-11, atol=1e-14, maxiter=100))]:
    y, ok, it = ls(J, prob.vec(x), -1.0, 1.0)
    print(name, "solve ok", ok, it, "err", np.abs(y.numpy() - ref).max() / np.abs(ref).max())
    for nev in (9, 10):
        for seed in (1234, 1):
            eig = hip.ShiftInvert(1.0, ls, tol=1e-9, maxiter=40, hermitian=False, save_vectors=False, seed=seed)
            vals, _, cv, nops = eig(J, nev)
            print("   ", name, nev, seed, cv, nops, np.round(vals, 4))
