"""CPU test of the product's HOST logic (bifurcationkit.jl_amd/continuation.py + the generic BorderingBLS of
hip.py): driven with a NumPy-backed vector type and oracle plugins it must reproduce the oracle's own
continuation run.  This is the plugin contract of test/continuation/test-cont-non-vector.jl:55-99,133-176:
a state type that is not an array, an opaque Jacobian, custom solvers returning (out, true, 1).
The NumPy vector type lives HERE (test infrastructure); the product has no CPU backend."""
import numpy as np
import pytest
import scipy.sparse.linalg as spla

from bk_amd import continuation as C
from bk_amd import hip
from oracle import bordered, krylov, operators, palc


class NumpyVec:
    """Minimal VectorInterface carrier for the host-logic tests (cf. VI.MinimalMVec)."""

    def __init__(self, a):
        self.a = np.asarray(a, dtype=float)

    def copy(self): return NumpyVec(self.a.copy())
    def zerovector(self): return NumpyVec(np.zeros_like(self.a))
    def similar(self): return NumpyVec(np.empty_like(self.a))
    def copyto_(self, s): self.a[:] = s.a; return self
    def scale_(self, c): self.a *= c; return self
    def add_(self, x, a=1.0, b=1.0): self.a = b * self.a + a * x.a; return self
    def inner(self, y): return float(self.a @ y.a)
    def norm(self): return float(np.linalg.norm(self.a))
    def norminf(self): return float(np.abs(self.a).max())
    def __len__(self): return self.a.size


class OpaqueJacobian:
    def __init__(self, M): self.M = M
    def __call__(self, dx): return NumpyVec(self.M @ dx.a)


class Prob:
    def __init__(self, sh, nu):
        self.sh, self.nu = sh, nu
        self.delta = palc.EPS_FD
    def residual(self, x, p): return NumpyVec(self.sh.F(x.a, p, self.nu))
    def jacobian(self, x, p): return OpaqueJacobian(self.sh.J(x.a, p, self.nu))


def direct_ls(J, rhs, a0=0.0, a1=1.0):
    x, ok, it = bordered.default_ls(J.M, rhs.a, a0, a1)
    return NumpyVec(x), ok, it


def test_generic_bordering_matches_oracle():
    rng = np.random.default_rng(0)
    n = 40
    M = np.eye(n) + 0.1 * rng.random((n, n))
    dR, dzu, R = rng.random(n), rng.random(n), rng.random(n)
    bls = hip.BorderingBLS(direct_ls, check_precision=True, k=2)
    dX, dl, ok, it = bls(OpaqueJacobian(M), NumpyVec(dR), NumpyVec(dzu), 0.3, NumpyVec(R), 0.7, 0.4, 0.6, shift=0.2)
    rX, rl, _, _ = bordered.bordering_bls(bordered.default_ls, M, dR, dzu, 0.3, R, 0.7, 0.4, 0.6, shift=0.2, k=2)
    assert ok and np.allclose(dX.a, rX, rtol=1e-12) and np.isclose(dl, rl, rtol=1e-12)


def test_continuation_driver_matches_oracle():
    sh = operators.SwiftHohenberg((9, 8), (3.0, 2.5))
    x0 = 0.8 * sh.guess()
    # oracle run
    oprob = palc.Problem(F=lambda x, p: sh.F(x, p, 1.3), J=lambda x, p: sh.J(x, p, 1.3))
    s0 = palc.newton(oprob, x0, -0.1, bordered.default_ls, tol=1e-10, max_iterations=30, normN=palc.norminf)
    assert s0["converged"]
    obls = lambda *a, **k: bordered.bordering_bls(bordered.default_ls, *a, check_precision=False, **k)
    eig_o = lambda J, nev: krylov.default_eig(J, nev)
    kw = dict(ds=0.002, dsmin=1e-4, dsmax=0.01, p_min=-0.3, p_max=0.3, max_steps=5, tol=1e-10, max_iterations=15)
    bo = palc.continuation(oprob, s0["u"], -0.1, ls=bordered.default_ls, bls=obls, tangent="bordered",
                           normC=palc.norminf, eig=eig_o, nev=4, **kw)
    # product host logic with NumPy vectors and the same plugins
    prob = Prob(sh, 1.3)
    eig_p = lambda J, nev: krylov.default_eig(J.M, nev)
    nopt = C.NewtonPar(tol=1e-10, max_iterations=15, linsolver=direct_ls, eigsolver=eig_p)
    cp = C.ContinuationPar(ds=0.002, dsmin=1e-4, dsmax=0.01, p_min=-0.3, p_max=0.3, max_steps=5, nev=4,
                           newton_options=nopt)
    alg = C.PALC(tangent="bordered", theta=0.5, bls=hip.BorderingBLS(None, check_precision=False))
    bp = C.continuation(prob, NumpyVec(s0["u"]), -0.1, alg, cp, normC=C.norminf)
    assert len(bp.param) == len(bo.param) == 6
    assert np.allclose(bp.param, bo.param, rtol=0, atol=1e-12)
    assert bp.itnewton == bo.itnewton
    assert bp.n_unstable == bo.n_unstable
    assert np.allclose(bp.ds, bo.ds)
    for a, b in zip(bp.residuals, bo.residuals):
        assert np.allclose(a, b, rtol=1e-6, atol=1e-13)


def test_solution_sampling_and_checkpoints(tmp_path):
    """save! (src/Continuation.jl:280-292): br.sol keeps (x, p, step) every save_sol_every_step steps and at the last
    point; save_to_file (ext/JLD2Ext/save.jl:8-30): one checkpoint per accepted step plus the branch record, and a run
    restarted from a checkpoint continues on the same branch."""
    sh = operators.SwiftHohenberg((9, 8), (3.0, 2.5))
    prob = Prob(sh, 1.3)
    oprob = palc.Problem(F=lambda x, p: sh.F(x, p, 1.3), J=lambda x, p: sh.J(x, p, 1.3))
    s0 = palc.newton(oprob, 0.8 * sh.guess(), -0.1, bordered.default_ls, tol=1e-10, max_iterations=30, normN=palc.norminf)
    eig_p = lambda J, nev: krylov.default_eig(J.M, nev)
    nopt = C.NewtonPar(tol=1e-10, max_iterations=15, linsolver=direct_ls, eigsolver=eig_p)
    cp = C.ContinuationPar(ds=0.002, dsmin=1e-4, dsmax=0.01, p_min=-0.3, p_max=0.3, max_steps=5, nev=4,
                           newton_options=nopt, save_sol_every_step=2, save_to_file=True)
    alg = C.PALC(tangent="secant", theta=0.5, bls=hip.BorderingBLS(None, check_precision=False))
    fn = str(tmp_path / "branch")
    br = C.continuation(prob, NumpyVec(s0["u"]), -0.1, alg, cp, normC=C.norminf, filename=fn)
    assert [s["step"] for s in br.sol] == [2, 4, 5]                         # mod_counter(step, 2) or the last point
    assert all(np.isclose(s["p"], br.param[s["step"]]) for s in br.sol)
    assert C.mod_counter(0, 1) is False and C.mod_counter(3, 0) is False and C.mod_counter(3, 1) and not C.mod_counter(3, 2)
    for i in range(1, 6):
        x, p = C.load_solution(fn, i, "fw")
        assert np.isclose(p, br.param[i])
        assert np.abs(sh.F(x, p, 1.3)).max() < 1e-9                         # every checkpoint is a point of the branch
    x4, p4 = C.load_solution(fn, 4)
    assert np.array_equal(x4, br.sol[1]["x"].a)
    rec = C.load_branch(fn)
    assert np.allclose(rec["param"], br.param) and rec["n_unstable"] == br.n_unstable and len(rec["eig"]) == len(br.param)
    # restart from the checkpoint of step 3: the first new point lies on the same solution curve, beyond the checkpoint
    x3, p3 = C.load_solution(fn, 3)
    cp2 = C.ContinuationPar(ds=0.002, dsmin=1e-4, dsmax=0.01, p_min=-0.3, p_max=0.3, max_steps=1, nev=4, newton_options=nopt)
    br2 = C.continuation(prob, NumpyVec(x3), p3, alg, cp2, normC=C.norminf)
    assert np.isclose(br2.param[0], p3) and br2.param[1] > p3
    cpb = C.ContinuationPar(ds=-0.002, dsmin=1e-4, dsmax=0.01, p_min=-0.3, p_max=0.3, max_steps=1, nev=4, newton_options=nopt,
                            save_to_file=True)
    C.continuation(prob, NumpyVec(x3), p3, alg, cpb, normC=C.norminf, filename=fn)
    assert C.load_solution(fn, 1, "bw")[1] < p3                             # backward branch: the `bw` group


class FoldProb:
    """F(x, p) = p - x_i^2 + 0.1 (x_i - mean x): a turning point at p ~ 0 (the cubic-with-a-fold pattern of
    test/newton/test_newton.jl:23-52 / test/continuation/simple_continuation.jl)."""
    delta = palc.EPS_FD

    def F(self, x, p):
        return p - x**2 + 0.1 * (x - x.mean())

    def Jm(self, x, p):
        n = x.size
        return np.diag(-2 * x + 0.1) - 0.1 / n * np.ones((n, n))

    def residual(self, x, p): return NumpyVec(self.F(x.a, p))
    def jacobian(self, x, p): return OpaqueJacobian(self.Jm(x.a, p))


def test_fold_detection_by_parameter_monotony_and_parameter_checks():
    """locate_fold! (src/Bifurcations.jl:32-69): with detect_bifurcation < 2 a fold is recorded where the parameter stops
    being monotone along the branch, at the step and with the interval the reference stores; with eigenvalue-based
    detection (>= 2) the fold test is off (Continuation.jl:524).  ContinuationPar refuses inconsistent settings
    (src/ContParameters.jl:89-99)."""
    prob = FoldProb()
    nopt = C.NewtonPar(tol=1e-11, max_iterations=15, linsolver=direct_ls, eigsolver=lambda J, nev: krylov.default_eig(J.M, nev))
    alg = C.PALC(tangent="secant", theta=0.5, bls=hip.BorderingBLS(None, check_precision=False))
    x0 = NumpyVec(np.full(5, 0.5))
    for level in (0, 1):
        cp = C.ContinuationPar(ds=-0.02, dsmin=1e-4, dsmax=0.03, p_min=-1.0, p_max=1.0, max_steps=40, nev=3,
                               detect_bifurcation=level, newton_options=nopt)
        br = C.continuation(prob, x0, 0.25, alg, cp, normC=C.norminf)
        folds = [sp for sp in br.specialpoint if sp.get("type") == "fold"]
        assert len(folds) == 1, br.specialpoint
        k = folds[0]["step"]
        # the reference runs locate_fold! BEFORE save! (Continuation.jl:524 vs :579): the three points are the last three
        # already recorded (steps k-2, k-1, k), `param` is the state computed after them, idx / interval the middle point
        assert C.detect_fold(br.param[k - 2], br.param[k - 1], br.param[k]) and folds[0]["param"] == br.param[k + 1]
        assert not C.detect_fold(br.param[k - 3], br.param[k - 2], br.param[k - 1])       # not flagged one step early
        assert folds[0]["idx"] == k - 1 and folds[0]["interval"] == (br.param[k - 1], br.param[k - 1])
        assert abs(br.param[k - 1]) < 0.01
        assert all(sp.get("type") == "fold" for sp in br.specialpoint)    # level 1 computes eigenvalues but flags nothing
    cp2 = C.ContinuationPar(ds=-0.02, dsmin=1e-4, dsmax=0.03, p_min=-1.0, p_max=1.0, max_steps=40, nev=3,
                            detect_bifurcation=2, newton_options=nopt)
    br2 = C.continuation(prob, x0, 0.25, alg, cp2, normC=C.norminf)
    assert not any(sp.get("type") == "fold" for sp in br2.specialpoint) and len(br2.specialpoint) >= 1
    for bad in (dict(ds=0.2, dsmax=0.1), dict(ds=1e-5, dsmin=1e-4), dict(p_min=1.0, p_max=0.0), dict(n_inversion=3),
                dict(detect_bifurcation=4), dict(tol_stability=-1.0)):
        with pytest.raises(ValueError):
            C.ContinuationPar(**bad)


def test_step_size_control_and_stability():
    cp = C.ContinuationPar(dsmin=1e-3, dsmax=0.1, a=0.5, newton_options=C.NewtonPar(max_iterations=10))
    ds, stop = C.step_size_control(0.01, True, 2, cp)
    assert np.isclose(ds, 0.01 * (1 + 0.5 * 0.8 ** 2)) and not stop
    ds, stop = C.step_size_control(0.01, False, 10, cp)
    assert np.isclose(ds, 0.005) and not stop
    ds, stop = C.step_size_control(-1e-3, False, 10, cp)
    assert stop
    ds, stop = C.step_size_control(-0.09, True, 0, cp)
    assert np.isclose(ds, -0.1)
    assert C.is_stable(np.array([0.5 + 1j, 0.5 - 1j, 1e-11, -0.2]), 1e-10) == (2, 2)


def test_bench_tiling_helpers_agree():
    """bench.py: the device tiling (tile_cell, torch) and the CPU baseline's NumPy tiling produce the same even-reflection
    layout, on the whole grid and on a z-slab; thread selection honours the affinity mask."""
    import os
    import sys
    import numpy as np
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    cx, cy, cz = bench.CELL
    rng = np.random.default_rng(0)
    cell = rng.standard_normal(cx * cy * cz)
    tiles = (2, 3, 2)
    idx = [np.concatenate([np.arange(nc) if c % 2 == 0 else np.arange(nc)[::-1] for c in range(T)])
           for nc, T in zip(bench.CELL, tiles)]
    ref = np.ascontiguousarray(cell.reshape(cz, cy, cx)[np.ix_(idx[2], idx[1], idx[0])])
    full = bench.tile_cell(torch.from_numpy(cell), tiles, (0, cz * tiles[2]), "cpu").numpy()
    assert np.array_equal(full, ref.reshape(-1))
    lo, hi = 7, 41                                                       # a slab that crosses the reflection plane
    part = bench.tile_cell(torch.from_numpy(cell), tiles, (lo, hi), "cpu").numpy()
    assert np.array_equal(part, ref[lo:hi].reshape(-1))
    # the tiled field continues the cell evenly: plane cz-1 == plane cz (reflection), so a Neumann-ghost stencil sees no seam
    assert np.array_equal(ref[cz - 1], ref[cz])
    assert bench.tiles_for(512) == (8, 16, 16)
    with pytest.raises(SystemExit):
        bench.tiles_for(100)
    n = bench.available_cpus()
    assert 1 <= n <= (os.cpu_count() or 1) and max(bench.thread_counts()) == n
