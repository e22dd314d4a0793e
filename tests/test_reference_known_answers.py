"""The oracle's continuation stack against known answers HELD BY THE REFERENCE'S OWN TESTS (tests/golden/reference_known_answers.json:
only the asserted numbers, with the file:line of each assertion).  The reference checks them with isapprox at its default tolerance
(rtol = sqrt(eps) = 1.5e-8), and they are results of the WHOLE path -- two Newton solves, secant start, PALC corrector with
MatrixBLS, Secant / Bordered tangent, step-size control driven by the Newton iteration counts, eigenvalue counts, bisection with
sign inversions, the both-sides merge -- so a restatement that deviates anywhere (Newton tolerance 1e-10 instead of the default
1e-12, the other tangent predictor) misses them by 1e-5 ... 1e-6: measured while writing this test.  The HIP path is compared with
the same oracle functions in tests/test_gpu_parity.py / test_gpu_configs.py, so these numbers pin it by transitivity."""
import json
import os

import numpy as np
import pytest

from oracle import bifurcations as bif
from oracle import bordered, palc

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_known_answers.json")))


def _eig(Jm, nev):
    """DefaultEig (src/EigSolver.jl): dense spectrum, rightmost first."""
    w = np.linalg.eigvals(Jm)
    return w[np.argsort(-w.real, kind="stable")][:nev], None, True, 1


def _matrix_bls(J, dR, dzu, dzp, R, n, xiu, xip, shift=None, dotp=None):
    """MatrixBLS as PALC calls it (src/LinearBorderSolver.jl:16-36, 231-264): last row xi_u * dzu / N, xi_p * dzp."""
    return bordered.matrix_bls(J, dR, dzu, dzp, R, n, xiu, xip, shift=shift, apply_xiu=lambda v: v / R.shape[0])


def _both_sides(prob, x0, p0, **kw):
    """continuation(...; bothside = true) (src/Continuation.jl:687-699) + _merge (src/Results.jl:464-489): the branch computed with
    -ds, reversed, followed by the branch computed with ds; returns the special points in that order (end points not included)."""
    runs = {}
    for sgn in (1, -1):
        cp = bif.ContPar(**{**kw, "ds": sgn * kw["ds"]})
        runs[sgn] = bif.continuation(prob, x0, p0, ls=bordered.default_ls, bls=_matrix_bls, eig=_eig, cp=cp, normC=palc.norminf)
    return list(reversed(runs[-1]["specialpoint"])) + runs[1]["specialpoint"], runs


def _comodel():
    q1, q3, q4, q5, q6, k = 2.5, 10.0, 0.0675, 1.0, 0.1, 0.4

    def F(u, q2):
        x, y, s = u
        z = 1 - x - y - s
        return np.array([2 * q1 * z**2 - 2 * q5 * x**2 - q3 * x * y, q2 * z - q6 * y - q3 * x * y, q4 * (z - k * s)])

    def J(u, q2):                                    # what ForwardDiff returns for F (the reference's default Jacobian)
        x, y, s = u
        z = 1 - x - y - s
        return np.array([[-4 * q1 * z - 4 * q5 * x - q3 * y, -4 * q1 * z - q3 * x, -4 * q1 * z],
                         [-q2 - q3 * y, -q2 - q6 - q3 * x, -q2],
                         [-q4, -q4, -q4 * (1 + k)]])
    return palc.Problem(F, J)


def test_comodel_special_points_match_the_reference_test():
    g = GOLD["comodel"]
    sp, runs = _both_sides(_comodel(), np.array([0.001137, 0.891483, 0.062345]), 1.0, ds=0.002, dsmax=0.01, p_min=0.5, p_max=2.3,
                           max_steps=100, nev=3, n_inversion=6, max_bisection_steps=25, tangent="secant")
    assert [s["type"] for s in sp] == ["hopf", "bp", "bp", "hopf"] and all(s["status"] == "converged" for s in sp)
    got = np.array([s["param"] for s in sp])
    want = np.array(g["specialpoint_param_2_to_5"])
    assert np.all(np.abs(got - want) <= g["rtol"] * np.abs(want)), (got, want)
    # the fold of this model, as the reference's minimally augmented Newton solve refines it (another test file, rtol 1e-4 there):
    # the point PALC + bisection stops at agrees to 1e-9 in the parameter and 1e-4 in the state
    f = GOLD["comodel_fold"]
    fold = sp[1]
    assert abs(fold["param"] - f["p"]) <= 1e-9 and np.abs(fold["x"] - np.array(f["u"])).max() <= f["rtol"] * np.abs(f["u"]).max()
    # sensitivity of the pin: the Bordered tangent, or Newton tolerance 1e-10, moves the Hopf points by more than the tolerance
    sp_b, _ = _both_sides(_comodel(), np.array([0.001137, 0.891483, 0.062345]), 1.0, ds=0.002, dsmax=0.01, p_min=0.5, p_max=2.3,
                          max_steps=100, nev=3, n_inversion=6, max_bisection_steps=25, tangent="bordered")
    assert abs(sp_b[0]["param"] - want[0]) > 100 * g["rtol"]


def test_lorenz84_bisection_intervals_match_the_reference_test():
    g = GOLD["lorenz84"]
    al, be, G, de, ga, T = 0.25, 1.0, 0.25, 1.04, 0.987, 0.04

    def F(u, Fp):
        X, Y, Z, U = u
        return np.array([-Y**2 - Z**2 - al * X + al * Fp - ga * U**2, X * Y - be * X * Z - Y + G, be * X * Y + X * Z - Z,
                         -de * U + ga * U * X + T])

    def J(u, Fp):
        X, Y, Z, U = u
        return np.array([[-al, -2 * Y, -2 * Z, -2 * ga * U], [Y - be * Z, X - 1, -be * X, 0.0], [be * Y + Z, be * X, X - 1, 0.0],
                         [ga * U, 0.0, 0.0, -de + ga * X]])
    z0 = np.array([2.9787004394953343, -0.03868302503393752, 0.058232737694740085, -0.02105288273117459])
    sp, runs = _both_sides(palc.Problem(F, J), z0, 3.0, ds=0.001, dsmax=0.025, p_min=-1.5, p_max=3.0, max_steps=252, nev=4,
                           n_inversion=6, max_bisection_steps=25, tangent="bordered")
    assert [s["type"] for s in sp] == ["hopf", "hopf", "hopf", "bp"] and all(s["status"] == "converged" for s in sp)
    got = np.array([s["interval"] for s in sp])
    want = np.array(g["specialpoint_interval_2_to_5"])
    assert np.all(np.abs(got - want) <= g["rtol"] * np.abs(want)), (got, want)
    assert len(runs[1]["param"]) == 2 and runs[1]["param"][-1] == 3.0            # the other side starts on p_max: one clamped step


def test_bifurcation_points_of_the_trivial_branch_with_their_dimensions():
    g = GOLD["detection"]
    diag = np.array([1.0 / i for i in range(1, 6) for _ in range(i)] + [1 / 6.0, 1 / 6.5, 1 / 6.75, 1 / 6.875])
    prob = palc.Problem(lambda x, lam: -x + (diag * x) * lam - x**3, lambda x, lam: np.diag(diag * lam - 3 * x**2 - 1))
    cp = bif.ContPar(ds=0.1, dsmax=0.1, p_min=-1.0, p_max=10.3, max_steps=150, nev=3, n_inversion=4, tol_bisection_eigenvalue=1e-7,
                     tangent="secant")
    r = bif.continuation(prob, np.zeros(diag.size), 0.0, ls=bordered.default_ls, bls=_matrix_bls, eig=_eig, cp=cp, normC=palc.norm2)
    sp = r["specialpoint"]
    got = np.array([s["param"] for s in sp])
    want = np.array(g["bifurcation_points"])
    # test_bif_detection.jl:92-98: every marked state lies AFTER its bifurcation point, within 3e-3, inside its interval, and
    # |delta n_unstable| is the dimension of the kernel
    assert got.shape == want.shape and np.all(got > want) and np.abs(got - want).max() < g["max_distance"]
    assert [abs(s["n_unstable"][0] - s["n_unstable"][1]) for s in sp] == g["dimension"]
    assert all(s["interval"][0] <= s["param"] <= s["interval"][1] for s in sp)
