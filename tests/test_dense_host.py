"""CPU test of the host-side dense routines (bifurcationkit.jl_amd/csrc/dense.h) used by the Krylov-Schur
eigensolver: Jacobi (symmetric) and complex shifted-QR (general) eigen-decompositions vs NumPy, plus the
reference's literal 5x5 golden eigenvalues (test/linear_solvers/test_linear.jl:595-614)."""
import json
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def exe(tmp_path_factory):
    out = tmp_path_factory.mktemp("dense") / "dense_check"
    subprocess.run(["g++", "-O2", "-std=c++17", os.path.join(ROOT, "tests", "cpp", "dense_check.cpp"), "-o", str(out)],
                   check=True)
    return str(out)


def _run(exe, mode, A):
    n = A.shape[0]
    inp = f"{mode} {n}\n" + "\n".join(" ".join(repr(float(x)) for x in row) for row in A)
    out = subprocess.run([exe], input=inp, capture_output=True, text=True, check=True).stdout.split("\n")
    st = int(out[0])
    if mode == 0:
        w = np.array([float(x) for x in out[1:1 + n]])
        Z = np.array([[float(x) for x in out[1 + n + i].split()] for i in range(n)])
        return st, w, Z
    w = np.array([complex(*map(float, out[1 + i].split())) for i in range(n)])
    Y = np.array([[float(x) for x in out[1 + n + i].split()] for i in range(n)])
    return st, w, Y[:, 0::2] + 1j * Y[:, 1::2]


@pytest.mark.parametrize("n", [1, 2, 5, 30, 63])
def test_jacobi_eigh(exe, n):
    rng = np.random.default_rng(n)
    A = rng.standard_normal((n, n))
    S = A + A.T
    st, w, Z = _run(exe, 0, S)
    assert st >= 0
    scale = max(1.0, np.abs(S).max()) * n
    assert np.abs(w - np.linalg.eigvalsh(S)).max() < 1e-14 * scale
    assert np.abs(S @ Z - Z * w).max() < 1e-14 * scale
    assert np.abs(Z.T @ Z - np.eye(n)).max() < 1e-13


@pytest.mark.parametrize("n", [1, 2, 5, 30, 63])
def test_eig_general(exe, n):
    rng = np.random.default_rng(100 + n)
    A = rng.standard_normal((n, n))
    st, w, Y = _run(exe, 1, A)
    assert st == 0
    ev = np.linalg.eigvals(A)
    d = np.abs(w[:, None] - ev[None, :])
    assert max(d.min(axis=1).max(), d.min(axis=0).max()) < 1e-12 * n
    assert np.abs(A @ Y - Y * w).max() < 1e-12 * n
    assert np.allclose(np.linalg.norm(Y, axis=0), 1.0)


def test_eig_general_golden_5x5(exe):
    g = json.load(open(os.path.join(ROOT, "tests", "golden", "eig5x5.json")))
    J0 = np.array(g["J0"])
    vals = np.array([complex(*v) for v in g["vals"]])
    st, w, Y = _run(exe, 1, J0)
    assert st == 0
    d = np.abs(w[:, None] - vals[None, :])
    assert d.min(axis=0).max() < 1.5e-8 * np.abs(vals).max() and d.min(axis=1).max() < 1.5e-8 * np.abs(vals).max()
