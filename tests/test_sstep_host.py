"""CPU test of the host algebra of the block Arnoldi step (bifurcationkit.jl_amd/csrc/sstep.h): the library's header, with the
two streaming passes replaced by plain loops (tests/cpp/sstep_check.cpp), against a textbook Arnoldi process."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def exe(tmp_path_factory):
    out = tmp_path_factory.mktemp("sstep") / "sstep_check"
    subprocess.run(["g++", "-O2", "-std=c++17", os.path.join(ROOT, "tests", "cpp", "sstep_check.cpp"), "-o", str(out)], check=True)
    return str(out)


def _arnoldi(A, b, m):
    n = b.size
    V = np.zeros((m + 1, n))
    H = np.zeros((m + 1, m))
    V[0] = b / np.linalg.norm(b)
    for j in range(m):
        w = A @ V[j]
        for _ in range(2):
            h = V[:j + 1] @ w
            w = w - V[:j + 1].T @ h
            H[:j + 1, j] += h
        H[j + 1, j] = np.linalg.norm(w)
        V[j + 1] = w / H[j + 1, j]
    return V, H


def _run(exe, A, b, m, s, shifts=()):
    n = b.size
    inp = f"{n} {m} {s}\n" + "\n".join(" ".join(repr(float(x)) for x in row) for row in A) + "\n" + " ".join(repr(float(x)) for x in b)
    inp += f"\n{len(shifts)} " + " ".join(repr(float(x)) for x in shifts)
    out = subprocess.run([exe], input=inp, capture_output=True, text=True, check=True).stdout.strip().split("\n")
    assert out[0].split()[0] == "ok", out[0]
    # the solution update folded through the last block's unformed vectors (sstep.h: fold_solution_coefficients; the deferred update
    # pass of solver.hip) equals the update through the explicit vectors
    assert float(out[0].split()[1]) < 1e-13, out[0]
    H = np.array([[float(x) for x in out[1 + a].split()] for a in range(m + 1)])
    Q = np.array([[float(x) for x in out[2 + m + a].split()] for a in range(m + 1)])
    return H, Q


@pytest.mark.parametrize("s", [1, 2, 3, 4])
@pytest.mark.parametrize("m", [5, 12, 30])
def test_block_arnoldi_reproduces_the_arnoldi_process(exe, s, m):
    """A = -I + a contraction with a spread spectrum (the shape of Pl^-1 J): same Hessenberg matrix, same basis, the Arnoldi
    relation to rounding, exact zeros below the subdiagonal."""
    rng = np.random.default_rng(10 * m + s)
    n = 80
    M = rng.standard_normal((n, n))
    A = -np.eye(n) + 0.6 * (M + M.T) / np.linalg.norm(M + M.T, 2) + 0.05 * M / np.linalg.norm(M, 2)
    b = rng.standard_normal(n)
    V, H0 = _arnoldi(A, b, m)
    H, Q = _run(exe, A, b, m, s)
    assert np.abs(np.tril(H, -2)).max() == 0.0
    # orthonormality INSIDE a block is eps * cond(projected monomial block)^2 (1e2 .. 3e3 for s = 4); across blocks the measured
    # Gram matrix keeps it from accumulating
    assert np.abs(Q @ Q.T - np.eye(m + 1)).max() < (1e-7 if s == 4 else 1e-9)
    assert np.abs(A @ Q[:m].T - Q.T @ H).max() < 1e-11
    # the basis is unique up to rounding amplified by the conditioning of the Krylov matrix: compare what GMRES uses
    beta = np.linalg.norm(b)
    for k in (m // 2, m):
        e = np.zeros(k + 1)
        e[0] = beta
        r0 = np.linalg.lstsq(H0[:k + 1, :k], e, rcond=None)
        r1 = np.linalg.lstsq(H[:k + 1, :k], e, rcond=None)
        x0, x1 = V[:k].T @ r0[0], Q[:k].T @ r1[0]
        assert np.linalg.norm(x0 - x1) <= 1e-8 * np.linalg.norm(x0)


def test_newton_shifts_give_the_same_hessenberg_with_a_better_conditioned_block(exe):
    """p_{i+1} = (A - theta_i) p_i with shifts inside the spectrum: the same Arnoldi relation and the same GMRES iterates, and a
    basis that is orthonormal to a tighter bound than the monomial block's."""
    rng = np.random.default_rng(7)
    n, m, s = 80, 12, 4
    M = rng.standard_normal((n, n))
    A = -np.eye(n) + 0.6 * (M + M.T) / np.linalg.norm(M + M.T, 2)
    b = rng.standard_normal(n)
    V, H0 = _arnoldi(A, b, m)
    ev = np.sort(np.linalg.eigvalsh(H0[:m, :m] * 0.5 + H0[:m, :m].T * 0.5))
    shifts = [ev[0], ev[-1], ev[m // 2], ev[m // 4]]
    Hm, Qm = _run(exe, A, b, m, s)
    Hn, Qn = _run(exe, A, b, m, s, shifts)
    assert np.abs(A @ Qn[:m].T - Qn.T @ Hn).max() < 1e-11 and np.abs(np.tril(Hn, -2)).max() == 0.0
    dm, dn = np.abs(Qm @ Qm.T - np.eye(m + 1)).max(), np.abs(Qn @ Qn.T - np.eye(m + 1)).max()
    assert dn < 1e-10 and dn <= dm
    e = np.zeros(m + 1)
    e[0] = np.linalg.norm(b)
    x0 = V[:m].T @ np.linalg.lstsq(H0[:m + 1, :m], e, rcond=None)[0]
    x1 = Qn[:m].T @ np.linalg.lstsq(Hn[:m + 1, :m], e, rcond=None)[0]
    assert np.linalg.norm(x0 - x1) <= 1e-9 * np.linalg.norm(x0)


def test_rank_deficient_block_is_truncated(exe):
    """An operator whose Krylov space closes inside the block (A^2 b in span{b, A b}): the block must be cut after its first
    vector -- the only one that adds a direction -- not factored to the end."""
    n = 20
    A = np.diag(np.r_[np.full(10, 2.0), np.full(10, -1.0)])
    b = np.ones(n)
    inp = f"{n} 4 4\n" + "\n".join(" ".join(repr(float(x)) for x in row) for row in A) + "\n" + " ".join(repr(float(x)) for x in b) + "\n0"
    out = subprocess.run([exe], input=inp, capture_output=True, text=True, check=True).stdout
    assert out.split("\n")[0] == "block at 0 truncated to 1"
