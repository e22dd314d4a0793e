"""ORACLE (test infrastructure) -- the communication-light distributed z-solve of the spectral preconditioner, restated in
NumPy to validate its algebra and stability before it was written in HIP (bifurcationkit.jl_amd/csrc/dct_slab.hip).

No reference counterpart: the reference is single-process; what it defines is the operator whose exact
inverse is applied, Pl = L1 + shift I with L1 = (I + Lap)^2 (examples/SH3d.jl:85-88, examples/SH2d-fronts.jl:121).

After the x / y DCTs every (kx, ky) line is an independent system in z,
    M w = f,     M = (c I + D)^2 + s I,     c = 1 + lam_x + lam_y,
D = Neumann-ghost second difference on ALL nz planes.  With the planes split into R slabs of nl = nz / R planes, let B be the
block-diagonal matrix of the same operator with a Neumann-ghost closure at every slab face, B_r = (c I + D_nl)^2 + s I -- each
block is diagonalised by the LOCAL DCT-II of length nl, so B^-1 is the existing fused z pass run on the slab.  Across the
face between planes p | q = p + 1,  D = D_loc - a w w',  w = e_p - e_q,  a = 1/h^2, hence with T = c I + D
    M - B = T^2 - T_loc^2 = -a (u w' + w u') + 2 a^2 w w',      u = T_loc w = a e_{p-1} + (c - a)(e_p - e_q) - a e_{q+1},
a rank-2 term per face: M = B + U G U' with U = [u_1 w_1 u_2 w_2 ...] and G = blockdiag([[0, -a], [-a, 2 a^2]]).  Woodbury:
    M^-1 f = B^-1 (f - U nu),      (G^-1 + U' B^-1 U) nu = U' B^-1 f,
a block-tridiagonal system with 2 x 2 blocks and R - 1 block rows per line whose entries only need B_r^-1 between the four
planes next to the slab faces -- six nl-term sums of the DCT basis.  Communication per line: 2 numbers per adjacent face each
way, instead of the two full transposes of the array.
"""
from __future__ import annotations

import numpy as np
import scipy.fft as sfft


def lam_neumann(n, a):
    return -(4.0 * a) * np.sin(np.pi * np.arange(n) / (2.0 * n)) ** 2


def face_green(c, s, nl, a):
    """(B_r^-1)[z, z'] for z, z' in the planes (0, 1, nl-2, nl-1), from the DCT-II eigenbasis of one slab:
    returns (A, C) with A[i, j] = sum_m phi_m(i) phi_m(j) sym_m and C[i, j] the same with (-1)^m (i, j in {0, 1})."""
    m = np.arange(nl)
    sym = 1.0 / ((c + lam_neumann(nl, a)) ** 2 + s)
    sk = np.where(m == 0, np.sqrt(1.0 / nl), np.sqrt(2.0 / nl))
    phi = np.stack([sk * np.cos(np.pi * (2 * z + 1) * m / (2.0 * nl)) for z in (0, 1)])       # [2][nl]
    A = (phi * sym) @ phi.T
    C = (phi * sym * (-1.0) ** m) @ phi.T
    return A, C


def reduced_blocks(c, s, nl, a):
    """Diagonal block D0 = G^-1 + K_ii and coupling block E = K_{i,i+1} of the capacitance system (uniform slabs)."""
    A, C = face_green(c, s, nl, a)
    # G_tt in plane order (nl-2, nl-1) = A with both indices reversed; G_bt[z, nl-1-z'] = C[z, z']
    Gtt = A[::-1, ::-1]
    Gbb = A
    Ptop = np.array([[a, 0.0], [c - a, 1.0]])            # rows: planes (nl-2, nl-1); columns: (u, w)
    Pbot = np.array([[-(c - a), -1.0], [-a, 0.0]])       # rows: planes (0, 1)
    Ginv = np.array([[-2.0, -1.0 / a], [-1.0 / a, 0.0]])
    D0 = Ginv + Ptop.T @ Gtt @ Ptop + Pbot.T @ Gbb @ Pbot
    Gbt = C[:, ::-1]                                     # rows planes (0, 1), columns planes (nl-2, nl-1)
    E = Pbot.T @ Gbt @ Ptop                              # face i (bottom part in slab i+1) x face i+1 (top part in slab i+1)
    return D0, E, Ptop, Pbot


def local_solve(f, c, s, nl, a):
    """B^-1 f on every slab: local DCT-II, symbol, inverse (f: [R, nl])."""
    sym = 1.0 / ((c + lam_neumann(nl, a)) ** 2 + s)
    return sfft.idct(sfft.dct(f, type=2, norm="ortho", axis=1) * sym, type=2, norm="ortho", axis=1)


def slab_zsolve(f, c, s, R, a):
    """M^-1 f for one z line split into R equal slabs, by the steps of the distributed HIP path."""
    n = f.shape[0]
    nl = n // R
    assert nl * R == n and nl >= 4
    F = f.reshape(R, nl).copy()
    Y = local_solve(F, c, s, nl, a)
    if R == 1:
        return Y.reshape(-1)
    D0, E, Ptop, Pbot = reduced_blocks(c, s, nl, a)
    # face i between slab i (top part) and slab i+1 (bottom part): g_i = U_i' y
    g = np.stack([Ptop.T @ Y[i, nl - 2:] + Pbot.T @ Y[i + 1, :2] for i in range(R - 1)])
    # block Thomas on  E' nu_{i-1} + D0 nu_i + E nu_{i+1} = g_i
    nf = R - 1
    Dk, gk = [None] * nf, [None] * nf
    Dk[0], gk[0] = D0, g[0]
    for i in range(1, nf):
        W = E.T @ np.linalg.inv(Dk[i - 1])
        Dk[i] = D0 - W @ E
        gk[i] = g[i] - W @ gk[i - 1]
    nu = [None] * nf
    nu[nf - 1] = np.linalg.solve(Dk[nf - 1], gk[nf - 1])
    for i in range(nf - 2, -1, -1):
        nu[i] = np.linalg.solve(Dk[i], gk[i] - E @ nu[i + 1])
    for i in range(nf):                                   # f' = f - U nu on the four planes next to each face
        F[i, nl - 2:] -= Ptop @ nu[i]
        F[i + 1, :2] -= Pbot @ nu[i]
    return local_solve(F, c, s, nl, a).reshape(-1)


def exact_zsolve(f, c, s, a):
    n = f.shape[0]
    return sfft.idct(sfft.dct(f, type=2, norm="ortho") / ((c + lam_neumann(n, a)) ** 2 + s), type=2, norm="ortho")


def slab_zsolve_split(f, c, s, R, a):
    """The same solve the way round 5's HIP path takes it (dct.hip: dct_apply_slab with dct_slab_split; dct_fast.hip: SLAB 1 / 2):
    ONE forward local transform, the face values of y = B^-1 f from sums over the spectrum, the capacitance system, and the correction
    applied in the z-spectral domain inside ONE inverse transform -- two single transforms instead of two round trips."""
    n = f.shape[0]
    nl = n // R
    assert nl * R == n and nl >= 4
    m = np.arange(nl)
    sym = 1.0 / ((c + lam_neumann(nl, a)) ** 2 + s)
    sk = np.where(m == 0, np.sqrt(1.0 / nl), np.sqrt(2.0 / nl))
    phi = np.stack([sk * np.cos(np.pi * (2 * z + 1) * m / (2.0 * nl)) for z in (0, 1)])       # [2][nl]: planes 0, 1
    sg = (-1.0) ** m
    Yh = sfft.dct(f.reshape(R, nl), type=2, norm="ortho", axis=1) * sym                      # forward half: y^ = sym .* f^
    if R == 1:
        return sfft.idct(Yh, type=2, norm="ortho", axis=1).reshape(-1)
    # y at planes (0, 1, nl-2, nl-1) of every slab: phi_k(nl-1-z) = (-1)^k phi_k(z)
    yf = np.stack([Yh @ phi[0], Yh @ phi[1], (Yh * sg) @ phi[1], (Yh * sg) @ phi[0]], axis=1)          # [R][4]
    D0, E, Ptop, Pbot = reduced_blocks(c, s, nl, a)
    g = np.stack([Ptop.T @ yf[i, 2:] + Pbot.T @ yf[i + 1, :2] for i in range(R - 1)])
    nf = R - 1
    Dk, gk = [None] * nf, [None] * nf
    Dk[0], gk[0] = D0, g[0]
    for i in range(1, nf):
        W = E.T @ np.linalg.inv(Dk[i - 1])
        Dk[i] = D0 - W @ E
        gk[i] = g[i] - W @ gk[i - 1]
    nu = [None] * nf
    nu[nf - 1] = np.linalg.solve(Dk[nf - 1], gk[nf - 1])
    for i in range(nf - 2, -1, -1):
        nu[i] = np.linalg.solve(Dk[i], gk[i] - E @ nu[i + 1])
    delta = np.zeros((R, 4))                              # -U nu at the planes (0, 1, nl-2, nl-1) of every slab
    for i in range(nf):
        delta[i, 2:] -= Ptop @ nu[i]
        delta[i + 1, :2] -= Pbot @ nu[i]
    # inverse half: a^_k = y^_k + sym_k (phi_k(0) d0 + phi_k(1) d1 + (-1)^k (phi_k(1) d2 + phi_k(0) d3))
    corr = delta[:, 0:1] * phi[0] + delta[:, 1:2] * phi[1] + sg * (delta[:, 2:3] * phi[1] + delta[:, 3:4] * phi[0])
    return sfft.idct(Yh + sym * corr, type=2, norm="ortho", axis=1).reshape(-1)
