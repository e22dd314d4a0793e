"""ORACLE (test infrastructure) -- PDE operators exactly as the reference examples
assemble them: sparse second differences, Kronecker sums, ``L1 = (I + Lap)^2`` as a
*matrix product*, and the residual / JVP expressions written on top of ``L1 @ u``.

Reference (paths relative to /root/reference):
  examples/SH3d.jl:16-53,69-86      Laplacian3D, F_sh, dF_sh, J_sh, grid, sol0
  examples/SH2d-fronts.jl:8-55      Laplacian2D, F_sh, dF_sh, hexagon guess
  examples/SHpde_snaking.jl:8-31    1-D cubic-quintic SH, Dirichlet
  examples/cGL2d.jl:6-54,57-91,262-318   Laplacian2D (Dirichlet), NL, Fcgl!, dFcgl, Jcgl
"""
from __future__ import annotations

import numpy as np
import scipy.sparse as sp

NEUMANN = "neumann"      # ghost-cell Neumann: corner diagonal -1/h^2  (SH3d.jl:25-32)
DIRICHLET = "dirichlet"  # plain truncation: corner diagonal -2/h^2    (cGL2d.jl:12-16)


def second_difference(N: int, l: float, bc: str) -> sp.csr_matrix:
    """tridiag(1,-2,1)/h^2 with h = 2l/N (SH3d.jl:18-23); Neumann-ghost variant overwrites
    the two corner diagonal entries with -1/h^2 (SH3d.jl:25-32); Dirichlet keeps -2/h^2
    (cGL2d.jl:12-16, SHpde_snaking.jl:16)."""
    h = 2.0 * l / N
    main = -2.0 * np.ones(N)
    off = np.ones(N - 1)
    D = sp.diags([off, main, off], [-1, 0, 1], format="lil") / h**2
    D = D.tolil()
    if bc == NEUMANN:
        D[0, 0] = -1.0 / h**2
        D[N - 1, N - 1] = -1.0 / h**2
    elif bc != DIRICHLET:
        raise ValueError(bc)
    return D.tocsr()


def laplacian(dims, ls, bc: str) -> sp.csr_matrix:
    """Kronecker-sum Laplacian, x fastest: ``kron(I_y, D2x) + kron(D2y, I_x)`` (+ z the same
    way) -- SH2d-fronts.jl:27, SH3d.jl:38-39.  ``dims=(Nx[,Ny[,Nz]])``, ``ls=(lx[,ly[,lz]])``."""
    dims = tuple(int(d) for d in dims)
    Ds = [second_difference(n, l, bc) for n, l in zip(dims, ls)]
    eye = lambda n: sp.identity(n, format="csr")
    if len(dims) == 1:
        return Ds[0]
    if len(dims) == 2:
        Nx, Ny = dims
        return (sp.kron(eye(Ny), Ds[0]) + sp.kron(Ds[1], eye(Nx))).tocsr()
    Nx, Ny, Nz = dims
    A2 = sp.kron(eye(Ny), Ds[0]) + sp.kron(Ds[1], eye(Nx))
    return (sp.kron(eye(Nz), A2) + sp.kron(sp.kron(Ds[2], eye(Ny)), eye(Nx))).tocsr()


def grid_axes(dims, ls):
    """``X = -lx .+ 2lx/Nx * (0:Nx-1)`` (right end point excluded) -- SH3d.jl:72-74."""
    return [-l + 2.0 * l / n * np.arange(n) for n, l in zip(dims, ls)]


class SwiftHohenberg:
    """2-D / 3-D quadratic-cubic Swift-Hohenberg, Neumann-ghost boundaries.

    ``F = -L1 u + l u + nu u^2 - u^3``                         (SH3d.jl:44-47)
    ``dF(u) du = -L1 du + (l + 2 nu u - 3 u^2) du``            (SH3d.jl:50-53)
    ``L1 = (I + Lap)^2`` as a sparse matrix product            (SH3d.jl:85, SH2d-fronts.jl:55)
    """

    def __init__(self, dims, ls):
        self.dims = tuple(int(d) for d in dims)
        self.ls = tuple(float(x) for x in ls)
        self.N = int(np.prod(self.dims))
        lap = laplacian(self.dims, self.ls, NEUMANN)
        A = sp.identity(self.N, format="csr") + lap
        self.lap = lap
        self.L1 = (A @ A).tocsr()

    def F(self, u, l, nu):
        return -(self.L1 @ u) + (l * u + nu * u**2 - u**3)

    def dF(self, u, l, nu, du):
        return -(self.L1 @ du) + (l + 2.0 * nu * u - 3.0 * u**2) * du

    def J(self, u, l, nu):
        """J_sh (SH3d.jl:56-59): ``-L1 + spdiagm(l + 2 nu u - 3 u^2)``."""
        return (-self.L1 + sp.diags(l + 2.0 * nu * u - 3.0 * u**2)).tocsr()

    def guess(self):
        """sol0 of SH3d.jl:77-80 (3-D) / SH2d-fronts.jl:47-51 (2-D), flattened x-fastest."""
        ax = grid_axes(self.dims, self.ls)
        if len(self.dims) == 3:
            X, Y, Z = np.meshgrid(*ax, indexing="ij")
            s = np.cos(X) * np.cos(Y) + 0.0 * Z
            s = s - s.min()
            s = s / s.max()
            s = s * 1.2
        else:
            X, Y = np.meshgrid(*ax, indexing="ij")
            s = np.cos(X) + np.cos(X / 2.0) * np.cos(np.sqrt(3.0) * Y / 2.0)
            s = s - s.min()
            s = s / s.max()
            s = s - 0.25
            s = s * 1.7
        # Julia `vec` of an [x, y, z] comprehension is column-major: x fastest.
        return np.ascontiguousarray(s.reshape(-1, order="F"))


class SwiftHohenberg1D:
    """1-D cubic-quintic SH of SHpde_snaking.jl:8-31, Dirichlet.
    ``L1 = -(I + Lap)^2``;  ``R = L1 u + lam u + nu u^3 - u^5``;  ``J = L1 + diag(lam + 3 nu u^2 - 5 u^4)``."""

    def __init__(self, N=200, l=6.0):
        self.N = int(N)
        self.l = float(l)
        lap = second_difference(self.N, self.l, DIRICHLET)
        A = sp.identity(self.N, format="csr") + lap
        self.L1 = (-(A @ A)).tocsr()
        self.X = grid_axes((self.N,), (self.l,))[0]

    def F(self, u, lam, nu):
        return self.L1 @ u + lam * u + nu * u**3 - u**5

    def dF(self, u, lam, nu, du):
        return self.L1 @ du + (lam + 3.0 * nu * u**2 - 5.0 * u**4) * du

    def J(self, u, lam, nu):
        return (self.L1 + sp.diags(lam + 3.0 * nu * u**2 - 5.0 * u**4)).tocsr()

    def guess(self):
        return 1.1 * np.cos(self.X)


class CGL2d:
    """2-D cubic-quintic complex Ginzburg-Landau, two real fields stacked SoA ``u = [u1; u2]``,
    Dirichlet.  cGL2d.jl:24-54 (NL, Fcgl!), :57-79 (Jcgl), :281-305 (dNL! closed form)."""

    def __init__(self, dims, ls):
        self.dims = tuple(int(d) for d in dims)
        self.ls = tuple(float(x) for x in ls)
        self.n = int(np.prod(self.dims))
        lap = laplacian(self.dims, self.ls, DIRICHLET)
        self.lap = lap
        self.Delta = sp.block_diag([lap, lap], format="csr")

    @staticmethod
    def default_params():
        return dict(r=0.5, mu=0.1, nu=1.0, c3=-1.0, c5=1.0, gamma=0.0)

    def NL(self, u, r, mu, nu, c3, c5, gamma=0.0):
        n = self.n
        u1, u2 = u[:n], u[n:]
        ua = u1**2 + u2**2
        f = np.empty_like(u)
        f[:n] = r * u1 - nu * u2 - ua * (c3 * u1 - mu * u2) - c5 * ua**2 * u1 + gamma
        f[n:] = r * u2 + nu * u1 - ua * (c3 * u2 + mu * u1) - c5 * ua**2 * u2
        return f

    def F(self, u, **p):
        return self.Delta @ u + self.NL(u, **p)

    def J(self, u, r, mu, nu, c3, c5, gamma=0.0):
        n = self.n
        u1, u2 = u[:n], u[n:]
        ua = u1**2 + u2**2
        f1u = r - 2 * u1 * (c3 * u1 - mu * u2) - c3 * ua - 4 * c5 * ua * u1**2 - c5 * ua**2
        f1v = -nu - 2 * u2 * (c3 * u1 - mu * u2) + mu * ua - 4 * c5 * ua * u1 * u2
        f2u = nu - 2 * u1 * (c3 * u2 + mu * u1) - mu * ua - 4 * c5 * ua * u1 * u2
        f2v = r - 2 * u2 * (c3 * u2 + mu * u1) - c3 * ua - 4 * c5 * ua * u2**2 - c5 * ua**2
        diag = np.concatenate([f1u, f2v])
        return (self.Delta + sp.diags([diag, f1v, f2u], [0, n, -n])).tocsr()

    def dF(self, u, du, **p):
        return self.J(u, **p) @ du


def dct_symbol(dims, ls, shift=0.0):
    """Eigenvalues ``(1 + lam_x + lam_y + lam_z)^2 + shift`` of ``L1 + shift I`` in the DCT basis, array axes (z, y, x).
    ``1 / min`` is the 2-norm of ``Pl^-1``: with ``Pl = cholesky(L1)`` (examples/SH3d.jl:88, shift 0) the modes next to the critical
    circle |k| = 1 of Swift-Hohenberg leave eigenvalues of order (h^2/12)^2 -- tests scale their rounding floors with it."""
    dims = tuple(int(d) for d in dims)
    lam = []
    for n, l in zip(dims, ls):
        h = 2.0 * l / n
        lam.append(-(4.0 / h**2) * np.sin(np.pi * np.arange(n) / (2.0 * n)) ** 2)
    grids = np.meshgrid(*lam[::-1], indexing="ij")
    return (1.0 + sum(grids)) ** 2 + shift


def dct_preconditioner(dims, ls, shift=0.0, workers=1):
    """Exact ``(L1 + shift I)^-1`` for the Neumann-ghost ``L1 = (I + Lap)^2`` through the orthonormal DCT-II
    (the 1-D operator of SH3d.jl:21-32 has eigenvectors cos(pi k (j+1/2)/N), eigenvalues -(4/h^2) sin^2(pi k/2N)).
    CPU stand-in for ``Pl = cholesky(Symmetric(L1))`` (examples/SH3d.jl:88) at sizes where a sparse factorisation
    no longer fits; mathematically the same operator.  Returns a callable v -> Pl \\ v on flat x-fastest vectors."""
    import scipy.fft as sfft
    dims = tuple(int(d) for d in dims)
    # array axes are (z, y, x) for a C-ordered reshape of an x-fastest vector
    shape = dims[::-1]
    sym = dct_symbol(dims, ls, shift)

    def apply(v):
        a = np.asarray(v, dtype=float).reshape(shape)
        s = sfft.dctn(a, type=2, norm="ortho", workers=workers)
        s /= sym
        return sfft.idctn(s, type=2, norm="ortho", workers=workers).reshape(-1)

    return apply


def dst_block_preconditioner_cgl(dims, ls, a, b, workers=1):
    """Exact ``(Lap (x) I_2 + [[a, -b], [b, a]])^-1`` on the stacked cGL fields through the orthonormal DST-I of the
    Dirichlet Laplacian (cGL2d.jl:6-22: eigenvectors sin(pi (k+1)(j+1)/(N+1)), eigenvalues -(4/h^2) sin^2(pi (k+1)/2(N+1))).
    With a = r, b = nu: the inverse of Jcgl at u = 0 (cGL2d.jl:57-79) -- the CPU stand-in for the sparse LU the reference
    uses on this problem.  Returns a callable on flat [u1; u2] vectors."""
    import scipy.fft as sfft
    dims = tuple(int(d) for d in dims)
    lam = []
    for n, l in zip(dims, ls):
        h = 2.0 * l / n
        lam.append(-(4.0 / h**2) * np.sin(np.pi * (np.arange(n) + 1) / (2.0 * (n + 1))) ** 2)
    shape = dims[::-1]
    m = sum(np.meshgrid(*lam[::-1], indexing="ij")) + a
    det = m * m + b * b
    nn = int(np.prod(dims))

    def apply(v):
        v = np.asarray(v, dtype=float)
        s1 = sfft.dstn(v[:nn].reshape(shape), type=1, norm="ortho", workers=workers)
        s2 = sfft.dstn(v[nn:].reshape(shape), type=1, norm="ortho", workers=workers)
        y1 = (m * s1 + b * s2) / det
        y2 = (m * s2 - b * s1) / det
        out = np.empty_like(v)
        out[:nn] = sfft.idstn(y1, type=1, norm="ortho", workers=workers).reshape(-1)
        out[nn:] = sfft.idstn(y2, type=1, norm="ortho", workers=workers).reshape(-1)
        return out

    return apply

