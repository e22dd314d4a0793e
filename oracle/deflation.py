"""ORACLE (test infrastructure) -- deflated Newton, restated on flat NumPy vectors.

  DeflationOperator      M(u) = prod_i ( <u - r_i, u - r_i>^-p + alpha )  (or the mean)
                                                     src/DeflationOperator.jl:61-141
  dM                     finite difference (M(u + delta du) - M(u)) / delta, delta = 1e-8        :160-169
  deflated_residual      M(u) F(u)                                                               :194-198
  custom_ls              DeflatedProblemCustomLS: two solves with the plain Jacobian + a Sherman-Morrison
                         style recombination                                                     :258-312
  deflated_newton        solve(prob, defOp, options, DeflatedProblemCustomLS())                  :340-355
                         (= _newton, src/Newton.jl:66-114, on the deflated functional)
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np

from .bordered import solve2
from .palc import norm2


@dataclass
class DeflationOperator:
    power: float
    alpha: float
    roots: list = field(default_factory=list)
    accumulator: str = "prod"            # Val(:Prod) | Val(:Mean)
    delta: float = 1e-8

    def __call__(self, u):               # :124-138
        if not self.roots:
            return 1.0
        M = lambda d: 1.0 / float(np.dot(d, d)) ** self.power + self.alpha
        out = M(u - self.roots[0])
        for r in self.roots[1:]:
            v = M(u - r)
            out = out * v if self.accumulator == "prod" else out + v
        if self.accumulator == "mean":
            out /= len(self.roots)
        return out

    def dM(self, u, du):                 # Val(:dMwithTmp), :160-169 (autodiff = false)
        if not self.roots:
            return 0.0
        return (self(u + self.delta * du) - self(u)) / self.delta


def custom_ls(ls, prob, defop, u, p, rhs):
    """(dfl::DeflatedProblemCustomLS)(J, rhs) with J = (u, p, defPb), :264-312 -> (h, True, (it1, it2))."""
    Fu = prob.F(u, p)
    Mu = defop(u)
    Ju = prob.J(u, p)
    if not defop.roots:
        h1, _, it1 = ls(Ju, rhs)
        return h1, True, (it1, 0)
    h1, h2, _, (it1, it2) = solve2(ls, Ju, rhs, Fu)
    z1 = defop.dM(u, h1)
    z2 = defop.dM(u, h2)
    z = z1 / (Mu + z2)
    return (h1 - z * h2) / Mu, True, (it1, it2)


def deflated_newton(prob, defop, x0, p, ls, *, tol=1e-12, max_iterations=25, normN=norm2):
    """_newton (src/Newton.jl:66-114) on the deflated functional M(u) F(u) with DeflatedProblemCustomLS."""
    x = x0.copy()
    fx = prob.F(x, p) * defop(x)
    res = normN(fx)
    residuals = [res]
    step, itlin = 0, 0
    while step < max_iterations and res > tol:
        h, _, it = custom_ls(ls, prob, defop, x, p, fx)
        itlin += int(np.sum(it))
        x = x - h
        fx = prob.F(x, p) * defop(x)
        res = normN(fx)
        residuals.append(res)
        step += 1
    return dict(u=x, residuals=residuals, converged=residuals[-1] < tol, itnewton=step, itlineartot=itlin)
