#!/usr/bin/env python3
"""Derive the order-5 WENO transform constants from first principles (exact rationals).

Build helper (tools/: it is neither the oracle nor built from the reference).  Emits `awfl_constants.h`,
once for the HIP product (`pam_amd/csrc/`) and once for the CPU oracle (`oracle/awfl_oracle.c`): the
numbers are mathematical constants, not code.  They restate the VALUES of the generated literals in the reference's
`dynamics/awfl/TransformMatrices.h`:

  sten_to_coefs<5,5>            TransformMatrices.h:970-997
  coefs_to_gll_lower<5,2>       TransformMatrices.h:1132-1144
  weno_lower_sten_to_coefs<3>   TransformMatrices.h:1218-1247
  coefs_to_tv<3>, <5>           TransformMatrices.h:188-193, :871-876
  get_gll_points/weights<9>     TransformMatrices.h:4113-4138

Derivation (unit cells centred on the stencil's middle cell):
  * coefs->stencil matrix  M(p,c) = cell average of x^p over cell c  (a Vandermonde of cell-average
    monomials); stencil->coefs = M^{-1}, index order (cell, coef) as the reference stores it.
  * edge evaluation  c2g(p,ind) = (-1/2)^p (ind=0, left edge) or (+1/2)^p (ind=1, right edge).
  * total variation  TV(a) = sum_{l>=1} integral_{-1/2}^{1/2} (d^l p/dx^l)^2 dx.
  * 9-point Gauss-Lobatto-Legendre nodes/weights on [-1/2, 1/2] (Newton on P8', 60 digits).

`python tools/gen_constants.py` rewrites `pam_amd/csrc/awfl_constants.h` and the copy under `oracle/`
(the oracle's Makefile calls it from here); `tests/test_oracle_kat.py` re-derives and checks them,
`tests/test_constants_vs_reference.py` compares them with the reference's literals.
"""
from fractions import Fraction as F
from decimal import Decimal, getcontext
import os
import sys


def cell_avg_monomial(p, lo, hi):
    # average of x^p over [lo, hi]
    return (hi ** (p + 1) - lo ** (p + 1)) / ((p + 1) * (hi - lo))


def inverse(mat):
    n = len(mat)
    a = [[F(x) for x in row] + [F(int(i == j)) for j in range(n)] for i, row in enumerate(mat)]
    for c in range(n):
        piv = next(r for r in range(c, n) if a[r][c] != 0)
        a[c], a[piv] = a[piv], a[c]
        d = a[c][c]
        a[c] = [x / d for x in a[c]]
        for r in range(n):
            if r != c and a[r][c] != 0:
                f = a[r][c]
                a[r] = [x - f * y for x, y in zip(a[r], a[c])]
    return [row[n:] for row in a]


def sten_to_coefs(edges):
    """edges: n+1 cell edges. returns S[c][p]: coefficient p contribution of cell c."""
    n = len(edges) - 1
    M = [[cell_avg_monomial(p, edges[c], edges[c + 1]) for c in range(n)] for p in range(n)]  # M[p][c]
    Minv = inverse(M)  # Minv[c][p]
    return Minv


def tv_form(n):
    """quadratic form Q[i][j] so that TV = sum_{i<=j} Q[i][j] a_i a_j (upper triangular, merged)."""
    Q = [[F(0)] * n for _ in range(n)]
    for l in range(1, n):
        # l-th derivative of x^p = p!/(p-l)! x^(p-l)
        def dcoef(p):
            c = F(1)
            for t in range(l):
                c *= (p - t)
            return c
        for p in range(l, n):
            for q in range(l, n):
                e = (p - l) + (q - l)
                integral = cell_avg_monomial(e, F(-1, 2), F(1, 2))
                Q[p][q] += dcoef(p) * dcoef(q) * integral
    U = [[F(0)] * n for _ in range(n)]
    for i in range(n):
        for j in range(i, n):
            U[i][j] = Q[i][j] if i == j else Q[i][j] + Q[j][i]
    return U


def gll9():
    getcontext().prec = 70
    n = 8  # P_n ; nodes are +-1 and roots of P_n'

    def legendre(nn, x):
        p0, p1 = Decimal(1), x
        if nn == 0:
            return p0
        for k in range(2, nn + 1):
            p0, p1 = p1, ((2 * k - 1) * x * p1 - (k - 1) * p0) / k
        return p1

    def dlegendre(nn, x):
        return nn * (x * legendre(nn, x) - legendre(nn - 1, x)) / (x * x - 1)

    import math
    nodes = [Decimal(-1)]
    for i in range(1, n):
        x = Decimal(-math.cos(math.pi * i / n))
        for _ in range(200):
            # Newton on q(x) = P_n'(x): q' from Legendre ODE: (1-x^2)P'' = 2xP' - n(n+1)P
            q = dlegendre(n, x)
            qp = (2 * x * q - n * (n + 1) * legendre(n, x)) / (1 - x * x)
            dx = q / qp
            x -= dx
            if abs(dx) < Decimal(10) ** -60:
                break
        nodes.append(x)
    nodes.append(Decimal(1))
    weights = [Decimal(2) / (n * (n + 1) * legendre(n, x) ** 2) for x in nodes]
    # map [-1,1] -> [-1/2,1/2]
    return [x / 2 for x in nodes], [w / 2 for w in weights]


def lit(x):
    """decimal literal that rounds to the correctly rounded double of the exact value."""
    if isinstance(x, F):
        return repr(x.numerator / x.denominator) if x != 0 else "0.0"
    return repr(float(x))


def build():
    edges5 = [F(2 * c - 5, 2) for c in range(6)]      # -5/2 .. 5/2
    S5 = sten_to_coefs(edges5)                        # S5[s][ii]
    # three quadratic sub-stencils, each in the coordinate of the 5-stencil's centre cell
    W3 = [sten_to_coefs(edges5[i:i + 4]) for i in range(3)]   # W3[i][s][ii]
    c2g = [[F(-1, 2) ** p, F(1, 2) ** p] for p in range(5)]
    tv3 = tv_form(3)
    tv5 = tv_form(5)
    assert tv3[1][1] == 1 and tv3[2][2] == F(13, 3) and tv3[1][2] == 0
    assert (tv5[1][1], tv5[2][2], tv5[1][3], tv5[3][3], tv5[2][4]) == \
        (1, F(13, 3), F(1, 2), F(3129, 80), F(21, 5))
    # Quirk Q9: the derivation gives 87617/140 = 625.8357... for a4*a4, but the reference literal
    # (TransformMatrices.h:873) is 625.8 = 3129/5 (it lacks the first-derivative term 1/28).
    # Parity with the reference wins: use the reference's value.
    assert tv5[4][4] == F(87617, 140) and F(87617, 140) - F(1, 28) == F(3129, 5)
    tv5[4][4] = F(3129, 5)
    assert S5[0][0] == F(3, 640) and S5[2][0] == F(1067, 960) and W3[0][2][0] == F(23, 24)
    pts, wts = gll9()
    return dict(S5=S5, W3=W3, c2g=c2g, tv3=tv3, tv5=tv5, gll_pts=pts, gll_wts=wts)


def emit(c):
    o = []
    o.append("/* GENERATED by tools/gen_constants.py -- do not edit.  Mathematical constants of the order-5")
    o.append(" * WENO transform (values restate dynamics/awfl/TransformMatrices.h:188,871,970,1132,1218,4113,4126). */")
    o.append("#ifndef AWFL_CONSTANTS_H\n#define AWFL_CONSTANTS_H\n")
    o.append("/* stencil -> polynomial coefficients, [s][ii] (TransformMatrices.h:970) */")
    o.append("#define AWFL_STEN_TO_COEFS_INIT { \\")
    for s in range(5):
        o.append("  { " + ", ".join(lit(c["S5"][s][ii]) for ii in range(5)) + " }, \\")
    o.append("}")
    o.append("/* three quadratic sub-stencils -> coefficients, [i][s][ii] (TransformMatrices.h:1218) */")
    o.append("#define AWFL_WENO_LOWER_INIT { \\")
    for i in range(3):
        o.append("  { " + ", ".join("{ " + ", ".join(lit(c["W3"][i][s][ii]) for ii in range(3)) + " }"
                                    for s in range(3)) + " }, \\")
    o.append("}")
    o.append("/* coefficients -> cell-edge values, [ii][ind] ind=0 left edge, 1 right edge (TransformMatrices.h:1132) */")
    o.append("#define AWFL_COEFS_TO_GLL_INIT { \\")
    for p in range(5):
        o.append("  { " + ", ".join(lit(x) for x in c["c2g"][p]) + " }, \\")
    o.append("}")
    o.append("/* total-variation forms (TransformMatrices.h:188, :871) */")
    o.append("#define AWFL_TV3_A2A2 " + lit(c["tv3"][2][2]))
    o.append("#define AWFL_TV5_A1A1 " + lit(c["tv5"][1][1]))
    o.append("#define AWFL_TV5_A2A2 " + lit(c["tv5"][2][2]))
    o.append("#define AWFL_TV5_A1A3 " + lit(c["tv5"][1][3]))
    o.append("#define AWFL_TV5_A3A3 " + lit(c["tv5"][3][3]))
    o.append("#define AWFL_TV5_A2A4 " + lit(c["tv5"][2][4]))
    o.append("#define AWFL_TV5_A4A4 " + lit(c["tv5"][4][4]))
    o.append("/* square roots of the a3^2 and a4^2 weights: the HIP tables carry the x^3 / x^4 rows pre-scaled by them */")
    o.append("#define AWFL_TV5_SQRT_A3A3 " + repr(float(c["tv5"][3][3]) ** 0.5))
    o.append("#define AWFL_TV5_SQRT_A4A4 " + repr(float(c["tv5"][4][4]) ** 0.5))
    o.append("/* WENO ideal weights / sigma before convexification (WenoLimiter.h:39-44) */")
    o.append("#define AWFL_WENO_SIGMA 0.73564225445964")
    o.append("#define AWFL_WENO_IDL_INIT { 1.0, 73.564225445964, 1.0, 1584.89319246111 }")
    o.append("/* 9-point GLL nodes / weights on [-1/2,1/2] (TransformMatrices.h:4113, :4126) */")
    o.append("#define AWFL_GLL9_PTS_INIT { " + ", ".join(lit(x) for x in c["gll_pts"]) + " }")
    o.append("#define AWFL_GLL9_WTS_INIT { " + ", ".join(lit(x) for x in c["gll_wts"]) + " }")
    o.append("\n#endif")
    return "\n".join(o) + "\n"


def main():
    here = os.path.dirname(os.path.abspath(__file__))
    text = emit(build())
    targets = [os.path.join(here, "..", "pam_amd", "csrc", "awfl_constants.h"),
               os.path.join(here, "..", "oracle", "awfl_constants.h")]
    for t in targets:
        os.makedirs(os.path.dirname(t), exist_ok=True)
        with open(t, "w") as f:
            f.write(text)
    if "-v" in sys.argv:
        print(text)


if __name__ == "__main__":
    main()
