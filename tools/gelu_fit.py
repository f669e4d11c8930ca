#!/usr/bin/env python
"""Minimax (Lawson) fits behind pa_common.h's gelu_fast2 / gelu_grad_fast2 (bf16 epilogues only):

    Phi(x)  - 1/2 = 0.5 erf(x / sqrt 2)            ~  xc * P(xc^2),   xc = clamp(x, -XC, XC)
    gelu'(x) - 1/2 = 0.5 erf(x / sqrt 2) + x phi(x) ~  xc * Q(xc^2)

Odd polynomials evaluated by Horner in f32 with packed FMAs (no v_exp / v_rcp: on CDNA4 the transcendental
ops are quarter rate, and the GELU epilogue of the fc1 / dgrad-fc2 GEMMs is VALU bound).  Prints the
coefficients (highest degree last) and the worst absolute error of the f32 evaluation over [-8, 8].
"""
import numpy as np
from numpy.polynomial import chebyshev as C, polynomial as P
from scipy.special import erf

XC, DEG_P, DEG_Q = 4.25, 8, 9


def lawson(s, xfac, y, deg, iters=600):
    V = C.chebvander(2 * s - 1, deg) * xfac[:, None]
    w = np.ones_like(y)
    for _ in range(iters):
        c, *_ = np.linalg.lstsq(V * w[:, None], y * w, rcond=None)
        e = np.abs(V @ c - y)
        w = w * (1 + 4 * e / e.max())
        w /= w.max()
    c, *_ = np.linalg.lstsq(V * w[:, None], y * w, rcond=None)
    return c


def mono_in_t(c, xc):
    p = C.cheb2poly(c)
    out = np.zeros(1)
    for k, a in enumerate(p):
        out = P.polyadd(out, a * P.polypow([-1, 2], k))       # u = 2 s - 1
    return np.array([a / xc ** (2 * k) for k, a in enumerate(out)])   # s = t / xc^2


def eval_f32(coef, x):
    x = x.astype(np.float32)
    xc = np.clip(x, np.float32(-XC), np.float32(XC))
    t = xc * xc
    p = np.full_like(t, np.float32(coef[-1]))
    for a in coef[-2::-1]:
        p = p * t + np.float32(a)
    return xc * p + np.float32(0.5)


def main():
    x = np.linspace(1e-6, XC, 60001)
    s = (x / XC) ** 2
    phi = np.exp(-x * x / 2) / np.sqrt(2 * np.pi)
    cp = mono_in_t(lawson(s, x, 0.5 * erf(x / np.sqrt(2)), DEG_P), XC)
    cq = mono_in_t(lawson(s, x, 0.5 * erf(x / np.sqrt(2)) + x * phi, DEG_Q), XC)
    xx = np.linspace(-8, 8, 400001)
    cdf = 0.5 + 0.5 * erf(xx / np.sqrt(2))
    dg = cdf + xx * np.exp(-xx * xx / 2) / np.sqrt(2 * np.pi)
    e_cdf = np.abs(eval_f32(cp, xx) - cdf).max()
    e_gelu = np.abs(xx.astype(np.float32) * eval_f32(cp, xx) - xx * cdf).max()
    e_dg = np.abs(eval_f32(cq, xx) - dg).max()
    print("XC", XC)
    print("P (Phi - 1/2 = xc P(t)):", ", ".join(f"{a:.9e}f" for a in cp))
    print("Q (gelu' - 1/2 = xc Q(t)):", ", ".join(f"{a:.9e}f" for a in cq))
    print(f"max |Phi err| {e_cdf:.2e}   max |gelu err| {e_gelu:.2e}   max |gelu' err| {e_dg:.2e}   (f32 Horner, x in [-8, 8])")


if __name__ == "__main__":
    main()
