"""Coefficients of the transcendental-free GELU of the GEGLU epilogue (csrc/common.h::geglu_pair).

Phi(x) ~ 0.5 + xc * P((xc / c)^2), xc = clamp(x, -c, c): P is the least-squares fit at Chebyshev nodes of
(Phi(x) - 0.5) / x on (0, c]; the script scans c and the number of terms and prints the float32-evaluated error
max |x Phi_hat(x) - x Phi(x)| / |x| over [-10, 10].  Development tool; the kernel carries the printed constants."""
import numpy as np
from numpy.polynomial import Polynomial, chebyshev as Ch
from scipy.special import erf


def Phi(x):
    return 0.5 * (1 + erf(x / np.sqrt(2)))


def fit(c, nterm):
    xs = c * np.cos(np.pi * (np.arange(6000) + 0.5) / 12000)
    u = (xs / c) ** 2
    coef = Ch.chebfit(2 * u - 1, (Phi(xs) - 0.5) / xs, nterm - 1)
    return Polynomial(Ch.cheb2poly(coef))(Polynomial([-1, 2])).coef


def error(c, pu):
    xt = np.linspace(-10, 10, 400001).astype(np.float32)
    xc = np.clip(xt, -c, c).astype(np.float32)
    uu = (xc * np.float32(1 / c)) ** 2
    acc = np.full_like(uu, np.float32(pu[-1]))
    for k in pu[-2::-1]:
        acc = acc * uu + np.float32(k)
    gel = xt * (np.float32(0.5) + xc * acc)
    ref = xt.astype(np.float64) * Phi(xt.astype(np.float64))
    return (np.abs(gel - ref) / np.maximum(np.abs(xt), 1e-3)).max()


if __name__ == "__main__":
    best = None
    for nterm in (8, 9, 10):
        for c in np.arange(4.1, 4.8, 0.05):
            pu = fit(c, nterm)
            e = error(c, pu)
            if best is None or e < best[0]:
                best = (e, nterm, c, pu)
    print("max |gelu err| / |x| = %.3e with %d terms, c = %.2f" % best[:3])
    print(", ".join("%.9ef" % v for v in best[3]))
