"""Fit of the two polynomials behind gelu_fast2 / gelu_and_grad_fast2 (csrc/gemm.hip) and their error report.

    Phi(x) - 0.5 = x P(x^2),   gelu'(x) - 0.5 = x R(x^2)      on |x| <= 4 (inputs are clamped to that range)
Weighted least squares on Chebyshev nodes (weight x: absolute error of Phi / gelu'), evaluated in fp32 Horner form."""
import numpy as np
from numpy.polynomial import chebyshev as Ch, polynomial as Po
from scipy.special import erf

X, DEG = 4.0, 8
Phi = lambda t: 0.5 * (1 + erf(t / np.sqrt(2)))          # noqa: E731
phi = lambda t: np.exp(-t * t / 2) / np.sqrt(2 * np.pi)  # noqa: E731

if __name__ == "__main__":
    x = np.cos(np.pi * (np.arange(8000) + 0.5) / 8000) * X / 2 + X / 2
    x = x[x > 1e-6]
    u = x * x
    P = Ch.Chebyshev.fit(u, (Phi(x) - 0.5) / x, DEG, domain=[0, X * X], w=x).convert(kind=Po.Polynomial)
    R = Ch.Chebyshev.fit(u, (Phi(x) - 0.5) / x + phi(x), DEG, domain=[0, X * X], w=x).convert(kind=Po.Polynomial)

    def horner32(c, uu):
        acc = np.full_like(uu, np.float32(c[-1]))
        for k in c[-2::-1]:
            acc = acc * uu + np.float32(k)
        return acc

    xt = np.linspace(-8, 8, 800001)
    x32 = xt.astype(np.float32)
    xc = np.clip(x32, np.float32(-X), np.float32(X))
    uu = xc * xc
    cdf = xc * horner32(P.coef, uu) + np.float32(0.5)
    g, gd = x32 * cdf, xc * horner32(R.coef, uu) + np.float32(0.5)
    inside = np.abs(xt) <= X
    print("P (c0..c8):", ", ".join(f"{c:.8e}f" for c in P.coef))
    print("R (c0..c8):", ", ".join(f"{c:.8e}f" for c in R.coef))
    print(f"max |Phi err|   on [-4, 4]: {np.abs(cdf - Phi(xt))[inside].max():.2e}")
    print(f"max |gelu err|  on [-4, 4]: {np.abs(g - xt * Phi(xt))[inside].max():.2e}   on [-8, 8]: {np.abs(g - xt * Phi(xt)).max():.2e}")
    ref = Phi(xt) + xt * phi(xt)
    print(f"max |gelu' err| on [-4, 4]: {np.abs(gd - ref)[inside].max():.2e}   on [-8, 8]: {np.abs(gd - ref).max():.2e}")
