"""Accuracy of the two erf-GELU formulations of csrc/sd_common.h against fp64 (numpy restatement of their fp32 op sequences).
python tools/exp/gelu_check.py"""
import math
import numpy as np

u = np.linspace(-8, 8, 2_000_001).astype(np.float32)
want = np.array([0.5 * x * (1.0 + math.erf(x / math.sqrt(2.0))) for x in u.astype(np.float64)[::50]])
u50 = u[::50]
f = np.float32


def gelu_as(u):
    z = np.abs(u) * f(0.70710678118654752440)
    t = f(1.0) / (z * f(0.3275911) + f(1.0))
    p = t * f(1.061405429) + f(-1.453152027)
    p = p * t + f(1.421413741)
    p = p * t + f(-0.284496736)
    p = p * t + f(0.254829592)
    p = p * t
    e = p * np.exp2(z * z * f(-1.44269504088896340736)).astype(np.float32)
    hu = u * f(0.5)
    return np.where(u >= 0, u - hu * e, hu * e).astype(np.float32)


def gelu_cheb(u):
    z = u * f(0.70710678118654752440)
    a = np.abs(z)
    t = f(1.0) / (a * f(0.5) + f(1.0))
    p = np.full_like(u, f(0.17087277))
    for c in (-0.82215223, 1.48851587, -1.13520398, 0.27886807, -0.18628806, 0.09678418, 0.37409196, 1.00002368, -1.26551223):
        p = p * t + f(c)
    e = t * np.exp2((p - a * a) * f(1.44269504088896340736)).astype(np.float32)
    w = np.where(z >= 0, f(2.0) - e, e)
    return (u * f(0.5) * w).astype(np.float32)


for name, fn in (("Abramowitz-Stegun 7.1.26 (gelu_erf_as2)", gelu_as), ("Chebyshev erfc fit (gelu_erf_fast2)", gelu_cheb)):
    got = fn(u50).astype(np.float64)
    err = np.abs(got - want)
    print(f"{name:44s} max abs error {err.max():.3e} at u = {u50[err.argmax()]:+.3f};  max error relative to max(|gelu|, 1e-3) {np.max(err / np.maximum(np.abs(want), 1e-3)):.3e}")
