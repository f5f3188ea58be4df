#!/usr/bin/env python3
"""Coefficients of gelu_erf (csrc/igemm_epi.h): log2 Phi(-a) on [0, 6] as a degree-6 polynomial, fitted by weighted least squares on
Chebyshev nodes (weight a Phi(-a): the error that reaches gelu(x) = max(x, 0) - |x| Phi(-|x|)), then checked in f32 arithmetic
against the f64 erf form over [-12, 12].  Prints the coefficients (constant term first) and the errors."""
import numpy as np
from numpy.polynomial import chebyshev as Ch, polynomial as P
from scipy.special import erf, log_ndtr

A, DEG = 6.0, 6
n = 4000
t = np.cos(np.pi * (np.arange(n) + 0.5) / n)
a = (t + 1) / 2 * A
q = log_ndtr(-a) / np.log(2)
c = Ch.chebfit(t, q, DEG, w=np.maximum(a * np.exp2(q), 1e-7))
pa = np.zeros(1)
for k, co in enumerate(Ch.cheb2poly(c)):
    pa = P.polyadd(pa, co * P.polypow([-1, 2 / A], k))
print("coefficients:", ", ".join("%.9g" % v for v in pa))
xs = np.linspace(-12, 12, 600001).astype(np.float32)
aa = np.minimum(np.abs(xs), np.float32(A))
c32 = pa.astype(np.float32)
p = np.full_like(aa, c32[-1])
for co in c32[-2::-1]:
    p = (p * aa + co).astype(np.float32)
out = (np.maximum(xs, 0) - np.abs(xs) * np.exp2(p).astype(np.float32)).astype(np.float32)
x64 = xs.astype(np.float64)
true = x64 * (0.5 + 0.5 * erf(x64 / np.sqrt(2)))
err = np.abs(out - true)
print("max abs error %.3g, max error relative to max(|gelu|, 1e-3) %.3g" % (err.max(), (err / np.maximum(np.abs(true), 1e-3)).max()))
