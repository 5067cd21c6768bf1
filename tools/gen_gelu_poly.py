"""Writes facialmmt_amd/csrc/gelu_poly_data.h: coefficients of the odd polynomials

    Phi(x)   ~ 0.5 + x * P(x^2)      |x| <= RP   (Phi = standard normal CDF; gelu(x) = x Phi(x), nn.GELU default)
    gelu'(x) ~ 0.5 + x * G(x^2)      |x| <= RG   (gelu'(x) = Phi(x) + x phi(x))

(round 4; clamped argument: |error| <= 5e-5 ABSOLUTE, hence no relative accuracy and no sign guarantee where the functions are smaller than
that: gelu below x ~ -3.5, gelu' below ~ -4) and, round 6, of the forms the bf16 kernels evaluate now -- relative accuracy on the negative tail:

    a = clamp(-|x|, -R, 0)
    gelu(x)  = max(x, 0) + a * 2^L(a)                    L(a) ~ log2 Phi(a), degree 6, plain polynomial in a        (gelu(x) = x + gelu(-x) for x > 0)
    gelu'(x) = x < 0 ? g : 1 - g,  g = 2^(c a^2) * S(a)  S(a) ~ Phi(a) / phi(a) / sqrt(2 pi) + a / sqrt(2 pi), degree 8, c = -log2(e) / 2
                                                          (gelu'(x) = 1 - gelu'(-x); S is smooth: the Gaussian factor carries the decay)

one v_exp_f32 each, the rest plain FMAs.  Fit: least squares on Chebyshev nodes in double precision (near-minimax), converted to the monomial basis in
u = 2 (x / R)^2 - 1 (|u| <= 1, coefficients <= 0.2: Horner in fp32 stays well conditioned -- in x^2 itself the 13-term gelu' fit loses
three digits); the script also runs the fp32 Horner form the kernels use over a dense grid and prints the measured maximum errors,
which are copied into the header's comment.
    python tools/gen_gelu_poly.py"""
import math
import os

import numpy as np
from numpy.polynomial import chebyshev as C


def phi_cdf(x):
    return np.array([0.5 * math.erfc(-v / math.sqrt(2.0)) for v in np.atleast_1d(x)])


def gelu_grad(x):
    x = np.atleast_1d(x)
    return phi_cdf(x) + x * np.exp(-0.5 * x * x) / math.sqrt(2.0 * math.pi)


def fit(f, R, nterms):
    n = 6001
    x = np.cos(np.pi * (np.arange(n) + 0.5) / n) * R
    x = x[np.abs(x) > 1e-9]
    u = 2.0 * (x / R) ** 2 - 1.0
    V = C.chebvander(u, nterms - 1) * x[:, None]
    coef, *_ = np.linalg.lstsq(V, f(x) - 0.5, rcond=None)
    return C.cheb2poly(coef)                                # chebyshev in u  ->  monomial in u


def eval_f32(mono_s, R, x):
    """the kernels' arithmetic in fp32 (fmmt_common.h, fmmt_odd_poly2): xc = clamp(x), u = xc * xc * (2 / R^2) - 1, Horner, 0.5 + xc * q"""
    f = np.float32
    xc = np.clip(x.astype(f), f(-R), f(R))
    s = ((xc * xc).astype(f) * f(2.0 / (R * R))).astype(f) - f(1.0)
    q = np.full_like(xc, f(mono_s[-1]))
    for c in mono_s[-2::-1]:
        q = (q * s).astype(f) + f(c)
    return (q * xc).astype(f) + f(0.5)


def report(name, f, mono_s, R):
    x = np.linspace(-10.0, 10.0, 400001)
    got = eval_f32(mono_s, R, x).astype(np.float64)
    want = f(x)
    err = np.abs(got - want)
    return f"{name}: |error| <= {err.max():.2e} on [-10, 10] (fp32 Horner, argument clamped to +-{R}); {len(mono_s)} terms"


def emit(name, mono_s):
    return f"constexpr float {name}[{len(mono_s)}] = {{" + ", ".join(f"{c:.9e}f" for c in mono_s) + "};\n"


RP, NP_, RG, NG = 4.0, 8, 4.5, 10
cp = fit(phi_cdf, RP, NP_)
cg = fit(gelu_grad, RG, NG)
lines = [report("Phi", phi_cdf, cp, RP), report("gelu'", gelu_grad, cg, RG)]

# ---- round 6: the exponential forms ----
from scipy.special import log_ndtr, ndtr

RL, DL, RS, DS = 9.0, 6, 8.5, 8


def fit_log2phi(R, deg, n=4001):
    t = np.cos(np.pi * (np.arange(n) + 0.5) / n)
    a = -R * (1.0 - t) / 2.0
    coef = C.chebfit(t, log_ndtr(a) / math.log(2.0), deg)
    mono_t = C.cheb2poly(coef)                              # in t = 2 a / R + 1
    from numpy.polynomial import polynomial as P
    pa, acc, lin = np.zeros(1), np.array([1.0]), np.array([1.0, 2.0 / R])
    for c in mono_t:
        pa = P.polyadd(pa, c * acc)
        acc = P.polymul(acc, lin)
    return pa                                               # monomial in a


def s_fun(a):
    return np.exp(log_ndtr(a) + 0.5 * a * a) + a / math.sqrt(2.0 * math.pi)


def fit_s(R, deg, n=8001):
    """weighted least squares: the error of g = e(a) S(a) is e(a) dS -- 5e-5 absolute near 0, and 1e-3 RELATIVE to S on the tail (a < -2)"""
    t = np.cos(np.pi * (np.arange(n) + 0.5) / n)
    a = -R * (1.0 - t) / 2.0
    e = np.exp(-0.5 * a * a)
    tol = 5e-5 / e
    tol = np.where(a < -2.0, np.minimum(tol, 1e-3 * np.abs(s_fun(a))), tol)
    sc = R ** np.arange(deg + 1)
    V = np.vander(a, deg + 1, increasing=True) / sc
    coef, *_ = np.linalg.lstsq(V / tol[:, None], s_fun(a) / tol, rcond=None)
    return coef / sc


def horner_f32(mono, a):
    f = np.float32
    q = np.full_like(a, f(mono[-1]))
    for c in mono[-2::-1]:
        q = (q * a).astype(f) + f(c)
    return q


def eval_gelu_exp(pl, x):
    f = np.float32
    xf = x.astype(f)
    a = np.maximum(-np.abs(xf), f(-RL))
    t = (a * np.exp2(horner_f32(pl, a).astype(np.float64)).astype(f)).astype(f)
    return (np.maximum(xf, f(0)) + t).astype(f)


def eval_grad_exp(ps, x):
    f = np.float32
    xf = x.astype(f)
    a = np.maximum(-np.abs(xf), f(-RS))
    e = np.exp2(((a * f(-0.5 / math.log(2.0))).astype(f) * a).astype(np.float64)).astype(f)
    g = (e * horner_f32(ps, a)).astype(f)
    return np.where(xf < 0, g, (f(1.0) - g).astype(f))


pl, ps = fit_log2phi(RL, DL), fit_s(RS, DS)
x = np.linspace(-10.0, 10.0, 800001)
tg = x * phi_cdf(x)
gg = eval_gelu_exp(pl, x).astype(np.float64)
neg = (x < -1e-3) & (x >= -8.0)
lines.append(f"gelu (exp form): |error| <= {np.abs(gg - tg).max():.2e} on [-10, 10], relative <= {(np.abs(gg[neg] - tg[neg]) / np.abs(tg[neg])).max():.2e} on [-8, 0), "
             f"sign exact (result <= 0 for x < 0: {bool(np.all(gg[x < 0] <= 0))}); degree {DL} in a = clamp(-|x|, -{RL}, 0)")
td = gelu_grad(x)
gd = eval_grad_exp(ps, x).astype(np.float64)
tail = (x < -1.5) & (x >= -8.0)
lines.append(f"gelu' (exp form): |error| <= {np.abs(gd - td).max():.2e} on [-10, 10], relative <= {(np.abs(gd[tail] - td[tail]) / np.abs(td[tail])).max():.2e} on [-8, -1.5], "
             f"sign exact there ({bool(np.all(np.sign(gd[tail]) == np.sign(td[tail])))}); degree {DS} in a = clamp(-|x|, -{RS}, 0)")


def eval_both_exp(pl, x):
    """gelu_both_exp_f (fmmt_common.h): gelu' from the SAME Phi polynomial and one more exponential"""
    f = np.float32
    xf = x.astype(f)
    a = np.maximum(-np.abs(xf), f(-RL))
    cdf = np.exp2(horner_f32(pl, a).astype(np.float64)).astype(f)
    u = ((a * f(-0.5 / math.log(2.0))).astype(f) * a).astype(f) + f(math.log2(1.0 / math.sqrt(2.0 * math.pi)))
    pdf = np.exp2(u.astype(np.float64)).astype(f)
    ga = ((a * pdf).astype(f) + cdf).astype(f)
    return np.where(xf < 0, ga, (f(1.0) - ga).astype(f))


gb = eval_both_exp(pl, x).astype(np.float64)
lines.append(f"gelu' (shared form, FMMT_EPI_GELU_DG): |error| <= {np.abs(gb - td).max():.2e} on [-10, 10], relative <= {(np.abs(gb[tail] - td[tail]) / np.abs(td[tail])).max():.2e} on [-8, -1.5], "
             f"sign exact there ({bool(np.all(np.sign(gb[tail]) == np.sign(td[tail])))}): Phi(a) + a phi(a) with Phi = 2^L(a)")
for ln in lines:
    print(ln)
here = os.path.dirname(os.path.abspath(__file__))
with open(os.path.join(here, "..", "facialmmt_amd", "csrc", "gelu_poly_data.h"), "w") as f:
    f.write("// Generated by tools/gen_gelu_poly.py -- do not edit.\n"
            "// What the bf16 kernels evaluate (round 6): a = clamp(-|x|, -R, 0); gelu(x) = max(x, 0) + a 2^L(a); gelu'(x) = x < 0 ? g : 1 - g, g = 2^(c a^2) S(a);\n"
            "//   L, S plain polynomials in a (lowest power first).  For the record, the first two lines: the round-4 odd polynomials these replaced\n"
            "//   (Phi(x) ~ 0.5 + x P(u), gelu'(x) ~ 0.5 + x G(u), u = 2 (x / R)^2 - 1, argument clamped to [-R, R]; the script still fits and measures them).\n")
    for ln in lines:
        f.write("//   " + ln + "\n")
    f.write("#pragma once\n")
    f.write(f"constexpr float FMMT_GELU_L_R = {RL}f, FMMT_GELU_S_R = {RS}f, FMMT_GELU_S_C = {-0.5 / math.log(2.0):.9e}f;\n")
    f.write(emit("fmmt_gelu_log2phi_poly", pl))
    f.write(emit("fmmt_gelu_grad_s_poly", ps))
