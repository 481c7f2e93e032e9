"""Fit of the GELU tail polynomial used by csrc/mdr_encoder.hip (gelu_erf2): Phi(-a) = 2^-(1 + a P(a)), P fitted to
-log2(erfc(a / sqrt 2)) / a on [0, A0] in a Chebyshev basis with weights p(a) * a (the error of Phi is p ln2 a dP), then evaluated
the way the kernel does (fp32 Horner, exp2) against scipy over [-8, 8]. Prints the error table and the n = 6 coefficients."""
import numpy as np
from scipy.special import erfc, erf
from numpy.polynomial import chebyshev as C, polynomial as P
# h(a) = -log2(erfc(a/sqrt2)) , a=|x| in [0, A0]; fit h(a) = a*(c1 + c2 a + ...)
def fit(A0, n, wt=True):
    k = np.arange(6000)
    t = np.cos(np.pi*(k+0.5)/6000)
    a = (t+1)/2*A0
    h = -np.log2(erfc(a/np.sqrt(2)))
    q = np.where(a>1e-9, h/np.maximum(a,1e-300), 2/np.sqrt(2*np.pi)/np.log(2))
    # weight: error in phi = p*ln2*dh = p*ln2*a*dq ; weight w = p*a
    p = 0.5*erfc(a/np.sqrt(2))
    w = (p*a + 1e-6) if wt else np.ones_like(a)
    V = C.chebvander(t, n-1)
    c, *_ = np.linalg.lstsq(V*w[:,None], q*w, rcond=None)
    pt = C.cheb2poly(c)
    aa, bb = 2/A0, -1.0
    pu = np.zeros(1)
    for i, ci in enumerate(pt):
        term = np.array([1.0])
        for _ in range(i): term = P.polymul(term, np.array([bb, aa]))
        pu = P.polyadd(pu, ci*term)
    return pu
def evalf32(pu, xs):
    ax = np.abs(xs).astype(np.float32)
    acc = np.full_like(ax, np.float32(pu[-1]))
    for ci in pu[-2::-1]:
        acc = acc*ax + np.float32(ci)
    e = acc*ax + np.float32(1.0)
    p = np.exp2(-e.astype(np.float32)).astype(np.float32)
    s = np.copysign(np.float32(0.5)-p, xs).astype(np.float32)
    g = (xs*s + np.float32(0.5)*xs).astype(np.float32)
    return g, np.where(xs>=0, 1-p, p)
xs = np.linspace(-8, 8, 1600001).astype(np.float32)
ref_phi = 0.5*erfc(-xs.astype(np.float64)/np.sqrt(2))
ref_g = xs.astype(np.float64)*ref_phi
for A0 in (5.0, 6.0):
    for n in range(4, 10):
        pu = fit(A0, n)
        g, ph = evalf32(pu, xs)
        print(f"A0={A0} n={n} max|phi err| {np.abs(ph-ref_phi).max():.2e}  max|gelu err| {np.abs(g-ref_g).max():.2e}  rel-to-fp16ulp {np.max(np.abs(g-ref_g)/np.maximum(np.abs(ref_g)*2**-11, 6e-8)):.3f}")
np.set_printoptions(precision=10)
pu = fit(6.0, 7); print(repr(pu))
pu = fit(6.0, 6); print("n=6:", ", ".join(f"{v:.9e}f" for v in pu))
g, ph = evalf32(pu, xs); print(np.abs(ph-ref_phi).max(), np.abs(g-ref_g).max())
# current erf_fast error for comparison
def erf_fast(x):
    ax = np.abs(x); t = 1/(1+0.3275911*ax)
    poly = t*(0.254829592+t*(-0.284496736+t*(1.421413741+t*(-1.453152027+t*1.061405429))))
    return np.copysign(1-poly*np.exp(-ax*ax), x)
g0 = 0.5*xs.astype(np.float64)*(1+erf_fast(xs.astype(np.float64)/np.sqrt(2)))
print("current formula (fp64 eval) max gelu err", np.abs(g0-ref_g).max())
