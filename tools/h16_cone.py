#!/usr/bin/env python3
"""How far from a ray's line can a sphere lie and still get a positive discriminant in binary16 (sphere.h:18-22 with every
operation rounded to binary16, the USE_FP16 contract)?  CPU only (numpy float16 arithmetic = float operation + one rounding).
The rigorous bound: disc > 0  =>  dist(c, line)^2 < r^2 + 19 u |o - c|^2, u = 2^-11; this samples
near-tangent configurations and prints the worst excess seen, in units of u.  It is the error cone a candidate cull for the
binary16 kernel would have to cover: relative to the distance from the ray's origin, not a constant of the scene."""
import numpy as np
rng = np.random.default_rng(1)
h = np.float16
def disc16(o, d, c, r2):
    # sphere.h:18-22 in binary16, one rounding per operation (numpy float16 arithmetic = float op + round)
    ocx, ocy, ocz = o[:,0]-c[:,0], o[:,1]-c[:,1], o[:,2]-c[:,2]
    a = (d[:,0]*d[:,0] + d[:,1]*d[:,1]) + d[:,2]*d[:,2]
    b = (ocx*d[:,0] + ocy*d[:,1]) + ocz*d[:,2]
    cc = ((ocx*ocx + ocy*ocy) + ocz*ocz) - r2
    return b*b - a*cc, a
worst = 0.0
tot = 0
for rep in range(60):
    n = 2_000_000
    o = rng.uniform(-12, 12, (n,3)); o[:,1] = rng.uniform(0, 3, n)
    c = o + rng.normal(size=(n,3)) * rng.choice([0.5, 2, 5, 12], (n,1))
    c[:,1] = rng.uniform(0.0, 1.2, n)
    r = rng.choice([0.05, 0.1, 0.2, 0.2, 0.2, 1.0], n)
    # aim near-tangent: direction towards a point at distance ~r(1+eps) from the centre
    tgt = c + rng.normal(size=(n,3)); v = tgt - c; v /= np.linalg.norm(v, axis=1)[:,None]
    tgt = c + v * (r * (1 + rng.normal(0, 0.4, n)))[:,None] * rng.choice([1.0, 1.0, 3.0], (n,1))
    d = (tgt - o); d *= (rng.uniform(0.3, 2.0, n) / np.maximum(np.linalg.norm(d, axis=1), 1e-9))[:,None]
    o16, d16, c16 = o.astype(h), d.astype(h), c.astype(h)
    r2_16 = (r.astype(h) * r.astype(h))
    D, a16 = disc16(o16, d16, c16, r2_16)
    pos = D > h(0)
    O, Dd, Cc = o16.astype(np.float64), d16.astype(np.float64), c16.astype(np.float64)
    oc = O - Cc
    a = (Dd*Dd).sum(1); bq = (oc*Dd).sum(1); oc2 = (oc*oc).sum(1)
    dist2 = np.maximum(oc2 - bq*bq/np.maximum(a, 1e-30), 0)
    ok = pos & (a16.astype(np.float64) > 2.0**-8) & (oc2 > 1e-6)
    ex = (dist2 - r2_16.astype(np.float64)) / oc2
    if ok.any():
        worst = max(worst, ex[ok].max()); tot += int(ok.sum())
print("positive discriminants: %d; worst (dist^2 - r^2)/|oc|^2 = %.6f = %.2f u (u = 2^-11); bound 19 u = %.6f" % (tot, worst, worst * 2048, 19/2048))
