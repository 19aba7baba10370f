#!/usr/bin/env python3
"""Numerical model of the matrix-core candidate filter: is `disc_filter >= 0` a superset of the exact f32 rule?

The filter evaluates   disc = (d.(C-o))^2 - |C-o|^2 + r^2   expanded into 11 bilinear (ray feature x sphere feature) terms, every
f32 factor split into three bf16 parts (H, M, L) with the six largest cross products kept, plus the margin
eps (|C|^2 + r^2 + |o|^2).  Here the MFMA is modelled pessimistically (f32 accumulation, one rounding per product, in slot order);
the script reports the worst filter error relative to the margin over random and deliberately grazing (ray, sphere) pairs and counts
false negatives against the exact rule `c < 0 | (disc > 0 & h > 0)` evaluated in f32 exactly as the kernels do.

CPU only, numpy only; not part of the product.
"""
import sys
import numpy as np

EPS = 2e-5
f32 = np.float32


def bf16_rn(x):
    u = x.astype(f32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) >> 16
    return (u.astype(np.uint32) << 16).view(f32)


def split3(x):
    x = x.astype(f32)
    h = bf16_rn(x)
    r1 = (x - h).astype(f32)
    m = bf16_rn(r1)
    l = bf16_rn((r1 - m).astype(f32))
    return h, m, l


COMBOS = [(0, 0), (0, 1), (1, 0), (0, 2), (2, 0), (1, 1)]      # (ray part, sphere part)


def filter_disc(o, d, C, r, eps=EPS, accumulate=f32):
    """o, d: [n,3] f32; C: [n,3] f32, r: [n] f32 (pairs).  Returns the modelled accumulator value."""
    C64 = C.astype(np.float64)
    c2 = (C64 ** 2).sum(1)
    r2 = r.astype(np.float64) ** 2
    sph = [C64[:, 0] ** 2, C64[:, 1] ** 2, C64[:, 2] ** 2, C64[:, 0] * C64[:, 1], C64[:, 0] * C64[:, 2], C64[:, 1] * C64[:, 2],
           C64[:, 0], C64[:, 1], C64[:, 2]]
    sph = [s.astype(f32) for s in sph]
    kj = ((r2 - c2) + eps * (c2 + r2)).astype(f32)
    od = (o[:, 0] * d[:, 0] + o[:, 1] * d[:, 1] + o[:, 2] * d[:, 2]).astype(f32)
    oo = (o[:, 0] * o[:, 0] + o[:, 1] * o[:, 1] + o[:, 2] * o[:, 2]).astype(f32)
    two = f32(2.0)
    ray = [d[:, 0] * d[:, 0], d[:, 1] * d[:, 1], d[:, 2] * d[:, 2], two * d[:, 0] * d[:, 1], two * d[:, 0] * d[:, 2], two * d[:, 1] * d[:, 2],
           two * (o[:, 0] - od * d[:, 0]), two * (o[:, 1] - od * d[:, 1]), two * (o[:, 2] - od * d[:, 2])]
    ray = [x.astype(f32) for x in ray]
    e = (od * od - oo * f32(1.0 - eps)).astype(f32)
    sp = [split3(s) for s in sph]
    rp = [split3(x) for x in ray]
    kp = split3(kj)
    ep = split3(e)
    acc = np.zeros(len(r), accumulate)
    for (a, b) in COMBOS:
        for t in range(9):
            acc = (acc + (rp[t][a].astype(accumulate) * sp[t][b].astype(accumulate))).astype(accumulate)
    for p in range(3):
        acc = (acc + kp[p].astype(accumulate)).astype(accumulate)
    for p in range(3):
        acc = (acc + ep[p].astype(accumulate)).astype(accumulate)
    return acc


def fma(a, b, c):
    return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(f32)   # double rounding is harmless for a sign census


def exact_rule(o, d, C, r2):
    cx, cy, cz = (C[:, 0] - o[:, 0]).astype(f32), (C[:, 1] - o[:, 1]).astype(f32), (C[:, 2] - o[:, 2]).astype(f32)
    h = fma(cz, d[:, 2], fma(cy, d[:, 1], (cx * d[:, 0]).astype(f32)))
    c = fma(cz, cz, fma(cy, cy, fma(cx, cx, -r2)))
    disc = fma(h, h, -c)
    return (c < 0) | ((disc > 0) & (h > 0)), disc


def unit(v):
    v = v / np.linalg.norm(v, axis=1, keepdims=True)
    v = v.astype(f32)
    n = np.sqrt((v.astype(f32) ** 2).sum(1, dtype=f32))
    return (v / n[:, None]).astype(f32)


def census(name, o, d, C, r):
    r2 = (r * r).astype(f32)
    cand, disc = exact_rule(o, d, C, r2)
    ff = filter_disc(o, d, C, r)
    f64 = filter_disc(o, d, C, r, accumulate=np.float64)
    scale = (C.astype(np.float64) ** 2).sum(1) + r.astype(np.float64) ** 2 + (o.astype(np.float64) ** 2).sum(1)
    # reference value of the filter: exact discriminant + margin
    oc = C.astype(np.float64) - o.astype(np.float64)
    dd = d.astype(np.float64)
    true = (oc * dd).sum(1) ** 2 - (oc ** 2).sum(1) * (dd ** 2).sum(1) + r.astype(np.float64) ** 2
    err32 = np.abs(ff.astype(np.float64) - (true + EPS * scale)) / (EPS * scale)
    err64 = np.abs(f64 - (true + EPS * scale)) / (EPS * scale)
    fn = cand & ~(ff >= 0)
    print("%-34s pairs %8d  exact candidates %8d  filter candidates %8d  FALSE NEGATIVES %d   error/margin: f32-acc max %.3f  exact-acc max %.3f"
          % (name, len(r), cand.sum(), (ff >= 0).sum(), fn.sum(), err32.max(), err64.max()))
    return int(fn.sum())


def main():
    rng = np.random.default_rng(7)
    n = 2_000_000
    bad = 0
    # (1) book scene scale: centres in [-11, 11] x 0.2 x [-11, 11], r = 0.2; rays from the camera and from surface points
    C = np.stack([rng.uniform(-11, 11, n), np.full(n, 0.2), rng.uniform(-11, 11, n)], 1).astype(f32)
    r = np.full(n, 0.2, f32)
    o = np.stack([rng.uniform(-13, 13, n), rng.uniform(0, 3, n), rng.uniform(-13, 13, n)], 1).astype(f32)
    d = unit(rng.normal(size=(n, 3)))
    bad += census("book scale, random rays", o, d, C, r)
    # (2) grazing: aim at a point at distance r (1 + delta) from the centre, delta in +-1e-3 .. 1e-7
    for mag in (1e-3, 1e-5, 1e-6, 1e-7, 0.0):
        t = unit(rng.normal(size=(n, 3))).astype(np.float64)
        delta = rng.uniform(-mag, mag, n)
        p = C.astype(np.float64) + t * (r.astype(np.float64) * (1 + delta))[:, None]
        v = p - o.astype(np.float64)
        # direction perpendicular to t through p: remove the component along t
        w = rng.normal(size=(n, 3))
        w = w - (w * t).sum(1, keepdims=True) * t
        og = (p - w / np.linalg.norm(w, axis=1, keepdims=True) * rng.uniform(0.5, 30, n)[:, None]).astype(f32)
        dg = unit(p - og.astype(np.float64))
        bad += census("book scale, grazing +-%g" % mag, og, dg, C, r)
    # (3) the ground sphere (C = (0,-1000,0), r = 1000) from points on and above it
    Cg = np.tile(np.array([[0, -1000, 0]], f32), (n, 1))
    rg = np.full(n, 1000, f32)
    og = np.stack([rng.uniform(-15, 15, n), rng.choice([0.0, 1e-4, 0.2, 2.0], n) * rng.uniform(0, 1, n), rng.uniform(-15, 15, n)], 1).astype(f32)
    dg = unit(np.stack([rng.normal(size=n), rng.normal(size=n) * rng.choice([1.0, 1e-2, 1e-4], n), rng.normal(size=n)], 1))
    bad += census("ground sphere r=1000", og, dg, Cg, rg)
    # (4) stress scale (config 4): centres in [-50,50] x [0.2,20] x [-100,0], r in [0.05,0.4]
    C4 = np.stack([rng.uniform(-50, 50, n), rng.uniform(0.2, 20, n), rng.uniform(-100, 0, n)], 1).astype(f32)
    r4 = rng.uniform(0.05, 0.4, n).astype(f32)
    t = unit(rng.normal(size=(n, 3))).astype(np.float64)
    p = C4.astype(np.float64) + t * (r4.astype(np.float64) * (1 + rng.uniform(-1e-5, 1e-5, n)))[:, None]
    w = rng.normal(size=(n, 3)); w = w - (w * t).sum(1, keepdims=True) * t
    o4 = (p - w / np.linalg.norm(w, axis=1, keepdims=True) * rng.uniform(0.5, 120, n)[:, None]).astype(f32)
    d4 = unit(p - o4.astype(np.float64))
    bad += census("stress scale, grazing +-1e-5", o4, d4, C4, r4)
    # (5) tiny and huge coordinates
    for s in (1e-3, 1e3):
        bad += census("book scale x %g, grazing" % s, (og * 0 + o * f32(s)).astype(f32), d, (C * f32(s)).astype(f32), (r * f32(s)).astype(f32))
    print("total false negatives:", bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
