"""Adversarial numerical check (not gpu) of the two float32 facts the culling grid rests on (rt_accel.h):

  (1) discriminant > 0 in the reference's operation order  ==>  dist(centre, line)^2 < r^2 + K2      (K2 for the near zone)
  (2) the float hit point o + t*d of an accepted root lies within R*(1 + 1e-4) of the centre, R^2 = r^2 + K2

numpy float32 arithmetic is IEEE binary32, one rounding per operation — the same arithmetic as the kernels (no FMA).
Rays and spheres are drawn from the near zone, with the sphere centre placed at a controlled distance from the ray's
line: just outside the inflated radius for (1), grazing / random for (2)."""
import numpy as np

F = np.float32
U = 2.0 ** -24
ZONE, CENTRE_BOUND = 24.0, 17.5
K2 = 2.0 * 16.1 * U * (ZONE + CENTRE_BOUND) ** 2              # accel_K2() of csrc/rt_accel.h


def disc32(o, d, c, r2):
    oc = [(o[k] - c[k]).astype(F) for k in range(3)]
    a = ((d[0] * d[0]).astype(F) + (d[1] * d[1]).astype(F)).astype(F) + (d[2] * d[2]).astype(F)
    b = ((oc[0] * d[0]).astype(F) + (oc[1] * d[1]).astype(F)).astype(F) + (oc[2] * d[2]).astype(F)
    cc = (((oc[0] * oc[0]).astype(F) + (oc[1] * oc[1]).astype(F)).astype(F) + (oc[2] * oc[2]).astype(F)).astype(F) - r2
    return ((b * b).astype(F) - (a.astype(F) * cc.astype(F)).astype(F)).astype(F), a.astype(F), b.astype(F)


def scenes(n, rng, dist_of_r):
    """rays with origin in the near zone, spheres with centre within CENTRE_BOUND of (0,1,0), at distance dist_of_r(r, |o-c|) from the line"""
    o = rng.normal(size=(n, 3)); o *= (rng.uniform(0, ZONE, n) / np.linalg.norm(o, axis=1))[:, None]; o[:, 1] += 1.0
    d = rng.normal(size=(n, 3)) * (10.0 ** rng.uniform(-3, 2, n))[:, None]
    r = rng.choice([0.03, 0.1, 0.2, 0.35], n)
    # a point on the line, then step perpendicular to the line by the wanted distance
    t = rng.uniform(-1, 3, n) * 10.0 / np.linalg.norm(d, axis=1)
    foot = o + t[:, None] * d
    perp = np.cross(d, rng.normal(size=(n, 3))); perp /= np.linalg.norm(perp, axis=1)[:, None]
    dist = dist_of_r(r, np.linalg.norm(foot - o, axis=1) + 1.0)
    c = foot + perp * dist[:, None]
    keep = np.linalg.norm(c - np.array([0, 1, 0]), axis=1) <= CENTRE_BOUND
    o, d, c, r = o[keep], d[keep], c[keep], r[keep]
    o32, d32, c32 = o.astype(F), d.astype(F), c.astype(F)
    r2 = (r.astype(F) * r.astype(F)).astype(F)
    return o32, d32, c32, r2


def exact_dist2(o32, d32, c32):
    o, d, c = o32.astype(np.float64), d32.astype(np.float64), c32.astype(np.float64)
    oc = o - c
    return (oc * oc).sum(1) - (oc * d).sum(1) ** 2 / (d * d).sum(1)


def test_no_positive_discriminant_outside_the_inflated_radius():
    rng = np.random.default_rng(11)
    worst = 0.0
    for _ in range(6):
        # centres just outside sqrt(r^2 + K2): 0 .. 1 % beyond it, where rounding is most likely to flip the sign
        o, d, c, r2 = scenes(400_000, rng, lambda r, far: np.sqrt(r * r + K2) * (1.0 + rng.uniform(0, 1e-2, r.size)))
        ex = exact_dist2(o, d, c)
        outside = ex >= r2.astype(np.float64) + K2
        disc, _, _ = disc32([o[:, k] for k in range(3)], [d[:, k] for k in range(3)], [c[:, k] for k in range(3)], r2)
        assert outside.sum() > 100_000
        assert not (disc[outside] > 0).any()
        # how much of the budget does float error actually use?  (positive disc just inside the bound)
        inside_pos = (disc > 0) & (ex > r2.astype(np.float64))
        if inside_pos.any():
            worst = max(worst, float(((ex - r2.astype(np.float64))[inside_pos]).max()))
    assert worst < K2 / 2          # the factor-2 safety in K2 is not consumed


def test_float_hit_point_stays_within_the_inflated_ball():
    rng = np.random.default_rng(12)
    o, d, c, r2 = scenes(1_000_000, rng, lambda r, far: r * rng.uniform(0.0, 1.05, r.size))     # hits, grazing hits, near misses
    disc, a, b = disc32([o[:, k] for k in range(3)], [d[:, k] for k in range(3)], [c[:, k] for k in range(3)], r2)
    pos = disc > 0
    sq = np.sqrt(disc[pos]).astype(F)
    R = np.sqrt(r2[pos].astype(np.float64) + K2)
    for sign in (-1.0, 1.0):
        t = ((-b[pos] + F(sign) * sq).astype(F) / a[pos]).astype(F)
        p = o[pos].astype(np.float64) + t.astype(np.float64)[:, None] * d[pos].astype(np.float64)
        assert (np.linalg.norm(p - c[pos].astype(np.float64), axis=1) <= R * (1 + 1e-4)).all()
