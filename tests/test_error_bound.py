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


def test_walk_prefilter_never_rejects_an_acceptable_sphere():
    """The square-root-free pre-filter of the grid walk (rt_kernels.hip, walk_lanes / walk_pool / walk_pool_dense) may only reject a
    sphere whose float roots sphere::hit (sphere.h:24-43) would reject too.  Emulated in float32 numpy on cases built to sit ON the
    decision boundaries: best hit within +-1e-3 (relative) of the near root, far root within +-1e-3 of t_min."""
    rng = np.random.default_rng(5)
    n = 1_500_000
    kap = F(1e-4)
    a = (10.0 ** rng.uniform(-6, 6, n)).astype(F)
    S = (10.0 ** rng.uniform(-4, 3, n))                                           # sqrt(disc) wanted, in units of a
    mode = rng.integers(0, 3, n)
    eps = rng.choice([-1.0, 1.0], n) * 10.0 ** rng.uniform(-8, -3, n)
    a64 = a.astype(np.float64)
    # mode 0: near root ~ best_t (b < 0);  mode 1: far root ~ t_min;  mode 2: anything
    t_near = 10.0 ** rng.uniform(-2, 2, n)
    b = np.where(mode == 0, -(t_near * a64 + S * a64), np.where(mode == 1, S * a64 - 0.001 * a64 * (1.0 + eps), rng.normal(size=n) * S * a64 * 2)).astype(F)
    disc = ((S * a64) ** 2).astype(F)
    best_t = np.where(mode == 0, t_near * (1.0 + eps), 10.0 ** rng.uniform(-3, 3, n)).astype(F)
    best_t[rng.random(n) < 0.05] = np.finfo(F).max                                # no hit yet
    with np.errstate(over="ignore", invalid="ignore"):
        # the reference's float roots and what it would offer
        sq = np.sqrt(disc).astype(F)
        t1 = ((-b - sq).astype(F) / a).astype(F)
        t2 = ((-b + sq).astype(F) / a).astype(F)
        cand = np.where(t1 > F(0.001), t1, np.where(t2 > F(0.001), t2, np.inf))
        acceptable = cand <= best_t                                               # "<" accepts, "==" must still be seen (tie detection)
        # the filter, as in the kernel (the fmaf is emulated in float64 and perturbed by an ulp either way below)
        abt0 = ((a * best_t).astype(F).astype(np.float64) * float(F(1.0001)) + (F(1e-6) * a).astype(F).astype(np.float64)).astype(F)
        atm = (a * F(F(0.001) * F(0.9999) - F(1e-6))).astype(F)
        # L = fma(|b|, -kap, -b) - abt, M = fma(|b|, -kap, b) + atm; they are never both positive, so the kernel tests the larger
        # one only: P = max(L, M), reject if P > 0 and P^2 > 1.0003 disc.  The older form (kb rounded on its own, two tests) is
        # checked alongside: it must decide alike up to the one rounding.
        b64 = b.astype(np.float64); ab64 = np.abs(b64) * float(kap)
        dk = (disc * F(1.0003)).astype(F)
        kb = (kap * np.abs(b)).astype(F)
        for ulp in (0, -1, 1):
            pre_l = (-b64 - ab64).astype(F); pre_m = (b64 - ab64).astype(F)
            if ulp: pre_l = np.nextafter(pre_l, F(ulp * np.inf)); pre_m = np.nextafter(pre_m, F(ulp * np.inf))
            M = (pre_m + atm).astype(F)
            for abt in (abt0, np.nextafter(abt0, F(0)), np.nextafter(abt0, F(np.inf))):
                L = (pre_l - abt).astype(F)
                assert not ((L > 0) & (M > 0)).any()                                  # one test of the larger covers both
                P = np.maximum(L, M)
                reject = (disc > 0) & (P > 0) & ((P * P).astype(F) > dk)
                assert not (reject & acceptable).any()
        M0 = ((b - kb).astype(F) + atm).astype(F); L0 = ((-b - kb).astype(F) - abt0).astype(F)
        reject0 = (disc > 0) & (((L0 > 0) & ((L0 * L0).astype(F) > dk)) | ((M0 > 0) & ((M0 * M0).astype(F) > dk)))
        assert not (reject0 & acceptable).any()
    # the filter does something: most unacceptable boundary cases further than its margin are rejected
    assert (reject & ~acceptable).sum() > 0.2 * (~acceptable).sum()


def test_brick_margin_puts_t_inside_the_float_slab_interval():
    """Brick eligibility (rt_kernels.hip in_brick): if the kernel's hit point, in level-3 cell coordinates,
    lies inside the brick bounds (margin 8e-4 cells in x/z, 8.5e-3 in y ~ 0.002 in distance), the hit's t lies inside the float
    slab interval [fl(fl(X0-o)/d), fl(fl(X1-o)/d)] of intersect_ray_aabb (acceleration_structure.h:226-244) for that axis.
    Adversarial: hit points placed within +-1e-4 of the margin line, origins in the near zone, |d| over five decades."""
    rng = np.random.default_rng(21)
    n = 2_000_000
    for cell, org, scale, margin in ((2.75, -11.0, F(1.0 / 2.75), 8e-4), (0.25, 0.0, F(4.0), 8.5e-3)):
        i0 = rng.integers(0, 8, n); i1 = np.minimum(7, i0 + rng.integers(0, 3, n))
        X0 = (org + cell * i0).astype(F); X1 = (org + cell * (i1 + 1)).astype(F)          # exact in float32
        o = rng.uniform(-25, 25, n).astype(F)
        d = (rng.choice([-1.0, 1.0], n) * 10.0 ** rng.uniform(-3, 2, n)).astype(F)
        # a hit point close to the margin line inside the lower or the upper face
        side = rng.integers(0, 2, n)
        target = np.where(side == 0, X0.astype(np.float64) + cell * margin, X1.astype(np.float64) - cell * margin) + rng.uniform(-1e-4, 1e-4, n)
        t = ((target - o.astype(np.float64)) / d.astype(np.float64)).astype(F)
        keep = t > F(0.001)
        # kernel side: P = fma(t, d, o), u = fma(P, scale, 4) (x/z) or P * 4 (y)
        P = (t.astype(np.float64) * d.astype(np.float64) + o.astype(np.float64)).astype(F)
        u = (P.astype(np.float64) * float(scale) + (4.0 if cell == 2.75 else 0.0)).astype(F)
        lo = (i0 + margin).astype(F); hi = (i1 + 1 - margin).astype(F)
        inside = keep & (u > lo) & (u < hi)
        # reference side
        ta = ((X0 - o).astype(F) / d).astype(F); tb = ((X1 - o).astype(F) / d).astype(F)
        tlo, thi = np.minimum(ta, tb), np.maximum(ta, tb)
        assert inside.sum() > 0.15 * n
        assert ((t >= tlo) & (t <= thi))[inside].all()
        # ... and the margin is not vacuous: without it, points this close to the face do fall outside sometimes
        bare = keep & (u > i0.astype(F)) & (u < (i1 + 1).astype(F))
        near = np.abs(np.where(side == 0, P.astype(np.float64) - X0, X1 - P.astype(np.float64))) < 1e-5
        assert bare.sum() >= inside.sum() and near.sum() == 0          # (the cases above keep >= 0.002 - 1e-4 from the faces)


def test_away_from_an_outer_sphere_no_float_root_passes():
    """closest_tree's ground shortcut (rt_kernels.hip): b > 0 and c > 0 in the reference's float operations ==> fl(sqrt(disc)) <= b,
    so t1 = (-b - sq)/a and t2 = (-b + sq)/a are both <= 0 < 0.001 — sphere::hit (sphere.h:17-46) cannot accept a root."""
    rng = np.random.default_rng(5)
    n = 2_000_000
    # the ground sphere of create_world and small spheres; origins just outside the surface (c tiny), directions from
    # grazing to radial, |d| over five decades (a*c down to underflow against b*b)
    big = rng.random(n) < 0.5
    c3 = np.where(big[:, None], np.array([0.0, -1000.0, -1.0]), rng.uniform(-10, 10, (n, 3)))
    rad = np.where(big, 1000.0, rng.choice([0.1, 0.2, 1.0], n))
    u = rng.normal(size=(n, 3)); u /= np.linalg.norm(u, axis=1)[:, None]
    o = c3 + u * (rad * (1.0 + 10.0 ** rng.uniform(-8, -1, n)))[:, None]
    tang = np.cross(u, rng.normal(size=(n, 3))); tang /= np.linalg.norm(tang, axis=1)[:, None]
    d = (tang + u * (10.0 ** rng.uniform(-8, 0, n))[:, None] * rng.choice([-1.0, 1.0], n)[:, None]) * (10.0 ** rng.uniform(-3, 2, n))[:, None]
    o32, d32, c32 = o.astype(F), d.astype(F), c3.astype(F)
    r2 = (rad.astype(F) * rad.astype(F)).astype(F)
    oc = [(o32[:, k] - c32[:, k]).astype(F) for k in range(3)]
    dd = [d32[:, k] for k in range(3)]
    disc, a, b = disc32([o32[:, k] for k in range(3)], dd, [c32[:, k] for k in range(3)], r2)
    cc = (((oc[0] * oc[0]).astype(F) + (oc[1] * oc[1]).astype(F)).astype(F) + (oc[2] * oc[2]).astype(F)).astype(F) - r2
    skip = (b > 0) & (cc > 0) & (disc > 0)
    assert skip.sum() > 200_000
    sq = np.sqrt(disc[skip]).astype(F)                               # IEEE sqrt, correctly rounded
    t1 = ((-b[skip] - sq).astype(F) / a[skip]).astype(F)
    t2 = ((-b[skip] + sq).astype(F) / a[skip]).astype(F)
    assert (sq <= b[skip]).all()
    assert not (t1 > F(0.001)).any() and not (t2 > F(0.001)).any()
