// rt_oracle.hpp — CPU ORACLE (test infrastructure, NOT product code).
//
// A from-scratch CPU restatement of the reference's render() hot path
// (MuellerNico/DD2360-RayTracing), used ONLY by tests/, __graft_entry__.smoke()
// and bench.py's cpu_baseline leg as the checker / baseline.  Nothing in the
// product path (dd2360-raytracing_amd/) may include, link or call this.
//
// PARITY STATUS: "parity unpinned".  The reference ships no tests, golden
// images or known-answer vectors, and it cannot be compiled in this image
// (needs nvcc, <curand_kernel.h>, <cuda_fp16.h>; writing stand-ins for those is
// not allowed).  The restatement is therefore pinned only by
//   (i)  reading the reference sources (every function cites file:line), and
//   (ii) the probe values recorded in SURVEY.md §8c / App. A (XORWOW KATs, world
//        counts, octree node/leaf counts, camera half_height bits, C1 PPM md5),
//        which tests/test_oracle_pins.py checks.
//
// Numeric contract restated here: IEEE binary32 per-op
// rounding, NO fused multiply-add, left-to-right evaluation of RNG draws that
// sit in argument lists, XORWOW as in cuRAND with subsequence 0 / offset 0.
// USE_FP16 mode (precision_types.h:8): real_t is binary16, every real_t
// operator = float op followed by one rounding to binary16 (the reference's
// host branches, precision_types.h:35-37 etc.).
//
// Every expression that mixes float and real_t in the reference is written out
// with explicit conversions so the template reads the same for float and h16.
#pragma once
#include <cstdint>
#include <cstring>
#include <cmath>
#include <cfloat>
#include <vector>
#include <cstdio>

namespace orc {

// ---------------------------------------------------------------- binary16
static inline uint32_t f32_bits(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }
static inline float bits_f32(uint32_t u) { float f; std::memcpy(&f, &u, 4); return f; }

// float -> half, round-to-nearest-even, IEEE (subnormals, inf, nan kept).
static inline uint16_t f32_to_f16(float f) {
    uint32_t x = f32_bits(f);
    const uint32_t sign = (x >> 16) & 0x8000u;
    x &= 0x7fffffffu;
    if (x >= 0x7f800000u) {                       // inf / nan
        if (x == 0x7f800000u) return (uint16_t)(sign | 0x7c00u);
        return (uint16_t)(sign | 0x7e00u | ((x >> 13) & 0x3ffu));
    }
    if (x >= 0x477ff000u) return (uint16_t)(sign | 0x7c00u);   // >= 65520 -> inf
    if (x < 0x38800000u) {                        // < 2^-14 : half subnormal / zero
        // value * 2^24 rounded to integer (RNE) via the float adder
        const float a = bits_f32(x);
        const float magic = bits_f32((127u - 14u + 23u - 10u) << 23);   // 2^-1 * 2^... see below
        // a + magic aligns the 10 subnormal mantissa bits at the bottom of the float mantissa
        // magic = 2^(-14+23-10) = 2^-1
        const float r = a + magic;
        return (uint16_t)(sign | (f32_bits(r) - f32_bits(magic)));
    }
    // normal range
    const uint32_t mant_odd = (x >> 13) & 1u;
    x += ((uint32_t)(15 - 127) << 23) + 0xfffu;   // rebias exponent, rounding bias part 1
    x += mant_odd;                                // rounding bias part 2 (ties to even)
    return (uint16_t)(sign | (x >> 13));
}

static inline float f16_to_f32(uint16_t h) {
    const uint32_t sign = ((uint32_t)h & 0x8000u) << 16;
    const uint32_t em = h & 0x7fffu;
    uint32_t out;
    if (em >= 0x7c00u) out = 0x7f800000u | ((em & 0x3ffu) << 13);          // inf / nan
    else if (em >= 0x0400u) out = (em << 13) + ((uint32_t)(127 - 15) << 23); // normal
    else {                                                                    // subnormal / zero
        const float v = (float)em * bits_f32((127u - 24u) << 23);             // em * 2^-24 (exact)
        out = f32_bits(v);
    }
    return bits_f32(out | sign);
}

// real_t of USE_FP16 (precision_types.h:16-160): storage binary16, each operator
// computes in float and rounds once.
struct h16 {
    uint16_t b;
    h16() : b(0) {}
    explicit h16(float f) : b(f32_to_f16(f)) {}
};
static inline float to_f(h16 x) { return f16_to_f32(x.b); }
static inline float to_f(float x) { return x; }
static inline h16 operator+(h16 a, h16 c) { return h16(to_f(a) + to_f(c)); }   // precision_types.h:31
static inline h16 operator-(h16 a, h16 c) { return h16(to_f(a) - to_f(c)); }   // :40
static inline h16 operator*(h16 a, h16 c) { return h16(to_f(a) * to_f(c)); }   // :49
static inline h16 operator/(h16 a, h16 c) { return h16(to_f(a) / to_f(c)); }   // :58
static inline bool operator<(h16 a, h16 c) { return to_f(a) < to_f(c); }       // :109
static inline bool operator>(h16 a, h16 c) { return to_f(a) > to_f(c); }       // :118
static inline bool operator<=(h16 a, h16 c) { return to_f(a) <= to_f(c); }     // :127
static inline bool operator>=(h16 a, h16 c) { return to_f(a) >= to_f(c); }     // :136

template <class R> inline R from_f(float f);
template <> inline float from_f<float>(float f) { return f; }
template <> inline h16 from_f<h16>(float f) { return h16(f); }                 // real_t(float) :22
template <class R> static inline R from_d(double d) { return from_f<R>((float)d); }   // real_t(double) :23 (double->float->half)
template <class R> static inline R from_i(int i) { return from_f<R>((float)i); }      // real_t(int) :24
// "-x" on a real_t goes through operator float (no unary minus in real_t): exact.
static inline float neg(float x) { return -x; }
static inline h16 neg(h16 x) { h16 r; r.b = (uint16_t)(x.b ^ 0x8000u); return r; }
// sqrt(real_t) resolves to sqrtf(float(x)), result converted back on assignment.
template <class R> static inline R sqrt_r(R x) { return from_f<R>(std::sqrt(to_f(x))); }

// ---------------------------------------------------------------- XORWOW
// cuRAND XORWOW, curand_init(seed, 0, 0) only (main.cu:80, :93).  SURVEY.md App. A.1.
struct Xorwow {                    // 48-byte curandStateXORWOW layout
    uint32_t d, v[5];
    int32_t boxmuller_flag, boxmuller_flag_double;
    float boxmuller_extra;
    uint32_t pad_;
    double boxmuller_extra_double;
};
static_assert(sizeof(Xorwow) == 48, "curandState is 48 bytes");

static inline void xorwow_init(Xorwow& s, uint64_t seed) {
    const uint32_t s0 = (uint32_t)seed ^ 0xaad26b49u;
    const uint32_t s1 = (uint32_t)(seed >> 32) ^ 0xf7dcefddu;
    const uint32_t t0 = 1099087573u * s0;
    const uint32_t t1 = 2591861531u * s1;
    s.d = 6615241u + t1 + t0;
    s.v[0] = 123456789u + t0;
    s.v[1] = 362436069u ^ t0;
    s.v[2] = 521288629u + t1;
    s.v[3] = 88675123u ^ t1;
    s.v[4] = 5783321u + t0;
    s.boxmuller_flag = 0; s.boxmuller_flag_double = 0;
    s.boxmuller_extra = 0.f; s.pad_ = 0; s.boxmuller_extra_double = 0.0;
}
static inline uint32_t xorwow_next(Xorwow& s) {
    const uint32_t t = s.v[0] ^ (s.v[0] >> 2);
    s.v[0] = s.v[1]; s.v[1] = s.v[2]; s.v[2] = s.v[3]; s.v[3] = s.v[4];
    s.v[4] = (s.v[4] ^ (s.v[4] << 4)) ^ (t ^ (t << 1));
    s.d += 362437u;
    return s.v[4] + s.d;
}
// curand_uniform: x * 2^-32 + 2^-33 with float rounding after each op; in (0,1].
static inline float uniform(Xorwow& s) {
    const float x = (float)xorwow_next(s);
    const float m = x * 2.3283064e-10f;
    return m + (2.3283064e-10f / 2.0f);
}

// draw counter (diagnostics for world generation; SURVEY §8c pins)
struct Counters { uint64_t rays = 0, sphere_tests = 0, slab_tests = 0, bucket_visits = 0, draws = 0, samples = 0; };

// ---------------------------------------------------------------- vec3 / ray  (vec3.h, ray.h)
template <class R> struct V3 { R e[3]; };
template <class R> static inline V3<R> mk(R a, R b, R c) { V3<R> v; v.e[0] = a; v.e[1] = b; v.e[2] = c; return v; }
template <class R> static inline V3<R> add(const V3<R>& a, const V3<R>& b) { return mk<R>(a.e[0] + b.e[0], a.e[1] + b.e[1], a.e[2] + b.e[2]); }   // vec3.h:63
template <class R> static inline V3<R> sub(const V3<R>& a, const V3<R>& b) { return mk<R>(a.e[0] - b.e[0], a.e[1] - b.e[1], a.e[2] - b.e[2]); }   // :67
template <class R> static inline V3<R> mulv(const V3<R>& a, const V3<R>& b) { return mk<R>(a.e[0] * b.e[0], a.e[1] * b.e[1], a.e[2] * b.e[2]); }  // :71
template <class R> static inline V3<R> scale(R t, const V3<R>& v) { return mk<R>(t * v.e[0], t * v.e[1], t * v.e[2]); }                            // :79 and :87 (both t*v.e[i])
template <class R> static inline V3<R> divs(const V3<R>& v, R t) { return mk<R>(v.e[0] / t, v.e[1] / t, v.e[2] / t); }                             // :83
template <class R> static inline R dot(const V3<R>& a, const V3<R>& b) { return a.e[0] * b.e[0] + a.e[1] * b.e[1] + a.e[2] * b.e[2]; }             // :91
template <class R> static inline V3<R> cross(const V3<R>& a, const V3<R>& b) {                                                                      // :95
    return mk<R>(a.e[1] * b.e[2] - a.e[2] * b.e[1], neg(a.e[0] * b.e[2] - a.e[2] * b.e[0]), a.e[0] * b.e[1] - a.e[1] * b.e[0]);
}
template <class R> static inline V3<R> negv(const V3<R>& a) { return mk<R>(neg(a.e[0]), neg(a.e[1]), neg(a.e[2])); }                               // :23
template <class R> static inline R sqlen(const V3<R>& a) { return a.e[0] * a.e[0] + a.e[1] * a.e[1] + a.e[2] * a.e[2]; }                            // :35
template <class R> static inline R length(const V3<R>& a) { return sqrt_r<R>(sqlen(a)); }                                                  // :34
template <class R> static inline V3<R> unit(const V3<R>& a) { return divs(a, length(a)); }                                                         // :146
// vec3::operator/=(real_t): k = 1.0/t in DOUBLE, rounded to real_t, then 3 multiplies (vec3.h:137-144)
template <class R> static inline V3<R> div_assign(const V3<R>& a, R t) { const R k = from_d<R>(1.0 / (double)to_f(t)); return mk<R>(a.e[0] * k, a.e[1] * k, a.e[2] * k); }

template <class R> struct Ray { V3<R> A, B; };
template <class R> static inline V3<R> point_at(const Ray<R>& r, R t) { return add(r.A, scale(t, r.B)); }                                           // ray.h:13

// ---------------------------------------------------------------- scene
enum MatKind : int32_t { MAT_NONE = -1, MAT_LAMBERTIAN = 0, MAT_METAL = 1, MAT_DIELECTRIC = 2 };

template <class R> struct Sphere {          // sphere.h:7-15 + the material it points to (material.h:52-116)
    V3<R> center; R radius;
    int32_t kind;                           // MAT_NONE = never-initialised "ghost" slot: skipped everywhere
    V3<R> albedo; R param;                  // param: metal fuzz (clamped) or dielectric ref_idx
};

template <class R> struct Camera {          // camera.h:51-56
    V3<R> origin, lower_left_corner, horizontal, vertical, u, v, w; R lens_radius;
};

template <class R> struct Hit { R t; V3<R> p, normal; int32_t sphere; };   // hitable.h:9-15 (mat_ptr -> sphere index)

// pow((1-cos),5) of material.h:14.  Neither CUDA powf nor glibc powf is pinned by the
// reference; the contract is x^5 in binary64 ((x*x)*(x*x))*x rounded once to binary32.
static inline float pow5(float x) { const double d = (double)x; const double d2 = d * d; return (float)((d2 * d2) * d); }

// ---------------------------------------------------------------- sphere::hit  (sphere.h:17-46)
// Returns candidate t for (t_min, t_max); identical op order to the reference.
template <class R> static inline bool sphere_hit(const Sphere<R>& s, const Ray<R>& r, R t_min, R t_max, Hit<R>& rec, int idx) {
    const V3<R> oc = sub(r.A, s.center);
    const R a = dot(r.B, r.B);
    const R b = dot(oc, r.B);
    const R c = dot(oc, oc) - s.radius * s.radius;
    const R disc = b * b - a * c;
    if (disc > from_i<R>(0)) {
        // fp32: (-b - sqrt(disc))/a.  fp16: real_t::sqrt(disc) rounds to half, but "-b - h" and "/a" are
        // builtin FLOAT ops (left operand is float after unary minus), rounded once on assignment (sphere.h:24-28).
        const float nb = -to_f(b);
        R temp = from_f<R>((nb - to_f(sqrt_r<R>(disc))) / to_f(a));
        if (temp < t_max && temp > t_min) {
            rec.t = temp; rec.p = point_at(r, rec.t); rec.normal = divs(sub(rec.p, s.center), s.radius); rec.sphere = idx;
            return true;
        }
        // far root: sqrt(discriminant) is the FLOAT sqrt of float(disc), not rounded to half (sphere.h:36)
        temp = from_f<R>((nb + std::sqrt(to_f(disc))) / to_f(a));
        if (temp < t_max && temp > t_min) {
            rec.t = temp; rec.p = point_at(r, rec.t); rec.normal = divs(sub(rec.p, s.center), s.radius); rec.sphere = idx;
            return true;
        }
    }
    return false;
}

// ---------------------------------------------------------------- octree (acceleration_structure.h)
template <class R> struct AABB { R lo[3], hi[3]; };                           // :23-26 (x_low,y_low,z_low,x_high,y_high,z_high)
template <class R> struct OctNode { int32_t level; AABB<R> box; int32_t children[8]; };   // :35-39
struct OctLeaf { std::vector<int32_t> idx; };                                 // :46-49 (capacity SPHERES_PER_LEAF)
template <class R> struct Octree {                                            // :57-62
    std::vector<OctNode<R>> nodes;      // capacity 585
    std::vector<OctLeaf> leaves;        // leaves[0] unused
    int32_t nodeCount = 0, leafCount = 1;
    int32_t spl = 30;
    int32_t dropped_full = 0, dropped_outside = 0;   // the two printf paths (:106, :135)
};

template <class R> static inline bool box_intersects(const Sphere<R>& s, AABB<R> bx) {   // :82-93
    for (int k = 0; k < 3; ++k) { bx.lo[k] = bx.lo[k] - s.radius; }
    for (int k = 0; k < 3; ++k) { bx.hi[k] = bx.hi[k] + s.radius; }
    return (s.center.e[0] > bx.lo[0] && s.center.e[0] <= bx.hi[0])
        && (s.center.e[1] >= bx.lo[1] && s.center.e[1] <= bx.hi[1])
        && (s.center.e[2] >= bx.lo[2] && s.center.e[2] <= bx.hi[2]);
}

template <class R> static int octree_insert(Octree<R>& T, int node_i, const Sphere<R>& s, int sidx) {   // :104-186
    if (!box_intersects(s, T.nodes[node_i].box)) { T.dropped_outside++; return 0; }
    if (T.nodes[node_i].level == 3) {
        for (int i = 0; i < 8; ++i) {
            int leaf = T.nodes[node_i].children[i];
            if (leaf == 0) { leaf = T.leafCount++; T.leaves.resize(T.leafCount); T.leaves[leaf].idx.clear(); T.nodes[node_i].children[i] = leaf; }
            if ((int)T.leaves[leaf].idx.size() < T.spl) { T.leaves[leaf].idx.push_back(sidx); return 1; }
        }
        T.dropped_full++;
        return 0;
    }
    int count = 0;
    // midpoints are computed in FLOAT (lambda takes floats) and converted back to real_t (:141-147)
    const AABB<R> pb = T.nodes[node_i].box;
    R half[3];
    for (int k = 0; k < 3; ++k) { const float lo = to_f(pb.lo[k]), hi = to_f(pb.hi[k]); half[k] = from_f<R>(lo + (hi - lo) / 2); }
    for (int i = 0; i < 8; ++i) {                       // child i = 4*(x high) + 2*(y high) + (z high)  (:149-165)
        AABB<R> cb;
        const int hx = (i >> 2) & 1, hy = (i >> 1) & 1, hz = i & 1;
        cb.lo[0] = hx ? half[0] : pb.lo[0]; cb.hi[0] = hx ? pb.hi[0] : half[0];
        cb.lo[1] = hy ? half[1] : pb.lo[1]; cb.hi[1] = hy ? pb.hi[1] : half[1];
        cb.lo[2] = hz ? half[2] : pb.lo[2]; cb.hi[2] = hz ? pb.hi[2] : half[2];
        if (box_intersects(s, cb)) {
            if (T.nodes[node_i].children[i] == 0) {
                const int ni = T.nodeCount++;
                T.nodes[node_i].children[i] = ni;
                OctNode<R> n; n.level = T.nodes[node_i].level + 1; n.box = cb; for (int k = 0; k < 8; ++k) n.children[k] = 0;
                T.nodes[ni] = n;
            }
            count += octree_insert(T, T.nodes[node_i].children[i], s, sidx);
        }
    }
    return count;
}

template <class R> static Octree<R> build_octree(const std::vector<Sphere<R>>& list, int spl) {     // :195-217
    Octree<R> T; T.spl = spl; T.nodes.resize(585); T.leaves.resize(1);
    for (auto& n : T.nodes) { n.level = 0; for (int k = 0; k < 3; ++k) { n.box.lo[k] = from_i<R>(0); n.box.hi[k] = from_i<R>(0); } for (int k = 0; k < 8; ++k) n.children[k] = 0; }
    OctNode<R>& root = T.nodes[0];
    root.level = 0;
    root.box.lo[0] = from_i<R>(-11); root.box.lo[1] = from_i<R>(0); root.box.lo[2] = from_i<R>(-11);
    root.box.hi[0] = from_i<R>(11);  root.box.hi[1] = from_i<R>(2); root.box.hi[2] = from_i<R>(11);
    T.nodeCount = 1;
    for (int i = 1; i < (int)list.size(); ++i) {        // ground sphere (0) stays out of the tree
        // ghost slots (kind MAT_NONE) are modelled as zero-filled spheres: they ARE inserted, exactly as the
        // reference would insert zeroed memory, but traversal never tests them.
        octree_insert(T, 0, list[i], i);
    }
    return T;
}

// intersect_ray_aabb (:226-244): arithmetic in real_t, results held in float.
template <class R> static inline bool ray_box(const Ray<R>& r, const AABB<R>& bx) {
    float tmin = to_f((bx.lo[0] - r.A.e[0]) / r.B.e[0]);
    float tmax = to_f((bx.hi[0] - r.A.e[0]) / r.B.e[0]);
    if (tmin > tmax) { const float t = tmin; tmin = tmax; tmax = t; }
    float tymin = to_f((bx.lo[1] - r.A.e[1]) / r.B.e[1]);
    float tymax = to_f((bx.hi[1] - r.A.e[1]) / r.B.e[1]);
    if (tymin > tymax) { const float t = tymin; tymin = tymax; tymax = t; }
    if ((tmin > tymax) || (tymin > tmax)) return false;
    if (tymin > tmin) tmin = tymin;
    if (tymax < tmax) tmax = tymax;
    float tzmin = to_f((bx.lo[2] - r.A.e[2]) / r.B.e[2]);
    float tzmax = to_f((bx.hi[2] - r.A.e[2]) / r.B.e[2]);
    if (tzmin > tzmax) { const float t = tzmin; tzmin = tzmax; tzmax = t; }
    if ((tmin > tzmax) || (tzmin > tmax)) return false;
    return true;
}

template <class R> struct World {
    std::vector<Sphere<R>> list;          // NUM_SPHERES slots (ghost slots have kind MAT_NONE)
    Camera<R> cam;
    Octree<R> tree;
    bool use_octree = false;
    int n_real = 0;
    uint64_t world_draws = 0;
    Xorwow world_rng_after;               // *rand_state = local_rand_state (main.cu:183)
};

template <class R> static void traverse(const World<R>& W, const Ray<R>& r, int node_i, bool& hit_any, Hit<R>& rec, R& closest, Counters* C) {   // :276-304
    const OctNode<R>& n = W.tree.nodes[node_i];
    if (C) C->slab_tests++;
    if (!ray_box(r, n.box)) return;
    if (n.level == 3) {
        for (int i = 0; i < 8; ++i) {
            const int leaf = n.children[i];
            if (leaf == 0) return;
            if (C) C->bucket_visits++;
            const OctLeaf& L = W.tree.leaves[leaf];
            for (size_t j = 0; j < L.idx.size(); ++j) {                   // processHit :254-265
                const int si = L.idx[j];
                if (si == 0) continue;
                if (W.list[si].kind == MAT_NONE) continue;   // ghost: unhittable by definition
                Hit<R> tmp;
                if (C) C->sphere_tests++;
                if (sphere_hit(W.list[si], r, from_f<R>(0.001f), closest, tmp, si)) { hit_any = true; closest = tmp.t; rec = tmp; }
            }
        }
        return;
    }
    for (int i = 0; i < 8; ++i) if (n.children[i] != 0) traverse(W, r, n.children[i], hit_any, rec, closest, C);
}

template <class R> static bool hit_tree(const World<R>& W, const Ray<R>& r, Hit<R>& rec, Counters* C) {    // :319-342
    Hit<R> g;
    if (C) C->sphere_tests++;
    const bool hit_ground = sphere_hit(W.list[0], r, from_f<R>(0.001f), from_f<R>(FLT_MAX), g, 0);
    bool hit_any = false; R closest = hit_ground ? g.t : from_f<R>(FLT_MAX);
    if (hit_ground) { hit_any = true; rec = g; }
    traverse(W, r, 0, hit_any, rec, closest, C);
    return hit_any;
}

template <class R> static bool hit_list(const World<R>& W, const Ray<R>& r, R t_min, R t_max, Hit<R>& rec, Counters* C) {   // hitable_list.h:16-31
    bool hit_any = false; R closest = t_max; Hit<R> tmp;
    for (int i = 0; i < (int)W.list.size(); ++i) {
        if (W.list[i].kind == MAT_NONE) continue;      // ghost slot
        if (C) C->sphere_tests++;
        if (sphere_hit(W.list[i], r, t_min, closest, tmp, i)) { hit_any = true; closest = tmp.t; rec = tmp; }
    }
    return hit_any;
}

template <class R> static inline bool closest_hit(const World<R>& W, const Ray<R>& r, Hit<R>& rec, Counters* C) {   // main.cu:51-55
    if (C) C->rays++;
    return W.use_octree ? hit_tree(W, r, rec, C) : hit_list(W, r, from_f<R>(0.001f), from_f<R>(FLT_MAX), rec, C);
}

// ---------------------------------------------------------------- materials (material.h)
template <class R> static inline V3<R> random_in_unit_sphere(Xorwow& s) {          // :35-41, draws x,y,z left-to-right
    V3<R> p;
    do {
        const float rx = uniform(s); const float ry = uniform(s); const float rz = uniform(s);
        const V3<R> rv = mk<R>(from_f<R>(rx), from_f<R>(ry), from_f<R>(rz));
        p = sub(scale(from_f<R>(2.0f), rv), mk<R>(from_i<R>(1), from_i<R>(1), from_i<R>(1)));
    } while (sqlen(p) >= from_f<R>(1.0f));
    return p;
}
template <class R> static inline V3<R> reflect(const V3<R>& v, const V3<R>& n) {    // :43-45  v - (2*dot(v,n))*n
    return sub(v, scale(from_f<R>(2.0f) * dot(v, n), n));
}
template <class R> static inline bool refract(const V3<R>& v, const V3<R>& n, R ni_over_nt, V3<R>& refracted) {   // :17-31
    const V3<R> uv = unit(v);
    const R dt = dot(uv, n);
    const R disc = from_f<R>(1.0f) - ni_over_nt * ni_over_nt * (from_f<R>(1.0f) - dt * dt);
    if (disc > from_i<R>(0)) {
        // fp32: sqrt(disc); fp16: real_t::sqrt(disc) -> both are "float sqrt then convert to real_t"
        refracted = sub(scale(ni_over_nt, sub(uv, scale(dt, n))), scale(sqrt_r<R>(disc), n));
        return true;
    }
    return false;
}
template <class R> static inline R schlick(R cosine, R ref_idx) {                   // :11-15
    R r0 = from_f<R>(1.0f - to_f(ref_idx)) / from_f<R>(1.0f + to_f(ref_idx));
    r0 = r0 * r0;
    return r0 + from_f<R>(1.0f - to_f(r0)) * from_f<R>(pow5(1.0f - to_f(cosine)));
}

// returns true if the ray continues; attenuation/scattered as in material::scatter (:49)
template <class R> static inline bool scatter(const Sphere<R>& m, const Ray<R>& r_in, const Hit<R>& rec, V3<R>& att, Ray<R>& sc, Xorwow& s) {
    if (m.kind == MAT_LAMBERTIAN) {                                                  // :55-60
        const V3<R> target = add(add(rec.p, rec.normal), random_in_unit_sphere<R>(s));
        sc.A = rec.p; sc.B = sub(target, rec.p); att = m.albedo; return true;
    }
    if (m.kind == MAT_METAL) {                                                       // :68-73
        const V3<R> refl = reflect(unit(r_in.B), rec.normal);
        sc.A = rec.p; sc.B = add(refl, scale(m.param, random_in_unit_sphere<R>(s)));
        att = m.albedo;
        return dot(sc.B, rec.normal) > from_f<R>(0.0f);
    }
    // dielectric :81-113
    V3<R> outward; const V3<R> reflected = reflect(r_in.B, rec.normal);
    R ni_over_nt; att = mk<R>(from_d<R>(1.0), from_d<R>(1.0), from_d<R>(1.0));
    V3<R> refracted = mk<R>(from_i<R>(0), from_i<R>(0), from_i<R>(0)); R reflect_prob; R cosine;
    const R ref_idx = m.param;
    if (dot(r_in.B, rec.normal) > from_f<R>(0.0f)) {
        outward = negv(rec.normal); ni_over_nt = ref_idx;
        cosine = dot(r_in.B, rec.normal) / length(r_in.B);
        cosine = sqrt_r<R>(from_f<R>(1.0f) - ref_idx * ref_idx * (from_f<R>(1.0f) - cosine * cosine));
    } else {
        outward = rec.normal; ni_over_nt = from_f<R>(1.0f) / ref_idx;
        // "-dot(..) / len": float negate, builtin float divide, one rounding on assignment
        cosine = from_f<R>(-to_f(dot(r_in.B, rec.normal)) / to_f(length(r_in.B)));
    }
    if (refract(r_in.B, outward, ni_over_nt, refracted)) reflect_prob = schlick(cosine, ref_idx);
    else reflect_prob = from_f<R>(1.0f);
    sc.A = rec.p;
    if (uniform(s) < to_f(reflect_prob)) sc.B = reflected; else sc.B = refracted;   // float < float (NaN -> refracted)
    return true;
}

// ---------------------------------------------------------------- camera (camera.h)
template <class R> static inline V3<R> random_in_unit_disk(Xorwow& s) {              // :12-18
    V3<R> p;
    do {
        const float rx = uniform(s); const float ry = uniform(s);
        p = sub(scale(from_f<R>(2.0f), mk<R>(from_f<R>(rx), from_f<R>(ry), from_i<R>(0))), mk<R>(from_i<R>(1), from_i<R>(1), from_i<R>(0)));
    } while (dot(p, p) >= from_f<R>(1.0f));
    return p;
}
template <class R> struct TanHalf;
// fp32: tan(arg) (tanf). Contract: correctly rounded tan via binary64 (matches SURVEY probe 0x3e8930a3).
template <> struct TanHalf<float> { static float f(float arg) { return (float)std::tan((double)arg); } };
// fp16 device branch (camera.h:29): real_t(hsin(arg)/hcos(arg)) -> half(sin), half(cos), half divide.
template <> struct TanHalf<h16> { static h16 f(h16 arg) { const h16 s((float)std::sin((double)to_f(arg))); const h16 c((float)std::cos((double)to_f(arg))); return s / c; } };

template <class R> static Camera<R> make_camera(V3<R> lookfrom, V3<R> lookat, V3<R> vup, R vfov, R aspect, R aperture, R focus) {   // :22-44
    Camera<R> c;
    c.lens_radius = aperture / from_f<R>(2.0f);
    const R theta = vfov * from_d<R>(3.14159265358979323846) / from_f<R>(180.0f);
    const R arg = theta / from_f<R>(2.0f);
    const R half_height = TanHalf<R>::f(arg);
    const R half_width = aspect * half_height;
    c.origin = lookfrom;
    c.w = unit(sub(lookfrom, lookat));
    c.u = unit(cross(vup, c.w));
    c.v = cross(c.w, c.u);
    c.lower_left_corner = sub(sub(sub(c.origin, scale(half_width * focus, c.u)), scale(half_height * focus, c.v)), scale(focus, c.w));
    c.horizontal = scale(from_f<R>(2.0f) * half_width * focus, c.u);
    c.vertical = scale(from_f<R>(2.0f) * half_height * focus, c.v);
    return c;
}
template <class R> static inline Ray<R> get_ray(const Camera<R>& c, R s, R t, Xorwow& st) {   // :45-49
    const V3<R> rd = scale(c.lens_radius, random_in_unit_disk<R>(st));
    const V3<R> offset = add(scale(rd.e[0], c.u), scale(rd.e[1], c.v));
    Ray<R> r;
    r.A = add(c.origin, offset);
    r.B = sub(sub(add(add(c.lower_left_corner, scale(s, c.horizontal)), scale(t, c.vertical)), c.origin), offset);
    return r;
}

// ---------------------------------------------------------------- create_world (main.cu:146-204)
template <class R> static World<R> create_world(int num_spheres, float sphere_radius, int nx, int ny, bool use_octree, int spl) {
    World<R> W; W.use_octree = use_octree;
    Xorwow rs; xorwow_init(rs, 1984);                      // rand_init main.cu:80
    uint64_t draws = 0;
    auto RND = [&]() { draws++; return uniform(rs); };
    Sphere<R> ghost; ghost.center = mk<R>(from_i<R>(0), from_i<R>(0), from_i<R>(0)); ghost.radius = from_i<R>(0); ghost.kind = MAT_NONE;
    ghost.albedo = ghost.center; ghost.param = from_i<R>(0);
    W.list.assign(num_spheres, ghost);
    auto set = [&](int i, V3<R> c, R rad, int kind, V3<R> alb, R prm) { Sphere<R> s; s.center = c; s.radius = rad; s.kind = kind; s.albedo = alb; s.param = prm; W.list[i] = s; };
    const V3<R> zero3 = mk<R>(from_i<R>(0), from_i<R>(0), from_i<R>(0));
    auto metal_fuzz = [&](R f) { return (f < from_f<R>(1.0f)) ? f : from_f<R>(1.0f); };   // material.h:67
    int i = 0;
    set(i++, mk<R>(from_i<R>(0), from_d<R>(-1000.0), from_i<R>(-1)), from_i<R>(1000), MAT_LAMBERTIAN, mk<R>(from_d<R>(0.5), from_d<R>(0.5), from_d<R>(0.5)), from_i<R>(0));
    if (num_spheres > 1) set(i++, mk<R>(from_i<R>(0), from_i<R>(1), from_i<R>(0)), from_d<R>(1.0), MAT_DIELECTRIC, zero3, from_d<R>(1.5));
    if (num_spheres > 2) set(i++, mk<R>(from_i<R>(-4), from_i<R>(1), from_i<R>(0)), from_d<R>(1.0), MAT_LAMBERTIAN, mk<R>(from_d<R>(0.4), from_d<R>(0.2), from_d<R>(0.1)), from_i<R>(0));
    if (num_spheres > 3) set(i++, mk<R>(from_i<R>(4), from_i<R>(1), from_i<R>(0)), from_d<R>(1.0), MAT_METAL, mk<R>(from_d<R>(0.7), from_d<R>(0.6), from_d<R>(0.5)), metal_fuzz(from_d<R>(0.0)));
    const int spheres_per_dim = (int)std::sqrt((float)num_spheres - 4);       // float sqrt, truncation (:160)
    const double spacing = 20. / spheres_per_dim;                             // :161
    for (double a = -10; a < 10; a += spacing) {
        for (double b = -10; b < 10 && i < num_spheres; b += spacing) {
            const R choose_mat = from_f<R>(RND());
            const float jx = RND();                                         // x draw before z draw (L->R, :166)
            const float jz = RND();
            const V3<R> center = mk<R>(from_d<R>(a + (double)jx), from_f<R>(sphere_radius), from_d<R>(b + (double)jz));
            if (choose_mat < from_f<R>(0.8f)) {
                const float r0 = RND(), r1 = RND(), r2 = RND(), r3 = RND(), r4 = RND(), r5 = RND();
                set(i++, center, from_f<R>(sphere_radius), MAT_LAMBERTIAN, mk<R>(from_f<R>(r0 * r1), from_f<R>(r2 * r3), from_f<R>(r4 * r5)), from_i<R>(0));
            } else if (choose_mat < from_f<R>(0.95f)) {
                const float r0 = RND(), r1 = RND(), r2 = RND(), r3 = RND();
                set(i++, center, from_f<R>(sphere_radius), MAT_METAL,
                    mk<R>(from_f<R>(0.5f * (1.0f + r0)), from_f<R>(0.5f * (1.0f + r1)), from_f<R>(0.5f * (1.0f + r2))), metal_fuzz(from_f<R>(0.5f * r3)));
            } else {
                set(i++, center, from_f<R>(sphere_radius), MAT_DIELECTRIC, zero3, from_d<R>(1.5));
            }
        }
    }
    W.n_real = i; W.world_draws = draws; W.world_rng_after = rs;
    const V3<R> lookfrom = mk<R>(from_i<R>(13), from_i<R>(2), from_i<R>(3));
    const V3<R> lookat = mk<R>(from_i<R>(0), from_i<R>(0), from_i<R>(0));
    W.cam = make_camera<R>(lookfrom, lookat, mk<R>(from_i<R>(0), from_i<R>(1), from_i<R>(0)), from_d<R>(30.0),
                           from_i<R>(nx) / from_i<R>(ny), from_d<R>(0.1), from_d<R>(10.0));
    if (use_octree) W.tree = build_octree(W.list, spl);      // main.cu:410
    return W;
}

// ---------------------------------------------------------------- color / render (main.cu:43-142)
template <class R> static V3<R> color(const World<R>& W, Ray<R> cur, Xorwow& s, Counters* C) {
    V3<R> att = mk<R>(from_d<R>(1.0), from_d<R>(1.0), from_d<R>(1.0));
    const V3<R> black = mk<R>(from_d<R>(0.0), from_d<R>(0.0), from_d<R>(0.0));
    for (int i = 0; i < 50; ++i) {
        Hit<R> rec;
        if (closest_hit(W, cur, rec, C)) {
            Ray<R> sc; V3<R> a;
            if (scatter(W.list[rec.sphere], cur, rec, a, sc, s)) { att = mulv(att, a); cur = sc; }
            else return black;
        } else {
            const V3<R> ud = unit(cur.B);
            const R t = from_f<R>(0.5f) * (ud.e[1] + from_f<R>(1.0f));
            // (1.0f - t) is a builtin float subtraction; "float * vec3" converts it to real_t (main.cu:70)
            const R omt = from_f<R>(1.0f - to_f(t));
            const V3<R> c = add(scale(omt, mk<R>(from_d<R>(1.0), from_d<R>(1.0), from_d<R>(1.0))), scale(t, mk<R>(from_d<R>(0.5), from_d<R>(0.7), from_d<R>(1.0))));
            return mulv(att, c);
        }
    }
    return black;
}

template <class R> static inline V3<R> sample_pixel(const World<R>& W, int i, int j, int max_x, int max_y, Xorwow& s, Counters* C) {   // main.cu:104-107
    const float du = uniform(s);
    const R u = from_f<R>((float)i + du) / from_i<R>(max_x);
    const float dv = uniform(s);
    const R v = from_f<R>((float)j + dv) / from_i<R>(max_y);
    const Ray<R> r = get_ray(W.cam, u, v, s);
    if (C) C->samples++;
    return color(W, r, s, C);
}

// render() for one pixel: ns samples, col/=ns, sqrt gamma (main.cu:96-117)
template <class R> static inline V3<R> render_pixel(const World<R>& W, int i, int j, int max_x, int max_y, int ns, Xorwow& s, Counters* C) {
    V3<R> col = mk<R>(from_i<R>(0), from_i<R>(0), from_i<R>(0));
    for (int k = 0; k < ns; ++k) col = add(col, sample_pixel(W, i, j, max_x, max_y, s, C));
    col = div_assign(col, from_i<R>(ns));
    for (int k = 0; k < 3; ++k) col.e[k] = sqrt_r<R>(col.e[k]);
    return col;
}

} // namespace orc
