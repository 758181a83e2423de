// rt_kernels_fp16.hip — the render path with real_t = binary16 (the reference's USE_FP16, precision_types.h:8).
//
// Same kernels as rt_kernels.hip (render_init is shared), written over rt::half_t: every real_t operator is one
// float operation followed by one rounding to binary16 (precision_types.h:31-143; on gfx950 the compiler may pick
// the native v_*_f16 instruction, which gives the same bits).  Expressions of the reference that mix float and
// real_t keep their C++ conversions: e.g. sphere::hit's `(-b - real_t::sqrt(disc))/a` is FLOAT arithmetic on the
// converted operands, rounded once on assignment (sphere.h:24-28), and `1.0f - t` in color() is a float subtraction
// (main.cu:70).  The tree is always walked with the reference scan here: the culling grid's error bounds are
// binary32 bounds.  Scene data arrive as floats holding exact binary16 images.
#include <hip/hip_runtime.h>
#include <float.h>
#include "rt_device.h"
#include "rt_real.h"
#include "rt_tuning.h"
#include "rt_divshared.h"

#pragma clang fp contract(off)

namespace rt {
namespace h16 {

#define RT_DEV static __device__ __forceinline__
typedef half_t R;

RT_DEV R rf(float f) { return half_t(f); }                 // real_t(float)
RT_DEV R rd(double d) { return half_t((float)d); }         // real_t(double): double -> float -> half
RT_DEV R ri(int i) { return half_t((float)i); }            // real_t(int)
RT_DEV float fl(R r) { return r.f(); }
RT_DEV R rsqrt_(R x) { return half_t(sqrtf(x.f())); }      // sqrt(real_t) / real_t::sqrt: float sqrt, converted back
RT_DEV R rneg(R x) { R r; r.bits = (uint16_t)(x.bits ^ 0x8000u); return r; }

struct Rng { uint32_t d, v0, v1, v2, v3, v4; };
RT_DEV float rng_uniform(Rng& s) {
    const uint32_t t = s.v0 ^ (s.v0 >> 2);
    s.v0 = s.v1; s.v1 = s.v2; s.v2 = s.v3; s.v3 = s.v4;
    s.v4 = (s.v4 ^ (s.v4 << 4)) ^ (t ^ (t << 1));
    s.d += 362437u;
    const float x = (float)(s.d + s.v4);
    const float m = x * 2.3283064e-10f;
    return m + (2.3283064e-10f / 2.0f);
}
RT_DEV float pow5(float x) { const double v = (double)x; const double v2 = v * v; return (float)((v2 * v2) * v); }

struct V { R x, y, z; };
RT_DEV V vadd(const V& a, const V& b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
RT_DEV V vsub(const V& a, const V& b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
RT_DEV V vmul(const V& a, const V& b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
RT_DEV V vscale(R t, const V& v) { return {t * v.x, t * v.y, t * v.z}; }
RT_DEV V vdiv(const V& v, R t) { return {v.x / t, v.y / t, v.z / t}; }
RT_DEV R vdot(const V& a, const V& b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
RT_DEV R vsqlen(const V& a) { return a.x * a.x + a.y * a.y + a.z * a.z; }
RT_DEV R vlen(const V& a) { return rsqrt_(vsqlen(a)); }
RT_DEV V vunit(const V& a) { return vdiv(a, vlen(a)); }
RT_DEV V vneg(const V& a) { return {rneg(a.x), rneg(a.y), rneg(a.z)}; }
RT_DEV V vload(const float* p) { return {rf(p[0]), rf(p[1]), rf(p[2])}; }
struct Ray { V o, d; };

// sphere::hit (sphere.h:17-46) for one candidate; `closest` is closest_so_far, `a` = dot(d,d) hoisted
RT_DEV void sphere_test(const Ray& r, R a, const float4 s, int id, R& closest, int& best) {
    const V c = {rf(s.x), rf(s.y), rf(s.z)};
    const V oc = vsub(r.o, c);
    const R b = vdot(oc, r.d);
    const R cc = vdot(oc, oc) - rf(s.w);                   // s.w = radius*radius, rounded to binary16 on the host
    const R disc = b * b - a * cc;
    if (disc > ri(0)) {
        const R tmin = rf(0.001f);
        const float nb = -fl(b);
        R t = rf((nb - fl(rsqrt_(disc))) / fl(a));         // float arithmetic on converted operands, one rounding
        if (t < closest && t > tmin) { closest = t; best = id; }
        else {
            t = rf((nb + sqrtf(fl(disc))) / fl(a));        // far root: float sqrt of float(disc), not rounded (sphere.h:36)
            if (t < closest && t > tmin) { closest = t; best = id; }
        }
    }
}
// the same for a bucket entry of a binary16 tree: four halves (cx, cy, cz, r^2) in 8 bytes (rt_api.hip, rt_octree_upload)
RT_DEV R hbits(uint32_t b) { R r; r.bits = (uint16_t)b; return r; }
RT_DEV void sphere_test(const Ray& r, R a, const uint2 pk, int id, R& closest, int& best) {
    const V c = {hbits(pk.x & 0xffffu), hbits(pk.x >> 16), hbits(pk.y & 0xffffu)};
    const V oc = vsub(r.o, c);
    const R b = vdot(oc, r.d);
    const R cc = vdot(oc, oc) - hbits(pk.y >> 16);                   // s.w = radius*radius, rounded to binary16 on the host
    const R disc = b * b - a * cc;
    if (disc > ri(0)) {
        const R tmin = rf(0.001f);
        const float nb = -fl(b);
        R t = rf((nb - fl(rsqrt_(disc))) / fl(a));         // float arithmetic on converted operands, one rounding
        if (t < closest && t > tmin) { closest = t; best = id; }
        else {
            t = rf((nb + sqrtf(fl(disc))) / fl(a));        // far root: float sqrt of float(disc), not rounded (sphere.h:36)
            if (t < closest && t > tmin) { closest = t; best = id; }
        }
    }
}

// intersect_ray_aabb (acceleration_structure.h:226-244): real_t arithmetic, float results
RT_DEV bool ray_box(const Ray& r, const float4 n0, const float4 n1) {
    float tmin = fl((rf(n0.x) - r.o.x) / r.d.x);
    float tmax = fl((rf(n0.w) - r.o.x) / r.d.x);
    if (tmin > tmax) { const float t = tmin; tmin = tmax; tmax = t; }
    float tymin = fl((rf(n0.y) - r.o.y) / r.d.y);
    float tymax = fl((rf(n1.x) - r.o.y) / r.d.y);
    if (tymin > tymax) { const float t = tymin; tymin = tymax; tymax = t; }
    if ((tmin > tymax) || (tymin > tmax)) return false;
    if (tymin > tmin) tmin = tymin;
    if (tymax < tmax) tmax = tymax;
    float tzmin = fl((rf(n0.z) - r.o.z) / r.d.z);
    float tzmax = fl((rf(n1.y) - r.o.z) / r.d.z);
    if (tzmin > tzmax) { const float t = tzmin; tzmin = tzmax; tzmax = t; }
    if ((tmin > tzmax) || (tzmin > tmax)) return false;
    return true;
}

RT_DEV void closest_list(const DevScene& S, const Ray& r, R a, R& closest, int& best) {
    const float4* __restrict__ hot = S.list_hot;
    int k = 0;
    for (; k + 8 <= S.n_list; k += 8) {                      // eight spheres per pass: their (scalar) loads in flight together
        float4 sv[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) sv[q] = hot[k + q];
#pragma unroll
        for (int q = 0; q < 8; ++q) sphere_test(r, a, sv[q], k + q, closest, best);
    }
    for (; k < S.n_list; ++k) sphere_test(r, a, hot[k], k, closest, best);
    if (best >= 0) best = S.list_id[best];
}

// ---------------------------------------------------------------------------------------------------- hitTree, binary16
// hitTree (acceleration_structure.h:319-342) for the 64 rays of a wave, in rounds of three phases:
//   1  every lane walks the tree — the visited set does not depend on the hits (traverseTree prunes by the slab test only) —
//      and puts the bucket range of each visited non-empty node into the WAVE's pool of segments (LDS);
//   2  the pool's sphere tests are dealt out evenly: lane w takes pairs [w*C, (w+1)*C) of the concatenated segments, whoever
//      the rays' owners are (a ray crossing eight cells and a ray crossing none cost the same to every lane; scanning each
//      lane's own buckets ran at 20 % lane utilisation, SQ_THREAD_CYCLES_VALU / 64 SQ_INSTS_VALU);
//   3  every owner picks up its ray's result.
// Why any order gives the reference's record.  What a sphere offers does not depend on closest_so_far: the near root t1 if
// t1 > t_min, else the far root t2 if t2 > t_min (sphere.h:24-43; t2 >= t1 also in this mixed arithmetic: nb - x <= nb + y for
// x, y >= 0, division by a > 0 and rounding are monotone), and it is accepted iff it is < closest_so_far.  The sequential scan
// therefore ends with the smallest offered t, the FIRST visited sphere among equal t.  Bucket entries are stored in the
// tree's pre-order — the visit order of every ray — so "first visited" = "lowest entry index": the result is the minimum of
// the 64-bit keys (t bits << 32 | entry index + 1) (t > t_min > 0: binary16 bits order like the values), seeded with
// (closest_so_far << 32 | 0) — what the ray holds already (the ground sphere, earlier rounds) wins ties like in the scan.
// Tests run two spheres at a time in packed binary16 (v_pk_add_f16 / v_pk_mul_f16: one rounding per operation, the same bits
// as the float-and-round form of rt_real.h).  A positive discriminant is rare (3 % of the tests) and expensive (correctly
// rounded sqrt, two IEEE divisions), so the test loop never evaluates it in place — 16 sub-slots per pass, each with one or two
// of the 64 lanes interested, would cost several times the packed arithmetic: a packed test marks the positive halves of a pass
// (pass_mask), their entries go into a queue of the wave (LDS: push_pass), and whenever 64 candidates have gathered all 64 lanes
// take one each (candidate_eval): b and the discriminant again from the pair, a cheap float filter against the owner's best so far,
// then the reference's roots, then an LDS atomic min on the owner's key.
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
RT_DEV h2 as_h2(uint32_t u) { return __builtin_bit_cast(h2, u); }
RT_DEV uint32_t h2_bits(h2 v) { return __builtin_bit_cast(uint32_t, v); }
RT_DEV uint32_t dup16(R r) { return (uint32_t)r.bits | ((uint32_t)r.bits << 16); }
typedef unsigned short us2 __attribute__((ext_vector_type(2)));

// The wave's pool of bucket ranges ("segments") per round.  The tree's level-3 nodes come in two kinds (C4: 64 bottom cells of 25-128
// pairs, 56 upper cells holding one large sphere each): BIG segments are concatenated and dealt out in equal spans, kPP pairs per
// lane and pass; SMALL ones (fewer than kSmallPairs pairs) would break every span they fall into — a pass cannot cross a segment
// boundary — so they go to a pool of their own and are tested one segment per lane.  (One pool for both: 1 pair per pass was the
// fastest setting, 80.0 ms against 96.6 with 4, because most segments were tiny; split: kPP (eight) pairs per pass on the big ones.)
constexpr int kBig = RT_H16_BIG;                              // big segments per wave and round (a multiple of 64)
constexpr int kSmall = RT_H16_SMALL;                          // small segments per wave and round
constexpr unsigned kSmallPairs = 8u;                          // a node with fewer pairs is a small segment (its count must fit 3 bits)
constexpr int kCand = 128;                                    // candidate queue of a wave
constexpr int kPP = RT_H16_PP;
constexpr int kTask = RT_H16_TASKS;                           // level-2 expansions a wave pools per round (regular trees)
#ifdef RT_H16_STATS            // diagnostic build (tools/h16_phases.py): cycles per phase, summed over waves
#define H16_TICK() ((unsigned long long)__builtin_amdgcn_s_memtime())
__device__ unsigned long long g_h16_cyc[8];
#ifdef RT_H16_COUNTS           // (the counts build keeps slots 1 and 2 for closest_tree's rounds and calls)
#define H16_ADD(k, t0) do { const unsigned long long now_ = H16_TICK(); if ((threadIdx.x & 63) == 0 && (k) != 1 && (k) != 2) atomicAdd(&g_h16_cyc[k], now_ - (t0)); (t0) = now_; } while (0)
#else
#define H16_ADD(k, t0) do { const unsigned long long now_ = H16_TICK(); if ((threadIdx.x & 63) == 0) atomicAdd(&g_h16_cyc[k], now_ - (t0)); (t0) = now_; } while (0)
#endif
#ifdef RT_H16_COUNTS           // (per-lane atomics: distort every timing of the same run)
#define H16_CNT(k, v) atomicAdd(&g_h16_cyc[k], (unsigned long long)(v))
#define H16_FIRST_ACTIVE() ((threadIdx.x & 63) == (unsigned)__builtin_ctzll(__ballot(true)))
#else
#define H16_CNT(k, v) ((void)0)
#define H16_FIRST_ACTIVE() false
#endif
#else
#define H16_CNT(k, v) ((void)0)
#define H16_FIRST_ACTIVE() false
#define H16_TICK() 0ull
#define H16_ADD(k, t0) ((void)0)
#endif
constexpr int kPlaneStride = 30;                              // binary16 slots per lane in the plane table (15 dwords: an odd stride)
struct WaveLds {
    uint2 seg[kBig];                     // big segments: (first pair | owner lane << 26, pairs)
    unsigned sseg[kSmall];               // small segments: first pair | pairs << 23 | owner lane << 26
    unsigned pref[kBig];                 // exclusive prefix of the big segments' pairs
    unsigned long long key[64];          // per owner: t bits << 32 | entry index + 1
    union {
        unsigned short tp[64 * kPlaneStride];   // phase 1: per lane, the ray's parameter at every box plane of the tree
        struct {
            uint4 ray[128];              // phase 2: per owner (ox,oy,oz,dx) (dy,dz,a,-), every value in both halves of its dword
            unsigned cq[kCand];          // candidates: entry index + 1 << 6 | owner
        } p2;
    } u;
    // regular trees: the level-2 nodes that are still to be expanded, whoever's ray they belong to — node | owner lane << 10 |
    // leaf children still to be pooled << 16 | expanded << 24 (closest_tree, phase 1b)
    unsigned task[kTask];
    unsigned count, scount, tcount, tbegin;                  // (tbegin: first task of the pool that is not done)
};
static_assert(sizeof(WaveLds) % 16 == 0, "WaveLds keeps 16-byte alignment");

RT_DEV void wave_sync() { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier(); }
RT_DEV unsigned wave_excl_scan(unsigned v, unsigned& total) {   // exclusive prefix sum over the 64 lanes: six DPP adds, no LDS
    int x = (int)v;
    x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, false);     // row_shr:1
    x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, false);     // row_shr:2
    x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, false);     // row_shr:4
    x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, false);     // row_shr:8
    x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false);     // row_bcast:15 into rows 1 and 3
    x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false);     // row_bcast:31 into rows 2 and 3
    total = (unsigned)__builtin_amdgcn_readlane(x, 63);
    return (unsigned)x - v;
}

struct PairRay { h2 ox, oy, oz, dx, dy, dz, a; };
RT_DEV PairRay load_pair_ray(const WaveLds& L, int owner) {
    const uint4 r0 = L.u.p2.ray[2 * owner], r1 = L.u.p2.ray[2 * owner + 1];
    PairRay q;
    q.ox = as_h2(r0.x); q.oy = as_h2(r0.y); q.oz = as_h2(r0.z); q.dx = as_h2(r0.w);
    q.dy = as_h2(r1.x); q.dz = as_h2(r1.y); q.a = as_h2(r1.z);
    return q;
}
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// Two spheres against one ray: the discriminants of sphere.h:18-22 in packed binary16 (17 instructions); disc > 0 marks a candidate.
// (A branch-free packed filter for spheres surely behind the origin — b > 0 and fl(b b) > fl(1.02 disc) — is exact but costs six
// packed instructions per pair to drop a fifth of the candidates: measured in round 2, 75.8 against 76.6 ms, not kept.)
RT_DEV void pair_math(const PairRay& q, const u32x4 p, h2& b, h2& disc) {
    const h2 cx = as_h2(p.x), cy = as_h2(p.y), cz = as_h2(p.z), r2 = as_h2(p.w);
    const h2 ocx = q.ox - cx, ocy = q.oy - cy, ocz = q.oz - cz;
    b = (ocx * q.dx + ocy * q.dy) + ocz * q.dz;
    const h2 cc = ((ocx * ocx + ocy * ocy) + ocz * ocz) - r2;
    const h2 bb = b * b;
    disc = bb - q.a * cc;
}

// One queued candidate: the offer of sphere::hit (sphere.h:24-43) for a positive discriminant, merged into its owner's key.
// (A record of the queue is the candidate's entry and its owner; b and the discriminant are computed again here, from the pair and the
// owner's ray — the same packed operations on the same operands, so the same bits — at 64 candidates per pass of the wave: cheaper
// than picking the two words out of a lane's pairs when the record is written: ~20 instructions of a 32-instruction turn then.)
RT_DEV void candidate_eval(WaveLds& L, const __amdgpu_buffer_rsrc_t ent, const uint32_t rec) {
    H16_CNT(5, 1);
    const int owner = (int)(rec & 63u);
    const uint32_t idx1 = rec >> 6;
    const uint32_t e0 = idx1 - 1u;                             // entry: pair e0 / 2, half e0 & 1
    const u32x4 pr = __builtin_amdgcn_raw_buffer_load_b128(ent, (int)((e0 >> 1) * 16u), 0, 0);
    const PairRay q = load_pair_ray(L, owner);
    h2 pb, pd;
    pair_math(q, pr, pb, pd);
    const uint32_t sh = (e0 & 1u) * 16u;
    R hb, hd, ha, hbest;
    hb.bits = (uint16_t)(h2_bits(pb) >> sh); hd.bits = (uint16_t)(h2_bits(pd) >> sh);
    ha.bits = (uint16_t)h2_bits(q.a);
    hbest.bits = (uint16_t)(((const unsigned*)&L.key[owner])[1]);
    const float b_f = fl(hb), d_f = fl(hd), A = fl(ha), best_f = fl(hbest);
    if (!(d_f > 0.0f)) return;                                // sphere.h:23 (the packed positive test of pass_mask lets a -0 through)
    const float nb = -b_f;
    // cheap filter, margin >= 2x its own error (half an ulp of binary16 on sqrt: 4.9e-4 s; float roundings ~1e-7)
    const float ra = __builtin_amdgcn_rcpf(A);
    const float s = __builtin_amdgcn_sqrtf(d_f);
    const float q1 = (nb - s) * ra, q2 = (nb + s) * ra;
    const float m = (fabsf(nb) + s) * ra * 1e-3f;
    if (q1 - m >= best_f) return;                             // near root surely not below the best: nothing it offers can win
    if (q2 + m <= 0.00099945068359375f) return;               // far root surely <= real_t(0.001f): offers nothing
    H16_CNT(6, 1);
    const R tmin = rf(0.001f);
    const R sq16 = rf(sqrtf(d_f));
    R t = rf((nb - fl(sq16)) / A);                            // float arithmetic on converted operands, one rounding
    if (!(t > tmin)) {
        t = rf((nb + sqrtf(d_f)) / A);                        // far root: float sqrt of float(disc), not rounded (sphere.h:36)
        if (!(t > tmin)) return;
    }
    atomicMin(&L.key[owner], ((unsigned long long)t.bits << 32) | idx1);
}

// all queued candidates, 64 at a time
RT_DEV void drain_candidates(WaveLds& L, const __amdgpu_buffer_rsrc_t ent, int lane, unsigned& qn) {
    wave_sync();
    const unsigned cnt = min(qn, (unsigned)kCand);
    for (unsigned base = 0; base < cnt; base += 64u) {
        const unsigned i = base + (unsigned)lane;
        if (i < cnt) candidate_eval(L, ent, L.u.p2.cq[i]);
    }
    qn = 0u;
    wave_sync();
}

// The positive discriminants of one pass — NP pairs per lane, i.e. up to 2 NP spheres — go into the wave's candidate queue.  One
// place for all of them: a packed test marks the positive halves (max(disc, 0) is 0 for a negative or NaN discriminant; min_u16 with
// 1 turns every other half into a 1), then every lane writes its own records, lowest sphere first.
// (Round 2: one ballot + mbcnt block per sphere slot, eight per pass, ~10 vector and ~11 scalar instructions each, and the candidate
// evaluation inlined in each of them for a full queue.  Until the end of round 3 a record also carried the pair's b and
// discriminant words, picked out of the lane's pairs by a chain of selects: see candidate_eval.)
// bit k (x halves) / 16 + k (y halves) for every positive discriminant among the NP pairs of a pass that lie inside the lane's range
// (the first `nb` of them: a pass may read past its segment, and what it computes there is masked here, once, not per pair)
template <int NP>
RT_DEV uint32_t pass_mask(const h2 (&d)[NP], unsigned nb) {
    static_assert(NP >= 1 && NP <= 16, "positions 0-15 (x halves) and 16-31 (y halves) of the mask");
    uint32_t acc = 0u;
#pragma unroll
    for (int k = 0; k < NP; ++k) {
        uint32_t pos, one;
        // v_pk_max_f16 with 0: 0 for a negative half and for a NaN one (maxNum: a discriminant is the result of an addition, a quiet NaN
        // if any); written out — the C form gets a canonicalising v_pk_max_f16 x, x in front for signalling NaNs that cannot occur here,
        // and the optimiser turns the min_u16 (1 in every half that is not 0) back into compares
        asm("v_pk_max_f16 %0, %1, 0" : "=v"(pos) : "v"(h2_bits(d[k])));
        asm("v_pk_min_u16 %0, %1, %2" : "=v"(one) : "v"(pos), "v"(0x00010001u));
        acc |= one << k;
    }
    return acc & (((1u << nb) - 1u) * 0x00010001u);
}

template <int STRIDE>
RT_DEV void push_pass(WaveLds& L, const __amdgpu_buffer_rsrc_t ent, int lane, uint32_t acc, unsigned pair0, int owner, unsigned& qn) {
    // one record per lane and turn, the lanes that still hold a positive taking consecutive slots (ballot + mbcnt): two turns a
    // pass on average; a turn that does not fit the queue drains it first — no candidate is evaluated in place
    while (true) {
        const unsigned long long m = __ballot(acc != 0u);
        if (m == 0ull) break;
        const unsigned n = (unsigned)__popcll(m);
        if (qn + n > (unsigned)kCand) drain_candidates(L, ent, lane, qn);
        if (acc != 0u) {
            const int p = __builtin_ctz(acc);
            acc &= acc - 1u;
            const int pr = p & 15, hi = p >> 4;               // pair of the pass, half of the pair
            const uint32_t idx1 = (pair0 + (unsigned)(pr * STRIDE)) * 2u + 1u + (unsigned)hi;       // (this lane's pairs of the pass are STRIDE apart)
            const unsigned slot = qn + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
            L.u.p2.cq[slot] = (idx1 << 6) | (uint32_t)owner;
        }
        qn += n;
    }
}

// div_prepare / div_by (rt_divshared.h): IEEE float division with the refined reciprocal shared per divisor — the header is also what
// tools/micro/div_shared.hip checks for all 2^32 pairs of binary16 operands.

// intersect_ray_aabb (acceleration_structure.h:226-244) with the six quotients looked up in the lane's plane table
RT_DEV bool ray_box_tab(const unsigned short* tp, uint32_t w) {
    typedef _Float16 H;                                          // (compared as binary16: see expand_node)
    auto T = [&](int q) { return __builtin_bit_cast(H, tp[(w >> (5 * q)) & 31u]); };
    H tmin = T(0), tmax = T(1);
    if (tmin > tmax) { const H t = tmin; tmin = tmax; tmax = t; }
    H tymin = T(2), tymax = T(3);
    if (tymin > tymax) { const H t = tymin; tymin = tymax; tymax = t; }
    if ((tmin > tymax) || (tymin > tmax)) return false;
    if (tymin > tmin) tmin = tymin;
    if (tymax < tmax) tmax = tymax;
    H tzmin = T(4), tzmax = T(5);
    if (tzmin > tzmax) { const H t = tzmin; tzmin = tzmax; tzmax = t; }
    if ((tmin > tzmax) || (tzmin > tmax)) return false;
    return true;
}

// The tree's nodes in LDS.  With a plane table (h16_np[0] > 0: every tree the library builds) the walk needs four words of a node —
// skip link, first pair, pairs, the six plane indices — and they are staged as ONE 16-byte record per node (one ds_read_b128 a
// visit instead of three, 2.5 KB instead of 7.5 KB at C4); a tree without the table keeps the full 48-byte nodes (boxes for ray_box).
// Returns the calling wave's pool, which follows the nodes.
constexpr int kPlaneLds4 = 8;                                 // float4 slots for the tree's planes in LDS (at most 30 planes: kPlaneStride)
RT_DEV bool regular_planes(const DevTree& T) { return T.h16_np[0] == 9 && T.h16_np[1] == 9 && T.h16_np[2] == 9; }
RT_DEV WaveLds* stage_tree(const DevTree& T, float4* s_nodes) {
    const bool compact = T.h16_np[0] > 0;
    if (compact) {
        for (int t = threadIdx.x; t < T.n_nodes; t += 256) {
            const float4 n1 = T.nodes4[t * 3 + 1], n2 = T.nodes4[t * 3 + 2];
            s_nodes[t] = make_float4(n1.z, n1.w, n2.x, n2.z);
        }
        __syncthreads();
        // behind the records: per node its eight children by octant (node index as 16 bits, 0 = no such child) — closest_tree expands
        // a node by testing all its children at once (below).  A child follows its parent in pre-order, its siblings follow at the
        // skip links; its octant is read off the plane indices: high half on an axis <=> its low plane is not the parent's.
        if (regular_planes(T)) {
            uint4* s_child = (uint4*)(s_nodes + T.n_nodes);
            for (int t = threadIdx.x; t < T.n_nodes; t += 256) {
                const uint32_t w = (uint32_t)__float_as_int(s_nodes[t].w);
                uint32_t ch[4] = {0u, 0u, 0u, 0u}, exist = 0u;
                const bool inner = ((w >> 5) & 31u) - (w & 31u) > 1u;   // its box spans more than one plane step
                if (inner) {
                    const int end = __float_as_int(s_nodes[t].x);
                    for (int c = t + 1; c < end; c = __float_as_int(s_nodes[c].x)) {
                        const uint32_t wc = (uint32_t)__float_as_int(s_nodes[c].w);
                        const int oct = ((wc & 31u) != (w & 31u) ? 4 : 0) | (((wc >> 10) & 31u) != ((w >> 10) & 31u) ? 2 : 0) | (((wc >> 20) & 31u) != ((w >> 20) & 31u) ? 1 : 0);
                        ch[oct >> 1] |= (uint32_t)c << (16 * (oct & 1));
                        exist |= 1u << oct;
                    }
                }
                s_child[t] = make_uint4(ch[0], ch[1], ch[2], ch[3]);
                if (inner) s_nodes[t].y = __int_as_float((int)exist);   // (an inner node has no bucket range: the word holds which children exist)
            }
        }
    } else {
        for (int t = threadIdx.x; t < T.n_nodes * 3; t += 256) s_nodes[t] = T.nodes4[t];
    }
    // behind that (trees with a plane table): the planes themselves, rounded to binary16 once — closest_tree forms every ray's quotients from them
    if (compact) {
        float* s_pl = (float*)(s_nodes + T.n_nodes * (regular_planes(T) ? 2 : 1));
        for (int t = threadIdx.x; t < T.h16_np[0] + T.h16_np[1] + T.h16_np[2]; t += 256) s_pl[t] = fl(rf(T.h16_planes[t]));
    }
    __syncthreads();
    return (WaveLds*)(s_nodes + T.n_nodes * (compact ? (regular_planes(T) ? 2 : 1) : 3) + (compact ? kPlaneLds4 : 0)) + (threadIdx.x >> 6);
}

// intersect_ray_aabb for the EIGHT children of a node at once (regular trees: a child box is a half of the parent's on every axis,
// cut at the parent's middle plane — nine plane parameters instead of 8 x 6, one LDS round trip).  Every child's test is the function
// ray_box_tab computes of its six parameters — the same comparisons in the same order, shared where children share a half.
// Returns the children that pass, bit 4 (x high) + 2 (y high) + (z high) — acceleration_structure.h:149-165.
RT_DEV unsigned expand_node(const unsigned short* tp, uint32_t w) {
    // (the parameters are compared as the binary16 values they are: the conversion to float is exact and order-preserving, NaNs included,
    // so v_cmp_*_f16 on the stored bits decides what the reference's float comparisons decide — nine conversions less per expansion)
    typedef _Float16 H;
    auto T = [&](uint32_t i) { return __builtin_bit_cast(H, tp[i]); };
    const uint32_t i0 = w & 31u, i1 = (w >> 5) & 31u, j0 = (w >> 10) & 31u, j1 = (w >> 15) & 31u, k0 = (w >> 20) & 31u, k1 = (w >> 25) & 31u;
    const H tx0 = T(i0), txm = T((i0 + i1) >> 1), tx1 = T(i1), ty0 = T(j0), tym = T((j0 + j1) >> 1), ty1 = T(j1), tz0 = T(k0), tzm = T((k0 + k1) >> 1), tz1 = T(k1);
    H lo[3][2], hi[3][2];                                        // [axis][half]: the slab interval after the reference's swap
    auto slab = [&](int ax, int h, H a, H b) { if (a > b) { const H t = a; a = b; b = t; } lo[ax][h] = a; hi[ax][h] = b; };
    slab(0, 0, tx0, txm); slab(0, 1, txm, tx1); slab(1, 0, ty0, tym); slab(1, 1, tym, ty1); slab(2, 0, tz0, tzm); slab(2, 1, tzm, tz1);
    unsigned mask = 0u;
#pragma unroll
    for (int xh = 0; xh < 2; ++xh)
#pragma unroll
        for (int yh = 0; yh < 2; ++yh) {
            H tmin = lo[0][xh], tmax = hi[0][xh];
            const H tymin = lo[1][yh], tymax = hi[1][yh];
            const bool xy = !((tmin > tymax) || (tymin > tmax));
            if (tymin > tmin) tmin = tymin;
            if (tymax < tmax) tmax = tymax;
#pragma unroll
            for (int zh = 0; zh < 2; ++zh) {
                const bool ok = xy && !((tmin > hi[2][zh]) || (lo[2][zh] > tmax));
                mask |= ok ? 1u << (4 * xh + 2 * yh + zh) : 0u;
            }
        }
    return mask;
}

RT_DEV void closest_tree(const DevScene& S, const DevTree& T, const float4* s_nodes, WaveLds& L, Ray& r, R& a, bool live, R& closest, int& best) {
    // the packed pairs (rt_api.hip, octree_upload) through a buffer descriptor: one offset register and immediate offsets serve the
    // kPP loads of a pass, and a pass may read past the last pair of the array (such loads return zeros; their results are masked)
    const __amdgpu_buffer_rsrc_t ent = __builtin_amdgcn_make_buffer_rsrc((void*)T.ent_hot, 0, T.n_entries * 8, 0x00020000);
    if (S.ground_valid) {
        int gb = -1;
        sphere_test(r, a, S.list_hot[0], 0, closest, gb);
        if (gb == 0) best = 0;
    }
    unsigned long long tph = H16_TICK(); (void)tph;
    const int lane = threadIdx.x & 63;
    int e_best = -1;
    int node = live ? 0 : T.n_nodes;
    const int n_nodes = T.n_nodes;
    const int np0 = T.h16_np[0], np1 = T.h16_np[1], np2 = T.h16_np[2];
    const bool regular = regular_planes(T);
    // (between the walks of two rounds a lane's state is ONE register, the wave's lives in LDS: the test loop needs every register it can get)
    uint32_t ws_m = (live ? 0u : 1u) << 26;                     // level-1 children to expand | children waiting for the task pool << 8 | their level-1 node << 16 | started << 26
    if (lane == 0) { L.tcount = 0u; L.tbegin = 0u; }
    const unsigned short* s_child16 = (const unsigned short*)(s_nodes + n_nodes);
    unsigned short* tp = L.u.tp + lane * kPlaneStride;
#ifdef RT_H16_COUNTS
    if (lane == 0) atomicAdd(&g_h16_cyc[2], 1ull);                   // (counts build: calls = wave-bounces)
#endif
    while (true) {
#ifdef RT_H16_COUNTS
        if (lane == 0) atomicAdd(&g_h16_cyc[1], 1ull);               // (counts build: rounds)
#endif
        if (lane == 0) { L.count = 0u; L.scount = 0u; }
        L.key[lane] = (unsigned long long)closest.bits << 32;
        if (np0 > 0) {
            // the ray's parameter at every box plane: the quotients intersect_ray_aabb forms, one division per plane
            const float* pl = (const float*)(s_nodes + n_nodes * (regular ? 2 : 1));       // (LDS: rf(plane) as a float, stage_tree)
            // (nine quotients per divisor: div_by — the compiler's IEEE division without its scaling steps and with the refined
            // reciprocal shared; bit-identical for every pair of binary16 operands, tools/micro/div_shared.hip)
            { const DivBy D = div_prepare(fl(r.d.x));
_Pragma("unroll 1") for (int k = 0; k < np0; ++k) tp[k] = rf(div_by(fl(rf(pl[k] - fl(r.o.x))), D)).bits; }
            { const DivBy D = div_prepare(fl(r.d.y));
_Pragma("unroll 1") for (int k = 0; k < np1; ++k) tp[np0 + k] = rf(div_by(fl(rf(pl[np0 + k] - fl(r.o.y))), D)).bits; }
            { const DivBy D = div_prepare(fl(r.d.z));
_Pragma("unroll 1") for (int k = 0; k < np2; ++k) tp[np0 + np1 + k] = rf(div_by(fl(rf(pl[np0 + np1 + k] - fl(r.o.z))), D)).bits; }
        }
        wave_sync();
        // ---- phase 1: walk; every visited non-empty level-3 node becomes a segment of one of the pools (or stalls the lane when that
        //      pool is full).  traverseTree (acceleration_structure.h:276-304) visits a node's non-zero children in index order,
        //      each after its own slab test; which nodes are visited does not depend on the order, and neither does the result (above).
        if (regular) {
            // Phase 1a, every lane for its own ray: the root's slab test and expansion, then the expansion of every level-1 child that
            // passed — each yields the level-2 children that pass, which are NOT expanded here: they go into the wave's task pool.
            // Phase 1b: the pooled level-2 expansions are dealt out evenly, 64 a pass, whoever's ray they belong to (the owner's plane
            // table is in LDS); the leaves that pass become segments.  A ray through the sphere field needs four or five expansions,
            // a ray into the sky one: walked lane by lane (rounds 2-3) the loop ran 13 trips a call at 27 of 64 lanes busy.
            // Which nodes are visited does not depend on who expands them or when, and neither does the result (above).
            unsigned wm0 = ws_m & 255u, pmask = (ws_m >> 8) & 255u; bool started = ((ws_m >> 26) & 1u) != 0u;
            int pnode = (int)((ws_m >> 16) & 1023u);                 // (a level-1 node's passing children that found the task pool full)
            if (!started) {
                started = true;
                const uint32_t w0 = (uint32_t)__float_as_int(s_nodes[0].w);
                if (ray_box_tab(tp, w0) && ((w0 >> 5) & 31u) - (w0 & 31u) > 1u) wm0 = expand_node(tp, w0) & (unsigned)__float_as_int(s_nodes[0].y);
            }
            bool room = true;
            auto push_tasks = [&]() {                               // pmask's level-2 children of pnode -> tasks; false: the pool is full
                const unsigned n = (unsigned)__popc(pmask);
                const unsigned slot = atomicAdd(&L.tcount, n);
                if (slot + n > (unsigned)kTask) { atomicSub(&L.tcount, n); return false; }
                unsigned k = slot;
                while (pmask != 0u) {
                    const int c = __builtin_ctz(pmask); pmask &= pmask - 1u;
                    L.task[k++] = (unsigned)s_child16[pnode * 8 + c] | ((unsigned)lane << 10) | (0xffu << 16);
                }
                return true;
            };
            if (pmask != 0u) room = push_tasks();
            while (room && wm0 != 0u) {
                H16_CNT(4, 1); if (H16_FIRST_ACTIVE()) H16_CNT(7, 1);       // (counts build: lane trips / wave trips of this loop)
                const int c = __builtin_ctz(wm0); wm0 &= wm0 - 1u;
                pnode = (int)s_child16[c];                              // (the root's child by octant)
                const float4 nd = s_nodes[pnode];                       // (skip, existing children, -, plane indices)
                pmask = expand_node(tp, (uint32_t)__float_as_int(nd.w)) & (unsigned)__float_as_int(nd.y);
                if (pmask != 0u) room = push_tasks();
            }
            ws_m = wm0 | (pmask << 8) | ((uint32_t)pnode << 16) | ((started ? 1u : 0u) << 26);
            wave_sync();
            // ---- phase 1b
            const unsigned n_task = __builtin_amdgcn_readfirstlane(L.tcount);
            unsigned t_begin = __builtin_amdgcn_readfirstlane(L.tbegin);
            bool stuck = false;
            while (t_begin < n_task) {
                const unsigned ti = t_begin + (unsigned)lane;
                unsigned e = ti < n_task ? L.task[ti] : (1u << 24);  // (no task: expanded, nothing left)
                const int tnode = (int)(e & 1023u), towner = (int)((e >> 10) & 63u);
                if (!((e >> 24) & 1u)) {
                    H16_CNT(4, 1);
                    const float4 nd = s_nodes[tnode];
                    const unsigned cm = expand_node(L.u.tp + towner * kPlaneStride, (uint32_t)__float_as_int(nd.w)) & (unsigned)__float_as_int(nd.y);
                    e = (e & 0xffffu) | (cm << 16) | (1u << 24);
                }
                if (H16_FIRST_ACTIVE()) H16_CNT(7, 1);
                unsigned lm = (e >> 16) & 255u;
                while (lm != 0u) {
                    const int c = __builtin_ctz(lm);
                    const float4 nd = s_nodes[(int)s_child16[tnode * 8 + c]];         // (skip, first pair, pairs, plane indices)
                    const uint32_t first = (uint32_t)__float_as_int(nd.y); const unsigned cnt = (unsigned)__float_as_int(nd.z);
                    if (cnt >= kSmallPairs) {
                        const unsigned slot = atomicAdd(&L.count, 1u);
                        if (slot >= (unsigned)kBig) break;
                        L.seg[slot] = make_uint2(first | ((uint32_t)towner << 26), cnt);
                    } else if (cnt > 0u) {
                        const unsigned slot = atomicAdd(&L.scount, 1u);
                        if (slot >= (unsigned)kSmall) break;
                        L.sseg[slot] = first | (cnt << 23) | ((uint32_t)towner << 26);
                    }
                    lm &= lm - 1u;
                }
                if (ti < n_task) L.task[ti] = (e & 0xff00ffffu) | (lm << 16);
                if (__ballot(lm != 0u) != 0ull) { stuck = true; break; }        // a segment pool is full: this pass again after the tests
                t_begin += 64u;
            }
            wave_sync();
            // all done: no lane has children left to expand and no task waits; then the pool starts over
            const bool lane_open = wm0 != 0u || pmask != 0u;
            const bool more = stuck || t_begin < n_task || __ballot(lane_open) != 0ull;
            if (!more) node = n_nodes;
            if (!stuck && t_begin >= n_task) { t_begin = 0u; if (lane == 0) L.tcount = 0u; }     // (every task done: a fresh pool for the lanes that still hold children)
            if (lane == 0) L.tbegin = t_begin;
        } else
        while (node < n_nodes) {
            bool pass; int skip; uint32_t first; unsigned cnt;
            if (np0 > 0) {
                const float4 nd = s_nodes[node];                     // (skip, first pair, pairs, plane indices)
                pass = ray_box_tab(tp, (uint32_t)__float_as_int(nd.w));
                skip = __float_as_int(nd.x); first = (uint32_t)__float_as_int(nd.y); cnt = (unsigned)__float_as_int(nd.z);
            } else {
                const float4 n0 = s_nodes[node * 3 + 0], n1 = s_nodes[node * 3 + 1], n2 = s_nodes[node * 3 + 2];
                pass = ray_box(r, n0, n1);
                skip = __float_as_int(n1.z); first = (uint32_t)__float_as_int(n1.w); cnt = (unsigned)__float_as_int(n2.x);
            }
            if (pass) {
                if (cnt >= kSmallPairs) {
                    const unsigned slot = atomicAdd(&L.count, 1u);
                    if (slot >= (unsigned)kBig) break;               // pool full: this node again next round
                    L.seg[slot] = make_uint2(first | ((uint32_t)lane << 26), cnt);
                } else if (cnt > 0u) {
                    const unsigned slot = atomicAdd(&L.scount, 1u);
                    if (slot >= (unsigned)kSmall) break;
                    L.sseg[slot] = first | (cnt << 23) | ((uint32_t)lane << 26);
                }
                node = node + 1;
            } else {
                node = skip;
            }
        }
        wave_sync();
        H16_ADD(0, tph);                                             // walk
        const unsigned n_seg = min(__builtin_amdgcn_readfirstlane(L.count), (unsigned)kBig);
        const unsigned n_small = min(__builtin_amdgcn_readfirstlane(L.scount), (unsigned)kSmall);
        if (n_seg == 0u && n_small == 0u) {
            // nothing to test this round.  Lanes stalled by a full SEGMENT pool cannot exist then; lanes whose level-2 nodes found
            // the TASK pool full can (regular trees): they go on next round
            if (__ballot(node < n_nodes) == 0ull) break;
            continue;
        }
        // this lane's ray for whoever tests its spheres (the plane table's space: the walk is over)
        L.u.p2.ray[2 * lane] = make_uint4(dup16(r.o.x), dup16(r.o.y), dup16(r.o.z), dup16(r.d.x));
        L.u.p2.ray[2 * lane + 1] = make_uint4(dup16(r.d.y), dup16(r.d.z), dup16(a), 0u);
        unsigned qn = 0u;                                            // candidates queued (wave-uniform)
        if (n_seg != 0u) {
            // ---- phase 2a: exclusive prefix of the big segments' pair counts, kBig/64 segments per lane
            unsigned total = 0u;
            {
                unsigned run = 0u;
#pragma unroll
                for (int k = 0; k < kBig / 64; ++k) {
                    const unsigned sidx = (unsigned)(k * 64 + lane);
                    const unsigned c = sidx < n_seg ? L.seg[sidx].y : 0u;       // pairs
                    unsigned tot;
                    const unsigned ex = wave_excl_scan(c, tot);
                    if (sidx < n_seg) L.pref[sidx] = run + ex;
                    run += tot;
                }
                total = run;
            }
            wave_sync();
            // every lane takes an equal span of the pool's concatenated pairs and steps through it kPP pairs a pass: kPP loads in
            // flight and kPP independent chains of packed arithmetic (a dependent v_pk_*_f16 issues every ~9 cycles, independent
            // ones every ~4.5: tools/micro/pk16_rate.hip); a pass stays inside one segment (one owner's ray).  (A span per QUAD of
            // lanes, so that every load of a pass reads 64 contiguous bytes per quad, measured the same: 46.7 against 46.0 ms —
            // the loop is not bound by the texture addresser — and loses more pair slots at segment ends.)
            const unsigned C = (total + 63u) / 64u;
            const unsigned begin = min((unsigned)lane * C, total), end = min(begin + C, total);
            unsigned cur = begin, seg_end = begin, base = 0u, sg = 0u;      // seg_end == cur: the first pass loads segment sg
            int owner = 0;
            PairRay q; q.ox = q.oy = q.oz = q.dx = q.dy = q.dz = q.a = as_h2(0u);
            if (begin < end) {                                       // last segment starting at or before `begin`
                unsigned lo = 0u, hi = n_seg;
#pragma unroll
                for (int it = 0; it < 7; ++it) {                     // kBig <= 128
                    const unsigned mid = (lo + hi) >> 1;
                    if (hi - lo > 1u) { if (L.pref[mid] <= begin) lo = mid; else hi = mid; }
                }
                sg = lo;
            }
            static_assert(kBig <= 128, "binary search depth");
            H16_ADD(1, tph);                                         // prefix + search
            while (true) {
                const bool act = cur < end;
                if (__ballot(act) == 0ull) break;
                if (act && cur >= seg_end) {
                    const uint2 sd = L.seg[sg];
                    const unsigned p0 = L.pref[sg];
                    ++sg;
                    seg_end = min(p0 + sd.y, end);
                    base = (sd.x & 0x3ffffffu) - p0;                 // pair `cur` of the pool is pair base + cur of the tree
                    owner = (int)(sd.x >> 26);
                    q = load_pair_ray(L, owner);
                }
                const unsigned nb = act ? min((unsigned)kPP, seg_end - cur) : 0u;
                const unsigned i0 = act ? base + cur : 0u;
                const int voff = (int)(i0 * 16u);
                u32x4 e[kPP];
#pragma unroll
                for (int k = 0; k < kPP; ++k) e[k] = __builtin_amdgcn_raw_buffer_load_b128(ent, voff, 16 * k, 0);
                h2 b[kPP], d[kPP];
#pragma unroll
                for (int k = 0; k < kPP; ++k) pair_math(q, e[k], b[k], d[k]);
                const uint32_t acc = pass_mask<kPP>(d, nb);
                if (__ballot(acc != 0u) != 0ull) push_pass<1>(L, ent, lane, acc, i0, owner, qn);
                cur += nb;
                if (qn >= 64u) drain_candidates(L, ent, lane, qn);
            }
        }
        // ---- phase 2b: the small segments, one per lane (their pairs one after the other: mostly one)
        for (unsigned sb = 0u; sb < n_small; sb += 64u) {
            const unsigned j = sb + (unsigned)lane;
            const unsigned w = j < n_small ? L.sseg[j] : 0u;
            const unsigned first = w & 0x7fffffu, cnt = (w >> 23) & 7u;       // (cnt 0: no segment for this lane)
            const int owner = (int)(w >> 26);
            const PairRay q = load_pair_ray(L, owner);
            for (unsigned k = 0u; __ballot(k < cnt) != 0ull; ++k) {
                const bool act = k < cnt;
                const unsigned ix = act ? first + k : 0u;
                const u32x4 e = __builtin_amdgcn_raw_buffer_load_b128(ent, (int)(ix * 16u), 0, 0);
                h2 b, d[1];
                pair_math(q, e, b, d[0]);
                const uint32_t acc = pass_mask<1>(d, act ? 1u : 0u);
                if (__ballot(acc != 0u) != 0ull) push_pass<1>(L, ent, lane, acc, ix, owner, qn);
                if (qn >= 64u) drain_candidates(L, ent, lane, qn);
            }
        }
        drain_candidates(L, ent, lane, qn);
        H16_ADD(2, tph);                                             // tests
        // ---- phase 3.  (The ray comes back from where phase 2 put it for the other lanes: between that store and this load its seven
        // values need no registers — the test loop spills otherwise — and the bits are the same.)
        {
            const uint4 q0 = L.u.p2.ray[2 * lane]; const uint2 q1 = *(const uint2*)&L.u.p2.ray[2 * lane + 1]; const unsigned qa = L.u.p2.ray[2 * lane + 1].z;
            r.o.x.bits = (uint16_t)q0.x; r.o.y.bits = (uint16_t)q0.y; r.o.z.bits = (uint16_t)q0.z; r.d.x.bits = (uint16_t)q0.w;
            r.d.y.bits = (uint16_t)q1.x; r.d.z.bits = (uint16_t)q1.y; a.bits = (uint16_t)qa;
        }
        const unsigned long long k = L.key[lane];
        closest.bits = (uint16_t)(k >> 32);
        if ((uint32_t)k != 0u) e_best = (int)(uint32_t)k - 1;
        if (__ballot(node < n_nodes) == 0ull) break;
    }
    if (e_best >= 0) best = T.ent_id[e_best];
}

RT_DEV V random_in_unit_sphere(Rng& s) {                  // material.h:35-41
    V p;
    const V one = {ri(1), ri(1), ri(1)};
    do {
        const float x = rng_uniform(s); const float y = rng_uniform(s); const float z = rng_uniform(s);
        const V q = {rf(x), rf(y), rf(z)};
        p = vsub(vscale(rf(2.0f), q), one);
    } while (vsqlen(p) >= rf(1.0f));
    return p;
}

struct Cam { V origin, llc, horizontal, vertical, u, v; R lens_radius; };
RT_DEV Cam load_camera(const rt_camera& c) {
    return {vload(c.origin), vload(c.lower_left_corner), vload(c.horizontal), vload(c.vertical), vload(c.u), vload(c.v), rf(c.lens_radius)};
}

RT_DEV Ray primary_ray(const Cam& c, int i, int j, int max_x, int max_y, Rng& s) {   // main.cu:104-106, camera.h:12-18,45-49
    const float du = rng_uniform(s);
    const R u = rf((float)i + du) / ri(max_x);
    const float dv = rng_uniform(s);
    const R v = rf((float)j + dv) / ri(max_y);
    V p;
    const V one0 = {ri(1), ri(1), ri(0)};
    do {
        const float x = rng_uniform(s); const float y = rng_uniform(s);
        const V q = {rf(x), rf(y), ri(0)};
        p = vsub(vscale(rf(2.0f), q), one0);
    } while (vdot(p, p) >= rf(1.0f));
    const V rdv = vscale(c.lens_radius, p);
    const V offset = vadd(vscale(rdv.x, c.u), vscale(rdv.y, c.v));
    Ray r;
    r.o = vadd(c.origin, offset);
    r.d = vsub(vsub(vadd(vadd(c.llc, vscale(u, c.horizontal)), vscale(v, c.vertical)), c.origin), offset);
    return r;
}

// material::scatter (material.h:55-113); false = absorbed
RT_DEV bool scatter(const DevScene& S, int sphere, R t, Ray& r, V& att, Rng& s) {
    const float4 g = S.shade[2 * sphere];
    const float4 m = S.shade[2 * sphere + 1];
    const int kind = S.kind8[sphere];
    const V c = {rf(g.x), rf(g.y), rf(g.z)};
    const V p = vadd(r.o, vscale(t, r.d));                                  // ray.h:13
    const V n = vdiv(vsub(p, c), rf(g.w));                                  // sphere.h:29
    const V albedo = {rf(m.x), rf(m.y), rf(m.z)};
    // As in the float kernel: lambertian and metal both draw exactly one random_in_unit_sphere and nothing else — one shared rejection
    // loop instead of one per branch; metal and dielectric both start from unit_vector(r_in.direction()) — formed once.  A wave holds
    // all three kinds and walks through every branch; the draws keep their order, the expressions their operands.
    V q = {ri(0), ri(0), ri(0)};
    if (kind != RT_MAT_DIELECTRIC) q = random_in_unit_sphere(s);
    V ud = {ri(0), ri(0), ri(0)};
    if (kind != RT_MAT_LAMBERTIAN) ud = vunit(r.d);
    if (kind == RT_MAT_LAMBERTIAN) {
        const V target = vadd(vadd(p, n), q);
        r.d = vsub(target, p); r.o = p;
        att = vmul(att, albedo);
        return true;
    }
    if (kind == RT_MAT_METAL) {
        const V refl = vsub(ud, vscale(rf(2.0f) * vdot(ud, n), n));
        r.d = vadd(refl, vscale(rf(m.w), q));
        r.o = p;
        att = vmul(att, albedo);
        return vdot(r.d, n) > rf(0.0f);
    }
    // dielectric
    const R ref_idx = rf(m.w);
    const V reflected = vsub(r.d, vscale(rf(2.0f) * vdot(r.d, n), n));
    V outward; R ni_over_nt, cosine;
    if (vdot(r.d, n) > rf(0.0f)) {
        outward = vneg(n); ni_over_nt = ref_idx;
        cosine = vdot(r.d, n) / vlen(r.d);
        cosine = rsqrt_(rf(1.0f) - ref_idx * ref_idx * (rf(1.0f) - cosine * cosine));
    } else {
        outward = n; ni_over_nt = rf(1.0f) / ref_idx;
        cosine = rf(-fl(vdot(r.d, n)) / fl(vlen(r.d)));                     // float negate, float divide, one rounding
    }
    // refract (material.h:17-31)
    const V uv = ud;
    const R dt = vdot(uv, outward);
    const R disc = rf(1.0f) - ni_over_nt * ni_over_nt * (rf(1.0f) - dt * dt);
    V refracted = {ri(0), ri(0), ri(0)};
    R reflect_prob;
    if (disc > ri(0)) {
        refracted = vsub(vscale(ni_over_nt, vsub(uv, vscale(dt, outward))), vscale(rsqrt_(disc), outward));
        R r0 = rf(1.0f - fl(ref_idx)) / rf(1.0f + fl(ref_idx));             // schlick (material.h:11-15)
        r0 = r0 * r0;
        reflect_prob = r0 + rf(1.0f - fl(r0)) * rf(pow5(1.0f - fl(cosine)));
    } else {
        reflect_prob = rf(1.0f);
    }
    r.o = p;
    if (rng_uniform(s) < fl(reflect_prob)) r.d = reflected; else r.d = refracted;
    return true;
}

RT_DEV V sky(const Ray& r, const V& att) {                 // main.cu:67-72
    const V ud = vunit(r.d);
    const R t = rf(0.5f) * (ud.y + rf(1.0f));
    const R omt = rf(1.0f - fl(t));                        // float subtraction, converted by `float * vec3`
    const V white = {rd(1.0), rd(1.0), rd(1.0)}, blue = {rd(0.5), rd(0.7), rd(1.0)};
    const V c = vadd(vscale(omt, white), vscale(t, blue));
    return vmul(att, c);
}

// LDS: the tree's nodes, then one WaveLds per wave.  (Staging the leaf sphere lists there too — C4's 89 KB of pairs fit next to
// eight wave areas, one 512-thread block per CU — was built and measured: 97.6 ms against 83.8 ms from L2 at the same pass width:
// two waves per SIMD instead of four cost the walk and the shading more than the scan gains; the pairs stay in L2.)
template <bool TREE, int MODE>
__global__ __launch_bounds__(256, RT_H16_MINWAVES) void k_render_h(RenderArgs A) {
    extern __shared__ float4 s_nodes[];
    WaveLds* wl = TREE ? stage_tree(A.tree, s_nodes) : nullptr;       // this wave's pool (TREE only)
    const int lane = threadIdx.x & 63;
    const long long n_slots = A.n_local_tiles * 64;
    const long long first_free = 0;                                // every slot is handed out by the work counter
    const int ns = (MODE == 0) ? A.ns : 1;
    const Cam cam = load_camera(A.scene.cam);
    // scheduling as in k_render (rt_kernels.hip): tiles in the pilot pass's longest-first order, slots interleaved over blocks of 64
    // tiles, pre-classified long chains first, RT_H16_LONG_PER_WAVE to a wave that does not refill its other lanes while one is alive
    const unsigned int n_long_raw = A.long_list ? A.queue[2] : 0u;
    const bool use_long = n_long_raw != 0u && (long long)n_long_raw * 64 <= n_slots;
    const unsigned int n_long = use_long ? n_long_raw : 0u;
    bool is_long = false, long_done = false, thin = false, retired = false, thin_counted = false;
    unsigned int iters = 0;                       // bounces spent on the current pixel (in-flight detection of long chains)
    const unsigned int thin_cap = gridDim.x * 4u / RT_H16_THIN_CAP_DEN;      // at most a quarter of the waves may go thin for chains found in flight

    long long slot = 0;
    int i = 0, j = 0; long long idx = 0;
    Rng s = {0, 0, 0, 0, 0, 0};
    V col = {ri(0), ri(0), ri(0)};
    V att = {rd(1.0), rd(1.0), rd(1.0)};
    Ray r; r.o = col; r.d = att;
    int sample = 0, depth = 0;
    bool live = false;

    auto start_pixel = [&]() {
        const rt_rand_state* st = A.rand_state + idx;
        s.d = st->d; s.v0 = st->v[0]; s.v1 = st->v[1]; s.v2 = st->v[2]; s.v3 = st->v[3]; s.v4 = st->v[4];
        col = {ri(0), ri(0), ri(0)}; att = {rd(1.0), rd(1.0), rd(1.0)}; sample = 0; depth = 0;
        r = primary_ray(cam, i, j, A.max_x, A.max_y, s);
    };
    auto begin_pixel = [&]() {
        live = false; is_long = false;
        while (slot < n_slots) {
            // consecutive slots are the same pixel position of 64 different tiles (in hand-out order): the pixels of a tile never travel together
            constexpr int kIl = RT_H16_INTERLEAVE;                   // (rt_tuning.h: tiles whose pixels interleave)
            const long long blk = slot / (64 * kIl);
            const long long tiles_in_blk = (A.n_local_tiles - blk * kIl) < kIl ? (A.n_local_tiles - blk * kIl) : kIl;
            const long long within = slot % (64 * kIl);
            const long long rank = blk * kIl + within % tiles_in_blk;
            const int l = (int)(within / tiles_in_blk);
            const long long local_tile = A.order ? (long long)A.order[rank] : rank;
            const long long tile = part_tile(local_tile, A.part, A.nparts, A.tile_begin, A.tile_end);
            const int tx = (int)(tile % A.tiles_x), ty = (int)(tile / A.tiles_x);
            i = tx * 8 + (l & 7); j = ty * 8 + (l >> 3);
            const bool taken = use_long && A.long_flag[local_tile * 64 + l];      // long chains are handed out separately
            if (i < A.max_x && j < A.max_y && !taken) { idx = part_whole(A.nparts, A.tile_begin, A.tile_end) ? (long long)j * A.max_x + i : local_tile * 64 + l; live = true; break; }
            slot = first_free + (long long)atomicAdd(A.queue, 1u);
        }
        if (!live) retired = true;
        if (live) start_pixel();
    };
    // the next pre-classified long chain (lanes 0..RT_H16_LONG_PER_WAVE-1), strided through the list as in k_render
    auto begin_long_pixel = [&]() -> bool {
        if (!use_long || long_done) return false;
        const unsigned int h = atomicAdd(A.queue + 3, 1u);
        if (h >= n_long) { long_done = true; return false; }
        const unsigned int stride = n_long % 257u ? 257u : (n_long % 263u ? 263u : 269u);
        const long long pid = (long long)A.long_list[(unsigned int)(((unsigned long long)h * stride) % n_long)];
        const long long local_tile = pid >> 6;
        const int l = (int)(pid & 63);
        const long long tile = part_tile(local_tile, A.part, A.nparts, A.tile_begin, A.tile_end);
        const int tx = (int)(tile % A.tiles_x), ty = (int)(tile / A.tiles_x);
        i = tx * 8 + (l & 7); j = ty * 8 + (l >> 3);
        idx = part_whole(A.nparts, A.tile_begin, A.tile_end) ? (long long)j * A.max_x + i : pid;
        live = true; is_long = true;
        start_pixel();
        return true;
    };
    auto end_pixel = [&]() {
        rt_rand_state* st = A.rand_state + idx;
        st->d = s.d; st->v[0] = s.v0; st->v[1] = s.v1; st->v[2] = s.v2; st->v[3] = s.v3; st->v[4] = s.v4;
        uint16_t* fb = (uint16_t*)A.fb + idx * 3;
        if (MODE == 0) {
            const R k = rd(1.0 / (double)fl(ri(A.ns)));                      // vec3::operator/=(real_t): k = 1.0/t in double
            col = {col.x * k, col.y * k, col.z * k};
            fb[0] = rsqrt_(col.x).bits; fb[1] = rsqrt_(col.y).bits; fb[2] = rsqrt_(col.z).bits;
        } else {
            if (A.ns == 1) { fb[0] = col.x.bits; fb[1] = col.y.bits; fb[2] = col.z.bits; }
            else {
                R f0, f1, f2; f0.bits = fb[0]; f1.bits = fb[1]; f2.bits = fb[2];
                fb[0] = (f0 + col.x).bits; fb[1] = (f1 + col.y).bits; fb[2] = (f2 + col.z).bits;
            }
        }
    };
    if (ns > 0) {
        if (lane < RT_H16_LONG_PER_WAVE) begin_long_pixel();
        if (__ballot(live) != 0ull) { thin = true; __builtin_amdgcn_s_setprio(3); }
        else { slot = first_free + (long long)atomicAdd(A.queue, 1u); begin_pixel(); }
    }
    unsigned long long tk0 = H16_TICK(); (void)tk0;

    while (true) {
#if RT_H16_LONG_RATE
        if (!thin && __ballot(live && is_long) != 0ull) {
            // a chain found in flight (below): the wave stops refilling its other lanes, unless too many waves do so already
            unsigned int prev = 0;
            if (lane == 0) prev = atomicAdd(A.queue + 1, 1u);
            prev = __builtin_amdgcn_readfirstlane(prev);
            if (prev < thin_cap) { thin = true; thin_counted = true; __builtin_amdgcn_s_setprio(3); }
            else { if (lane == 0) atomicSub(A.queue + 1, 1u); is_long = false; }
        }
#endif
        if (thin && __ballot(live && is_long) == 0ull) {                                                       // its chains have ended: refill
            thin = false; __builtin_amdgcn_s_setprio(0);
            if (thin_counted && lane == 0) atomicSub(A.queue + 1, 1u);
            thin_counted = false;
        }
        if (!thin && !live && !retired && ns > 0) { slot = first_free + (long long)atomicAdd(A.queue, 1u); begin_pixel(); }
        if (__ballot(live) == 0ull) break;
        R a = vdot(r.d, r.d);
        R closest = rf(FLT_MAX); int best = -1;                              // real_t(FLT_MAX) = +inf in binary16
        if (TREE) closest_tree(A.scene, A.tree, s_nodes, *wl, r, a, live, closest, best);
        else closest_list(A.scene, r, a, closest, best);
        if (live) {
            bool done;
            ++iters;
            if (best >= 0) {
                const bool cont = scatter(A.scene, best, closest, r, att, s);
                ++depth;
                done = !cont || depth >= 50;
            } else {
                col = vadd(col, sky(r, att));
                done = true;
            }
            if (done) {
                ++sample; depth = 0; att = {rd(1.0), rd(1.0), rd(1.0)};
                if (sample < ns) {
                    r = primary_ray(cam, i, j, A.max_x, A.max_y, s);
#if RT_H16_LONG_RATE
                    if (MODE == 0 && (sample & (RT_H16_LONG_CHECK - 1)) == 0 && sample + 8 <= ns && iters >= (unsigned int)(RT_H16_LONG_RATE * sample)) is_long = true;
#endif
                } else {
                    end_pixel();
                    live = false; is_long = false; iters = 0;
                    if (lane < RT_H16_LONG_PER_WAVE && begin_long_pixel()) { /* the next long chain */ }
                    else if (!thin) { slot = first_free + (long long)atomicAdd(A.queue, 1u); begin_pixel(); }
                }
            }
        }
    }
    H16_ADD(3, tk0);                                                 // whole loop
}

// The pilot pass of the scheduling (k_tile_cost of rt_kernels.hip in binary16): two samples per 2x2 pixel block on a private RNG
// stream, in adjacent lanes, cut at RT_H16_PILOT_CAP bounces; only the bounce counts are kept — per tile for the hand-out order,
// per block for k_long_select.  Changes WHEN a pixel is rendered, never the pixel.
template <bool TREE>
__global__ __launch_bounds__(256, RT_H16_MINWAVES) void k_tile_cost_h(RenderArgs A, int* __restrict__ cost, unsigned char* __restrict__ pilot) {
    extern __shared__ float4 s_nodes[];
    WaveLds* wl = TREE ? stage_tree(A.tree, s_nodes) : nullptr;
    const int lane = threadIdx.x & 63;
    const Cam cam = load_camera(A.scene.cam);
    // a wave covers two tiles: 16 blocks x 2 samples each
    const long long local_tile = ((long long)blockIdx.x * 4 + (threadIdx.x >> 6)) * 2 + lane / 32;
    const bool tile_ok = local_tile < A.n_local_tiles;
    const long long tile = part_tile(tile_ok ? local_tile : 0, A.part, A.nparts, A.tile_begin, A.tile_end);
    const int tx = (int)(tile % A.tiles_x), ty = (int)(tile / A.tiles_x);
    const int sub = (lane % 32) / 2, smp = lane % 2;
    const int lx = 2 * (sub & 3), ly = 2 * (sub >> 2);
    const int i = tx * 8 + lx, j = ty * 8 + ly;
    const bool inside = tile_ok && (i < A.max_x) && (j < A.max_y);
    // a private stream: any state that is not all zero will do (this is not curand_init; nothing is compared with it)
    Rng ps;
    {
        unsigned long long z = 0x5deece66dull + (unsigned long long)((long long)j * A.max_x + i) * 0x9e3779b97f4a7c15ull + (unsigned long long)smp * 0xbf58476d1ce4e5b9ull;
        auto mix = [&]() { z += 0x9e3779b97f4a7c15ull; unsigned long long x = z; x = (x ^ (x >> 30)) * 0xbf58476d1ce4e5b9ull; x = (x ^ (x >> 27)) * 0x94d049bb133111ebull; return (uint32_t)((x ^ (x >> 31)) >> 16); };
        ps.v0 = mix() | 1u; ps.v1 = mix(); ps.v2 = mix(); ps.v3 = mix(); ps.v4 = mix(); ps.d = mix();
    }
    V att = {rd(1.0), rd(1.0), rd(1.0)};
    Ray r; r.o = {ri(0), ri(0), ri(0)}; r.d = {ri(0), ri(1), ri(0)};
    bool live = inside;
    if (live) r = primary_ray(cam, i, j, A.max_x, A.max_y, ps);
    int bounces = 0;
    while (__ballot(live) != 0ull) {
        R a = vdot(r.d, r.d);
        R closest = rf(FLT_MAX); int best = -1;
        if (TREE) closest_tree(A.scene, A.tree, s_nodes, *wl, r, a, live, closest, best);
        else closest_list(A.scene, r, a, closest, best);
        if (live) {
            ++bounces;
            bool done = true;
            if (best >= 0) { const bool cont = scatter(A.scene, best, closest, r, att, ps); done = !cont || bounces >= RT_H16_PILOT_CAP; }
            if (done) live = false;
        }
    }
    int pix = inside ? bounces : 0;
    pix += __shfl_xor(pix, 1);                                      // the block's two samples
    if (pilot && tile_ok && smp == 0) pilot[local_tile * 16 + sub] = (unsigned char)(pix < 255 ? pix : 255);
    int w = inside ? bounces : 0;
    for (int off = 16; off > 0; off >>= 1) w += __shfl_xor(w, off);                      // the tile's 16 blocks x 2 samples
    if (lane % 32 == 0 && tile_ok) cost[local_tile] = w * 4;
}

template <bool TREE>
__global__ __launch_bounds__(256) void k_trace_h(DevScene S, DevTree T, const float* rays, long long n, rt_hit_record* out) {
    extern __shared__ float4 s_nodes[];
    WaveLds* wl = TREE ? stage_tree(T, s_nodes) : nullptr;
    const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
    const bool live = gid < n;
    Ray r; r.o = {ri(0), ri(0), ri(0)}; r.d = {ri(0), ri(1), ri(0)};
    if (live) { const float* p = rays + gid * 6; r.o = vload(p); r.d = vload(p + 3); }
    R a = vdot(r.d, r.d);
    R closest = rf(FLT_MAX); int best = -1;
    if (TREE) closest_tree(S, T, s_nodes, *wl, r, a, live, closest, best);
    else closest_list(S, r, a, closest, best);
    if (!live) return;
    rt_hit_record h;
    h.sphere = best; h.t = 0.f; h.p[0] = h.p[1] = h.p[2] = 0.f; h.normal[0] = h.normal[1] = h.normal[2] = 0.f;
    if (best >= 0) {
        const float4 g = S.geom[best];
        const V c = {rf(g.x), rf(g.y), rf(g.z)};
        const V p = vadd(r.o, vscale(closest, r.d));
        const V nn = vdiv(vsub(p, c), rf(g.w));
        h.t = fl(closest);
        h.p[0] = fl(p.x); h.p[1] = fl(p.y); h.p[2] = fl(p.z);
        h.normal[0] = fl(nn.x); h.normal[1] = fl(nn.y); h.normal[2] = fl(nn.z);
    }
    out[gid] = h;
}

__global__ __launch_bounds__(256) void k_assemble_h(uint16_t* full, const uint16_t* parts, int max_x, int max_y, int tiles_x, int nparts, long long part_stride_px, long long n_tiles) {
    const int lane = threadIdx.x & 63;
    const long long tile = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (tile >= n_tiles) return;
    const int tx = (int)(tile % tiles_x), ty = (int)(tile / tiles_x);
    const int i = tx * 8 + (lane & 7), j = ty * 8 + (lane >> 3);
    if (i >= max_x || j >= max_y) return;
    int owner; long long local_tile;
    part_owner(tile, nparts, owner, local_tile);
    const long long src = owner * part_stride_px + local_tile * 64 + lane;
    const long long dst = (long long)j * max_x + i;
    full[dst * 3 + 0] = parts[src * 3 + 0]; full[dst * 3 + 1] = parts[src * 3 + 1]; full[dst * 3 + 2] = parts[src * 3 + 2];
}

} // namespace h16

static unsigned resident_blocks_h(const void* kernel, size_t lds) {
    int dev = 0, cus = 0, per_cu = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 1024u;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) return 1024u;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, 256, lds) != hipSuccess || per_cu <= 0) per_cu = 4;
    return (unsigned)(cus * per_cu);
}

const char* render_kernel_name_h(bool tree, int mode) {
    return tree ? (mode == 0 ? "k_render_h<true,0>" : "k_render_h<true,1>") : (mode == 0 ? "k_render_h<false,0>" : "k_render_h<false,1>");
}

// LDS of a block of the binary16 tree kernels: the nodes (stage_tree), then one WaveLds per wave
static size_t h16_lds_bytes(bool tree, const DevTree& T) {
    const bool regular = T.h16_np[0] == 9 && T.h16_np[1] == 9 && T.h16_np[2] == 9;
    return tree ? (size_t)T.n_nodes * (T.h16_np[0] > 0 ? (regular ? 2 : 1) * sizeof(float4) : sizeof(DevNode)) + (T.h16_np[0] > 0 ? h16::kPlaneLds4 * sizeof(float4) : 0) + 4 * sizeof(h16::WaveLds) : 0;
}

hipError_t launch_select_and_order(const RenderArgs& A, int* cost, unsigned int* order, unsigned char* flags, unsigned int* long_list, hipStream_t st, int long_sum, int solo_sum);   // rt_kernels.hip

// the pilot pass alone (rt_split_balanced): per tile 4 x the bounces of its pilot samples
hipError_t launch_pilot_h(const RenderArgs& A, bool tree, int* cost, hipStream_t st) {
    if (A.n_local_tiles <= 0) return hipSuccess;
    const unsigned blocks = (unsigned)((A.n_local_tiles + 7) / 8);                 // a wave covers two tiles
    const size_t lds = h16_lds_bytes(tree, A.tree);
    if (tree) hipLaunchKernelGGL((h16::k_tile_cost_h<true>), dim3(blocks), dim3(256), lds, st, A, cost, (unsigned char*)nullptr);
    else hipLaunchKernelGGL((h16::k_tile_cost_h<false>), dim3(blocks), dim3(256), lds, st, A, cost, (unsigned char*)nullptr);
    return hipGetLastError();
}

// the scheduling pre-pass of a binary16 render: pilot pass in binary16, then the precision-independent selection and ordering
hipError_t launch_tile_order_h(const RenderArgs& A, bool tree, int* cost, unsigned int* order, unsigned char* flags, unsigned int* long_list, hipStream_t st) {
    if (A.n_local_tiles <= 0) return hipSuccess;
    const unsigned blocks = (unsigned)((A.n_local_tiles + 7) / 8);                 // a wave covers two tiles
    const size_t lds = h16_lds_bytes(tree, A.tree);
    unsigned char* pilot = flags ? flags + (size_t)A.n_local_tiles * 64 : nullptr;
    if (tree) hipLaunchKernelGGL((h16::k_tile_cost_h<true>), dim3(blocks), dim3(256), lds, st, A, cost, pilot);
    else hipLaunchKernelGGL((h16::k_tile_cost_h<false>), dim3(blocks), dim3(256), lds, st, A, cost, pilot);
    return launch_select_and_order(A, cost, order, flags, long_list, st, RT_H16_PILOT_LONG_SUM, 0x7fffffff);
}

hipError_t launch_render_h(const RenderArgs& A, bool tree, int mode, hipStream_t st) {
    if (A.n_local_tiles <= 0) return hipSuccess;
    const unsigned need = (unsigned)((A.n_local_tiles + 3) / 4);
    const size_t lds = h16_lds_bytes(tree, A.tree);
    void (*k)(RenderArgs) = tree ? (mode == 0 ? h16::k_render_h<true, 0> : h16::k_render_h<true, 1>)
                                 : (mode == 0 ? h16::k_render_h<false, 0> : h16::k_render_h<false, 1>);
    const unsigned cap = resident_blocks_h((const void*)k, lds);
    const unsigned blocks = need < cap ? need : cap;
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), lds, st, A);
    return hipGetLastError();
}

hipError_t launch_trace_h(const DevScene& S, const DevTree& T, bool tree, const float* rays, long long n, rt_hit_record* out, hipStream_t st) {
    if (n <= 0) return hipSuccess;
    const unsigned blocks = (unsigned)((n + 255) / 256);
    const size_t lds = h16_lds_bytes(tree, T);
    if (tree) hipLaunchKernelGGL((h16::k_trace_h<true>), dim3(blocks), dim3(256), lds, st, S, T, rays, n, out);
    else hipLaunchKernelGGL((h16::k_trace_h<false>), dim3(blocks), dim3(256), lds, st, S, T, rays, n, out);
    return hipGetLastError();
}

#ifdef RT_H16_STATS
hipError_t read_h16_stats(unsigned long long* out, int reset) {
    hipError_t e = hipMemcpyFromSymbol(out, HIP_SYMBOL(h16::g_h16_cyc), sizeof(unsigned long long) * 8);
    if (e != hipSuccess) return e;
    if (reset) { unsigned long long z[8] = {0}; e = hipMemcpyToSymbol(HIP_SYMBOL(h16::g_h16_cyc), z, sizeof(z)); }
    return e;
}
#endif

hipError_t launch_assemble_h(void* full, const void* parts, int max_x, int max_y, int nparts, hipStream_t st) {
    const int tiles_x = (max_x + 7) / 8, tiles_y = (max_y + 7) / 8;
    const long long tiles = (long long)tiles_x * tiles_y;
    const long long per_part = part_local_tiles(tiles, 0, nparts) * 64;
    const unsigned blocks = (unsigned)((tiles + 3) / 4);
    hipLaunchKernelGGL(h16::k_assemble_h, dim3(blocks), dim3(256), 0, st, (uint16_t*)full, (const uint16_t*)parts, max_x, max_y, tiles_x, nparts, per_part, tiles);
    return hipGetLastError();
}

} // namespace rt
