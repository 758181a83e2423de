// rt_kernels.hip — hand-written HIP kernels of the render path for gfx950 (MI355X, CDNA4).
//
// Replaces the reference's device code: render_init / render / render_progressive (main.cu:84-142), color()
// (main.cu:43-75), hitTree / traverseTree / intersect_ray_aabb / processHit (acceleration_structure.h:226-342),
// hitable_list::hit (hitable_list.h:16-31), sphere::hit (sphere.h:17-46), the three material::scatter
// (material.h:55-113) and camera::get_ray (camera.h:45-49).
//
// Shape:
//   * one lane = one pixel (the per-pixel XORWOW stream is strictly serial); persistent waves pull pixels from a global
//     queue (one pixel position of 64 different tiles per wave: long chains cluster), most expensive tiles first;
//   * the reference's `for sample { for bounce {..} }` nest is flattened into ONE loop per lane: a lane whose path
//     ended starts its next sample at once instead of idling until the slowest path of the wave finishes;
//   * octree on: candidates come from an exact culling grid (rt_accel.h), the reference's traversal is the fallback; a wave's
//     sphere tests are pooled and dealt out evenly over its 64 lanes — walk_pool on sparse grids (two columns per ray and round),
//     walk_pool_dense on dense ones (one column, four entries per lane and pass, the owner's best hit re-read every pass);
//     waves holding long pixel chains stop refilling ("thin");
//   * no virtual calls, no device heap, no recursion: materials are a tag + 4 floats, the octree is a pre-order
//     node array with skip links staged in LDS, bucket contents are pre-gathered (centre, r^2) float4 streams;
//   * hitable_list path: the sphere index is wave-uniform, so sphere data comes through scalar loads (SGPR operands).
//
// Numeric contract: every floating-point operation below is one IEEE binary32 operation in the reference's
// order; FMA contraction is off for the whole file (and on the command line); division and sqrt are the
// correctly-rounded forms (hipcc default).  Values that are merely hoisted (a = d.d, radius^2) are bitwise the
// values the reference recomputes.
#include <hip/hip_runtime.h>
#include <float.h>
#include "rt_device.h"
#include "rt_tuning.h"

// RT_TU_CONTRACT (rt_kernels_contract.hip): the same source once more with FMA contraction ALLOWED — what nvcc's default -fmad=true
// does to the reference (Makefile:9) — in a namespace of its own, for rt_world_set_arith(RT_ARITH_CONTRACT).  Never the parity mode.
#ifdef RT_TU_CONTRACT
#pragma clang fp contract(fast)
#else
#pragma clang fp contract(off)
#endif

namespace rt {
#ifdef RT_TU_CONTRACT
namespace fmac {
#endif

#define RT_DEV static __device__ __forceinline__

#include "rt_stats.h"      // the diagnostic build's counters (-DRT_STATS): STAT / WPASS / RT_STATS_ONLY — nothing in a product build

struct Rng { uint32_t d, v0, v1, v2, v3, v4; };

// curand (XORWOW) + curand_uniform: x * 2^-32 + 2^-33, one rounding per operation
RT_DEV float rng_uniform(Rng& s) {
    const uint32_t t = s.v0 ^ (s.v0 >> 2);
    s.v0 = s.v1; s.v1 = s.v2; s.v2 = s.v3; s.v3 = s.v4;
    s.v4 = (s.v4 ^ (s.v4 << 4)) ^ (t ^ (t << 1));
    s.d += 362437u;
    const float x = (float)(s.d + s.v4);
    const float m = x * 2.3283064e-10f;
    return m + (2.3283064e-10f / 2.0f);
}

RT_DEV void rng_seed(Rng& s, unsigned long long seed) {       // curand_init(seed, 0, 0)
    const uint32_t lo = (uint32_t)seed ^ 0xaad26b49u, hi = (uint32_t)(seed >> 32) ^ 0xf7dcefddu;
    const uint32_t t0 = 1099087573u * lo, t1 = 2591861531u * hi;
    s.d = 6615241u + t1 + t0;
    s.v0 = 123456789u + t0; s.v1 = 362436069u ^ t0; s.v2 = 521288629u + t1; s.v3 = 88675123u ^ t1; s.v4 = 5783321u + t0;
}

struct V3 { float x, y, z; };
RT_DEV float dot3(const V3& a, const V3& b) { return a.x * b.x + a.y * b.y + a.z * b.z; }     // vec3.h:91, left to right

struct RayF { V3 o, d; };

// pow((1 - cosine), 5) of material.h:14 — contract: binary64 ((x*x)*(x*x))*x rounded once to binary32
RT_DEV float pow5(float x) { const double v = (double)x; const double v2 = v * v; return (float)((v2 * v2) * v); }

// ---------------------------------------------------------------------------------------------------- sphere::hit
// One candidate test against (cx,cy,cz,r2).  closest is closest_so_far; on acceptance it is lowered and `best`
// records which entry won.  Operation order of sphere.h:18-41 with a = d.d hoisted by the caller.
RT_DEV void sphere_test(const RayF& r, float a, float cx, float cy, float cz, float r2, int id, float& closest, int& best) {
    const float ocx = r.o.x - cx, ocy = r.o.y - cy, ocz = r.o.z - cz;
    const float b = ocx * r.d.x + ocy * r.d.y + ocz * r.d.z;
    const float c = (ocx * ocx + ocy * ocy + ocz * ocz) - r2;
    const float disc = b * b - a * c;
    if (disc > 0.0f) {
        const float sq = sqrtf(disc);
        float t = (-b - sq) / a;
        if (t < closest && t > 0.001f) { closest = t; best = id; }
        else {
            t = (-b + sq) / a;
            if (t < closest && t > 0.001f) { closest = t; best = id; }
        }
    }
}

// intersect_ray_aabb — acceleration_structure.h:226-244 (comparisons kept literally: NaN/inf fall through as there)
RT_DEV bool ray_box(const RayF& r, float lox, float loy, float loz, float hix, float hiy, float hiz) {
    float tmin = (lox - r.o.x) / r.d.x;
    float tmax = (hix - r.o.x) / r.d.x;
    if (tmin > tmax) { const float t = tmin; tmin = tmax; tmax = t; }
    float tymin = (loy - r.o.y) / r.d.y;
    float tymax = (hiy - r.o.y) / r.d.y;
    if (tymin > tymax) { const float t = tymin; tymin = tymax; tymax = t; }
    if ((tmin > tymax) || (tymin > tmax)) return false;
    if (tymin > tmin) tmin = tymin;
    if (tymax < tmax) tmax = tymax;
    float tzmin = (loz - r.o.z) / r.d.z;
    float tzmax = (hiz - r.o.z) / r.d.z;
    if (tzmin > tzmax) { const float t = tzmin; tzmin = tzmax; tzmax = t; }
    if ((tmin > tzmax) || (tzmin > tmax)) return false;
    return true;
}

// ---------------------------------------------------------------------------------------------------- closest hit
// Every kernel that traces rays takes a RenderArgs as its FIRST argument.  The launch arguments again, read from the kernarg segment at the point of use.  Fields taken from the by-value argument
// live in SGPRs for the whole kernel; the render loop is short of SGPRs (spills cost VALU slots: v_readlane/v_writelane),
// so everything that is only needed between pixels or once per bounce (camera, queue, order, framebuffer, materials) is
// re-read through this pointer (scalar loads, no VALU) — the empty asm keeps the compiler from hoisting those loads.
RT_DEV const RenderArgs* cold_args() {
    auto p = __builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(p));
    return (const RenderArgs*)p;
}

RT_DEV float bcast(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }
// Minimum over each group of LG = 64, 32 or 16 consecutive lanes, in every lane of the group.  DPP row shifts and row
// broadcasts (~10 cycles a step) instead of dependent ds_bpermute round trips (~100 cycles each): the cooperative paths run in
// waves with nothing else to do while a reduction is in flight.
template <int LG>
RT_DEV float group_min(float v) {
    const int inf = 0x7f800000;
    int x = __float_as_int(v);
#define RT_DPP_MIN(ctrl, rmask) x = __float_as_int(fminf(__int_as_float(x), __int_as_float(__builtin_amdgcn_update_dpp(inf, x, ctrl, rmask, 0xf, false))))
    RT_DPP_MIN(0x111, 0xf);     // row_shr:1
    RT_DPP_MIN(0x112, 0xf);     // row_shr:2
    RT_DPP_MIN(0x114, 0xf);     // row_shr:4
    RT_DPP_MIN(0x118, 0xf);     // row_shr:8   -> lane 15 of each row of 16 holds the row's minimum
    if (LG >= 32) RT_DPP_MIN(0x142, 0xa);     // row_bcast:15 into rows 1 and 3 -> lanes 31 and 63 hold their half's minimum
    if (LG >= 64) RT_DPP_MIN(0x143, 0xc);     // row_bcast:31 into rows 2 and 3 -> lane 63 holds the wave's minimum
#undef RT_DPP_MIN
    const int q3 = __builtin_amdgcn_readlane(x, 63);
    if (LG == 64) return __int_as_float(q3);
    const int q1 = __builtin_amdgcn_readlane(x, 31);
    if (LG == 32) return __int_as_float((threadIdx.x & 32) ? q3 : q1);
    const int q0 = __builtin_amdgcn_readlane(x, 15), q2 = __builtin_amdgcn_readlane(x, 47);
    const int g = (threadIdx.x >> 4) & 3;
    return __int_as_float(g == 0 ? q0 : g == 1 ? q1 : g == 2 ? q2 : q3);
}
RT_DEV int wave_min_int(int x) {                              // the same for int32, whole wave
    const int big = 0x7fffffff;
#define RT_DPP_MIN(ctrl, rmask) x = min(x, __builtin_amdgcn_update_dpp(big, x, ctrl, rmask, 0xf, false))
    RT_DPP_MIN(0x111, 0xf); RT_DPP_MIN(0x112, 0xf); RT_DPP_MIN(0x114, 0xf); RT_DPP_MIN(0x118, 0xf); RT_DPP_MIN(0x142, 0xa); RT_DPP_MIN(0x143, 0xc);
#undef RT_DPP_MIN
    return __builtin_amdgcn_readlane(x, 63);
}
RT_DEV int bcast(int v, int l) { return __builtin_amdgcn_readlane(v, l); }

// sphere::hit reduced to "which t would this sphere offer": the near root if it is > t_min, else the far root if that
// is > t_min, else none (+inf).  The reference accepts exactly when that value is < closest_so_far (the far root is
// never below the near root, so a rejected near root in range cannot be followed by an accepted far root).
RT_DEV float sphere_candidate(const RayF& r, float a, const float4 s) {
    const float ocx = r.o.x - s.x, ocy = r.o.y - s.y, ocz = r.o.z - s.z;
    const float b = ocx * r.d.x + ocy * r.d.y + ocz * r.d.z;
    const float c = (ocx * ocx + ocy * ocy + ocz * ocz) - s.w;
    const float disc = b * b - a * c;
    float cand = __builtin_inff();
    if (disc > 0.0f) {
        const float sq = sqrtf(disc);
        const float t1 = (-b - sq) / a;
        if (t1 > 0.001f) cand = t1;
        else {
            const float t2 = (-b + sq) / a;
            if (t2 > 0.001f) cand = t2;
        }
    }
    return cand;
}

// hitable_list::hit over the hittable spheres in list order.
// Full waves: the sphere index `k` is wave-uniform -> scalar loads, every lane tests its own ray against sphere k.
// Thin waves (few live rays, e.g. the tail of a frame): one ray at a time with the 64 lanes testing 64 spheres per step;
// the winner is the smallest offered t, the lowest list index among equal t — exactly what the sequential scan keeps.
RT_DEV void closest_list(const DevScene& S, const RayF& r, float a, bool live, float& closest, int& best) {
    const float4* __restrict__ hot = S.list_hot;
    const int n = S.n_list;
    unsigned long long todo = __ballot(live);
    if ((int)__popcll(todo) * RT_LIST_COOP_COST <= n) {
        const int lane = threadIdx.x & 63;
        while (todo != 0ull) {
            const int L = __ffsll((long long)todo) - 1;
            todo &= todo - 1ull;
            RayF q;
            q.o.x = bcast(r.o.x, L); q.o.y = bcast(r.o.y, L); q.o.z = bcast(r.o.z, L);
            q.d.x = bcast(r.d.x, L); q.d.y = bcast(r.d.y, L); q.d.z = bcast(r.d.z, L);
            const float qa = bcast(a, L);
            float my_t = FLT_MAX; int my_k = 0x7fffffff;
            for (int base = 0; base < n; base += 256) {          // four spheres per lane and pass, their loads in flight together
                float4 sv[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) { const int k = base + u * 64 + lane; sv[u] = hot[k < n ? k : n - 1]; }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int k = base + u * 64 + lane;
                    if (k < n) {
                        const float cand = sphere_candidate(q, qa, sv[u]);
                        if (cand < my_t) { my_t = cand; my_k = k; }
                    }
                }
            }
            const float mn = group_min<64>(my_t);
            const int km = wave_min_int((my_t == mn && my_k != 0x7fffffff) ? my_k : 0x7fffffff);
            if (lane == L && km != 0x7fffffff) { closest = mn; best = S.list_id[km]; }
        }
        return;
    }
    int k = 0;
    for (; k + RT_LIST_BATCH <= n; k += RT_LIST_BATCH) {         // several spheres per pass: their (scalar) loads in flight together
        float4 sv[RT_LIST_BATCH];
#pragma unroll
        for (int q = 0; q < RT_LIST_BATCH; ++q) sv[q] = hot[k + q];
#pragma unroll
        for (int q = 0; q < RT_LIST_BATCH; ++q) sphere_test(r, a, sv[q].x, sv[q].y, sv[q].z, sv[q].w, k + q, closest, best);
    }
    for (; k < n; ++k) {
        const float4 s = hot[k];
        sphere_test(r, a, s.x, s.y, s.z, s.w, k, closest, best);
    }
    if (best >= 0) best = S.list_id[best];
}

// The reference scan: the tree in the reference's depth-first order (pre-order array + skip links), every bucket entry
// of every visited level-3 node.  While-while form: each lane first advances to its next non-empty level-3 node that
// passes the slab test, then scans that node's entries.  `closest`/`best` come in holding the ground-sphere result.
RT_DEV void tree_scan(const float4* s_nodes, const RayF& r, float a, bool live, float& closest, int& best) {
    const DevTree& T = cold_args()->tree;
    if (T.n_nodes == 1) {
        // the list seen as one unbounded node (rt_api.hip build_list_tree): all n_entries spheres are this ray's to test.  Few
        // rays come here (outside the near zone, ties), so one at a time with lanes = spheres as in closest_list — the smallest
        // offered t, the lowest entry among equal t, and only if it beats what the ray holds (the ground was tested first)
        const float4* __restrict__ hot = T.ent_hot;
        const int n = T.n_entries;
        const int lane = threadIdx.x & 63;
        unsigned long long todo = __ballot(live);
        if ((int)__popcll(todo) * RT_LIST_COOP_COST <= n) {
            while (todo != 0ull) {
                const int L = __ffsll((long long)todo) - 1;
                todo &= todo - 1ull;
                RayF q;
                q.o.x = bcast(r.o.x, L); q.o.y = bcast(r.o.y, L); q.o.z = bcast(r.o.z, L);
                q.d.x = bcast(r.d.x, L); q.d.y = bcast(r.d.y, L); q.d.z = bcast(r.d.z, L);
                const float qa = bcast(a, L);
                float my_t = FLT_MAX; int my_k = 0x7fffffff;
                for (int base = 0; base < n; base += 256) {
                    float4 sv[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) { const int k = base + u * 64 + lane; sv[u] = hot[k < n ? k : n - 1]; }
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int k = base + u * 64 + lane;
                        if (k < n) {
                            const float cand = sphere_candidate(q, qa, sv[u]);
                            if (cand < my_t) { my_t = cand; my_k = k; }
                        }
                    }
                }
                const float mn = group_min<64>(my_t);
                const int km = wave_min_int((my_t == mn && my_k != 0x7fffffff) ? my_k : 0x7fffffff);
                if (lane == L && km != 0x7fffffff && mn < closest) { closest = mn; best = T.ent_id[km]; }
            }
            return;
        }
    }
    int e_best = -1;
    int node = live ? 0 : T.n_nodes;
    const int n_nodes = T.n_nodes;
    while (true) {
        int e = 0, e_end = 0;
        while (node < n_nodes) {
            const float4 n0 = s_nodes[node * 3 + 0];
            const float4 n1 = s_nodes[node * 3 + 1];
            const float4 n2 = s_nodes[node * 3 + 2];
            const int skip = __float_as_int(n1.z);
            if (ray_box(r, n0.x, n0.y, n0.z, n0.w, n1.x, n1.y)) {
                const int cnt = __float_as_int(n2.x);
                node = node + 1;
                if (cnt > 0) { e = __float_as_int(n1.w); e_end = e + cnt; break; }
            } else {
                node = skip;
            }
        }
        if (e >= e_end) break;          // no node left for this lane (it waits at the loop exit for the rest of the wave)
        while (e < e_end) {                             // four entries per pass, their loads in flight together; tested in order
            const int m = e_end - e;
            const float4 s0 = T.ent_hot[e], s1 = T.ent_hot[m > 1 ? e + 1 : e], s2 = T.ent_hot[m > 2 ? e + 2 : e], s3 = T.ent_hot[m > 3 ? e + 3 : e];
            sphere_test(r, a, s0.x, s0.y, s0.z, s0.w, e, closest, e_best);
            if (m > 1) sphere_test(r, a, s1.x, s1.y, s1.z, s1.w, e + 1, closest, e_best);
            if (m > 2) sphere_test(r, a, s2.x, s2.y, s2.z, s2.w, e + 2, closest, e_best);
            if (m > 3) sphere_test(r, a, s3.x, s3.y, s3.z, s3.w, e + 3, closest, e_best);
            e += m > 4 ? 4 : m;
        }
    }
    if (e_best >= 0) best = T.ent_id[e_best];
}

// Is the sphere stored in a level-3 node that the reference's traversal visits for this ray?  With no zero direction
// component a node's slab test passing implies all its ancestors' tests pass (their intervals contain the child's),
// so the node's own test — the reference's arithmetic — decides.
RT_DEV bool eligible(const float4* s_nodes, const RayF& r, float cand, int sphere STAT_ARG) {
    const DevTree& T = cold_args()->tree;
    STAT(st, ST_ELIG, 1); WPASS(WP_ELIG_FN);
    // Shortcut for spheres stored in several nodes (the big ones): the level-3 cell that contains the hit point.  If the
    // point keeps 0.012 from all six faces of that cell, the hit's t lies inside all three float slab intervals of the
    // cell (rounding errors are ~1e-5), so that node's slab test passes; it remains to see that the sphere is stored
    // there (one bit of its membership bitmap).  Two dependent loads instead of a scan of the membership list.
    {
        const int row = T.acc.bits_index[sphere];
        const float px = r.o.x + cand * r.d.x, py = r.o.y + cand * r.d.y, pz = r.o.z + cand * r.d.z;
        const int ix = (int)floorf((px + kRootHalfXZf) * kInvCellXZf), iy = (int)floorf(py * kInvCellYf), iz = (int)floorf((pz + kRootHalfXZf) * kInvCellXZf);
        if (row >= 0 && ix >= 0 && ix < 8 && iy >= 0 && iy < 8 && iz >= 0 && iz < 8) {
            const int cell = ix * 64 + iy * 8 + iz;
            const int node = T.acc.cellnode[cell];
            const uint32_t word = T.acc.cellbits[row * 16 + (cell >> 5)];
            if (node >= 0 && ((word >> (cell & 31)) & 1u)) {
                const float4 n0 = s_nodes[node * 3 + 0];
                const float4 n1 = s_nodes[node * 3 + 1];
                const float m = 0.012f;
                if (px > n0.x + m && px < n0.w - m && py > n0.y + m && py < n1.x - m && pz > n0.z + m && pz < n1.y - m) return true;
            }
        }
    }
    const int mb = T.acc.memb_start[sphere], me = T.acc.memb_start[sphere + 1];
    for (int k = mb; k < me; ++k) {
        const int node = T.acc.memb_cell[k];
        STAT(st, ST_ELIG_NODES, 1); WPASS(WP_ELIG_LIST);
        const float4 n0 = s_nodes[node * 3 + 0];
        const float4 n1 = s_nodes[node * 3 + 1];
        if (ray_box(r, n0.x, n0.y, n0.z, n0.w, n1.x, n1.y)) return true;
    }
    return false;
}

// Does the hit at `cand` lie inside the sphere's brick (rt_accel.h: the box of level-3 cells that ALL store the sphere, in
// cell coordinates, margins included)?  Then it lies in a stored cell whose three float slab intervals contain the hit's t
// (the hit point keeps 0.002 from the brick's outer faces, rounding moves an interval end by < 1e-5; inner faces are shared
// planes, so the cells' intervals tile the brick's without gaps): that node passes the reference's slab test, and with it
// its ancestors — no division needed.  Approximate arithmetic is fine here: its error is part of the 1e-5.
RT_DEV bool in_brick(const RayF& r, float cand, const float4 blo, const float4 bhi) {
    const float ux = __builtin_fmaf(__builtin_fmaf(cand, r.d.x, r.o.x), kInvCellXZf, 4.0f);      // (x + 11) / 2.75 = x / 2.75 + 4: cell coordinates 0..8
    const float uy = __builtin_fmaf(cand, r.d.y, r.o.y) * kInvCellYf;
    const float uz = __builtin_fmaf(__builtin_fmaf(cand, r.d.z, r.o.z), kInvCellXZf, 4.0f);
    return ux > blo.x && ux < bhi.x && uy > blo.y && uy < bhi.y && uz > blo.z && uz < bhi.z;
}

// blo/bhi: the candidate's DevAccel::brick pair (brick bounds, world-list index, single storing node or -1)
RT_DEV void offer(const DevTree& T, const float4* s_nodes, const RayF& r, float cand, const float4 blo, const float4 bhi, float& best_t, int& best, bool& tie STAT_ARG) {
    STAT(st, ST_OFFERS, 1);
    const int id = __float_as_int(blo.w);
    if (cand < best_t) {
        bool ok = in_brick(r, cand, blo, bhi);
        if (!ok) {
            // hit point near an outer face of the brick, or no brick (a bucket was full): the reference's own slab test
            const int node1 = __float_as_int(bhi.w);
            if (node1 >= 0) {
                STAT(st, ST_ELIG, 1); STAT(st, ST_ELIG_NODES, 1); WPASS(WP_OFFER_RAYBOX);
                const float4 n0 = s_nodes[node1 * 3 + 0];
                const float4 n1 = s_nodes[node1 * 3 + 1];
                ok = ray_box(r, n0.x, n0.y, n0.z, n0.w, n1.x, n1.y);
            } else {
                ok = eligible(s_nodes, r, cand, id STAT_PASS);
            }
        }
        if (ok) { best_t = cand; best = id; }
    } else if (cand == best_t && id != best && best > 0) {
        tie = true;                      // two different tree spheres at the same float t: visit order decides -> reference scan
    }
}

// Fast path: large spheres directly, small spheres through the (x,z) grid along the ray's projection,
// clipped to the y-slab that holds them, front to back, ending at the column that lies beyond the best hit.  The walks
// (walk_pool for sparse grids, walk_pool_dense for dense ones) pool the sphere tests of a wave's rays and deal them out evenly
// over its 64 lanes; an exact tie between two tree spheres sends the ray to the reference scan.
struct Walk {                // a ray's walk over the grid, in cell units along its major axis
    int i, iend, coff;      // next column, end (exclusive, in travel direction), offset of the x- or z-major grid copy
    float om_c, on_c, slope, dm_c;
    bool fwd, walking;
    unsigned work, cols;    // grid entries this ray's columns have put into the wave's pool, and the columns it stepped through (read by the pilot pass only:
                            // a tile's cost in time — a ray along the horizon crosses hundreds of columns, most of them empty)
};

// last column worth visiting once a hit at best_t is known: the one that holds the hit point.  A sphere that offers a smaller t has its
// own (float) hit point before this one on the line and within R' of its centre — inside its registered extent,
// so it is filed under the column of that point, which the walk has passed or is in.  (Rounds 1-3 walked on until a column's entry
// edge lay R' beyond the hit, and began R' behind the origin: one column too many at either end of every walk — C5 450 -> 354 ms.)
RT_DEV void walk_clip(Walk& W, const DevAccel& A, float best_t) {
    const float fG = (float)A.G, back_c = 2e-3f * A.inv_h;     // (the float error of pm: ~1e-5 cells)
    const float pm = fminf(fmaxf(W.om_c + best_t * W.dm_c, -4.0f), fG + 4.0f);
    if (W.fwd) W.iend = min(W.iend, (int)floorf(pm + back_c) + 1);
    else W.iend = max(W.iend, (int)ceilf(pm - back_c - 1.0f) - 1);
    if (W.fwd ? (W.i >= W.iend) : (W.i <= W.iend)) W.i = W.iend;
}

RT_DEV Walk walk_setup(const DevAccel& A, const RayF& r, float best_t, int best) {
    Walk W;
    W.work = 0u; W.cols = 0u;
    const int G = A.G;
    const float fG = (float)G;
    const float slack = 2e-3f;
    const bool xmajor = fabsf(r.d.x) >= fabsf(r.d.z);
    const float om = xmajor ? r.o.x : r.o.z, on = xmajor ? r.o.z : r.o.x;
    const float dm = xmajor ? r.d.x : r.d.z, dn = xmajor ? r.d.z : r.d.x;
    W.coff = xmajor ? 0 : A.zoff;
    // everything below is in cell units (cell i spans [i, i+1) along either axis)
    W.om_c = (om - A.g0) * A.inv_h; W.on_c = (on - A.g0) * A.inv_h;
    W.dm_c = dm * A.inv_h;
    // the columns that matter are those in which a hit point can lie (a sphere is filed wherever its hit points can be): the stretch of
    // the line inside the spheres' y-slab, from the origin on; s_c covers the float error of these few operations
    const float s_c = slack * A.inv_h;
    // where the line crosses the planes y = ylo / y = yhi, measured along the major axis
    // the walk's own parameters need no exact division: a few ulp are far inside the rasterisation slack
    const float rmy = W.dm_c * __builtin_amdgcn_rcpf(r.d.y);
    const float mA = W.om_c + (A.ylo - r.o.y) * rmy, mB = W.om_c + (A.yhi - r.o.y) * rmy;
    float mlo = fminf(mA, mB) - s_c, mhi = fmaxf(mA, mB) + s_c;
    W.fwd = dm > 0.0f;
    if (W.fwd) mlo = fmaxf(mlo, W.om_c - s_c); else mhi = fminf(mhi, W.om_c + s_c);     // no hit point lies behind the origin (t > t_min > 0)
    int ilo = (int)floorf(fminf(fmaxf(mlo, -1.0f), fG)), ihi = (int)floorf(fminf(fmaxf(mhi, -1.0f), fG));
    W.walking = !(ihi < 0 || ilo > G - 1 || !(mlo <= mhi));
    ilo = max(ilo, 0); ihi = min(ihi, G - 1);
    W.i = W.fwd ? ilo : ihi;
    W.iend = (W.fwd ? ihi : ilo) + (W.fwd ? 1 : -1);
    W.slope = dn * __builtin_amdgcn_rcpf(dm);
    if (best >= 0) walk_clip(W, A, best_t);
    if (W.i == W.iend || (W.fwd ? (W.i > W.iend) : (W.i < W.iend))) W.walking = false;
    return W;
}

// ---------------------------------------------------------------------------------------------------- the walk of a full wave
// Letting every lane test the spheres of ITS columns (rounds 1-2: walk_lanes, removed in round 3 with the cooperative single-ray
// walk once no launch reached them) ran at 47 % busy lanes, 3.5 of 6 test slots used, lanes holding a candidate waiting for a
// vote.  walk_pool spreads the wave's sphere tests evenly instead (the scheme of the binary16 scan, rt_kernels_fp16.hip): per round every walking lane puts the entry ranges of its next RT_POOL_COLS columns
// into the wave's pool (LDS); a prefix over the ranges and a binary search give every lane an equal span of the concatenated
// entries, which it tests — 18-operation discriminant and square-root-free pre-filter as in phase A — against the OWNER's ray
// (LDS) whoever that is; survivors go into a queue of the wave and are resolved 64 at a time (exact roots, brick / slab
// eligibility as in phase B), the owner's best hit kept as a 64-bit key (t bits << 32 | sphere index) by an LDS atomic
// min.  An atomic min that meets an equal t from another sphere flags the exact tie (-> reference scan), as `offer` does.
// Nothing here decides a hit differently: the same candidates' exact values, merged by a minimum instead of one after the other;
// testing MORE spheres than a clipped per-ray walk would (no clipping inside a round) cannot change the minimum (App. A.4).
static_assert(RT_POOL_COLS_THIN >= RT_POOL_COLS, "a thin round takes at least as many columns as a full one");
constexpr int kWalkPool = 64 * RT_POOL_COLS;                  // segments per wave and round
constexpr int kWalkCand = 128;
struct WalkLds {
    uint2 seg[kWalkPool];                // (first entry, count | owner << 26)
    unsigned pref[kWalkPool];            // exclusive prefix of the counts
    float4 ray[128];                     // per owner: (o.x, o.y, o.z, d.x) (d.y, d.z, a, f_abt)
    unsigned long long key[64];          // per owner: bits of the best t << 32 | sphere index (low word ~0: none yet)
    uint4 cq[kWalkCand];                 // candidates: (bits of b, bits of disc, entry, owner)
    unsigned count, tie_lo, tie_hi, pad_;
};
struct PoolSpan { unsigned cur, end, seg_end, base, sg; int owner; RayF q; float a, abt, atm; };      // a lane's walk through one span of the pool
RT_DEV void walk_sync() { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier(); }
// exclusive prefix sum over the 64 lanes: row_shr 1/2/4/8 (a row of 16), then row_bcast 15 and 31 — six DPP adds, no LDS
RT_DEV unsigned walk_excl_scan(unsigned v, unsigned& total) {
    int x = (int)v;
    x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false);     // row_bcast:15 into rows 1 and 3
    x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false);     // row_bcast:31 into rows 2 and 3
    total = (unsigned)__builtin_amdgcn_readlane(x, 63);
    return (unsigned)x - v;
}

// one queued candidate: sphere.h:24-43's offer, eligibility as in `offer`, merged into the owner's key
RT_DEV void pool_candidate_b(const DevTree& T, const float4* s_nodes, WalkLds& L, const uint4 c, const float4 blo, const float4 bhi STAT_ARG) {
    const int owner = (int)c.w;
    const float b = __uint_as_float(c.x), disc = __uint_as_float(c.y);
    const float4 r0 = L.ray[2 * owner], r1 = L.ray[2 * owner + 1];
    RayF q; q.o = {r0.x, r0.y, r0.z}; q.d = {r0.w, r1.x, r1.y};
    const float a = r1.z;
    const unsigned long long k0 = L.key[owner];
    const float best_t = __uint_as_float((unsigned)(k0 >> 32));
    const int best = (int)(unsigned)k0;
    const float sq = sqrtf(disc);
    float cand = __builtin_inff();
    const float t1 = (-b - sq) / a;
    if (t1 > 0.001f) cand = t1;
    else { const float t2 = (-b + sq) / a; if (t2 > 0.001f) cand = t2; }
    if (!(cand <= best_t)) return;
    const int id = __float_as_int(blo.w);
    STAT(st, ST_OFFERS, 1);
    bool tie = false;
    if (cand < best_t) {
        bool ok = in_brick(q, cand, blo, bhi);
        if (!ok) {
            const int node1 = __float_as_int(bhi.w);
            if (node1 >= 0) {
                const float4 n0 = s_nodes[node1 * 3 + 0], n1 = s_nodes[node1 * 3 + 1];
                ok = ray_box(q, n0.x, n0.y, n0.z, n0.w, n1.x, n1.y);
            } else ok = eligible(s_nodes, q, cand, id STAT_PASS);
        }
        if (ok) {
            const unsigned long long old = atomicMin(&L.key[owner], ((unsigned long long)__float_as_uint(cand) << 32) | (unsigned)id);
            // an equal t from another tree sphere (index > 0: the ground wins its ties in the reference too): the visit order decides
            tie = (unsigned)(old >> 32) == __float_as_uint(cand) && (int)(unsigned)old != id && (int)(unsigned)old > 0;
        }
    } else tie = id != best && best > 0;
    if (tie) { if (owner < 32) atomicOr(&L.tie_lo, 1u << owner); else atomicOr(&L.tie_hi, 1u << (owner - 32)); }
}

RT_DEV void pool_candidate(const DevTree& T, const float4* s_nodes, WalkLds& L, const uint4 c STAT_ARG) {
    // (the brick is asked for before the roots are formed: its round trip overlaps the square root and the divisions — a candidate that
    // cannot win has fetched it for nothing, a chain alone in its wave has one dependent round trip less per candidate)
    const int e = (int)c.z;
    const float4 blo = T.acc.brick[2 * e], bhi = T.acc.brick[2 * e + 1];
    pool_candidate_b(T, s_nodes, L, c, blo, bhi STAT_PASS);
}

RT_DEV void pool_drain(const DevTree& T, const float4* s_nodes, WalkLds& L, int lane, unsigned& qn STAT_ARG) {
    walk_sync();
    const unsigned cnt = min(qn, (unsigned)kWalkCand);
    for (unsigned base = 0; base < cnt; base += 64u) {
        STAT(st, ST_B_ROUNDS_WAVE, 1);
        const unsigned i = base + (unsigned)lane;
        if (i < cnt) { STAT(st, ST_B_LANES, 1); pool_candidate(T, s_nodes, L, L.cq[i] STAT_PASS); }
    }
    qn = 0u;
    walk_sync();
}

template <int RT_QUORUM_DEN>
RT_DEV void walk_pool(const DevTree& T, const float4* s_nodes, WalkLds& L, const RayF& r, float a, Walk& W, float& best_t, int& best, bool& tie STAT_ARG) {
    const DevAccel& A = T.acc;
    const int32_t* __restrict__ cs = A.cs;
    const float4* __restrict__ hot = A.hot;
    const int lane = threadIdx.x & 63;
    // a column's bins (rt_accel.h): Gf fine bins of width 1 / F cells, entries keyed by their centre — the query grows by q_c
    const int Gf = A.Gf;
    const float fF = (float)A.F, fGf = (float)Gf, fGfm = fGf - 0.5f, q_c = A.rq_c;
    const int step = W.fwd ? 1 : -1;
    const float f_atm = a * (0.001f * 0.9999f - 1e-6f);
    bool walking = W.walking && W.i != W.iend;
    const int nw0 = __popcll(__ballot(walking));
    L.ray[2 * lane] = make_float4(r.o.x, r.o.y, r.o.z, r.d.x);
    while (true) {
        // ---- this lane's state for whoever tests its spheres; pool and flags cleared
        L.ray[2 * lane + 1] = make_float4(r.d.y, r.d.z, a, __builtin_fmaf(a * best_t, 1.0001f, 1e-6f * a));
        L.key[lane] = ((unsigned long long)__float_as_uint(best_t) << 32) | (unsigned)best;
        if (lane == 0) { L.tie_lo = 0u; L.tie_hi = 0u; }
        walk_sync();
        // ---- phase 1: the entry ranges of the next RT_POOL_COLS columns of every walking lane (their cell-start loads in flight together)
        // (a wave with few walkers — a thin wave's chains, the frame's critical path — takes RT_POOL_COLS_THIN columns per round: as
        // many cell-start loads in flight, fewer rounds and their dependent round trips; the pool holds 64 x RT_POOL_COLS ranges)
        const int cols_now = nw0 * RT_POOL_COLS_THIN <= kWalkPool ? RT_POOL_COLS_THIN : RT_POOL_COLS;
        int e0[RT_POOL_COLS_THIN], e1[RT_POOL_COLS_THIN];
#pragma unroll
        for (int c = 0; c < RT_POOL_COLS_THIN; ++c) {
            e0[c] = 0; e1[c] = 0;
            if (c < cols_now && walking && W.i != W.iend) {
                STAT(st, ST_COLS, 1);
                const float u0 = W.on_c + ((float)W.i - W.om_c) * W.slope, u1 = u0 + W.slope;      // the line at the column's two edges
                const float lo = (fminf(u0, u1) - q_c) * fF, hi = (fmaxf(u0, u1) + q_c) * fF;      // in fine bins (F is a power of two)
                if (hi >= 0.0f && lo < fGf) {
                    const int k0 = (int)fmaxf(lo, 0.0f), k1 = (int)fminf(hi, fGfm);
                    const unsigned cbase = (unsigned)(W.coff + W.i * Gf);
                    e0[c] = cs[cbase + (unsigned)k0]; e1[c] = cs[cbase + (unsigned)k1 + 1u];
                }
                W.i += step; ++W.cols;
            }
        }
        unsigned n_seg = 0u;                                         // (<= 64 x RT_POOL_COLS: the pool cannot overflow)
#pragma unroll
        for (int c = 0; c < RT_POOL_COLS_THIN; ++c) {
            if (c >= RT_POOL_COLS && c >= cols_now) break;          // (wave-uniform)
            const bool has = e1[c] > e0[c];
            const unsigned long long m = __ballot(has);
            if (has) {
                const unsigned slot = n_seg + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
                L.seg[slot] = make_uint2((unsigned)e0[c], (unsigned)(e1[c] - e0[c]) | ((unsigned)lane << 26));
                W.work += (unsigned)(e1[c] - e0[c]);
            }
            n_seg += (unsigned)__popcll(m);
        }
        walk_sync();
        if (n_seg != 0u) {
            // ---- phase 2: prefix of the ranges, an equal span of the concatenated entries per lane
            unsigned total = 0u;
            {
                unsigned run = 0u;
#pragma unroll
                for (int k = 0; k < RT_POOL_COLS; ++k) {
                    const unsigned sidx = (unsigned)(k * 64 + lane);
                    const unsigned cnt = sidx < n_seg ? (L.seg[sidx].y & 0x3ffffffu) : 0u;
                    unsigned tot;
                    const unsigned ex = walk_excl_scan(cnt, tot);
                    if (sidx < n_seg) L.pref[sidx] = run + ex;
                    run += tot;
                }
                total = run;
            }
            walk_sync();
            // RT_POOL_SPANS spans per lane, stepped through side by side: as many entry loads in flight per lane
            const unsigned C = (total + 64u * RT_POOL_SPANS - 1u) / (64u * RT_POOL_SPANS);
            PoolSpan sp[RT_POOL_SPANS];
#pragma unroll
            for (int j = 0; j < RT_POOL_SPANS; ++j) {
                PoolSpan& S = sp[j];
                const unsigned begin = min((unsigned)(lane + 64 * j) * C, total);
                S.end = min(begin + C, total); S.cur = begin; S.seg_end = begin; S.base = 0u; S.sg = 0u; S.owner = -1;
                S.q.o = {0.f, 0.f, 0.f}; S.q.d = {0.f, 1.f, 0.f}; S.a = 1.0f; S.abt = 0.0f; S.atm = 0.0f;
                if (n_seg <= 4u) {
                    // a thin wave's round: a handful of ranges — three independent LDS reads instead of a chain of eight
                    const unsigned p1 = L.pref[min(1u, n_seg - 1u)], p2 = L.pref[min(2u, n_seg - 1u)], p3 = L.pref[min(3u, n_seg - 1u)];
                    S.sg = (n_seg > 1u && p1 <= begin ? 1u : 0u) + (n_seg > 2u && p2 <= begin ? 1u : 0u) + (n_seg > 3u && p3 <= begin ? 1u : 0u);
                } else if (begin < S.end) {
                    unsigned lo = 0u, hi = n_seg;
#pragma unroll
                    for (int it = 0; it < 8; ++it) {                 // kWalkPool <= 256
                        const unsigned mid = (lo + hi) >> 1;
                        if (hi - lo > 1u) { if (L.pref[mid] <= begin) lo = mid; else hi = mid; }
                    }
                    S.sg = lo;
                }
            }
            // ---- phase 3: the tests
            unsigned qn = 0u;
            if (RT_POOL_DIRECT && RT_POOL_SPANS == 1 && total <= 64u) {
                // A round of at most one entry per lane (a thin wave's chains: the frame's critical path): the entry comes with its brick,
                // and the lane that finds a candidate resolves it on the spot — no queue, and the brick's round trip rides on the entry's
                // (a lone bounce: three dependent cache round trips in the walk become two).
                PoolSpan& S = sp[0];
                const bool act = S.cur < S.end;
                STAT(st, ST_A_ITERS_WAVE, 1);
                unsigned he = 0u;
                if (act) {
                    const uint2 sd = L.seg[S.sg];
                    const unsigned p0 = L.pref[S.sg];
                    he = sd.x - p0 + S.cur;
                    S.owner = (int)(sd.y >> 26);
                    const float4 r0 = L.ray[2 * S.owner], r1 = L.ray[2 * S.owner + 1];
                    S.q.o = {r0.x, r0.y, r0.z}; S.q.d = {r0.w, r1.x, r1.y}; S.a = r1.z; S.abt = r1.w;
                    S.atm = S.a * (0.001f * 0.9999f - 1e-6f);
                }
                const float4 s4 = hot[he], blo = A.brick[2 * he], bhi = A.brick[2 * he + 1];
                STAT(st, ST_TESTS, act ? 1 : 0); STAT(st, ST_A_LANE_STEPS, act ? 1 : 0);
                const float ocx = S.q.o.x - s4.x, ocy = S.q.o.y - s4.y, ocz = S.q.o.z - s4.z;
                const float b = ocx * S.q.d.x + ocy * S.q.d.y + ocz * S.q.d.z;
                const float c = (ocx * ocx + ocy * ocy + ocz * ocz) - s4.w;
                const float disc = b * b - S.a * c;
                const float ab = fabsf(b);
                const float Lm = __builtin_fmaf(ab, -1e-4f, -b) - S.abt, M = __builtin_fmaf(ab, -1e-4f, b) + S.atm;
                const float P = fmaxf(Lm, M);
                if (act && disc > 0.0f && !(P > 0.0f && P * P > disc * 1.0003f)) {
                    STAT(st, ST_DISCPOS, 1);
                    pool_candidate_b(T, s_nodes, L, make_uint4(__float_as_uint(b), __float_as_uint(disc), he, (unsigned)S.owner), blo, bhi STAT_PASS);
                }
            } else                                          // (pool_drain below, with nothing queued, is the fence before the owners pick up)
            while (true) {
                bool any = false;
#pragma unroll
                for (int j = 0; j < RT_POOL_SPANS; ++j) any = any || sp[j].cur < sp[j].end;
                if (__ballot(any) == 0ull) break;
                STAT(st, ST_A_ITERS_WAVE, 1);
                bool act[RT_POOL_SPANS]; unsigned he[RT_POOL_SPANS]; float4 s4[RT_POOL_SPANS];
#pragma unroll
                for (int j = 0; j < RT_POOL_SPANS; ++j) {
                    PoolSpan& S = sp[j];
                    act[j] = S.cur < S.end;
                    if (act[j] && S.cur >= S.seg_end) {
                        const uint2 sd = L.seg[S.sg];
                        const unsigned p0 = L.pref[S.sg];
                        ++S.sg;
                        S.seg_end = p0 + (sd.y & 0x3ffffffu);
                        S.base = sd.x - p0;
                        const int ow = (int)(sd.y >> 26);
                        if (ow != S.owner) {
                            S.owner = ow;
                            const float4 r0 = L.ray[2 * ow], r1 = L.ray[2 * ow + 1];
                            S.q.o = {r0.x, r0.y, r0.z}; S.q.d = {r0.w, r1.x, r1.y}; S.a = r1.z; S.abt = r1.w;
                            S.atm = S.a * (0.001f * 0.9999f - 1e-6f);
                        }
                    }
                    he[j] = act[j] ? S.base + S.cur : 0u;
                    s4[j] = hot[he[j]];
                }
                bool hold[RT_POOL_SPANS]; float hb[RT_POOL_SPANS], hd[RT_POOL_SPANS];
                bool anyhold = false;
#pragma unroll
                for (int j = 0; j < RT_POOL_SPANS; ++j) {
                    const PoolSpan& S = sp[j];
                    STAT(st, ST_TESTS, act[j] ? 1 : 0); STAT(st, ST_A_LANE_STEPS, act[j] ? 1 : 0);
                    const float ocx = S.q.o.x - s4[j].x, ocy = S.q.o.y - s4[j].y, ocz = S.q.o.z - s4[j].z;
                    const float b = ocx * S.q.d.x + ocy * S.q.d.y + ocz * S.q.d.z;
                    const float c = (ocx * ocx + ocy * ocy + ocz * ocz) - s4[j].w;
                    const float disc = b * b - S.a * c;
                    // App. A.5: near root beyond the best hit (Lm) / far root behind t_min (M); never both positive: one test of the larger
                    const float ab = fabsf(b);
                    const float Lm = __builtin_fmaf(ab, -1e-4f, -b) - S.abt, M = __builtin_fmaf(ab, -1e-4f, b) + S.atm;
                    const float P = fmaxf(Lm, M);
                    hold[j] = act[j] && disc > 0.0f && !(P > 0.0f && P * P > disc * 1.0003f);
                    hb[j] = b; hd[j] = disc;
                    anyhold = anyhold || hold[j];
                    if (act[j]) ++sp[j].cur;
                }
                if (__ballot(anyhold) != 0ull) {
#pragma unroll
                    for (int j = 0; j < RT_POOL_SPANS; ++j) {
                        const unsigned long long hm = __ballot(hold[j]);
                        if (hm != 0ull) {
                            const unsigned slot = qn + __builtin_amdgcn_mbcnt_hi((unsigned)(hm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)hm, 0u));
                            if (hold[j]) {
                                STAT(st, ST_DISCPOS, 1);
                                const uint4 ce = make_uint4(__float_as_uint(hb[j]), __float_as_uint(hd[j]), he[j], (unsigned)sp[j].owner);
                                if (slot < (unsigned)kWalkCand) L.cq[slot] = ce;
                                else pool_candidate(T, s_nodes, L, ce STAT_PASS);
                            }
                            qn += (unsigned)__popcll(hm);
                        }
                    }
                }
                if (qn >= 64u) pool_drain(T, s_nodes, L, lane, qn STAT_PASS);
            }
            pool_drain(T, s_nodes, L, lane, qn STAT_PASS);
            (void)f_atm;
            // ---- every owner picks up its result
            const unsigned long long k = L.key[lane];
            best_t = __uint_as_float((unsigned)(k >> 32)); best = (int)(unsigned)k;
            const unsigned tmask = lane < 32 ? L.tie_lo : L.tie_hi;
            if ((tmask >> (lane & 31)) & 1u) tie = true;
        }
        if (walking && best >= 0) walk_clip(W, A, best_t);
        walking = walking && W.i != W.iend;
        const int left = __popcll(__ballot(walking));
        if (left == 0) break;
        if (RT_QUORUM_DEN > 0 && nw0 >= RT_QUORUM_MIN && left * RT_QUORUM_DEN <= nw0) break;     // the stragglers resume next bounce (ts.pending)
    }
    W.walking = walking;
}

// The pooled walk for DENSE grids (C5: ~30 entries per cell, 60-100 per column).  One column per walking lane and round; a lane
// steps through its span RT_DENSE_PB entries at a time (their loads in flight together) and filters
// against the owner's CURRENT best hit, re-read from the wave's LDS every pass — on a dense grid most of a column lies behind the
// first hit, and a filter that only knew the hit of the previous round would send all of it to the candidate queue.  The queue is
// drained from RT_DENSE_DRAIN candidates on, so that a found hit starts rejecting soon.
template <int RT_QUORUM_DEN>
RT_DEV void walk_pool_dense(const DevTree& T, const float4* s_nodes, WalkLds& L, const RayF& r, float a, Walk& W, float& best_t, int& best, bool& tie STAT_ARG) {
    constexpr int PB = RT_DENSE_PB;
    const DevAccel& A = T.acc;
    const int32_t* __restrict__ cs = A.cs;
    const float4* __restrict__ hot = A.hot;
    const int lane = threadIdx.x & 63;
    const int Gf = A.Gf;
    const float fF = (float)A.F, fGf = (float)Gf, fGfm = fGf - 0.5f, q_c = A.rq_c;
    const int step = W.fwd ? 1 : -1;
    bool walking = W.walking && W.i != W.iend;
    const int nw0 = __popcll(__ballot(walking));
    L.ray[2 * lane] = make_float4(r.o.x, r.o.y, r.o.z, r.d.x);
    while (true) {
        L.ray[2 * lane + 1] = make_float4(r.d.y, r.d.z, a, __builtin_fmaf(a * best_t, 1.0001f, 1e-6f * a));
        L.key[lane] = ((unsigned long long)__float_as_uint(best_t) << 32) | (unsigned)best;
        if (lane == 0) { L.tie_lo = 0u; L.tie_hi = 0u; }
        walk_sync();
        // ---- phase 1: the entry range of the next column of every walking lane
        int e0 = 0, e1 = 0;
        if (walking && W.i != W.iend) {
            STAT(st, ST_COLS, 1);
            const float u0 = W.on_c + ((float)W.i - W.om_c) * W.slope, u1 = u0 + W.slope;
            const float lo = (fminf(u0, u1) - q_c) * fF, hi = (fmaxf(u0, u1) + q_c) * fF;
            if (hi >= 0.0f && lo < fGf) {
                const int k0 = (int)fmaxf(lo, 0.0f), k1 = (int)fminf(hi, fGfm);
                const unsigned cbase = (unsigned)(W.coff + W.i * Gf);
                e0 = cs[cbase + (unsigned)k0]; e1 = cs[cbase + (unsigned)k1 + 1u];
            }
            W.i += step; ++W.cols;
        }
        const bool has = e1 > e0;
        const unsigned long long m = __ballot(has);
        const unsigned n_seg = (unsigned)__popcll(m);
        if (has) {
            const unsigned slot = __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
            L.seg[slot] = make_uint2((unsigned)e0, (unsigned)(e1 - e0) | ((unsigned)lane << 26));
            W.work += (unsigned)(e1 - e0);
        }
        walk_sync();
        if (n_seg != 0u) {
            // ---- phase 2: prefix of the ranges, an equal span of the concatenated entries per lane
            unsigned total;
            {
                const unsigned cnt = (unsigned)lane < n_seg ? (L.seg[lane].y & 0x3ffffffu) : 0u;
                const unsigned ex = walk_excl_scan(cnt, total);
                if ((unsigned)lane < n_seg) L.pref[lane] = ex;
            }
            walk_sync();
            const unsigned C = (total + 63u) / 64u;
            PoolSpan S;
            const unsigned begin = min((unsigned)lane * C, total);
            S.end = min(begin + C, total); S.cur = begin; S.seg_end = begin; S.base = 0u; S.sg = 0u; S.owner = lane;
            S.q.o = {0.f, 0.f, 0.f}; S.q.d = {0.f, 1.f, 0.f}; S.a = 1.0f; S.abt = 0.0f; S.atm = 0.0f;
            if (begin < S.end) {
                unsigned lo = 0u, hi = n_seg;
#pragma unroll
                for (int it = 0; it < 6; ++it) {                     // n_seg <= 64
                    const unsigned mid = (lo + hi) >> 1;
                    if (hi - lo > 1u) { if (L.pref[mid] <= begin) lo = mid; else hi = mid; }
                }
                S.sg = lo;
            }
            // ---- phase 3: the tests
            unsigned qn = 0u;
            while (true) {
                const bool act = S.cur < S.end;
                if (__ballot(act) == 0ull) break;
                STAT(st, ST_A_ITERS_WAVE, 1);
                if (act && S.cur >= S.seg_end) {
                    const uint2 sd = L.seg[S.sg];
                    const unsigned p0 = L.pref[S.sg];
                    ++S.sg;
                    S.seg_end = p0 + (sd.y & 0x3ffffffu);
                    S.base = sd.x - p0;
                    S.owner = (int)(sd.y >> 26);
                    const float4 r0 = L.ray[2 * S.owner], r1 = L.ray[2 * S.owner + 1];
                    S.q.o = {r0.x, r0.y, r0.z}; S.q.d = {r0.w, r1.x, r1.y}; S.a = r1.z;
                    S.atm = S.a * (0.001f * 0.9999f - 1e-6f);
                }
                const unsigned nb = act ? min((unsigned)PB, min(S.seg_end, S.end) - S.cur) : 0u;   // (a batch stays within one owner's range)
                const unsigned he0 = act ? S.base + S.cur : 0u;
                const float4* __restrict__ hp = hot + he0;                            // (one address, the batch by immediate offsets)
                float4 s4[PB];
#pragma unroll
                for (int k = 0; k < PB; ++k) s4[k] = hp[k];                           // (padded arrays: reading past a range is harmless, holding is not)
                // the owner's best hit as of now
                const float bt_now = __uint_as_float((unsigned)(L.key[S.owner] >> 32));
                S.abt = __builtin_fmaf(S.a * bt_now, 1.0001f, 1e-6f * S.a);
                if (act) { STAT(st, ST_TESTS, nb); STAT(st, ST_A_LANE_STEPS, 1); }
                bool hold[PB]; float hb[PB], hd[PB];
                bool anyhold = false;
#pragma unroll
                for (int k = 0; k < PB; ++k) {
                    const float ocx = S.q.o.x - s4[k].x, ocy = S.q.o.y - s4[k].y, ocz = S.q.o.z - s4[k].z;
                    const float b = ocx * S.q.d.x + ocy * S.q.d.y + ocz * S.q.d.z;
                    const float c = (ocx * ocx + ocy * ocy + ocz * ocz) - s4[k].w;
                    const float disc = b * b - S.a * c;
                    // App. A.5: "near root beyond the best hit" (Lm) and "far root behind t_min" (M).  The two cannot both be
                    // positive (Lm > 0 needs b + kb < -abt, M > 0 needs b - kb > -atm, and abt > atm): one test of the larger.
                    // (the margins are fused here — one rounding less than the analysis allows for)
                    const float ab = fabsf(b);
                    const float Lm = __builtin_fmaf(ab, -1e-4f, -b) - S.abt, M = __builtin_fmaf(ab, -1e-4f, b) + S.atm;
                    const float P = fmaxf(Lm, M);
                    hold[k] = (unsigned)k < nb && disc > 0.0f && !(P > 0.0f && P * P > disc * 1.0003f);
                    hb[k] = b; hd[k] = disc;
                    anyhold = anyhold || hold[k];
                }
                if (__ballot(anyhold) != 0ull) {
#pragma unroll
                    for (int k = 0; k < PB; ++k) {
                        const unsigned long long hm = __ballot(hold[k]);
                        if (hm != 0ull) {
                            const unsigned slot = qn + __builtin_amdgcn_mbcnt_hi((unsigned)(hm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)hm, 0u));
                            if (hold[k]) {
                                STAT(st, ST_DISCPOS, 1);
                                const uint4 ce = make_uint4(__float_as_uint(hb[k]), __float_as_uint(hd[k]), he0 + (unsigned)k, (unsigned)S.owner);
                                if (slot < (unsigned)kWalkCand) L.cq[slot] = ce;
                                else pool_candidate(T, s_nodes, L, ce STAT_PASS);
                            }
                            qn += (unsigned)__popcll(hm);
                        }
                    }
                }
                S.cur += nb;
                if (qn >= (unsigned)RT_DENSE_DRAIN) pool_drain(T, s_nodes, L, lane, qn STAT_PASS);
            }
            pool_drain(T, s_nodes, L, lane, qn STAT_PASS);
            const unsigned long long k = L.key[lane];
            best_t = __uint_as_float((unsigned)(k >> 32)); best = (int)(unsigned)k;
            const unsigned tmask = lane < 32 ? L.tie_lo : L.tie_hi;
            if ((tmask >> (lane & 31)) & 1u) tie = true;
        }
        if (walking && best >= 0) walk_clip(W, A, best_t);
        walking = walking && W.i != W.iend;
        const int left = __popcll(__ballot(walking));
        if (left == 0) break;
        if (RT_QUORUM_DEN > 0 && nw0 >= RT_QUORUM_MIN && left * RT_QUORUM_DEN <= nw0) break;
    }
    W.walking = walking;
}

// hitTree (acceleration_structure.h:319-342): ground sphere first, then the tree.
// A pooled walk returns once most of the wave's rays are done (quorum): a lane whose walk is not finished comes back with
// ts.pending set and resumes on the next call with the same ray, while the lanes that are done go on to shade and start new
// rays — the wave does not wait for its longest walk.  `closest`/`best` persist with the caller.
// COOPG selects the walk: 4 = walk_pool (sparse grids), 5 = the same walk, the pre-classified chains start alone in their waves (very sparse grids), 2 = walk_pool_dense, 1 = none — trees without a candidate grid (and the
// reference traversal mode) take the literal scan for every ray.
// The spheres every bounce of every ray meets before its walk — the ground sphere and the first kHotLarge of the grid's large spheres
// (geometry and brick) — are staged in LDS behind the waves' pools by the kernels with a pooled walk: as uniform global loads in the
// serial part of a bounce each cost a full cache round trip (C2 / C3 are as long as their longest pixel: a lone bounce is what counts).
constexpr int kHotLarge = 8;
constexpr int kHotSlots = 1 + 3 * kHotLarge;                   // float4 slots: ground | large_hot[k] | large_brick[2k], [2k + 1]
RT_DEV float4* hot_spheres(const float4* s_nodes, int n_nodes) { return (float4*)((WalkLds*)(s_nodes + n_nodes * 3) + 4); }
template <bool POOL>
RT_DEV void stage_tree_fp32(const DevScene& S, const DevTree& T, float4* s_nodes) {
    const int n4 = T.n_nodes * 3;
    for (int t = threadIdx.x; t < n4; t += 256) s_nodes[t] = T.nodes4[t];
    if (POOL) {
        float4* h = hot_spheres(s_nodes, T.n_nodes);
        const int nl = min(T.acc.n_large, kHotLarge);
        if (threadIdx.x == 0) h[0] = S.ground_valid ? S.list_hot[0] : make_float4(0.f, 0.f, 0.f, 0.f);
        if ((int)threadIdx.x < nl) h[1 + threadIdx.x] = T.acc.large_hot[threadIdx.x];
        if ((int)threadIdx.x < 2 * nl) h[1 + kHotLarge + threadIdx.x] = T.acc.large_brick[threadIdx.x];
    }
    __syncthreads();
}

struct TreeState { Walk W; float g_t; int g_id; bool tie, pending; unsigned work, cols; };
RT_DEV void tree_state_init(TreeState& ts) {
    ts.pending = false; ts.tie = false; ts.g_t = FLT_MAX; ts.g_id = -1; ts.work = 0u; ts.cols = 0u; ts.W.work = 0u; ts.W.cols = 0u;
    ts.W.walking = false; ts.W.i = 0; ts.W.iend = 0; ts.W.coff = 0; ts.W.om_c = 0.f; ts.W.on_c = 0.f; ts.W.slope = 0.f; ts.W.dm_c = 0.f; ts.W.fwd = true;
}

template <int COOPG>
RT_DEV void closest_tree(const DevScene& S, const DevTree& T, const float4* s_nodes, const RayF& r, float a, bool live, float& closest, int& best, TreeState& ts STAT_ARG) {
    const bool fresh = live && !ts.pending;
    STAT(st, ST_RAYS, fresh ? 1 : 0);
    if (fresh) { closest = FLT_MAX; best = -1; }
    if (S.ground_valid && fresh) {
        WPASS(WP_GROUND);
        const float4 g = COOPG != 1 ? hot_spheres(s_nodes, T.n_nodes)[0] : S.list_hot[0];
        int gb = -1;
        // origin outside the sphere (c > 0) and heading away from its centre (b > 0): disc <= fl(b*b), so sqrt(disc) <= b and
        // both roots are <= 0 — the float test cannot pass (no margin involved); saves the exact sqrt and two divisions
        { const float ocx = r.o.x - g.x, ocy = r.o.y - g.y, ocz = r.o.z - g.z;
          const float b = ocx * r.d.x + ocy * r.d.y + ocz * r.d.z;
          const float c = (ocx * ocx + ocy * ocy + ocz * ocz) - g.w;
          if (__ballot(fresh && !(b > 0.0f && c > 0.0f)) != 0ull) sphere_test(r, a, g.x, g.y, g.z, g.w, 0, closest, gb); }
        if (gb == 0) best = 0;
    }
    RT_STATS_ONLY(
    const unsigned long long tG = TICK(); st.cyc[4] += tG;      // (entry stamp is subtracted by the caller's tC0)
    )
    bool slow = fresh;
    if (COOPG != 1 && T.acc.enabled) {
        // preconditions of the exactness argument; any NaN/inf makes a comparison false and sends the ray to the scan
        const float zx = r.o.x, zy = r.o.y - 1.0f, zz = r.o.z;
        const bool fast = fresh && (a >= 9.094947e-13f) && (a <= 1.0995116e12f) && (r.d.x != 0.0f) && (r.d.z != 0.0f)
                          && (fabsf(r.d.y) >= 9.094947e-13f) && (zx * zx + zy * zy + zz * zz <= T.acc.zone2);
        slow = fresh && !fast;
        STAT(st, ST_FAST, fast ? 1 : 0); STAT(st, ST_SLOW, slow ? 1 : 0);
        if (fast) {
            ts.g_t = closest; ts.g_id = best; ts.tie = false;
            const float ra = __builtin_amdgcn_rcpf(a);
            for (int k = 0; k < T.acc.n_large; ++k) {
                // same cheap pre-filter as in the walk: exact roots only for a sphere that can still win
                const float4* hs = hot_spheres(s_nodes, T.n_nodes);
                const float4 sp = k < kHotLarge ? hs[1 + k] : T.acc.large_hot[k];
                WPASS(WP_LARGE_K);
                const float ocx = r.o.x - sp.x, ocy = r.o.y - sp.y, ocz = r.o.z - sp.z;
                const float b = ocx * r.d.x + ocy * r.d.y + ocz * r.d.z;
                const float c = (ocx * ocx + ocy * ocy + ocz * ocz) - sp.w;
                const float disc = b * b - a * c;
                if (disc > 0.0f) {
                    const float sqa = __builtin_amdgcn_sqrtf(disc);
                    const float m = 1e-4f * ((fabsf(b) + sqa) * ra) + 1e-6f;
                    const bool behind = (sqa - b) * ra + m < 0.001f;
                    const bool beyond = (-b - sqa) * ra - m > closest;
                    if (!behind && !beyond) {
                        WPASS(WP_LARGE_EXACT);
                        const float sq = sqrtf(disc);
                        float cand = __builtin_inff();
                        const float t1 = (-b - sq) / a;
                        if (t1 > 0.001f) cand = t1;
                        else { const float t2 = (-b + sq) / a; if (t2 > 0.001f) cand = t2; }
                        offer(T, s_nodes, r, cand, k < kHotLarge ? hs[1 + kHotLarge + 2 * k] : T.acc.large_brick[2 * k], k < kHotLarge ? hs[2 + kHotLarge + 2 * k] : T.acc.large_brick[2 * k + 1], closest, best, ts.tie STAT_PASS);
                    }
                }
            }
            WPASS(WP_SETUP);
            ts.W = walk_setup(T.acc, r, closest, best);
        }
        RT_STATS_ONLY(
        const unsigned long long tL = TICK(); st.cyc[5] += tL - tG;       // large spheres + walk set-up
        )
        const bool walker = fast || (live && ts.pending);
        const int nw = __popcll(__ballot(walker));
        if (nw > 0) {
            // every lane of the wave takes part: lanes without a walk of their own test other lanes' spheres
            WalkLds& L = *((WalkLds*)(s_nodes + T.n_nodes * 3) + (threadIdx.x >> 6));
            Walk Wl = ts.W; Wl.walking = walker && ts.W.walking;
            float bt = walker ? closest : FLT_MAX; int bi = walker ? best : -1; bool tt = false;
            // (sparse grids — C3: 3.6 entries per cell — pool two columns per ray and round; on dense ones — C5: 37 per cell, ~100
            // per column — that walk filters against a stale best hit: 1281 ms against 670 for walk_pool_dense, which re-reads it)
            if (COOPG >= 4) walk_pool<RT_QUORUM_SPARSE>(T, s_nodes, L, r, a, Wl, bt, bi, tt STAT_PASS);
            else walk_pool_dense<RT_QUORUM_DENSE>(T, s_nodes, L, r, a, Wl, bt, bi, tt STAT_PASS);
            if (walker) { ts.work += Wl.work; ts.cols += Wl.cols; Wl.work = 0u; Wl.cols = 0u; ts.W = Wl; closest = bt; best = bi; ts.tie = ts.tie || tt; }
        }
        if (walker) {
            ts.pending = ts.W.walking;
            if (!ts.pending && ts.tie) { closest = ts.g_t; best = ts.g_id; slow = true; STAT(st, ST_TIE, 1); }
        }
    }
    RT_STATS_ONLY(
    const unsigned long long tS0 = TICK(); st.cyc[1] += tS0 - tG;     // fast path (large spheres + setup + walk), wave time
    )
    if (__ballot(slow) != 0ull) { WPASS(WP_SCAN); tree_scan(s_nodes, r, a, slow, closest, best); }
    RT_STATS_ONLY(
    st.cyc[3] += TICK() - tS0;
    )
}

// ---------------------------------------------------------------------------------------------------- sampling
RT_DEV V3 random_in_unit_sphere(Rng& s) {               // material.h:35-41
    V3 p;
    do {
        WPASS(WP_REJ_ITER);
        const float x = rng_uniform(s); const float y = rng_uniform(s); const float z = rng_uniform(s);
        p.x = 2.0f * x - 1.0f; p.y = 2.0f * y - 1.0f; p.z = 2.0f * z - 1.0f;
    } while (p.x * p.x + p.y * p.y + p.z * p.z >= 1.0f);
    return p;
}

// camera::get_ray + the two pixel-jitter draws (main.cu:104-106, camera.h:12-18, :45-49)
RT_DEV RayF primary_ray(const rt_camera& c, int i, int j, int max_x, int max_y, Rng& s) {
    const float u = ((float)i + rng_uniform(s)) / (float)max_x;
    const float v = ((float)j + rng_uniform(s)) / (float)max_y;
    float px, py;
    WPASS(WP_PRIMARY);
    do {
        WPASS(WP_DISK_ITER);
        const float x = rng_uniform(s); const float y = rng_uniform(s);
        px = 2.0f * x - 1.0f; py = 2.0f * y - 1.0f;
    } while (px * px + py * py + 0.0f >= 1.0f);
    const float rdx = c.lens_radius * px, rdy = c.lens_radius * py;
    RayF r;
    const float ox = rdx * c.u[0] + rdy * c.v[0], oy = rdx * c.u[1] + rdy * c.v[1], oz = rdx * c.u[2] + rdy * c.v[2];
    r.o.x = c.origin[0] + ox; r.o.y = c.origin[1] + oy; r.o.z = c.origin[2] + oz;
    r.d.x = (((c.lower_left_corner[0] + u * c.horizontal[0]) + v * c.vertical[0]) - c.origin[0]) - ox;
    r.d.y = (((c.lower_left_corner[1] + u * c.horizontal[1]) + v * c.vertical[1]) - c.origin[1]) - oy;
    r.d.z = (((c.lower_left_corner[2] + u * c.horizontal[2]) + v * c.vertical[2]) - c.origin[2]) - oz;
    return r;
}

// material::scatter for the sphere that was hit.  Returns false when the path is absorbed (metal, material.h:72).
RT_DEV bool scatter(const DevScene& S, int sphere, float t, RayF& r, V3& att, Rng& s) {
    const float4 g = S.shade[2 * sphere];
    const float4 m = S.shade[2 * sphere + 1];
    const int kind = S.kind8[sphere];
    V3 p, n;
    p.x = r.o.x + t * r.d.x; p.y = r.o.y + t * r.d.y; p.z = r.o.z + t * r.d.z;            // ray.h:13
    n.x = (p.x - g.x) / g.w; n.y = (p.y - g.y) / g.w; n.z = (p.z - g.z) / g.w;            // sphere.h:29
    // lambertian and metal both draw exactly one random_in_unit_sphere and nothing else: one shared rejection loop
    // (the slowest lane of a wave sets its length) instead of one per material branch; the draw order is unchanged
    V3 q = {0.0f, 0.0f, 0.0f};
    WPASS(WP_SC_ANY);
    if (kind != RT_MAT_DIELECTRIC) q = random_in_unit_sphere(s);
    // metal and dielectric both start from unit_vector(r_in.direction()) (material.h:69, :18): formed once for the lanes of either —
    // a wave that holds both kinds (most do: 15 % metal, 5 % glass) walks through both branches, and paid the square root and the
    // three divisions in each.  Same expression, same operands, same bits.
    float len = 1.0f; V3 ud = {0.0f, 0.0f, 0.0f};
    if (kind != RT_MAT_LAMBERTIAN) {
        len = sqrtf(r.d.x * r.d.x + r.d.y * r.d.y + r.d.z * r.d.z);
        ud.x = r.d.x / len; ud.y = r.d.y / len; ud.z = r.d.z / len;
    }
    if (kind == RT_MAT_LAMBERTIAN) {                                                          // material.h:55-60
        WPASS(WP_SC_LAMB);
        const float tx = (p.x + n.x) + q.x, ty = (p.y + n.y) + q.y, tz = (p.z + n.z) + q.z;
        r.d.x = tx - p.x; r.d.y = ty - p.y; r.d.z = tz - p.z;
        r.o = p;
        att.x *= m.x; att.y *= m.y; att.z *= m.z;
        return true;
    }
    if (kind == RT_MAT_METAL) {                                                               // material.h:68-73
        WPASS(WP_SC_METAL);
        const float k = 2.0f * dot3(ud, n);
        const float rx = ud.x - k * n.x, ry = ud.y - k * n.y, rz = ud.z - k * n.z;
        r.d.x = rx + m.w * q.x; r.d.y = ry + m.w * q.y; r.d.z = rz + m.w * q.z;
        r.o = p;
        att.x *= m.x; att.y *= m.y; att.z *= m.z;
        return dot3(r.d, n) > 0.0f;
    }
    // dielectric — material.h:81-113 (attenuation (1,1,1): the multiply is exact and omitted)
    WPASS(WP_SC_DIEL);
    const float ri = m.w;
    const float dn = dot3(r.d, n);
    const float k = 2.0f * dn;
    const float rx = r.d.x - k * n.x, ry = r.d.y - k * n.y, rz = r.d.z - k * n.z;            // reflect(dir, normal), dir not normalised
    V3 on; float ni, cosine;
    if (dn > 0.0f) {
        on.x = -n.x; on.y = -n.y; on.z = -n.z; ni = ri;
        cosine = dn / len;
        cosine = sqrtf(1.0f - ri * ri * (1.0f - cosine * cosine));
    } else {
        on = n; ni = 1.0f / ri;
        cosine = -dn / len;
    }
    const V3 uv = ud;                                                                         // refract(), material.h:17-31
    const float dt = dot3(uv, on);
    const float disc = 1.0f - ni * ni * (1.0f - dt * dt);
    float fx = 0.0f, fy = 0.0f, fz = 0.0f, reflect_prob;
    if (disc > 0.0f) {
        const float sq = sqrtf(disc);
        fx = ni * (uv.x - dt * on.x) - sq * on.x;
        fy = ni * (uv.y - dt * on.y) - sq * on.y;
        fz = ni * (uv.z - dt * on.z) - sq * on.z;
        float r0 = (1.0f - ri) / (1.0f + ri);                                                // schlick(), material.h:11-15
        r0 = r0 * r0;
        reflect_prob = r0 + (1.0f - r0) * pow5(1.0f - cosine);
    } else {
        reflect_prob = 1.0f;
    }
    r.o = p;
    if (rng_uniform(s) < reflect_prob) { r.d.x = rx; r.d.y = ry; r.d.z = rz; }
    else { r.d.x = fx; r.d.y = fy; r.d.z = fz; }
    return true;
}

// background gradient of color() — main.cu:67-72
RT_DEV V3 sky(const RayF& r, const V3& att) {
    WPASS(WP_SKY);
    const float len = sqrtf(r.d.x * r.d.x + r.d.y * r.d.y + r.d.z * r.d.z);
    const float uy = r.d.y / len;
    const float t = 0.5f * (uy + 1.0f);
    const float omt = 1.0f - t;
    V3 c;
    c.x = att.x * (omt + t * 0.5f);
    c.y = att.y * (omt + t * 0.7f);
    c.z = att.z * (omt + t * 1.0f);
    return c;
}

// ---------------------------------------------------------------------------------------------------- kernels
#ifndef RT_TU_LIST
__global__ __launch_bounds__(256) void k_render_init(rt_rand_state* rand_state, int max_x, int max_y, int tiles_x, int part, int nparts, long long begin, long long end, long long n_local_tiles) {
    const int lane = threadIdx.x & 63;
    const long long local_tile = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (local_tile >= n_local_tiles) return;
    const long long tile = part_tile(local_tile, part, nparts, begin, end);
    const int tx = (int)(tile % tiles_x), ty = (int)(tile / tiles_x);
    const int i = tx * 8 + (lane & 7), j = ty * 8 + (lane >> 3);
    const bool inside = (i < max_x) && (j < max_y);
    const bool whole = part_whole(nparts, begin, end);
    if (whole && !inside) return;
    const long long idx = whole ? (long long)j * max_x + i : local_tile * 64 + lane;
    const int pixel_index = j * max_x + i;
    Rng s; rng_seed(s, 1984ull + (unsigned long long)(long long)pixel_index);
    rt_rand_state out;
    out.d = s.d; out.v[0] = s.v0; out.v[1] = s.v1; out.v[2] = s.v2; out.v[3] = s.v3; out.v[4] = s.v4;
    out.boxmuller_flag = 0; out.boxmuller_flag_double = 0; out.boxmuller_extra = 0.f; out.pad_ = 0; out.boxmuller_extra_double = 0.0;
    rand_state[idx] = out;
}
#endif

// MODE 0: render (ns samples, /ns, sqrt).  MODE 1: render_progressive (one sample, accumulate).
//
// Persistent waves: the grid is sized to what the chip holds at once.  Pixel slots are numbered tile-major
// (slot = local_tile*64 + ly*8 + lx, the 8x8 block shape of the reference); every lane starts on slot
// (wave*64 + lane) and, when its pixel is finished, pulls the next unclaimed slot from a global counter, so no lane
// waits for the slowest pixel of "its" tile.  Which lane renders a pixel does not affect the pixel (the RNG is per pixel).
//
// Long chains.  A pixel is a strictly serial chain of ns x bounces iterations, and a few pixels (crevices between a
// sphere and the ground, glass) need 10-15x the average.  One iteration of a full wave costs ~40 k cycles, of a wave
// with a single live lane ~16 k, so the frame cannot end before (longest chain) x (iteration time of ITS wave).
// Therefore: slots are interleaved over blocks of 64 tiles (the 64 pixels of a tile go to 64 different lanes/waves: long
// pixels cluster); a lane whose pixel shows >= RT_LONG_RATE bounces per sample (checked every RT_LONG_CHECK samples) marks it long;
// a wave holding a long pixel stops refilling its other lanes ("thin" wave, raised issue priority) until that pixel is
// finished, then resumes.  A global counter caps the
// number of thin waves at a quarter of the grid, so a scene made of long pixels only keeps its throughput.
template <bool TREE, int MODE, int COOPG>
__global__ __launch_bounds__(256, RT_RENDER_WAVES) void k_render(RenderArgs A) {
    extern __shared__ float4 s_nodes[];
    if (TREE) stage_tree_fp32<COOPG != 1>(A.scene, A.tree, s_nodes);
    const int lane = threadIdx.x & 63;
    const long long n_slots = A.n_local_tiles * 64;
    const int ns = (MODE == 0) ? A.ns : 1;

    const long long n_waves = (long long)gridDim.x * 4;
    long long slot = 0;
    int i = 0, j = 0; long long idx = 0;
    unsigned int iters = 0;             // bounces spent on the current pixel
    bool is_long = false;               // the current pixel has been classified long
    bool is_med = false;                // ... medium: its wave issues at priority 1
    bool retired = false;               // the queue was empty when this lane last asked
    bool thin = false;                  // wave-uniform: this wave holds a long pixel and does not refill
    bool thin_counted = false;          // wave-uniform: this wave is included in the global thin-wave count
    bool long_done = false;             // the list of pre-classified long chains is exhausted
    // pre-classified long chains are honoured only while they are rare (<= 1/64 of the pixels): a scene made of long
    // chains only must keep its throughput.  The same value is read by every wave, so the decision is grid-wide.
    const unsigned int n_long_raw = A.long_list ? A.queue[2] : 0u, n_solo_raw = A.long_list ? A.queue[4] : 0u;
    const bool use_long = (n_long_raw + n_solo_raw) != 0u && (long long)(n_long_raw + n_solo_raw) * 64 <= n_slots;
    const unsigned int n_long = use_long ? n_long_raw : 0u, n_solo = use_long ? n_solo_raw : 0u;
    bool solo = false, solo_done = false;      // this wave started with one of the longest chains, alone (lane 0); that list is exhausted
    Rng s = {0, 0, 0, 0, 0, 0};
    V3 col = {0.0f, 0.0f, 0.0f};
    V3 att = {1.0f, 1.0f, 1.0f};
    RayF r; r.o = {0.f, 0.f, 0.f}; r.d = {0.f, 1.f, 0.f};
    int sample = 0, depth = 0;
    bool live = false;
    RT_STATS_ONLY(
    unsigned int pix_iters = 0;
    unsigned long long pix_t0 = 0;
    )

    // The next unclaimed slot.  render: one request to the global work counter per lane and pixel (a pixel is ns samples: requests are
    // rare, and the fine grain balances the frame's tail).  render_progressive hands out a pixel per lane and BOUNCE or so: 1.2 M
    // requests per C3 pass, one atomic per wave and iteration on one address — 0.3 ms of a 1.0 ms pass.  There a wave owns its first
    // kProgOwn slots outright (the counter's values start behind all waves' own slots) and then takes the rest in chunks of 64, at
    // the one place of the main loop where the wave is convergent (`refill_progressive`).
    constexpr bool kChunked = MODE == 1;
    constexpr long long kProgOwn = RT_PROG_OWN;
    const long long first_free = (kChunked && kProgOwn > 0 && n_waves * kProgOwn <= n_slots) ? n_waves * kProgOwn : 0;
    long long pool_next = 0, pool_end = 0;              // wave-uniform: this wave's current chunk [pool_next, pool_end)
    if (kChunked && first_free > 0) { pool_next = ((long long)blockIdx.x * 4 + (threadIdx.x >> 6)) * kProgOwn; pool_end = pool_next + kProgOwn; }
    auto take_slot = [&]() -> long long {
        return first_free + (long long)atomicAdd(cold_args()->queue, 1u);
    };
    // claim `slot` (skipping slots that fall outside the frame in edge tiles) and set the lane up for that pixel
    auto begin_pixel = [&]() {
        live = false;
        const RenderArgs& A = *cold_args();
#ifdef RT_ONLY_LANE
        if (lane != RT_ONLY_LANE) return;                  // probe build: one pixel chain alone in its wave
#endif
        iters = 0; is_long = false; is_med = false;
        while (slot < n_slots) {
            // Slots are interleaved over blocks of 64 tiles (in hand-out order): consecutive slots are the same pixel
            // position of 64 different tiles, so the pixels of one tile (long chains cluster) never travel together.
            // The END of the queue is handed out per pixel, most expensive 2x2 pilot block first (k_tail_hist / k_tail_scatter): a launch with few pixels
            // per lane ends when its last-started pixels do, and inside a tile of a cheap class sit pixels of three times its mean.
            long long local_tile; int l;
            // On sparse grids that list is handed out FIRST (its first `head` entries, k_tail_scatter: all of it but the blocks outside the
            // frame), then the tiles, most expensive class first: C3 13.9 -> 13.2 ms, C2 10.4 -> 9.3 (rt_tuning.h RT_HEAD_SUM_SPARSE).
            const long long tail0 = (MODE == 0 && A.tail_list) ? (long long)A.queue[kQueueThr + 2] : 0;
            const long long head = tail0 > 0 ? (long long)A.queue[kQueueThr + 3] : 0;
            if (tail0 > 0 && (slot < head || slot >= tail0 - 1 + head)) {
                // (head taken from the list's cheap END — queue[kQueueThr + 5] — : the sky first, the rest of the list, most expensive first, last)
                const bool from_end = A.queue[kQueueThr + 5] != 0u;
                const long long n_tail = n_slots - (tail0 - 1);
                const long long k = slot < head ? (from_end ? n_tail - head + slot : slot) : (from_end ? slot - (tail0 - 1) - head : slot - (tail0 - 1));
                const unsigned int pid = A.tail_list[k];
                local_tile = (long long)(pid >> 6); l = (int)(pid & 63u);
            } else {
                constexpr int kIl = COOPG == 2 ? RT_INTERLEAVE_DENSE : (COOPG == 5 ? RT_INTERLEAVE_SOLO : RT_INTERLEAVE);   // tiles whose pixels interleave (consecutive slots: one pixel position of kIl tiles)
                const long long ms = slot - head;                                  // the tiles' own slots
                const long long blk = ms / (64 * kIl);
                const long long tiles_in_blk = (A.n_local_tiles - blk * kIl) < kIl ? (A.n_local_tiles - blk * kIl) : kIl;
                const long long within = ms % (64 * kIl);
                const long long rank = blk * kIl + within % tiles_in_blk;          // position in the hand-out order
                l = (int)(within / tiles_in_blk);
                local_tile = A.order ? (long long)A.order[rank] : rank;
            }
            const long long tile = part_tile(local_tile, A.part, A.nparts, A.tile_begin, A.tile_end);
            const int tx = (int)(tile % A.tiles_x), ty = (int)(tile / A.tiles_x);
            i = tx * 8 + (l & 7); j = ty * 8 + (l >> 3);
            const bool taken = use_long && A.long_flag[local_tile * 64 + l];      // long chains are handed out separately
            if (i < A.max_x && j < A.max_y && !taken) {
                idx = part_whole(A.nparts, A.tile_begin, A.tile_end) ? (long long)j * A.max_x + i : local_tile * 64 + l;
                live = true;
                RT_STATS_ONLY(
                pix_t0 = __builtin_amdgcn_s_memrealtime();
                )
                break;
            }
            slot = take_slot();
        }
        if (!live) retired = true;
        if (live) {
            const rt_rand_state* st = A.rand_state + idx;
            s.d = st->d; s.v0 = st->v[0]; s.v1 = st->v[1]; s.v2 = st->v[2]; s.v3 = st->v[3]; s.v4 = st->v[4];
            col = {0.0f, 0.0f, 0.0f}; att = {1.0f, 1.0f, 1.0f}; sample = 0; depth = 0;
            r = primary_ray(A.scene.cam, i, j, A.max_x, A.max_y, s);
        }
    };
    // take the next pre-classified long chain, if any is left (lanes 0..RT_LONG_PER_WAVE-1 only)
    // (from_solo: the list of the chains that start one per wave, k_render<true,*,5>)
    auto begin_long_pixel = [&](bool from_solo) -> bool {
        if (!use_long || (from_solo ? solo_done : long_done)) return false;
        const RenderArgs& A = *cold_args();
        long long pid;
        if (from_solo) {
            const unsigned int h = atomicAdd(A.queue + 5, 1u);
            if (h >= n_solo) { solo_done = true; return false; }
            pid = (long long)A.long_list[n_slots - 1 - (long long)h];
        } else {
            const unsigned int h = atomicAdd(A.queue + 3, 1u);
            if (h >= n_long) { long_done = true; return false; }
            // the list is in the pilot's order, the four pixels of a 2x2 block and the blocks of a tile next to each other — and
            // neighbours are chains of like length: a stride (a prime that does not divide the count, so a permutation) puts
            // them into different waves, where each is left alone with its wave once the shorter ones have ended
            const unsigned int stride = n_long % 257u ? 257u : (n_long % 263u ? 263u : 269u);
            pid = (long long)A.long_list[(unsigned int)(((unsigned long long)h * stride) % n_long)];
        }
        const long long local_tile = pid >> 6;
        const int l = (int)(pid & 63);
        const long long tile = part_tile(local_tile, A.part, A.nparts, A.tile_begin, A.tile_end);
        const int tx = (int)(tile % A.tiles_x), ty = (int)(tile / A.tiles_x);
        i = tx * 8 + (l & 7); j = ty * 8 + (l >> 3);
        idx = part_whole(A.nparts, A.tile_begin, A.tile_end) ? (long long)j * A.max_x + i : pid;
        live = true; iters = 0; is_long = true;
        RT_STATS_ONLY(
        pix_t0 = __builtin_amdgcn_s_memrealtime();
        )
        const rt_rand_state* st = A.rand_state + idx;
        s.d = st->d; s.v0 = st->v[0]; s.v1 = st->v[1]; s.v2 = st->v[2]; s.v3 = st->v[3]; s.v4 = st->v[4];
        col = {0.0f, 0.0f, 0.0f}; att = {1.0f, 1.0f, 1.0f}; sample = 0; depth = 0;
        r = primary_ray(A.scene.cam, i, j, A.max_x, A.max_y, s);
        return true;
    };
    // rand_state[pixel_index] = local_rand_state; fb[pixel_index] = ...  (main.cu:110-115 / :133-141)
    auto end_pixel = [&]() {
        const RenderArgs& A = *cold_args();
        rt_rand_state* st_out = A.rand_state + idx;
        st_out->d = s.d; st_out->v[0] = s.v0; st_out->v[1] = s.v1; st_out->v[2] = s.v2; st_out->v[3] = s.v3; st_out->v[4] = s.v4;
        float* fb = (float*)A.fb + idx * 3;
        if (MODE == 0) {
            const float k = (float)(1.0 / (double)(float)A.ns);          // vec3::operator/=(real_t): 1.0/t in double (vec3.h:137)
            col.x *= k; col.y *= k; col.z *= k;
            fb[0] = sqrtf(col.x); fb[1] = sqrtf(col.y); fb[2] = sqrtf(col.z);
            RT_STATS_ONLY(
            // diagnostic build: chain length and end / start time (100 MHz ticks mod 2^24) instead of colour
            fb[0] = (float)pix_iters; fb[1] = (float)((__builtin_amdgcn_s_memrealtime() >> 4) & 0xffffffull); fb[2] = (float)((pix_t0 >> 4) & 0xffffffull);
            )
        } else {
            if (A.ns == 1) { fb[0] = col.x; fb[1] = col.y; fb[2] = col.z; }
            else { fb[0] += col.x; fb[1] += col.y; fb[2] += col.z; }
        }
    };
    RT_STATS_ONLY(
    Stats st; for (int q = 0; q < ST_N; ++q) st.c[q] = 0;
    for (int q = 0; q < 8; ++q) st.cyc[q] = 0;
    const unsigned long long tK0 = TICK(), rK0 = __builtin_amdgcn_s_memrealtime();
    unsigned int dbg_thin_iters = 0, dbg_long = 0;
    unsigned long long th[4] = {0, 0, 0, 0};
    unsigned long long dbg_thin_cyc = 0, dbg_thin_closest = 0, dbg_thin1_cyc = 0; unsigned int dbg_thin1_iters = 0; unsigned long long dbg_t_prev = TICK();
    )
    // start: pre-classified long chains first, RT_LONG_PER_WAVE per wave, in waves that are thin from the beginning
    constexpr bool kSolo = TREE && COOPG == 5;       // (the pre-classified chains alone in their waves: very sparse grids)
    if (kSolo && lane == 0 && begin_long_pixel(true)) solo = true;
    if (__ballot(live) == 0ull && lane < RT_LONG_PER_WAVE) begin_long_pixel(false);
    if (__ballot(live) != 0ull) { thin = true; __builtin_amdgcn_s_setprio(3); }
    else if (!kChunked) { slot = take_slot(); begin_pixel(); }

    const unsigned int thin_cap = (unsigned int)(n_waves / RT_THIN_CAP_DEN);
    float closest = FLT_MAX; int best = -1;
    TreeState ts; tree_state_init(ts);
    while (true) {
        // ---- wave-level bookkeeping (uniform)
        const unsigned long long m_live = __ballot(live);
        const unsigned long long m_long = __ballot(live && is_long);
        if (!thin && m_long != 0ull) {
            // try to become a thin wave; if the cap is reached these pixels simply stay ordinary
            unsigned int prev = 0;
            if (lane == 0) prev = atomicAdd(cold_args()->queue + 1, 1u);
            prev = __builtin_amdgcn_readfirstlane(prev);
            if (prev < thin_cap) { thin = true; thin_counted = true; __builtin_amdgcn_s_setprio(3); }       // the chain is on the critical path: win issue arbitration
            else { if (lane == 0) atomicSub(cold_args()->queue + 1, 1u); is_long = false; }
        } else if (thin && m_long == 0ull) {
            thin = false; __builtin_amdgcn_s_setprio(0);
            if (thin_counted && lane == 0) atomicSub(A.queue + 1, 1u);
            thin_counted = false;
        }
        if (RT_MED_RATE > 0 && !thin) { if (__ballot(live && is_med) != 0ull) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0); }
        if (!thin) {
            // idle lanes (their pixel ended while the wave was thin — or, progressive passes, in the previous iteration) go back to the queue
            if (kChunked) {
                const bool need = !live && !retired;
                const unsigned long long mneed = __ballot(need);
                if (mneed != 0ull) {
                    const long long n = (long long)__popcll(mneed);
                    const long long left = pool_end - pool_next;                 // what the current chunk still holds
                    const long long rank = (long long)__builtin_amdgcn_mbcnt_hi((unsigned)(mneed >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mneed, 0u));
                    long long fresh = 0;
                    if (n > left) {                                               // one more chunk for the wave (lane 0 asks)
                        unsigned int b = 0;
                        if (lane == 0) b = atomicAdd(cold_args()->queue, 64u);
                        fresh = first_free + (long long)__builtin_amdgcn_readfirstlane(b);
                    }
                    if (need) slot = rank < left ? pool_next + rank : fresh + (rank - left);
                    if (n > left) { pool_next = fresh + (n - left); pool_end = fresh + 64; } else pool_next += n;
                    if (need) begin_pixel();
                }
            } else if (!live && !retired) { slot = take_slot(); begin_pixel(); }
        }
        // thin implies a live long pixel; otherwise every lane that is not live has just found the queue empty
        if (__ballot(live) == 0ull) break;
        (void)m_live;
        STAT(st, ST_LOOP_ITERS_WAVE, 1);
        RT_STATS_ONLY(
        { const unsigned long long now = TICK(); const int nl2 = __popcll(__ballot(live));
          if (thin) { ++dbg_thin_iters; dbg_thin_cyc += now - dbg_t_prev; if (nl2 <= 2) { ++dbg_thin1_iters; dbg_thin1_cyc += now - dbg_t_prev; } }
          dbg_t_prev = now; }
        )
        RT_STATS_ONLY(
        { const int nl = __popcll(__ballot(live)); STAT(st, nl >= 56 ? ST_LIVE_GE56 : nl >= 32 ? ST_LIVE_32 : nl >= 8 ? ST_LIVE_8 : ST_LIVE_LT8, 1); }
        )
        const float a = dot3(r.d, r.d);
        if (!TREE) { closest = FLT_MAX; best = -1; }
        RT_STATS_ONLY(
        const unsigned long long tC0 = TICK();
        const unsigned long long c4_ = st.cyc[4], c5_ = st.cyc[5], c1_ = st.cyc[1], c3_ = st.cyc[3];
        const bool thin12 = thin && __popcll(__ballot(live)) <= 2;
        )
        if (TREE) closest_tree<COOPG>(A.scene, A.tree, s_nodes, r, a, live, closest, best, ts STAT_PASS);
        else closest_list(A.scene, r, a, live, closest, best);
        RT_STATS_ONLY(
        const unsigned long long tC1 = TICK(); st.cyc[0] += tC1 - tC0; st.cyc[2] += tC0; if (thin) dbg_thin_closest += tC1 - tC0;
        if (TREE && thin12) { th[0] += (st.cyc[4] - c4_) - tC0; th[1] += st.cyc[5] - c5_; th[2] += (st.cyc[1] - c1_) - (st.cyc[5] - c5_); th[3] += st.cyc[3] - c3_; }
        if (live) { ++pix_iters; }
        )
        if (live && !(TREE && ts.pending)) {
            ++iters;
            bool done;                                     // this sample's path has ended
            if (best >= 0) {
                const bool cont = scatter(cold_args()->scene, best, closest, r, att, s);
                ++depth;
                done = !cont || depth >= 50;               // absorbed, or 50 bounces used up: contributes (0,0,0)
            } else {
                const V3 c = sky(r, att);
                col.x += c.x; col.y += c.y; col.z += c.z;
                done = true;
            }
            if (done) {
                ++sample; depth = 0; att = {1.0f, 1.0f, 1.0f};
                if (sample < ns) {
                    { const RenderArgs& C = *cold_args(); r = primary_ray(C.scene.cam, i, j, C.max_x, C.max_y, s); }
                    // classify after every 4th sample while enough of the chain is left for it to matter
                    // ... long from RT_LONG_RATE bounces per sample on, or — k_tile_order — when the chain this rate predicts (rate x ns) is a sizeable
                    // part of what ONE lane works through in this launch (queue[kQueueThr]; 0 when no scheduling pass ran)
                    bool now_long = false;
                    if ((sample & (RT_LONG_CHECK - 1)) == 0 && sample + 8 <= ns) {
                        const unsigned int long_thr = MODE == 0 ? cold_args()->queue[kQueueThr] : 0u;
                        now_long = iters >= (unsigned int)((COOPG == 2 ? RT_LONG_RATE_DENSE : RT_LONG_RATE) * sample) ||
                                   (long_thr != 0u && (unsigned long long)iters * (unsigned int)ns >= (unsigned long long)long_thr * (unsigned int)sample);
                    }
                    if (now_long) {
                        RT_STATS_ONLY(
                        if (!is_long) ++dbg_long;
                        )
                        is_long = true;
                    }
                    else if (RT_MED_RATE > 0 && (sample & 7) == 0 && iters >= (unsigned int)(RT_MED_RATE * sample)) is_med = true;
                } else {
                    WPASS(WP_ENDPIX);
                    end_pixel();
                    RT_STATS_ONLY(
                    pix_iters = 0;
                    )
                    STAT(st, ST_SWITCHES, 1);
                    live = false; is_long = false; is_med = false;
                    if (kSolo && solo && begin_long_pixel(true)) { /* the next of the longest chains, alone again */ }
                    else if (lane < RT_LONG_PER_WAVE && begin_long_pixel(false)) { solo = false; /* next long chain */ }
                    else if (!thin && !kChunked) { slot = take_slot(); begin_pixel(); }      // (progressive passes refill at the top of the loop)
                }
            }
        }
    }

    RT_STATS_ONLY(
    for (int q = 0; q < ST_N; ++q) {
        const bool wave_level = (q == ST_LOOP_ITERS_WAVE || (q >= ST_LIVE_GE56 && q <= ST_LIVE_LT8));
        const bool lane_max = (q == ST_A_ITERS_WAVE || q == ST_B_ROUNDS_WAVE);       // single-wave probes: max over lanes = wave count
        if (lane_max) { int m = (int)st.c[q]; for (int off = 32; off > 0; off >>= 1) m = max(m, __shfl_xor(m, off)); if (lane == 0) atomicAdd(&g_stats[q], (unsigned long long)m); }
        else if (st.c[q] && (!wave_level || lane == 0)) atomicAdd(&g_stats[q], (unsigned long long)st.c[q]);
    }
    { int nl = (int)dbg_long; for (int off = 32; off > 0; off >>= 1) nl += __shfl_xor(nl, off); dbg_long = (unsigned int)nl; }
    if (lane == 0) {
        const long long wv = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
        if (wv < 8192) {
            g_wave_dbg[wv * 4 + 0] = __builtin_amdgcn_s_memrealtime();
            g_wave_dbg[wv * 4 + 1] = st.c[ST_LOOP_ITERS_WAVE]; g_wave_dbg[wv * 4 + 2] = dbg_thin_iters; g_wave_dbg[wv * 4 + 3] = dbg_long;
            atomicAdd(&g_stats[TH_GROUND], th[0]); atomicAdd(&g_stats[TH_LARGE_SETUP], th[1]); atomicAdd(&g_stats[TH_WALK], th[2]); atomicAdd(&g_stats[TH_SCAN], th[3]);
            atomicAdd(&g_stats[ST_SPARE0], dbg_thin_cyc); atomicAdd(&g_stats[ST_SPARE1], dbg_thin_closest); atomicAdd(&g_stats[ST_SPARE2], dbg_thin1_cyc); atomicAdd(&g_stats[ST_SPARE3], (unsigned long long)dbg_thin1_iters); atomicAdd(&g_stats[ST_SPARE4], (unsigned long long)dbg_thin_iters);
        }
        atomicAdd(&g_stats[ST_SAMPLES], 1ull);
        const unsigned long long tot = TICK() - tK0;
        atomicAdd(&g_stats[ST_CYC_TOTAL], tot);
        atomicAdd(&g_stats[ST_CYC_CLOSEST], st.cyc[0]); atomicAdd(&g_stats[ST_CYC_WALK_A], st.cyc[1]); atomicAdd(&g_stats[ST_CYC_WALK_B], st.cyc[4] - st.cyc[2]);
        atomicAdd(&g_stats[ST_CYC_SCAN], st.cyc[3]); atomicAdd(&g_stats[ST_CYC_SHADE], tot - st.cyc[0]); atomicAdd(&g_stats[ST_SPARE4 + 1], st.cyc[5]); atomicAdd(&g_stats[ST_SPARE4 + 2], st.cyc[6]);
        atomicAdd(&g_stats[ST_REALTIME], __builtin_amdgcn_s_memrealtime() - rK0);
    }
    )
}

// ---------------------------------------------------------------------------------------------------- scheduling
// Longest-processing-time-first hand-out order for the persistent render kernel.  A pixel is a strictly serial chain
// (ns samples x bounces on one RNG stream); with only a few pixels per resident lane, a long chain picked up late keeps
// a nearly empty wave running.  k_tile_cost traces one PILOT sample per pixel on a private RNG stream and counts its
// bounces: the tile sums feed k_tile_order (8 cost classes, most expensive first, stable), and pixels whose pilot path
// neighbourhood reaches RT_PILOT_LONG_SUM bounces (k_long_select) are listed as long chains, which the render kernel starts first, in thin waves.
// All of this changes only WHICH lane renders a pixel and WHEN, never the pixel.
template <bool TREE, int COOPG = 1>
__global__ __launch_bounds__(256) void k_tile_cost(RenderArgs A, int* __restrict__ cost, unsigned char* __restrict__ pilot, int* __restrict__ work) {
    extern __shared__ float4 s_nodes[];
    if (TREE) stage_tree_fp32<COOPG != 1>(A.scene, A.tree, s_nodes);
    const int lane = threadIdx.x & 63;
    // quarter resolution: one pilot pixel per 2x2 block (long chains cluster), its RT_PILOT_SAMPLES samples in adjacent
    // lanes (the pass is as long as its longest serial chain); a wave covers 4 / RT_PILOT_SAMPLES tiles
    constexpr int kPerTile = 16 * RT_PILOT_SAMPLES, kTilesPerWave = 64 / kPerTile;
    static_assert(RT_PILOT_SAMPLES == 1 || RT_PILOT_SAMPLES == 2 || RT_PILOT_SAMPLES == 4, "pilot samples per pixel");
    const long long local_tile = ((long long)blockIdx.x * 4 + (threadIdx.x >> 6)) * kTilesPerWave + lane / kPerTile;
    const bool tile_ok = local_tile < A.n_local_tiles;
    const long long tile = part_tile(tile_ok ? local_tile : 0, A.part, A.nparts, A.tile_begin, A.tile_end);
    const int tx = (int)(tile % A.tiles_x), ty = (int)(tile / A.tiles_x);
    const int sub = (lane % kPerTile) / RT_PILOT_SAMPLES, smp = lane % RT_PILOT_SAMPLES;
    const int lx = 2 * (sub & 3), ly = 2 * (sub >> 2);
    const int i = tx * 8 + lx, j = ty * 8 + ly;
    const bool inside = tile_ok && (i < A.max_x) && (j < A.max_y);
    // PILOT path: one sample per pixel on a private RNG stream (seeded away from the pixel's own 1984 + pixel_index
    // stream, which is not touched), same camera / closest-hit / scatter code as the render kernel; only the number
    // of bounces is kept.
    Rng ps; rng_seed(ps, 0x5deece66dull + (unsigned long long)((long long)j * A.max_x + i) + (unsigned long long)smp * 0x9e3779b97f4a7c15ull);
    RayF r; r.o = {0.f, 0.f, 0.f}; r.d = {0.f, 1.f, 0.f};
    V3 att = {1.0f, 1.0f, 1.0f};
    bool live = inside;
    if (live) r = primary_ray(A.scene.cam, i, j, A.max_x, A.max_y, ps);
    int bounces = 0;
    RT_STATS_ONLY(
    Stats st; for (int q = 0; q < ST_N; ++q) st.c[q] = 0;
    for (int q = 0; q < 8; ++q) st.cyc[q] = 0;
    )
    float closest = FLT_MAX; int best = -1;
    TreeState ts; tree_state_init(ts);
    int lane_iters = 0;          // passes of this loop the path was alive for: its bounces plus the passes a long walk stayed pending (quorum) — the lane-time it costs
    while (__ballot(live) != 0ull) {
        const float a = dot3(r.d, r.d);
        if (!TREE) { closest = FLT_MAX; best = -1; }
        if (TREE) closest_tree<COOPG>(A.scene, A.tree, s_nodes, r, a, live, closest, best, ts STAT_PASS);
        else closest_list(A.scene, r, a, live, closest, best);
        if (live) ++lane_iters;
        if (live && !(TREE && ts.pending)) {
            ++bounces;
            bool done = true;
            if (best >= 0) { const bool cont = scatter(A.scene, best, closest, r, att, ps); done = !cont || bounces >= RT_PILOT_CAP; }
            if (done) live = false;
        }
    }
    int pix = inside ? bounces : 0;
    for (int off = RT_PILOT_SAMPLES / 2; off > 0; off >>= 1) pix += __shfl_xor(pix, off);       // sum over the pixel's samples
    RT_STATS_ONLY(
    if (tile_ok && smp == 0 && local_tile * 16 + sub < (1 << 20)) g_pilot_dbg[local_tile * 16 + sub] = inside ? pix : -1;
    )
    if (pilot && tile_ok && smp == 0) pilot[local_tile * 16 + sub] = (unsigned char)(pix < 255 ? pix : 255);       // per 2x2 block, for k_long_select
    int w = inside ? bounces : 0;
    for (int off = kPerTile / 2; off > 0; off >>= 1) w += __shfl_xor(w, off);         // sum over the tile's 16 pilot pixels x samples
    if (lane % kPerTile == 0 && tile_ok) cost[local_tile] = w * 4;
    if (work) {                                                                       // (rt_split_balanced: what the tile's walks tested)
        int tw = inside ? (int)ts.work : 0;
        for (int off = kPerTile / 2; off > 0; off >>= 1) tw += __shfl_xor(tw, off);
        int tc = inside ? lane_iters : 0;
        for (int off = kPerTile / 2; off > 0; off >>= 1) tc += __shfl_xor(tc, off);
        if (lane % kPerTile == 0 && tile_ok) { work[local_tile] = tw; work[A.n_local_tiles + local_tile] = tc; }       // [tiles] tests, [tiles] lane passes
        (void)ts.cols;
    }
}

RT_DEV int cost_class(int w) { const int c = (w - 64) / 64; return c < 0 ? 0 : (c > 7 ? 7 : c); }

#ifndef RT_TU_LIST
// one thread per 2x2 block: the pilot counts of the block and its eight neighbours decide whether its pixels start as long chains.
// A neighbour outside the frame, or in a tile of another part of a partitioned frame, counts as the block itself.
__global__ __launch_bounds__(256) void k_long_select(RenderArgs A, const unsigned char* __restrict__ pilot, unsigned char* __restrict__ long_flag,
                                                    unsigned int* __restrict__ long_list, int long_sum, int solo_sum, unsigned char* __restrict__ sum8) {
    const long long g = (long long)blockIdx.x * 256 + threadIdx.x;
    if (g >= A.n_local_tiles * 16) return;
    const long long local_tile = g >> 4;
    const int sub = (int)(g & 15);
    const long long tile = part_tile(local_tile, A.part, A.nparts, A.tile_begin, A.tile_end);
    const int tx = (int)(tile % A.tiles_x), ty = (int)(tile / A.tiles_x);
    const int bx = tx * 4 + (sub & 3), by = ty * 4 + (sub >> 2);
    const int own = pilot[g];
    int sum = 0;
#pragma unroll
    for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
        for (int dx = -1; dx <= 1; ++dx) {
            const int nx_ = bx + dx, ny_ = by + dy;
            int v = own;
            if (nx_ >= 0 && ny_ >= 0 && nx_ < A.tiles_x * 4 && ny_ < A.tiles_y * 4) {
                long long nl;
                if (part_has((long long)(ny_ >> 2) * A.tiles_x + (nx_ >> 2), A.part, A.nparts, A.tile_begin, A.tile_end, nl)) v = pilot[nl * 16 + (ny_ & 3) * 4 + (nx_ & 3)];
            }
            sum += v;
        }
    if (sum8) sum8[g] = (unsigned char)(sum < 255 ? sum : 255);                    // for the tail sort (k_tail_hist / k_tail_scatter)
    // (k_tile_order, which runs first, may have lowered the threshold: a chain is long relative to the launch's load per lane)
    const int abs_sum = (int)A.queue[kQueueThr + 1];
    const bool is_long = sum >= (abs_sum > 0 && abs_sum < long_sum ? abs_sum : long_sum);
    const int lx = 2 * (sub & 3), ly = 2 * (sub >> 2);
#pragma unroll
    for (int q = 0; q < 4; ++q) {                          // the 2x2 block this pilot pixel stands for
        const int px = lx + (q & 1), py = ly + (q >> 1);
        const bool in_img = (tx * 8 + px < A.max_x) && (ty * 8 + py < A.max_y);
        const long long pid = local_tile * 64 + py * 8 + px;
        long_flag[pid] = (is_long && in_img) ? 1 : 0;
        // the longest of them (3x3 sum >= solo_sum) are listed apart, from the end of the array: k_render starts each in a wave of its own
        if (is_long && in_img) {
            if (sum >= solo_sum) { const unsigned int pos = atomicAdd(A.queue + 4, 1u); long_list[A.n_local_tiles * 64 - 1 - pos] = (unsigned int)pid; }
            else { const unsigned int pos = atomicAdd(A.queue + 2, 1u); long_list[pos] = (unsigned int)pid; }
        }
    }
}

// one block of 16 waves: stable counting sort of the tiles by cost class, descending (each wave takes a contiguous chunk)
//
// It also sets the launch's yardstick for "long": the pilot's bounce counts predict the launch's iterations (a tile's cost is 4 x
// the bounces of its 32 pilot samples; 64 pixels x ns samples follow them), and iterations / lanes of the persistent grid is what
// one lane will work through — the LOAD.  A pixel whose chain is a sizeable fraction of the load decides when the launch ends
// unless it runs in a thin wave from early on; whether 3000 bounces are long depends on the launch (C5 whole frame: load 18 000;
// one part of eight: 2 250).  queue[kQueueThr] = in-flight threshold in iterations, queue[kQueueThr + 1] = the same as a 3x3 pilot sum (18 samples).
__global__ __launch_bounds__(1024) void k_tile_order(const int* __restrict__ cost, unsigned int* __restrict__ order, long long n, unsigned int* __restrict__ queue,
                                                    int ns, int n_lanes, float f_inflight, float f_static, float f_tail, long long tail_cap_tiles, unsigned int* __restrict__ tail_ws, int head_sum, float head_min_load) {
    if (tail_ws && threadIdx.x < 512) tail_ws[threadIdx.x] = 0u;   // (the tail sort's counts and cursors: k_tail_hist / k_tail_scatter run after this kernel)
    __shared__ int s_cnt[16][8];
    __shared__ long long s_sum[16];
    __shared__ long long s_ccost[16][8];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long per = ((n + 15) / 16 + 63) / 64 * 64;          // tiles per wave, a multiple of 64
    const long long lo = wave * per, hi = (lo + per < n) ? lo + per : n;
    int cnt[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    long long csum = 0;
    long long ccost[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    // (eight loads in flight per lane: one block sorts the whole frame's tiles, and a pass of dependent round trips per 64 tiles is what it costs)
    for (long long t0 = lo + lane; t0 < hi; t0 += 512) {
        int wv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) wv[u] = (t0 + 64 * u < hi) ? cost[t0 + 64 * u] : -1;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int w = wv[u] < 0 ? 0 : wv[u];
            const int c = wv[u] < 0 ? -1 : cost_class(w);
            csum += w;
#pragma unroll
            for (int k = 0; k < 8; ++k) { cnt[k] += (c == k) ? 1 : 0; ccost[k] += (c == k) ? w : 0; }
        }
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        for (int off = 32; off > 0; off >>= 1) ccost[k] += __shfl_xor(ccost[k], off);
        if (lane == 0) s_ccost[wave][k] = ccost[k];
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        for (int off = 32; off > 0; off >>= 1) cnt[k] += __shfl_xor(cnt[k], off);
        if (lane == 0) s_cnt[wave][k] = cnt[k];
    }
    for (int off = 32; off > 0; off >>= 1) csum += __shfl_xor(csum, off);
    if (lane == 0) s_sum[wave] = csum;
    __syncthreads();
    if (threadIdx.x == 0 && queue && n_lanes > 0 && ns > 0) {       // (s_cnt / s_sum / s_ccost are complete: the barrier above)
        long long tot = 0;
        for (int w = 0; w < 16; ++w) tot += s_sum[w];
        const double load = (double)tot * 0.5 * (double)ns / (double)n_lanes;          // predicted iterations per lane
        // (floors: in a launch with a pixel or two per lane every pixel is "long" by this measure — a chain must also be long as chains go)
        double ti = (double)f_inflight * load, tsum = 18.0 * (double)f_static * load / (double)ns;
        if (ti < (double)RT_LONG_RATE_MIN * ns) ti = (double)RT_LONG_RATE_MIN * ns;
        if (tsum < (double)RT_PILOT_LONG_SUM_MIN) tsum = (double)RT_PILOT_LONG_SUM_MIN;
        // the tail of the queue: the last tiles of the order that hold f_tail of the predicted work (tiles of one class taken as alike),
        // from a multiple of 64 on — their pixels are handed out by the tail sort's list (k_tail_hist / k_tail_scatter)
        unsigned int tail_mark = 0u;
        if (f_tail > 0.f && tot > 0) {
            const double want = (1.0 - (double)f_tail) * (double)tot;
            double cum = 0.0; long long r0 = n;
            long long start = 0;
#pragma unroll 1
            for (int k = 7; k >= 0; --k) {
                long long nk = 0, ck = 0;
#pragma unroll 1
                for (int w = 0; w < 16; ++w) { nk += s_cnt[w][k]; ck += s_ccost[w][k]; }
                if (nk > 0 && cum + (double)ck >= want) { r0 = start + (long long)((want - cum) / ((double)ck / (double)nk)); break; }
                cum += (double)ck; start += nk;
            }
            r0 = (r0 / 64) * 64;
            if (r0 < n && (n - r0) <= tail_cap_tiles) tail_mark = (unsigned int)(r0 * 64 + 1);
        }
        queue[kQueueThr + 2] = tail_mark;
        // the tail's pixels whose 3x3 pilot sum reaches head_sum are handed out before the tiles (k_tail_scatter counts them; 0 = none)
        // (head_sum < 0: the pixels whose sum is AT MOST -head_sum — the list's cheap end, the sky — are handed out first instead;
        // launches below head_min_load iterations per lane keep the whole list at the end: they need their cheapest pixels for the drain)
        if (load < (double)head_min_load) head_sum = 0;
        queue[kQueueThr + 4] = (tail_mark != 0u && head_sum > 0) ? (unsigned int)(head_sum > 255 ? 255 : head_sum) : (tail_mark != 0u && head_sum < 0) ? (unsigned int)(-head_sum >= 254 ? 255 : -head_sum + 1) : 0u;
        queue[kQueueThr + 5] = (tail_mark != 0u && head_sum < 0) ? 1u : 0u;
        queue[kQueueThr] = f_inflight > 0.f ? (unsigned int)(ti < 1.0 ? 1.0 : (ti > 4.0e9 ? 4.0e9 : ti)) : 0u;
        queue[kQueueThr + 1] = f_static > 0.f ? (unsigned int)(tsum < 1.0 ? 1.0 : (tsum > 1.0e9 ? 1.0e9 : tsum)) : 0u;
    }
    // class k: after every higher class, behind the earlier waves' share.  (Computed by 128 threads into LDS: with every thread summing
    // the 16 x 8 counts itself the kernel needed 1.1 KB of scratch per lane — for a grid of one block the runtime still sets scratch
    // aside for the whole chip, on a slow path that cost ~80 us a launch.)
    __shared__ int s_tot[8];
    __shared__ int s_base[16][8];
    if (threadIdx.x < 8) {
        int all = 0;
#pragma unroll 1
        for (int w = 0; w < 16; ++w) all += s_cnt[w][threadIdx.x];
        s_tot[threadIdx.x] = all;
    }
    __syncthreads();
    if (threadIdx.x < 128) {
        const int w0 = threadIdx.x >> 3, k0 = threadIdx.x & 7;
        int b = 0;
#pragma unroll 1
        for (int k = k0 + 1; k < 8; ++k) b += s_tot[k];
#pragma unroll 1
        for (int w = 0; w < w0; ++w) b += s_cnt[w][k0];
        s_base[w0][k0] = b;
    }
    __syncthreads();
    int base[8];                                                   // (positions: the frame's tiles fit 31 bits many times over)
#pragma unroll
    for (int k = 0; k < 8; ++k) base[k] = s_base[wave][k];
    const unsigned long long lt = (1ull << lane) - 1ull;
    for (long long t0 = lo; t0 < hi; t0 += 512) {
        int wv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) wv[u] = (t0 + 64 * u + lane < hi) ? cost[t0 + 64 * u + lane] : -1;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (t0 + 64 * u >= hi) break;                           // (wave-uniform)
            const long long t = t0 + 64 * u + lane;
            const int c = wv[u] < 0 ? -1 : cost_class(wv[u]);
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const unsigned long long m = __ballot(c == k);
                if (c == k) order[base[k] + __popcll(m & lt)] = (unsigned int)t;
                base[k] += __popcll(m);
            }
        }
    }
}

// The pixels of the queue's tail (tiles of rank >= R0 in the hand-out order), sorted by the pilot's count for their 2x2 block and its
// eight neighbours (k_long_select's sum8), most expensive first: a counting sort over the 256 values in two launches of many blocks
// (k_tail_hist: the histogram; k_tail_scatter: every block ranks its 1024 blocks among themselves in LDS and reserves its stretch of
// each value's range with one atomic) — as one block of 1024 threads it took 158 us of C3's 14.5 ms step, whatever its inner loop did.
// ws: 256 counts + 256 cursors, zeroed by k_tile_order.  Which lane renders a pixel and when never changes the pixel; the order inside
// one value is left to the atomics.
constexpr int kTailChunk = 1024;                                     // 2x2 blocks per thread block (four per thread)
RT_DEV int tail_key(const unsigned int* __restrict__ order, const unsigned char* __restrict__ sum8, long long r0, long long b, long long n_blocks, long long& tile) {
    tile = b < n_blocks ? (long long)order[r0 + (b >> 4)] : -1;
    return tile >= 0 ? 255 - (int)sum8[tile * 16 + (b & 15)] : -1;
}
__global__ __launch_bounds__(256) void k_tail_hist(const unsigned int* __restrict__ order, const unsigned char* __restrict__ sum8, long long n, const unsigned int* __restrict__ queue, unsigned int* __restrict__ ws) {
    __shared__ unsigned int s_bin[256];
    const unsigned int mark = queue[kQueueThr + 2];
    if (mark == 0u) return;
    const long long r0 = (long long)(mark - 1u) / 64, n_blocks = (n - r0) * 16;
    const long long b0 = (long long)blockIdx.x * kTailChunk;
    if (b0 >= n_blocks) return;
    s_bin[threadIdx.x] = 0u;
    __syncthreads();
    int key[4]; long long tile;
#pragma unroll
    for (int u = 0; u < 4; ++u) key[u] = tail_key(order, sum8, r0, b0 + 256 * u + threadIdx.x, n_blocks, tile);
#pragma unroll
    for (int u = 0; u < 4; ++u) if (key[u] >= 0) atomicAdd(&s_bin[key[u]], 4u);
    __syncthreads();
    const unsigned int c = s_bin[threadIdx.x];
    if (c != 0u) atomicAdd(&ws[threadIdx.x], c);
}
__global__ __launch_bounds__(256) void k_tail_scatter(const unsigned int* __restrict__ order, const unsigned char* __restrict__ sum8, unsigned int* __restrict__ tail_list,
                                                     long long n, unsigned int* __restrict__ queue, unsigned int* __restrict__ ws) {
    __shared__ unsigned int s_bin[256], s_base[256], s_wave[4];
    const unsigned int mark = queue[kQueueThr + 2];
    if (mark == 0u) return;
    const long long r0 = (long long)(mark - 1u) / 64, n_blocks = (n - r0) * 16;
    const long long b0 = (long long)blockIdx.x * kTailChunk;
    if (b0 >= n_blocks) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    s_bin[threadIdx.x] = 0u;
    // where value `threadIdx.x` begins in the list: exclusive prefix over the 256 counts (a scan per wave, then the waves' totals)
    const unsigned int cnt = ws[threadIdx.x];
    unsigned int wtot;
    const unsigned int ex = walk_excl_scan(cnt, wtot);
    if (lane == 0) s_wave[wave] = wtot;
    int key[4]; long long tile[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) key[u] = tail_key(order, sum8, r0, b0 + 256 * u + threadIdx.x, n_blocks, tile[u]);
    __syncthreads();
    unsigned int start = ex;
    for (int w = 0; w < wave; ++w) start += s_wave[w];
    // the list's first entries — every pixel whose 3x3 sum reaches the head threshold (values 255 ... thr = bins 0 ... 255 - thr) — are
    // handed out before the tiles: their count is where bin 256 - thr begins
    // (from the cheap end, queue[kQueueThr + 5]: every entry whose sum is below thr — what follows the entries counted above)
    { const unsigned int thr = queue[kQueueThr + 4];
      if (blockIdx.x == 0 && thr != 0u && threadIdx.x == 256u - thr) queue[kQueueThr + 3] = queue[kQueueThr + 5] != 0u ? (unsigned int)(n_blocks * 4) - start : start; }
    unsigned int local[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) local[u] = key[u] >= 0 ? atomicAdd(&s_bin[key[u]], 4u) : 0u;
    __syncthreads();
    const unsigned int mine = s_bin[threadIdx.x];
    s_base[threadIdx.x] = mine != 0u ? start + atomicAdd(&ws[256 + threadIdx.x], mine) : 0u;
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        if (key[u] < 0) continue;
        const int sub = (int)((b0 + 256 * u + threadIdx.x) & 15);
        const unsigned int pos = s_base[key[u]] + local[u];
        const int lx = 2 * (sub & 3), ly = 2 * (sub >> 2);
#pragma unroll
        for (int q = 0; q < 4; ++q) tail_list[pos + q] = (unsigned int)(tile[u] * 64 + (ly + (q >> 1)) * 8 + lx + (q & 1));
    }
}
#endif

// hitTree / hitable_list::hit for a batch of rays (one lane per ray)
template <bool TREE, int COOPG = 1>
__global__ __launch_bounds__(256) void k_trace(RenderArgs A, const float* rays, long long n, rt_hit_record* out) {
    const DevScene& S = A.scene; const DevTree& T = A.tree;
    extern __shared__ float4 s_nodes[];
    if (TREE) stage_tree_fp32<COOPG != 1>(S, T, s_nodes);
    const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
    const bool live = gid < n;
    RayF r; r.o = {0.f, 0.f, 0.f}; r.d = {0.f, 1.f, 0.f};
    if (live) { const float* p = rays + gid * 6; r.o = {p[0], p[1], p[2]}; r.d = {p[3], p[4], p[5]}; }
    const float a = dot3(r.d, r.d);
    float closest = FLT_MAX; int best = -1;
    RT_STATS_ONLY(
    Stats st; for (int q = 0; q < ST_N; ++q) st.c[q] = 0;
    for (int q = 0; q < 8; ++q) st.cyc[q] = 0;
    )
    if (TREE) {
        TreeState ts; tree_state_init(ts);
        bool act = live;
        do { closest_tree<COOPG>(S, T, s_nodes, r, a, act, closest, best, ts STAT_PASS); act = live && ts.pending; } while (__ballot(act) != 0ull);
    } else closest_list(S, r, a, live, closest, best);
    if (!live) return;
    rt_hit_record h;
    h.sphere = best; h.t = 0.f; h.p[0] = h.p[1] = h.p[2] = 0.f; h.normal[0] = h.normal[1] = h.normal[2] = 0.f;
    if (best >= 0) {
        const float4 g = S.geom[best];
        h.t = closest;
        h.p[0] = r.o.x + closest * r.d.x; h.p[1] = r.o.y + closest * r.d.y; h.p[2] = r.o.z + closest * r.d.z;
        h.normal[0] = (h.p[0] - g.x) / g.w; h.normal[1] = (h.p[1] - g.y) / g.w; h.normal[2] = (h.p[2] - g.z) / g.w;
    }
    out[gid] = h;
}

// gather of tile-major part buffers into the row-major frame (after the multi-GPU all-gather)
#ifndef RT_TU_LIST
__global__ __launch_bounds__(256) void k_assemble(float* full, const float* parts, int max_x, int max_y, int tiles_x, int nparts, long long part_stride_px, long long n_tiles) {
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
#endif

// the same for the bands of a balanced split (rt_split_balanced): band p holds the tiles [starts.s[p], starts.s[p + 1]), its buffer begins
// part_stride_px elements behind band p - 1's.  One kernel for both precisions (T = float / uint16_t).
#ifndef RT_TU_LIST
template <class T>
__global__ __launch_bounds__(256) void k_assemble_split(T* full, const T* parts, int max_x, int max_y, int tiles_x, int nparts, long long part_stride_px, SplitStarts starts) {
    const int lane = threadIdx.x & 63;
    const long long tile = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (tile >= starts.s[nparts]) return;
    const int tx = (int)(tile % tiles_x), ty = (int)(tile / tiles_x);
    const int i = tx * 8 + (lane & 7), j = ty * 8 + (lane >> 3);
    if (i >= max_x || j >= max_y) return;
    int owner = 0;
    for (int p = 1; p < nparts; ++p) owner += tile >= starts.s[p] ? 1 : 0;
    const long long src = owner * part_stride_px + (tile - starts.s[owner]) * 64 + lane;
    const long long dst = (long long)j * max_x + i;
    full[dst * 3 + 0] = parts[src * 3 + 0]; full[dst * 3 + 1] = parts[src * 3 + 1]; full[dst * 3 + 2] = parts[src * 3 + 2];
}
hipError_t launch_assemble_split(void* full, const void* parts, int max_x, int max_y, int nparts, const long long* starts, long long part_stride_px, bool half, hipStream_t st) {
    SplitStarts S;
    for (int p = 0; p <= nparts; ++p) S.s[p] = starts[p];
    const int tiles_x = (max_x + 7) / 8;
    const unsigned blocks = (unsigned)((starts[nparts] + 3) / 4);
    if (half) hipLaunchKernelGGL((k_assemble_split<uint16_t>), dim3(blocks), dim3(256), 0, st, (uint16_t*)full, (const uint16_t*)parts, max_x, max_y, tiles_x, nparts, part_stride_px, S);
    else hipLaunchKernelGGL((k_assemble_split<float>), dim3(blocks), dim3(256), 0, st, (float*)full, (const float*)parts, max_x, max_y, tiles_x, nparts, part_stride_px, S);
    return hipGetLastError();
}
#endif

// ---------------------------------------------------------------------------------------------------- launchers
#ifndef RT_TU_LIST
// Zeroes the work counters of a launch.  A kernel of our own instead of hipMemsetAsync: captured into a hipGraph, the memset
// node took effect on the first replay only (ROCm 7.2) — later replays found the queue exhausted and rendered nothing.
__global__ void k_zero_counters(unsigned int* p, int n) { if ((int)threadIdx.x < n) p[threadIdx.x] = 0u; }
hipError_t launch_zero_counters(unsigned int* p, int n, hipStream_t st) {
    hipLaunchKernelGGL(k_zero_counters, dim3(1), dim3(64), 0, st, p, n);
    return hipGetLastError();
}

hipError_t launch_render_init(rt_rand_state* rs, int max_x, int max_y, int part, int nparts, long long begin, long long end, hipStream_t st) {
    const int tiles_x = (max_x + 7) / 8, tiles_y = (max_y + 7) / 8;
    const long long tiles = (long long)tiles_x * tiles_y;
    const long long local = part_local_tiles(tiles, part, nparts, begin, end);
    if (local <= 0) return hipSuccess;
    const unsigned blocks = (unsigned)((local + 3) / 4);
    hipLaunchKernelGGL(k_render_init, dim3(blocks), dim3(256), 0, st, rs, max_x, max_y, tiles_x, part, nparts, begin, end, local);
    return hipGetLastError();
}
#endif

// LDS of a block of the tree kernels: the nodes, then one WalkLds per wave
static size_t tree_lds_bytes(int n_nodes, bool pool = false) { return (size_t)n_nodes * sizeof(DevNode) + (pool ? 4 * sizeof(WalkLds) + kHotSlots * sizeof(float4) : 0); }

// blocks the chip holds at once for one render kernel variant (occupancy query, cached); the persistent grid is never
// larger than that, and never larger than the work
template <class K> static unsigned resident_blocks(K kernel, size_t lds) {
    int dev = 0, cus = 0, per_cu = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 1024u;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) return 1024u;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, 256, lds) != hipSuccess || per_cu <= 0) per_cu = 4;
    return (unsigned)(cus * per_cu);
}

// The hitable_list instantiations live in their own translation unit (rt_kernels_list.hip = this file with RT_TU_LIST):
// SLP vectorisation (packed fp32, v_pk_*_f32) makes the list scan 10 % faster and the tree walk 4 % slower on gfx950, so
// the two are compiled with different flags (Makefile).  Without RT_SPLIT_LIST (diagnostic build) everything is here.
#if defined(RT_TU_LIST) || !defined(RT_SPLIT_LIST)
#ifdef RT_TU_LIST
#define RT_LIST_FN(name) name##_list
#else
#define RT_LIST_FN(name) static name##_list
#endif
hipError_t RT_LIST_FN(launch_tile_cost)(const RenderArgs& A, unsigned blocks, int* cost, unsigned char* pilot, hipStream_t st) {
    hipLaunchKernelGGL((k_tile_cost<false>), dim3(blocks), dim3(256), 0, st, A, cost, pilot, (int*)nullptr);
    return hipGetLastError();
}
hipError_t RT_LIST_FN(launch_render)(const RenderArgs& A, int mode, hipStream_t st) {
    const unsigned need = (unsigned)((A.n_local_tiles + 3) / 4);
    const unsigned cap = mode == 0 ? resident_blocks(k_render<false, 0, 1>, 0) : resident_blocks(k_render<false, 1, 1>, 0);
    const unsigned blocks = need < cap ? need : cap;
    if (mode == 0) hipLaunchKernelGGL((k_render<false, 0, 1>), dim3(blocks), dim3(256), 0, st, A);
    else hipLaunchKernelGGL((k_render<false, 1, 1>), dim3(blocks), dim3(256), 0, st, A);
    return hipGetLastError();
}
hipError_t RT_LIST_FN(launch_trace)(const RenderArgs& A, unsigned blocks, const float* rays, long long n, rt_hit_record* out, hipStream_t st) {
    hipLaunchKernelGGL((k_trace<false>), dim3(blocks), dim3(256), 0, st, A, rays, n, out);
    return hipGetLastError();
}
#else
hipError_t launch_tile_cost_list(const RenderArgs& A, unsigned blocks, int* cost, unsigned char* pilot, hipStream_t st);
hipError_t launch_render_list(const RenderArgs& A, int mode, hipStream_t st);
hipError_t launch_trace_list(const RenderArgs& A, unsigned blocks, const float* rays, long long n, rt_hit_record* out, hipStream_t st);
#endif

#ifndef RT_TU_LIST
// Which k_render instantiation a call launches — the ONE place that decides (launch_render and rt_render_kernel_name):
// 0 = k_render<false,MODE,1> (list scan), 1 = k_render<true,MODE,1> (per-lane walk: trees without a candidate grid, reference
// traversal), 4 = k_render<true,MODE,4> (sparse grids: walk_pool), 2 = k_render<true,MODE,2> (dense grids: walk_pool_dense) —
// render (MODE 0) and render_progressive (MODE 1) walk the same way (rt_accel.h: coop_groups)
static int render_variant(bool tree, int mode, const DevAccel& acc) {
    (void)mode;
    if (!tree) return 0;
    if (acc.enabled) return acc.coop_groups >= 4 ? (acc.solo_chains ? 5 : 4) : 2;
    return 1;
}
const char* render_kernel_name(bool tree, int mode, const DevAccel& acc) {
    switch (render_variant(tree, mode, acc)) {
        case 0: return mode == 0 ? "k_render<false,0,1>" : "k_render<false,1,1>";
        case 4: return mode == 0 ? "k_render<true,0,4>" : "k_render<true,1,4>";
        case 5: return mode == 0 ? "k_render<true,0,5>" : "k_render<true,1,5>";
        case 2: return mode == 0 ? "k_render<true,0,2>" : "k_render<true,1,2>";
        default: return mode == 0 ? "k_render<true,0,1>" : "k_render<true,1,1>";
    }
}

// after a pilot pass (k_tile_cost here, k_tile_cost_h in rt_kernels_fp16.hip) has written the tile costs and, behind the pixel flags,
// the per-block counts: the long-chain list (if asked for) and the hand-out order of the tiles
// (long_sum: the 3x3 pilot sum from which a block's pixels start as long chains; 0 = this translation unit's RT_PILOT_LONG_SUM)
hipError_t launch_select_and_order(const RenderArgs& A, int* cost, unsigned int* order, unsigned char* flags, unsigned int* long_list, hipStream_t st, int long_sum, int solo_sum) {
    // (the order kernel first: it also derives the launch's thresholds for long chains, which the selection reads)
    // flags: 64 bytes per local tile (one per pixel), 16 (the pilot's count per 2x2 block), 16 (that count summed over the block's 3x3 neighbourhood)
    const bool tail = flags && A.tail_list && A.tail_ws && A.f_tail > 0.f;
    hipLaunchKernelGGL(k_tile_order, dim3(1), dim3(1024), 0, st, (const int*)cost, order, (long long)A.n_local_tiles, A.queue, (int)A.ns, (int)A.n_lanes, A.f_inflight, A.f_static,
                       tail ? A.f_tail : 0.f, (long long)A.n_local_tiles, tail ? A.tail_ws : (unsigned int*)nullptr, tail ? A.head_sum : 0, A.head_min_load);
    if (flags) {
        const unsigned char* pilot = flags + (size_t)A.n_local_tiles * 64;
        unsigned char* sum8 = flags + (size_t)A.n_local_tiles * 80;
        hipLaunchKernelGGL(k_long_select, dim3((unsigned)((A.n_local_tiles * 16 + 255) / 256)), dim3(256), 0, st, A, pilot, flags, long_list, long_sum > 0 ? long_sum : RT_PILOT_LONG_SUM, solo_sum, sum8);
        if (tail) {
            const unsigned nb = (unsigned)((A.n_local_tiles * 16 + kTailChunk - 1) / kTailChunk);      // (the tail's size is known on the device only: blocks beyond it return at once)
            hipLaunchKernelGGL(k_tail_hist, dim3(nb), dim3(256), 0, st, (const unsigned int*)order, (const unsigned char*)sum8, (long long)A.n_local_tiles, (const unsigned int*)A.queue, A.tail_ws);
            hipLaunchKernelGGL(k_tail_scatter, dim3(nb), dim3(256), 0, st, (const unsigned int*)order, (const unsigned char*)sum8, (unsigned int*)A.tail_list, (long long)A.n_local_tiles, A.queue, A.tail_ws);
        }
    }
    return hipGetLastError();
}

static unsigned render_grid_blocks(const RenderArgs& A, int variant, int mode);
// the pilot pass alone: per tile the bounces (x 4) and — trees with a candidate grid — the grid entries its samples' walks pooled
hipError_t launch_pilot(const RenderArgs& A, bool tree, int* cost, unsigned char* pilot, int* work, hipStream_t st) {
    if (A.n_local_tiles <= 0) return hipSuccess;
    const long long per_block = 4 * (64 / (16 * RT_PILOT_SAMPLES));       // a wave covers 4 / RT_PILOT_SAMPLES tiles
    const unsigned blocks = (unsigned)((A.n_local_tiles + per_block - 1) / per_block);
    // (the pilot paths walk the grid like the render kernel's waves do, through the wave's pool)
    const int variant = render_variant(tree, 0, A.tree.acc);
    if (variant == 5) hipLaunchKernelGGL((k_tile_cost<true, 5>), dim3(blocks), dim3(256), tree_lds_bytes(A.tree.n_nodes, true), st, A, cost, pilot, work);
    else if (variant == 4) hipLaunchKernelGGL((k_tile_cost<true, 4>), dim3(blocks), dim3(256), tree_lds_bytes(A.tree.n_nodes, true), st, A, cost, pilot, work);
    else if (variant == 2) hipLaunchKernelGGL((k_tile_cost<true, 2>), dim3(blocks), dim3(256), tree_lds_bytes(A.tree.n_nodes, true), st, A, cost, pilot, work);
    else if (tree) hipLaunchKernelGGL((k_tile_cost<true, 1>), dim3(blocks), dim3(256), tree_lds_bytes(A.tree.n_nodes), st, A, cost, pilot, work);
    else return launch_tile_cost_list(A, blocks, cost, pilot, st);
    return hipGetLastError();
}

hipError_t launch_tile_order(const RenderArgs& A, bool tree, int* cost, unsigned int* order, unsigned char* flags, unsigned int* long_list, hipStream_t st) {
    if (A.n_local_tiles <= 0) return hipSuccess;
    // flags: 64 bytes per local tile (one per pixel) followed by 16 per local tile (the pilot counts per 2x2 block) and 16 more (k_long_select)
    unsigned char* pilot = flags ? flags + (size_t)A.n_local_tiles * 64 : nullptr;
    const int variant = render_variant(tree, 0, A.tree.acc);
    RenderArgs B = A;
    B.n_lanes = tree ? (int)render_grid_blocks(A, variant, 0) * 256 : 0;      // (list scans keep the rate rule alone)
    if (variant == 2 && A.f_inflight_dense > 0.f) B.f_inflight = A.f_inflight_dense;
    B.head_sum = variant == 2 ? A.head_sum_dense : A.head_sum;
    B.head_min_load = variant == 2 ? A.head_min_load : 0.f;
    { const hipError_t e = launch_pilot(A, tree, cost, pilot, nullptr, st); if (e != hipSuccess) return e; }
    // (chains in waves of their own: the variant for very sparse grids)
    return launch_select_and_order(B, cost, order, flags, long_list, st, 0, variant == 5 ? RT_PILOT_SOLO_SUM : 0x7fffffff);
}

// blocks of the persistent grid of k_render<true, MODE, COOPG> for this launch: what the chip holds, never more than the work
template <int MODE, int COOPG>
static unsigned tree_grid_blocks(const RenderArgs& A, size_t lds) {
    const unsigned need = (unsigned)((A.n_local_tiles + 3) / 4);
    // the occupancy query costs as much host time as the launch: remembered per thread for the last (device, LDS size) —
    // a progressive loop issues the same launch hundreds of times
    thread_local int c_dev = -1; thread_local size_t c_lds = 0; thread_local unsigned c_cap = 0;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = -2;
    if (dev != c_dev || lds != c_lds || c_cap == 0) { c_cap = resident_blocks(k_render<true, MODE, COOPG>, lds); c_dev = dev; c_lds = lds; }
    const unsigned cap = c_cap;
    return need < cap ? need : cap;
}
template <int MODE, int COOPG>
static hipError_t launch_render_tree(const RenderArgs& A, size_t lds, hipStream_t st) {
    hipLaunchKernelGGL((k_render<true, MODE, COOPG>), dim3(tree_grid_blocks<MODE, COOPG>(A, lds)), dim3(256), lds, st, A);
    return hipGetLastError();
}
// the same for a tree variant chosen at run time (the scheduling pass wants the grid's lanes before the launch)
static unsigned render_grid_blocks(const RenderArgs& A, int variant, int mode) {
    const size_t lds = tree_lds_bytes(A.tree.n_nodes, variant != 1);
    if (variant == 5) return mode == 0 ? tree_grid_blocks<0, 5>(A, lds) : tree_grid_blocks<1, 5>(A, lds);
    if (variant == 4) return mode == 0 ? tree_grid_blocks<0, 4>(A, lds) : tree_grid_blocks<1, 4>(A, lds);
    if (variant == 2) return mode == 0 ? tree_grid_blocks<0, 2>(A, lds) : tree_grid_blocks<1, 2>(A, lds);
    return mode == 0 ? tree_grid_blocks<0, 1>(A, lds) : tree_grid_blocks<1, 1>(A, lds);
}

hipError_t launch_render(const RenderArgs& A, bool tree, int mode, hipStream_t st) {
    if (A.n_local_tiles <= 0) return hipSuccess;
    const int variant = render_variant(tree, mode, A.tree.acc);
    if (variant == 0) return launch_render_list(A, mode, st);
    const size_t lds = tree_lds_bytes(A.tree.n_nodes, variant != 1);                       // (the variants with a pooled walk)
    if (variant == 5) return mode == 0 ? launch_render_tree<0, 5>(A, lds, st) : launch_render_tree<1, 5>(A, lds, st);
    if (variant == 4) return mode == 0 ? launch_render_tree<0, 4>(A, lds, st) : launch_render_tree<1, 4>(A, lds, st);
    if (variant == 2) return mode == 0 ? launch_render_tree<0, 2>(A, lds, st) : launch_render_tree<1, 2>(A, lds, st);
    return mode == 0 ? launch_render_tree<0, 1>(A, lds, st) : launch_render_tree<1, 1>(A, lds, st);
}

hipError_t launch_trace(const DevScene& S, const DevTree& T, bool tree, const float* rays, long long n, rt_hit_record* out, hipStream_t st) {
    if (n <= 0) return hipSuccess;
    const unsigned blocks = (unsigned)((n + 255) / 256);
    RenderArgs A{};
    A.scene = S; A.tree = T;
    if (!tree) return launch_trace_list(A, blocks, rays, n, out, st);
    // the walk the render kernel would use for this tree (sparse grids: the pooled walk), so that per-ray parity checks cover it
    if (render_variant(true, 0, T.acc) == 5) hipLaunchKernelGGL((k_trace<true, 5>), dim3(blocks), dim3(256), tree_lds_bytes(T.n_nodes, true), st, A, rays, n, out);
    else if (render_variant(true, 0, T.acc) == 4) hipLaunchKernelGGL((k_trace<true, 4>), dim3(blocks), dim3(256), tree_lds_bytes(T.n_nodes, true), st, A, rays, n, out);
    else if (render_variant(true, 0, T.acc) == 2) hipLaunchKernelGGL((k_trace<true, 2>), dim3(blocks), dim3(256), tree_lds_bytes(T.n_nodes, true), st, A, rays, n, out);
    else hipLaunchKernelGGL((k_trace<true>), dim3(blocks), dim3(256), tree_lds_bytes(T.n_nodes), st, A, rays, n, out);
    return hipGetLastError();
}

hipError_t launch_assemble(float* full, const float* parts, int max_x, int max_y, int nparts, hipStream_t st) {
    const int tiles_x = (max_x + 7) / 8, tiles_y = (max_y + 7) / 8;
    const long long tiles = (long long)tiles_x * tiles_y;
    const long long per_part = part_local_tiles(tiles, 0, nparts) * 64;
    const unsigned blocks = (unsigned)((tiles + 3) / 4);
    hipLaunchKernelGGL(k_assemble, dim3(blocks), dim3(256), 0, st, full, parts, max_x, max_y, tiles_x, nparts, per_part, tiles);
    return hipGetLastError();
}
#endif

RT_STATS_READERS

#ifdef RT_TU_CONTRACT
} // namespace fmac
#endif
} // namespace rt
