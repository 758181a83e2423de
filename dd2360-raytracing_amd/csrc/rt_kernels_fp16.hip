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

// hitTree (acceleration_structure.h:319-342) in two phases per round, so that the 64 rays of a wave do not wait for each
// other at every level-3 node: (1) walk the tree — the visited set does not depend on the hits (traverseTree prunes by the
// slab test only) — and note the bucket ranges of up to kRanges visited non-empty nodes per lane in LDS; (2) scan them as ONE
// flat sequence of sphere tests per lane, in the reference's order (nodes in traversal order, entries in bucket order).
// Walking and scanning in lock step per node ran at 14 % lane utilisation (every node a barrier for the whole wave).
constexpr int kRanges = 24;                                   // nodes noted per lane and round (24 x 256 x 2 B of LDS)
RT_DEV void closest_tree(const DevScene& S, const DevTree& T, const float4* s_nodes, const Ray& r, R a, bool live, R& closest, int& best) {
    if (S.ground_valid) {
        int gb = -1;
        sphere_test(r, a, S.list_hot[0], 0, closest, gb);
        if (gb == 0) best = 0;
    }
    unsigned short* noted = (unsigned short*)(s_nodes + T.n_nodes * 3) + threadIdx.x;      // noted[k * 256]: this lane's k-th visited non-empty node
    int e_best = -1;
    int node = live ? 0 : T.n_nodes;
    const int n_nodes = T.n_nodes;
    while (true) {
        int nr = 0;
        while (node < n_nodes && nr < kRanges) {
            const float4 n0 = s_nodes[node * 3 + 0];
            const float4 n1 = s_nodes[node * 3 + 1];
            const float4 n2 = s_nodes[node * 3 + 2];
            if (ray_box(r, n0, n1)) {
                const int cnt = __float_as_int(n2.x);
                node = node + 1;
                if (cnt > 0) { noted[nr * 256] = (unsigned short)(node - 1); ++nr; }
            } else {
                node = __float_as_int(n1.z);
            }
        }
        if (__ballot(nr > 0) == 0ull) break;
        int k = 0, e = 0, e_end = 0;
        while (true) {
            if (e >= e_end && k < nr) {
                const int nd = (int)noted[k * 256]; ++k;
                e = __float_as_int(s_nodes[nd * 3 + 1].w); e_end = e + __float_as_int(s_nodes[nd * 3 + 2].x);
            }
            const bool has = e < e_end;
            if (__ballot(has) == 0ull) break;
            if (has) {
                // up to four entries of the range per pass, their loads in flight together; tested strictly in order
                const int m = e_end - e;
                const uint2* __restrict__ ent = (const uint2*)T.ent_hot;
                const uint2 s0 = ent[e], s1 = ent[m > 1 ? e + 1 : e], s2 = ent[m > 2 ? e + 2 : e], s3 = ent[m > 3 ? e + 3 : e];
                sphere_test(r, a, s0, e, closest, e_best);
                if (m > 1) sphere_test(r, a, s1, e + 1, closest, e_best);
                if (m > 2) sphere_test(r, a, s2, e + 2, closest, e_best);
                if (m > 3) sphere_test(r, a, s3, e + 3, closest, e_best);
                e += m > 4 ? 4 : m;
            }
        }
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
    const float4 g = S.geom[sphere];
    const float4 m = S.mat[sphere];
    const int kind = S.kind[sphere];
    const V c = {rf(g.x), rf(g.y), rf(g.z)};
    const V p = vadd(r.o, vscale(t, r.d));                                  // ray.h:13
    const V n = vdiv(vsub(p, c), rf(g.w));                                  // sphere.h:29
    const V albedo = {rf(m.x), rf(m.y), rf(m.z)};
    if (kind == RT_MAT_LAMBERTIAN) {
        const V target = vadd(vadd(p, n), random_in_unit_sphere(s));
        r.d = vsub(target, p); r.o = p;
        att = vmul(att, albedo);
        return true;
    }
    if (kind == RT_MAT_METAL) {
        const V ud = vunit(r.d);
        const V refl = vsub(ud, vscale(rf(2.0f) * vdot(ud, n), n));
        r.d = vadd(refl, vscale(rf(m.w), random_in_unit_sphere(s)));
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
    const V uv = vunit(r.d);
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

template <bool TREE, int MODE>
__global__ __launch_bounds__(256) void k_render_h(RenderArgs A) {
    extern __shared__ float4 s_nodes[];
    if (TREE) {
        const int n4 = A.tree.n_nodes * 3;
        for (int t = threadIdx.x; t < n4; t += 256) s_nodes[t] = A.tree.nodes4[t];
        __syncthreads();
    }
    const int lane = threadIdx.x & 63;
    const long long n_slots = A.n_local_tiles * 64;
    const long long first_free = (long long)gridDim.x * 256;
    const int ns = (MODE == 0) ? A.ns : 1;
    const Cam cam = load_camera(A.scene.cam);

    long long slot = ((long long)blockIdx.x * 4 + (threadIdx.x >> 6)) * 64 + lane;
    int i = 0, j = 0; long long idx = 0;
    Rng s = {0, 0, 0, 0, 0, 0};
    V col = {ri(0), ri(0), ri(0)};
    V att = {rd(1.0), rd(1.0), rd(1.0)};
    Ray r; r.o = col; r.d = att;
    int sample = 0, depth = 0;
    bool live = false;

    auto begin_pixel = [&]() {
        live = false;
        while (slot < n_slots) {
            const long long local_tile = slot >> 6;
            const int l = (int)(slot & 63);
            const long long tile = A.part + local_tile * A.nparts;
            const int tx = (int)(tile % A.tiles_x), ty = (int)(tile / A.tiles_x);
            i = tx * 8 + (l & 7); j = ty * 8 + (l >> 3);
            if (i < A.max_x && j < A.max_y) { idx = (A.nparts == 1) ? (long long)j * A.max_x + i : slot; live = true; break; }
            slot = first_free + (long long)atomicAdd(A.queue, 1u);
        }
        if (live) {
            const rt_rand_state* st = A.rand_state + idx;
            s.d = st->d; s.v0 = st->v[0]; s.v1 = st->v[1]; s.v2 = st->v[2]; s.v3 = st->v[3]; s.v4 = st->v[4];
            col = {ri(0), ri(0), ri(0)}; att = {rd(1.0), rd(1.0), rd(1.0)}; sample = 0; depth = 0;
            r = primary_ray(cam, i, j, A.max_x, A.max_y, s);
        }
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
    if (ns > 0) begin_pixel();

    while (__ballot(live) != 0ull) {
        const R a = vdot(r.d, r.d);
        R closest = rf(FLT_MAX); int best = -1;                              // real_t(FLT_MAX) = +inf in binary16
        if (TREE) closest_tree(A.scene, A.tree, s_nodes, r, a, live, closest, best);
        else closest_list(A.scene, r, a, closest, best);
        if (live) {
            bool done;
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
                if (sample < ns) r = primary_ray(cam, i, j, A.max_x, A.max_y, s);
                else {
                    end_pixel();
                    slot = first_free + (long long)atomicAdd(A.queue, 1u);
                    begin_pixel();
                }
            }
        }
    }
}

template <bool TREE>
__global__ __launch_bounds__(256) void k_trace_h(DevScene S, DevTree T, const float* rays, long long n, rt_hit_record* out) {
    extern __shared__ float4 s_nodes[];
    if (TREE) {
        const int n4 = T.n_nodes * 3;
        for (int t = threadIdx.x; t < n4; t += 256) s_nodes[t] = T.nodes4[t];
        __syncthreads();
    }
    const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
    const bool live = gid < n;
    Ray r; r.o = {ri(0), ri(0), ri(0)}; r.d = {ri(0), ri(1), ri(0)};
    if (live) { const float* p = rays + gid * 6; r.o = vload(p); r.d = vload(p + 3); }
    const R a = vdot(r.d, r.d);
    R closest = rf(FLT_MAX); int best = -1;
    if (TREE) closest_tree(S, T, s_nodes, r, a, live, closest, best);
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
    const long long src = (tile % nparts) * part_stride_px + (tile / nparts) * 64 + lane;
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

hipError_t launch_render_h(const RenderArgs& A, bool tree, int mode, hipStream_t st) {
    if (A.n_local_tiles <= 0) return hipSuccess;
    const unsigned need = (unsigned)((A.n_local_tiles + 3) / 4);
    const size_t lds = tree ? (size_t)A.tree.n_nodes * sizeof(DevNode) + (size_t)h16::kRanges * 256 * sizeof(unsigned short) : 0;
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
    const size_t lds = tree ? (size_t)T.n_nodes * sizeof(DevNode) + (size_t)h16::kRanges * 256 * sizeof(unsigned short) : 0;
    if (tree) hipLaunchKernelGGL((h16::k_trace_h<true>), dim3(blocks), dim3(256), lds, st, S, T, rays, n, out);
    else hipLaunchKernelGGL((h16::k_trace_h<false>), dim3(blocks), dim3(256), lds, st, S, T, rays, n, out);
    return hipGetLastError();
}

hipError_t launch_assemble_h(void* full, const void* parts, int max_x, int max_y, int nparts, hipStream_t st) {
    const int tiles_x = (max_x + 7) / 8, tiles_y = (max_y + 7) / 8;
    const long long tiles = (long long)tiles_x * tiles_y;
    const long long per_part = (tiles + nparts - 1) / nparts * 64;
    const unsigned blocks = (unsigned)((tiles + 3) / 4);
    hipLaunchKernelGGL(h16::k_assemble_h, dim3(blocks), dim3(256), 0, st, (uint16_t*)full, (const uint16_t*)parts, max_x, max_y, tiles_x, nparts, per_part, tiles);
    return hipGetLastError();
}

} // namespace rt
