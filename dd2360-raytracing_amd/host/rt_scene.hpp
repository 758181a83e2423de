// rt_scene.hpp — host-side mirror of the reference's scene interfaces for the MI355X render path.
//
// Same names and constructor arguments as the reference (vec3.h, ray.h, hitable.h, sphere.h, hitable_list.h,
// material.h, camera.h, acceleration_structure.h) so that a main.cu-style program builds its scene the same way;
// the objects describe the scene and serialise to the PODs of include/rt_amd.h.  The interfaces are callable on the host
// as well — hitable::hit (hitable.h:19), material::scatter (material.h:49), camera::get_ray (camera.h:45), with the
// reference's argument lists and the numeric contract of the kernels — for host-side picking, scene checks and tests;
// rendering itself runs only on the GPU (csrc/rt_kernels.hip): no rt_* entry point ever calls them.
//
// Everything is a template over real_t (float or rt::half_t) because the reference selects real_t at compile
// time (precision_types.h:8) while this library selects it per call.
//
// Numeric contract: binary32 per-op rounding, no FMA contraction, RNG draws consumed left to right.
#pragma once
#include <vector>
#include <memory>
#include <cstdio>
#include <cmath>
#include "../csrc/rt_real.h"
#include "../csrc/rt_octgeom.h"
#include "../../include/rt_amd.h"

namespace rt {

// ---------------------------------------------------------------------------------------------- vec3 (vec3.h:9-148)
template <class R> class vec3_t {
public:
    vec3_t() {}
    vec3_t(R e0, R e1, R e2) { e[0] = e0; e[1] = e1; e[2] = e2; }
    R x() const { return e[0]; }
    R y() const { return e[1]; }
    R z() const { return e[2]; }
    R operator[](int i) const { return e[i]; }
    R& operator[](int i) { return e[i]; }
    R squared_length() const { return e[0] * e[0] + e[1] * e[1] + e[2] * e[2]; }
    R length() const { return real_from<R>(sqrtf(as_float(squared_length()))); }   // sqrt(real_t) is the float sqrt
    R e[3];
};
template <class R> vec3_t<R> operator+(const vec3_t<R>& a, const vec3_t<R>& b) { return vec3_t<R>(a.e[0] + b.e[0], a.e[1] + b.e[1], a.e[2] + b.e[2]); }
template <class R> vec3_t<R> operator-(const vec3_t<R>& a, const vec3_t<R>& b) { return vec3_t<R>(a.e[0] - b.e[0], a.e[1] - b.e[1], a.e[2] - b.e[2]); }
template <class R> vec3_t<R> operator*(R t, const vec3_t<R>& v) { return vec3_t<R>(t * v.e[0], t * v.e[1], t * v.e[2]); }
template <class R> vec3_t<R> operator/(const vec3_t<R>& v, R t) { return vec3_t<R>(v.e[0] / t, v.e[1] / t, v.e[2] / t); }
template <class R> R dot(const vec3_t<R>& a, const vec3_t<R>& b) { return a.e[0] * b.e[0] + a.e[1] * b.e[1] + a.e[2] * b.e[2]; }
template <class R> R negate(R v) { return real_from<R>(-as_float(v)); }     // unary minus goes through float: exact
template <class R> vec3_t<R> cross(const vec3_t<R>& a, const vec3_t<R>& b) {
    return vec3_t<R>(a.e[1] * b.e[2] - a.e[2] * b.e[1], negate<R>(a.e[0] * b.e[2] - a.e[2] * b.e[0]), a.e[0] * b.e[1] - a.e[1] * b.e[0]);
}
template <class R> vec3_t<R> unit_vector(const vec3_t<R>& v) { return v / v.length(); }
template <class R> vec3_t<R> operator*(const vec3_t<R>& a, const vec3_t<R>& b) { return vec3_t<R>(a.e[0] * b.e[0], a.e[1] * b.e[1], a.e[2] * b.e[2]); }

// ---------------------------------------------------------------------------------------------- ray (ray.h:6-17), hit_record (hitable.h:9-15)
template <class R> class ray_t {
public:
    ray_t() {}
    ray_t(const vec3_t<R>& a, const vec3_t<R>& b) : A(a), B(b) {}
    vec3_t<R> origin() const { return A; }
    vec3_t<R> direction() const { return B; }
    vec3_t<R> point_at_parameter(R t) const { return A + t * B; }
    vec3_t<R> A, B;
};
template <class R> class material_t;
template <class R> struct hit_record_t {
    R t;
    vec3_t<R> p, normal;
    const material_t<R>* mat_ptr;
};

struct xorwow;
template <class R> vec3_t<R> random_in_unit_sphere(rt_rand_state* local_rand_state);       // material.h:35-41 (below, after the RNG)
// pow((1.0f - cosine), 5.0f) of schlick (material.h:14).  Contract shared with the kernels: binary64 ((x*x)*(x*x))*x, rounded once.
inline float pow5(float x) { const double v = (double)x; const double v2 = v * v; return (float)((v2 * v2) * v); }

// ---------------------------------------------------------------------------------------------- materials (material.h:47-116)
// Expressions of the reference that mix float and real_t are written out with their C++ conversions (as_float / real_from),
// so one body serves both real types: with R = float every conversion is the identity.
template <class R> R sqrt_real(R x) { return real_from<R>(sqrtf(as_float(x))); }            // real_t::sqrt / sqrt(real_t)
template <class R> vec3_t<R> reflect(const vec3_t<R>& v, const vec3_t<R>& n) { return v - (real_from<R>(2.0f) * dot(v, n)) * n; }   // material.h:43-45
template <class R> bool refract(const vec3_t<R>& v, const vec3_t<R>& n, R ni_over_nt, vec3_t<R>& refracted) {                       // material.h:17-31
    const vec3_t<R> uv = unit_vector(v);
    const R dt = dot(uv, n);
    const R discriminant = real_from<R>(1.0f) - ni_over_nt * ni_over_nt * (real_from<R>(1.0f) - dt * dt);
    if (discriminant > real_from_int<R>(0)) { refracted = ni_over_nt * (uv - dt * n) - sqrt_real(discriminant) * n; return true; }
    return false;
}
template <class R> R schlick(R cosine, R ref_idx) {                                                                                 // material.h:11-15
    R r0 = real_from<R>(1.0f - as_float(ref_idx)) / real_from<R>(1.0f + as_float(ref_idx));
    r0 = r0 * r0;
    return r0 + real_from<R>(1.0f - as_float(r0)) * real_from<R>(pow5(1.0f - as_float(cosine)));
}
float curand_uniform_of(rt_rand_state* s);      // curand_uniform on the XORWOW state (below)

template <class R> class material_t {
public:
    virtual ~material_t() {}
    virtual void describe(rt_sphere& out) const = 0;
    // material::scatter (material.h:49): attenuation and scattered ray of a hit; false = absorbed
    virtual bool scatter(const ray_t<R>& r_in, const hit_record_t<R>& rec, vec3_t<R>& attenuation, ray_t<R>& scattered, rt_rand_state* local_rand_state) const = 0;
};
template <class R> class lambertian_t : public material_t<R> {
public:
    explicit lambertian_t(const vec3_t<R>& a) : albedo(a) {}
    void describe(rt_sphere& o) const override { o.material = RT_MAT_LAMBERTIAN; for (int k = 0; k < 3; ++k) o.albedo[k] = as_float(albedo.e[k]); o.param = 0.f; }
    bool scatter(const ray_t<R>&, const hit_record_t<R>& rec, vec3_t<R>& attenuation, ray_t<R>& scattered, rt_rand_state* local_rand_state) const override {   // material.h:55-60
        const vec3_t<R> target = (rec.p + rec.normal) + random_in_unit_sphere<R>(local_rand_state);
        scattered = ray_t<R>(rec.p, target - rec.p);
        attenuation = albedo;
        return true;
    }
    vec3_t<R> albedo;
};
template <class R> class metal_t : public material_t<R> {
public:
    metal_t(const vec3_t<R>& a, R f) : albedo(a) { if (f < real_from<R>(1.0f)) fuzz = f; else fuzz = real_from<R>(1.0f); }   // material.h:66
    void describe(rt_sphere& o) const override { o.material = RT_MAT_METAL; for (int k = 0; k < 3; ++k) o.albedo[k] = as_float(albedo.e[k]); o.param = as_float(fuzz); }
    bool scatter(const ray_t<R>& r_in, const hit_record_t<R>& rec, vec3_t<R>& attenuation, ray_t<R>& scattered, rt_rand_state* local_rand_state) const override {   // material.h:68-73
        const vec3_t<R> reflected = reflect(unit_vector(r_in.direction()), rec.normal);
        scattered = ray_t<R>(rec.p, reflected + fuzz * random_in_unit_sphere<R>(local_rand_state));      // the draw happens even with fuzz 0
        attenuation = albedo;
        return dot(scattered.direction(), rec.normal) > real_from<R>(0.0f);
    }
    vec3_t<R> albedo; R fuzz;
};
template <class R> class dielectric_t : public material_t<R> {
public:
    explicit dielectric_t(R ri) : ref_idx(ri) {}
    void describe(rt_sphere& o) const override { o.material = RT_MAT_DIELECTRIC; o.albedo[0] = o.albedo[1] = o.albedo[2] = 0.f; o.param = as_float(ref_idx); }
    bool scatter(const ray_t<R>& r_in, const hit_record_t<R>& rec, vec3_t<R>& attenuation, ray_t<R>& scattered, rt_rand_state* local_rand_state) const override {   // material.h:81-113
        vec3_t<R> outward_normal;
        const vec3_t<R> reflected = reflect(r_in.direction(), rec.normal);       // the direction is NOT normalised here
        R ni_over_nt;
        attenuation = vec3_t<R>(real_from_double<R>(1.0), real_from_double<R>(1.0), real_from_double<R>(1.0));
        vec3_t<R> refracted(real_from_int<R>(0), real_from_int<R>(0), real_from_int<R>(0));
        R reflect_prob, cosine;
        if (dot(r_in.direction(), rec.normal) > real_from<R>(0.0f)) {
            outward_normal = vec3_t<R>(negate<R>(rec.normal.e[0]), negate<R>(rec.normal.e[1]), negate<R>(rec.normal.e[2]));
            ni_over_nt = ref_idx;
            cosine = dot(r_in.direction(), rec.normal) / r_in.direction().length();
            cosine = sqrt_real(real_from<R>(1.0f) - ref_idx * ref_idx * (real_from<R>(1.0f) - cosine * cosine));   // no clamp: NaN possible, kept
        } else {
            outward_normal = rec.normal;
            ni_over_nt = real_from<R>(1.0f) / ref_idx;
            cosine = real_from<R>(-as_float(dot(r_in.direction(), rec.normal)) / as_float(r_in.direction().length()));   // float negate, float divide
        }
        if (refract(r_in.direction(), outward_normal, ni_over_nt, refracted)) reflect_prob = schlick(cosine, ref_idx);
        else reflect_prob = real_from<R>(1.0f);
        if (curand_uniform_of(local_rand_state) < as_float(reflect_prob)) scattered = ray_t<R>(rec.p, reflected);     // exactly one draw
        else scattered = ray_t<R>(rec.p, refracted);
        return true;
    }
    R ref_idx;
};

// ---------------------------------------------------------------------------------------------- hitable / sphere / hitable_list
template <class R> class hitable_t {             // hitable.h:17-20
public:
    virtual ~hitable_t() {}
    virtual void describe(rt_sphere& out) const = 0;
    virtual bool hit(const ray_t<R>& r, R t_min, R t_max, hit_record_t<R>& rec) const = 0;     // hitable.h:19
};
template <class R> class sphere_t : public hitable_t<R> {      // sphere.h:7-15
public:
    sphere_t() : radius(real_from_int<R>(0)), mat_ptr(nullptr) { center = vec3_t<R>(radius, radius, radius); }
    sphere_t(vec3_t<R> cen, R r, std::shared_ptr<material_t<R>> m) : center(cen), radius(r), mat_ptr(std::move(m)) {}
    void describe(rt_sphere& o) const override {
        for (int k = 0; k < 3; ++k) o.center[k] = as_float(center.e[k]);
        o.radius = as_float(radius);
        if (mat_ptr) mat_ptr->describe(o);
        else { o.material = RT_MAT_NONE; o.albedo[0] = o.albedo[1] = o.albedo[2] = 0.f; o.param = 0.f; }
    }
    // sphere::hit (sphere.h:17-46).  A slot without a material (a "ghost" of the world list) is never hit.
    bool hit(const ray_t<R>& r, R t_min, R t_max, hit_record_t<R>& rec) const override {
        if (!mat_ptr) return false;
        const vec3_t<R> oc = r.origin() - center;
        const R a = dot(r.direction(), r.direction());
        const R b = dot(oc, r.direction());
        const R c = dot(oc, oc) - radius * radius;
        const R discriminant = b * b - a * c;
        if (discriminant > real_from_int<R>(0)) {
            R temp = real_from<R>((-as_float(b) - as_float(sqrt_real(discriminant))) / as_float(a));     // float arithmetic on converted operands, one rounding
            if (temp < t_max && temp > t_min) { fill(rec, r, temp); return true; }
            temp = real_from<R>((-as_float(b) + sqrtf(as_float(discriminant))) / as_float(a));            // the far root's sqrt is not rounded to real_t (sphere.h:36)
            if (temp < t_max && temp > t_min) { fill(rec, r, temp); return true; }
        }
        return false;
    }
    void fill(hit_record_t<R>& rec, const ray_t<R>& r, R t) const {
        rec.t = t;
        rec.p = r.point_at_parameter(rec.t);
        rec.normal = (rec.p - center) / radius;
        rec.mat_ptr = mat_ptr.get();
    }
    vec3_t<R> center; R radius; std::shared_ptr<material_t<R>> mat_ptr;
};
template <class R> class hitable_list_t : public hitable_t<R> {   // hitable_list.h:7-14
public:
    hitable_list_t() : list(nullptr), list_size(0) {}
    hitable_list_t(hitable_t<R>** l, int n) : list(l), list_size(n) {}
    void describe(rt_sphere&) const override {}
    void serialise(rt_sphere* out) const { for (int i = 0; i < list_size; ++i) list[i]->describe(out[i]); }
    // hitable_list::hit (hitable_list.h:16-31): every entry in list order, closest so far as t_max
    bool hit(const ray_t<R>& r, R t_min, R t_max, hit_record_t<R>& rec) const override {
        hit_record_t<R> temp_rec;
        bool hit_anything = false;
        R closest_so_far = t_max;
        for (int i = 0; i < list_size; i++) {
            if (list[i]->hit(r, t_min, closest_so_far, temp_rec)) { hit_anything = true; closest_so_far = temp_rec.t; rec = temp_rec; }
        }
        return hit_anything;
    }
    hitable_t<R>** list; int list_size;
};

// ---------------------------------------------------------------------------------------------- camera (camera.h:20-57)
template <class R> struct tan_of_half_angle;
// fp32: tan(arg).  Contract: tan evaluated in binary64 and rounded once (equals glibc tanf here: 0x3e8930a3 for vfov 30).
template <> struct tan_of_half_angle<float> { static float eval(float arg) { return (float)tan((double)arg); } };
// fp16 (camera.h:29): real_t(hsin(arg) / hcos(arg)) — binary16 sine, binary16 cosine, binary16 divide.
template <> struct tan_of_half_angle<half_t> {
    static half_t eval(half_t arg) { const half_t s((float)sin((double)arg.f())), c((float)cos((double)arg.f())); return s / c; }
};

template <class R> class camera_t {
public:
    camera_t() {}
    camera_t(vec3_t<R> lookfrom, vec3_t<R> lookat, vec3_t<R> vup, R vfov, R aspect, R aperture, R focus_dist) {
        lens_radius = aperture / real_from<R>(2.0f);
        const R theta = vfov * real_from_double<R>(3.14159265358979323846) / real_from<R>(180.0f);
        const R arg = theta / real_from<R>(2.0f);
        const R half_height = tan_of_half_angle<R>::eval(arg);
        const R half_width = aspect * half_height;
        origin = lookfrom;
        w = unit_vector(lookfrom - lookat);
        u = unit_vector(cross(vup, w));
        v = cross(w, u);
        lower_left_corner = origin - (half_width * focus_dist) * u - (half_height * focus_dist) * v - focus_dist * w;
        horizontal = (real_from<R>(2.0f) * half_width * focus_dist) * u;
        vertical = (real_from<R>(2.0f) * half_height * focus_dist) * v;
    }
    // camera::get_ray (camera.h:45-49) with random_in_unit_disk (camera.h:12-18): two draws per try
    ray_t<R> get_ray(R s, R t, rt_rand_state* local_rand_state) const {
        vec3_t<R> p;
        do {
            const float x = curand_uniform_of(local_rand_state);            // x before y
            const float y = curand_uniform_of(local_rand_state);
            p = real_from<R>(2.0f) * vec3_t<R>(real_from<R>(x), real_from<R>(y), real_from_int<R>(0)) - vec3_t<R>(real_from_int<R>(1), real_from_int<R>(1), real_from_int<R>(0));
        } while (dot(p, p) >= real_from<R>(1.0f));
        const vec3_t<R> rd = lens_radius * p;
        const vec3_t<R> offset = rd.x() * u + rd.y() * v;
        return ray_t<R>(origin + offset, lower_left_corner + s * horizontal + t * vertical - origin - offset);
    }
    void serialise(rt_camera& c) const {
        for (int k = 0; k < 3; ++k) {
            c.origin[k] = as_float(origin.e[k]); c.lower_left_corner[k] = as_float(lower_left_corner.e[k]);
            c.horizontal[k] = as_float(horizontal.e[k]); c.vertical[k] = as_float(vertical.e[k]);
            c.u[k] = as_float(u.e[k]); c.v[k] = as_float(v.e[k]); c.w[k] = as_float(w.e[k]);
        }
        c.lens_radius = as_float(lens_radius);
    }
    vec3_t<R> origin, lower_left_corner, horizontal, vertical, u, v, w; R lens_radius;
};

// ---------------------------------------------------------------------------------------------- world RNG (cuRAND XORWOW, main.cu:80)
struct xorwow {
    static void init(rt_rand_state& s, unsigned long long seed) {
        const uint32_t lo = (uint32_t)seed ^ 0xaad26b49u, hi = (uint32_t)(seed >> 32) ^ 0xf7dcefddu;
        const uint32_t t0 = 1099087573u * lo, t1 = 2591861531u * hi;
        s.d = 6615241u + t1 + t0;
        s.v[0] = 123456789u + t0; s.v[1] = 362436069u ^ t0; s.v[2] = 521288629u + t1; s.v[3] = 88675123u ^ t1; s.v[4] = 5783321u + t0;
        s.boxmuller_flag = 0; s.boxmuller_flag_double = 0; s.boxmuller_extra = 0.f; s.pad_ = 0; s.boxmuller_extra_double = 0.0;
    }
    static uint32_t next(rt_rand_state& s) {
        const uint32_t t = s.v[0] ^ (s.v[0] >> 2);
        s.v[0] = s.v[1]; s.v[1] = s.v[2]; s.v[2] = s.v[3]; s.v[3] = s.v[4];
        s.v[4] = (s.v[4] ^ (s.v[4] << 4)) ^ (t ^ (t << 1));
        s.d += 362437u;
        return s.d + s.v[4];
    }
    static float uniform(rt_rand_state& s) {                 // curand_uniform: (0,1]
        const float x = (float)next(s);
        const float scaled = x * 2.3283064e-10f;
        return scaled + (2.3283064e-10f / 2.0f);
    }
};

inline float curand_uniform_of(rt_rand_state* s) { return xorwow::uniform(*s); }
template <class R> vec3_t<R> random_in_unit_sphere(rt_rand_state* local_rand_state) {       // material.h:35-41: three draws per try, x y z
    vec3_t<R> p;
    do {
        const float x = curand_uniform_of(local_rand_state);
        const float y = curand_uniform_of(local_rand_state);
        const float z = curand_uniform_of(local_rand_state);
        p = real_from<R>(2.0f) * vec3_t<R>(real_from<R>(x), real_from<R>(y), real_from<R>(z)) - vec3_t<R>(real_from_int<R>(1), real_from_int<R>(1), real_from_int<R>(1));
    } while (p.squared_length() >= real_from<R>(1.0f));
    return p;
}

// ---------------------------------------------------------------------------------------------- create_world (main.cu:146-204)
template <class R> struct world_t {
    std::vector<sphere_t<R>> d_list;                 // sphere (*d_list)[NUM_SPHERES]
    std::vector<hitable_t<R>*> d_hitable;            // hitable** handed to hitable_list
    hitable_list_t<R> d_world;
    camera_t<R> d_camera;
    int created = 0;
};

// One pass over create_world's RNG chain (main.cu:150-182).  emit(i, center, radius, tag, albedo, param) receives every filled
// slot in order; returns 4 + k*k, the number of slots the reference fills (never more than num_spheres).  Draws sit in
// constructor argument lists in the reference: consumed left to right here, explicitly.
template <class R, class Emit> int generate_world(int num_spheres, float sphere_radius, rt_rand_state* rand_state, Emit emit) {
    typedef vec3_t<R> vec3;
    rt_rand_state local_rand_state = *rand_state;
    auto RND = [&]() { return xorwow::uniform(local_rand_state); };
    auto real = [](double d) { return real_from_double<R>(d); };
    auto ireal = [](int i) { return real_from_int<R>(i); };
    const vec3 none(ireal(0), ireal(0), ireal(0));
    int i = 0;
    auto put = [&](const vec3& c, R r, int tag, const vec3& alb, R param) { if (i < num_spheres) emit(i, c, r, tag, alb, param); ++i; };
    put(vec3(ireal(0), real(-1000.0), ireal(-1)), ireal(1000), RT_MAT_LAMBERTIAN, vec3(real(0.5), real(0.5), real(0.5)), ireal(0));
    put(vec3(ireal(0), ireal(1), ireal(0)), real(1.0), RT_MAT_DIELECTRIC, none, real(1.5));
    put(vec3(ireal(-4), ireal(1), ireal(0)), real(1.0), RT_MAT_LAMBERTIAN, vec3(real(0.4), real(0.2), real(0.1)), ireal(0));
    put(vec3(ireal(4), ireal(1), ireal(0)), real(1.0), RT_MAT_METAL, vec3(real(0.7), real(0.6), real(0.5)), real(0.0));
    if (i > num_spheres) i = num_spheres;
    const int spheres_per_dim = (int)sqrtf((float)num_spheres - 4);
    const double spacing = 20. / spheres_per_dim;
    const R radius = real_from<R>(sphere_radius);
    for (double a = -10; a < 10; a += spacing) {
        for (double b = -10; b < 10 && i < num_spheres; b += spacing) {
            const R choose_mat = real_from<R>(RND());
            const float dx = RND();                   // the x jitter is drawn before the z jitter
            const float dz = RND();
            const vec3 center(real(a + dx), radius, real(b + dz));
            if (choose_mat < real_from<R>(0.8f)) {
                float p[6]; for (int k = 0; k < 6; ++k) p[k] = RND();
                emit(i++, center, radius, RT_MAT_LAMBERTIAN, vec3(real_from<R>(p[0] * p[1]), real_from<R>(p[2] * p[3]), real_from<R>(p[4] * p[5])), ireal(0));
            } else if (choose_mat < real_from<R>(0.95f)) {
                float p[4]; for (int k = 0; k < 4; ++k) p[k] = RND();
                const vec3 alb(real_from<R>(0.5f * (1.0f + p[0])), real_from<R>(0.5f * (1.0f + p[1])), real_from<R>(0.5f * (1.0f + p[2])));
                emit(i++, center, radius, RT_MAT_METAL, alb, real_from<R>(0.5f * p[3]));
            } else {
                emit(i++, center, radius, RT_MAT_DIELECTRIC, none, real(1.5));
            }
        }
    }
    *rand_state = local_rand_state;
    return i;
}

// the camera of create_world (main.cu:192-202)
template <class R> camera_t<R> world_camera(int nx, int ny) {
    typedef vec3_t<R> vec3;
    auto real = [](double d) { return real_from_double<R>(d); };
    auto ireal = [](int i) { return real_from_int<R>(i); };
    const vec3 lookfrom(ireal(13), ireal(2), ireal(3)), lookat(ireal(0), ireal(0), ireal(0));
    const R dist_to_focus = real(10.0), aperture = real(0.1);
    return camera_t<R>(lookfrom, lookat, vec3(ireal(0), ireal(1), ireal(0)), real(30.0), ireal(nx) / ireal(ny), aperture, dist_to_focus);
}

// create_world as objects: spheres with their materials, the hitable_list over them, the camera
template <class R> void create_world(world_t<R>& W, int num_spheres, float sphere_radius, int nx, int ny, rt_rand_state* rand_state) {
    W.d_list.assign(num_spheres, sphere_t<R>());      // unfilled slots stay material-less ("ghosts")
    W.created = generate_world<R>(num_spheres, sphere_radius, rand_state, [&](int i, const vec3_t<R>& c, R r, int tag, const vec3_t<R>& alb, R param) {
        std::shared_ptr<material_t<R>> m;
        if (tag == RT_MAT_LAMBERTIAN) m = std::make_shared<lambertian_t<R>>(alb);
        else if (tag == RT_MAT_METAL) m = std::make_shared<metal_t<R>>(alb, param);
        else m = std::make_shared<dielectric_t<R>>(param);
        W.d_list[i] = sphere_t<R>(c, r, std::move(m));
    });
    W.d_hitable.resize(num_spheres);
    for (int k = 0; k < num_spheres; ++k) W.d_hitable[k] = &W.d_list[k];
    W.d_world = hitable_list_t<R>(W.d_hitable.data(), num_spheres);
    W.d_camera = world_camera<R>(nx, ny);
}

// ... and straight into the PODs of the C-ABI (what rt_create_world returns): the same values as serialising the objects,
// without 100 000 heap-allocated materials on the way (N = 100 000: 5.7 -> 1.3-2.5 ms on the GPU box's host)
template <class R> int create_world_pods(rt_sphere* list, int num_spheres, float sphere_radius, rt_camera* cam, int nx, int ny, rt_rand_state* rand_state) {
    for (int k = 0; k < num_spheres; ++k) {            // sphere_t() of an unfilled slot: zero centre and radius, no material
        rt_sphere& o = list[k];
        o.center[0] = o.center[1] = o.center[2] = 0.f; o.radius = 0.f; o.material = RT_MAT_NONE; o.albedo[0] = o.albedo[1] = o.albedo[2] = 0.f; o.param = 0.f;
    }
    const int created = generate_world<R>(num_spheres, sphere_radius, rand_state, [&](int i, const vec3_t<R>& c, R r, int tag, const vec3_t<R>& alb, R param) {
        rt_sphere& o = list[i];
        for (int k = 0; k < 3; ++k) o.center[k] = as_float(c.e[k]);
        o.radius = as_float(r);
        o.material = tag;
        const bool has_albedo = tag != RT_MAT_DIELECTRIC;
        for (int k = 0; k < 3; ++k) o.albedo[k] = has_albedo ? as_float(alb.e[k]) : 0.f;
        // metal: the constructor's clamp of fuzz to <= 1 (material.h:66); lambertian carries no parameter
        o.param = tag == RT_MAT_METAL ? as_float(param < real_from<R>(1.0f) ? param : real_from<R>(1.0f)) : (tag == RT_MAT_DIELECTRIC ? as_float(param) : 0.f);
    });
    world_camera<R>(nx, ny).serialise(*cam);
    return created;
}

// ---------------------------------------------------------------------------------------------- Octree (acceleration_structure.h)
// Reference layout with SPHERES_PER_LEAF as a run-time field.
struct Octree {
    std::vector<rt_octnode> nodes;          // [585], aabb held as float images of real_t
    std::vector<int32_t> leaf_count;        // OctLeaf::index_count per leaf; leaf 0 is never used
    std::vector<int32_t> leaf_indices;      // OctLeaf::sphere_indices, leaf-major, spl per leaf
    int nodeCount = 0, leafCount = 1, spl = 30;
    int dropped_full = 0, dropped_outside = 0;      // the two printf paths of insert()
};

template <class R> struct box_t { R lo[3], hi[3]; };

// intersects(sphere, AABB) — acceleration_structure.h:82-93: centre inside the box grown by the radius, x_low strict.
template <class R> bool intersects(const rt_sphere& obj, box_t<R> bx) {
    return sphere_touches_box<R>(real_from<R>(obj.center[0]), real_from<R>(obj.center[1]), real_from<R>(obj.center[2]), real_from<R>(obj.radius), bx.lo, bx.hi);
}

template <class R> box_t<R> box_of(const rt_octnode& n) {
    box_t<R> b; for (int k = 0; k < 3; ++k) { b.lo[k] = real_from<R>(n.aabb[k]); b.hi[k] = real_from<R>(n.aabb[3 + k]); } return b;
}

// insert() — acceleration_structure.h:104-186
template <class R> int insert(Octree& T, int node, const rt_sphere& obj, int sphereidx) {
    if (!intersects<R>(obj, box_of<R>(T.nodes[node]))) { T.dropped_outside++; return 0; }
    if (T.nodes[node].level == 3) {
        for (int i = 0; i < 8; ++i) {
            int leaf = T.nodes[node].children[i];
            if (leaf == 0) {
                leaf = T.leafCount++;
                T.leaf_count.resize(T.leafCount, 0);
                T.leaf_indices.resize((size_t)T.leafCount * T.spl, 0);
                T.nodes[node].children[i] = leaf;
            }
            if (T.leaf_count[leaf] < T.spl) { T.leaf_indices[(size_t)leaf * T.spl + T.leaf_count[leaf]++] = sphereidx; return 1; }
        }
        T.dropped_full++;
        return 0;
    }
    int inserted = 0;
    const box_t<R> pb = box_of<R>(T.nodes[node]);
    R mid[3];
    for (int k = 0; k < 3; ++k) { const float lo = as_float(pb.lo[k]), hi = as_float(pb.hi[k]); mid[k] = real_from<R>(lo + (hi - lo) / 2); }   // float midpoint (:141)
    for (int i = 0; i < 8; ++i) {                 // octant i: bit2 = x high, bit1 = y high, bit0 = z high (:149-165)
        box_t<R> cb;
        for (int k = 0; k < 3; ++k) {
            const bool high = (i >> (2 - k)) & 1;
            cb.lo[k] = high ? mid[k] : pb.lo[k];
            cb.hi[k] = high ? pb.hi[k] : mid[k];
        }
        if (!intersects<R>(obj, cb)) continue;
        if (T.nodes[node].children[i] == 0) {
            const int created = T.nodeCount++;
            T.nodes[node].children[i] = created;
            rt_octnode n; n.level = T.nodes[node].level + 1;
            for (int k = 0; k < 3; ++k) { n.aabb[k] = as_float(cb.lo[k]); n.aabb[3 + k] = as_float(cb.hi[k]); }
            for (int k = 0; k < 8; ++k) n.children[k] = 0;
            T.nodes[created] = n;
        }
        inserted += insert<R>(T, T.nodes[node].children[i], obj, sphereidx);
    }
    return inserted;
}

// buildOctree() — acceleration_structure.h:195-217.  Slots without a material are inserted as they stand
// (zero-filled spheres, the model of the reference's uninitialised memory) but are never hittable.
template <class R> Octree* buildOctree(const rt_sphere* d_list, const int num_hitables, int spheres_per_leaf) {
    Octree* octree = new Octree();
    octree->spl = spheres_per_leaf;
    rt_octnode zero; zero.level = 0; for (int k = 0; k < 6; ++k) zero.aabb[k] = 0.f; for (int k = 0; k < 8; ++k) zero.children[k] = 0;
    octree->nodes.assign(RT_OCTREE_MAX_NODES, zero);
    octree->leaf_count.assign(1, 0);
    octree->leaf_indices.assign((size_t)spheres_per_leaf, 0);
    rt_octnode& root = octree->nodes[0];
    const float root_box[6] = {-11, 0, -11, 11, 2, 11};
    for (int k = 0; k < 6; ++k) root.aabb[k] = root_box[k];
    octree->nodeCount++;
    for (int i = 1; i < num_hitables; i++) insert<R>(*octree, 0, d_list[i], i);    // ground sphere (idx 0) is not in the tree
    return octree;
}

} // namespace rt
