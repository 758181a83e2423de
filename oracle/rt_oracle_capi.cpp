// rt_oracle_capi.cpp — C entry points of the CPU ORACLE (test infrastructure only; see rt_oracle.hpp header).
// Loaded through ctypes by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.  Never by the product.
#include "rt_oracle.hpp"
#include <thread>
#include <atomic>
#include <string>
#include <memory>

using namespace orc;

struct orc_scene {
    int fp16 = 0;
    int num_spheres = 0, nx = 0, ny = 0, spl = 30;
    World<float> w32;
    World<h16> w16;
};

template <class F> static auto with_world(orc_scene* s, F&& f) { return s->fp16 ? f(s->w16) : f(s->w32); }

template <class Body> static void parallel_rows(int rows, int nthreads, Body body) {
    if (nthreads <= 1) { for (int j = 0; j < rows; ++j) body(j, 0); return; }
    std::atomic<int> next(0);
    std::vector<std::thread> th;
    for (int t = 0; t < nthreads; ++t) th.emplace_back([&, t]() { for (;;) { const int j = next.fetch_add(1); if (j >= rows) break; body(j, t); } });
    for (auto& x : th) x.join();
}

// A world given by the caller instead of create_world (tests of arbitrary scenes): geom N x (cx,cy,cz,r),
// mat N x (albedo rgb, param), kind N (MAT_*; -1 = ghost slot), cam = 22 floats in camera.h field order.
// Values are converted to real_t exactly as the product converts them (float images).
template <class R> static void fill_custom(World<R>& W, int n, const float* geom, const float* mat, const int32_t* kind, const float* cam, bool use_octree, int spl) {
    W.use_octree = use_octree;
    W.list.resize(n);
    int real = 0;
    for (int i = 0; i < n; ++i) {
        Sphere<R>& sp = W.list[i];
        for (int k = 0; k < 3; ++k) { sp.center.e[k] = from_f<R>(geom[i * 4 + k]); sp.albedo.e[k] = from_f<R>(mat[i * 4 + k]); }
        sp.radius = from_f<R>(geom[i * 4 + 3]); sp.param = from_f<R>(mat[i * 4 + 3]); sp.kind = kind[i];
        if (kind[i] != MAT_NONE) ++real;
    }
    W.n_real = real; W.world_draws = 0;
    xorwow_init(W.world_rng_after, 1984);
    int o = 0;
    auto get = [&](V3<R>& v) { for (int k = 0; k < 3; ++k) v.e[k] = from_f<R>(cam[o++]); };
    get(W.cam.origin); get(W.cam.lower_left_corner); get(W.cam.horizontal); get(W.cam.vertical); get(W.cam.u); get(W.cam.v); get(W.cam.w);
    W.cam.lens_radius = from_f<R>(cam[o++]);
    if (use_octree) W.tree = build_octree(W.list, spl);
}

extern "C" {

orc_scene* orc_scene_create(int num_spheres, float radius, int nx, int ny, int fp16, int use_octree, int spl) {
    orc_scene* s = new orc_scene();
    s->fp16 = fp16; s->num_spheres = num_spheres; s->nx = nx; s->ny = ny; s->spl = spl;
    if (fp16) s->w16 = create_world<h16>(num_spheres, radius, nx, ny, use_octree != 0, spl);
    else s->w32 = create_world<float>(num_spheres, radius, nx, ny, use_octree != 0, spl);
    return s;
}
orc_scene* orc_scene_create_custom(int n, const float* geom, const float* mat, const int32_t* kind, const float* cam, int nx, int ny, int fp16, int use_octree, int spl) {
    orc_scene* s = new orc_scene();
    s->fp16 = fp16; s->num_spheres = n; s->nx = nx; s->ny = ny; s->spl = spl;
    if (fp16) fill_custom(s->w16, n, geom, mat, kind, cam, use_octree != 0, spl);
    else fill_custom(s->w32, n, geom, mat, kind, cam, use_octree != 0, spl);
    return s;
}

void orc_scene_destroy(orc_scene* s) { delete s; }

// out: [0] slots [1] real [2] world draws [3] nodeCount [4] leafCount [5] leaf entries [6] dropped (buckets full) [7] dropped (outside root)
void orc_scene_info(orc_scene* s, int64_t* out) {
    with_world(s, [&](auto& W) {
        out[0] = (int64_t)W.list.size(); out[1] = W.n_real; out[2] = (int64_t)W.world_draws;
        out[3] = W.tree.nodeCount; out[4] = W.tree.leafCount;
        int64_t e = 0; for (auto& L : W.tree.leaves) e += (int64_t)L.idx.size();
        out[5] = e; out[6] = W.tree.dropped_full; out[7] = W.tree.dropped_outside;
        return 0;
    });
}
// geom: N x 4 (cx,cy,cz,r); mat: N x 4 (albedo rgb, param); kind: N
void orc_scene_spheres(orc_scene* s, float* geom, float* mat, int32_t* kind) {
    with_world(s, [&](auto& W) {
        for (size_t i = 0; i < W.list.size(); ++i) {
            auto& sp = W.list[i];
            for (int k = 0; k < 3; ++k) { geom[i * 4 + k] = to_f(sp.center.e[k]); mat[i * 4 + k] = to_f(sp.albedo.e[k]); }
            geom[i * 4 + 3] = to_f(sp.radius); mat[i * 4 + 3] = to_f(sp.param); kind[i] = sp.kind;
        }
        return 0;
    });
}
// 22 floats: origin, lower_left_corner, horizontal, vertical, u, v, w, lens_radius  (camera.h:51-56 order)
void orc_scene_camera(orc_scene* s, float* out) {
    with_world(s, [&](auto& W) {
        auto& c = W.cam; int o = 0;
        auto put = [&](auto& v) { for (int k = 0; k < 3; ++k) out[o++] = to_f(v.e[k]); };
        put(c.origin); put(c.lower_left_corner); put(c.horizontal); put(c.vertical); put(c.u); put(c.v); put(c.w);
        out[o++] = to_f(c.lens_radius);
        return 0;
    });
}
void orc_scene_world_rng(orc_scene* s, void* out48) { with_world(s, [&](auto& W) { std::memcpy(out48, &W.world_rng_after, 48); return 0; }); }

// reference-layout octree dump. level[585], box[585*6] (x_low,y_low,z_low,x_high,y_high,z_high), children[585*8]
void orc_scene_octree_nodes(orc_scene* s, int32_t* level, float* box, int32_t* children) {
    with_world(s, [&](auto& W) {
        for (int i = 0; i < 585; ++i) {
            auto& n = W.tree.nodes[i]; level[i] = n.level;
            for (int k = 0; k < 3; ++k) { box[i * 6 + k] = to_f(n.box.lo[k]); box[i * 6 + 3 + k] = to_f(n.box.hi[k]); }
            for (int k = 0; k < 8; ++k) children[i * 8 + k] = n.children[k];
        }
        return 0;
    });
}
// counts[leafCount], indices[leafCount*spl] (unused slots = 0, as the zero-initialised reference leaves)
void orc_scene_octree_leaves(orc_scene* s, int32_t* counts, int32_t* indices) {
    with_world(s, [&](auto& W) {
        const int spl = W.tree.spl;
        for (int l = 0; l < W.tree.leafCount; ++l) {
            auto& L = W.tree.leaves[l]; counts[l] = (int32_t)L.idx.size();
            for (int k = 0; k < spl; ++k) indices[(size_t)l * spl + k] = k < (int)L.idx.size() ? L.idx[k] : 0;
        }
        return 0;
    });
}

// closest hit for n rays (6 floats each: origin, direction). mode 0 = scene default, 1 = hitable_list, 2 = hitTree.
void orc_trace(orc_scene* s, int64_t n, const float* rays, int mode, int32_t* hit, int32_t* sph, float* t, float* p, float* nrm) {
    with_world(s, [&](auto& W) {
        using R = decltype(W.cam.lens_radius);
        const bool tree = mode == 0 ? W.use_octree : (mode == 2);
        for (int64_t i = 0; i < n; ++i) {
            Ray<R> r;
            for (int k = 0; k < 3; ++k) { r.A.e[k] = from_f<R>(rays[i * 6 + k]); r.B.e[k] = from_f<R>(rays[i * 6 + 3 + k]); }
            Hit<R> rec; rec.t = from_i<R>(0); rec.sphere = -1; rec.p = r.A; rec.normal = r.A;
            const bool h = tree ? hit_tree(W, r, rec, nullptr) : hit_list(W, r, from_f<R>(0.001f), from_f<R>(FLT_MAX), rec, nullptr);
            hit[i] = h ? 1 : 0; sph[i] = h ? rec.sphere : -1; t[i] = h ? to_f(rec.t) : 0.f;
            for (int k = 0; k < 3; ++k) { p[i * 3 + k] = h ? to_f(rec.p.e[k]) : 0.f; nrm[i * 3 + k] = h ? to_f(rec.normal.e[k]) : 0.f; }
        }
        return 0;
    });
}

// render_init (main.cu:84-94): seed 1984 + absolute pixel_index; states is a compact array for rows [row0,row0+rows)
void orc_render_init(int max_x, int max_y, int row0, int rows, void* states) {
    (void)max_y;
    Xorwow* st = (Xorwow*)states;
    for (int j = 0; j < rows; ++j)
        for (int i = 0; i < max_x; ++i) xorwow_init(st[(size_t)j * max_x + i], 1984ull + (uint64_t)((row0 + j) * max_x + i));
}

// render (main.cu:96-117) for rows [row0,row0+rows); fb and states are compact over those rows.
// fb: rows*max_x*3 floats (fp16 scenes: exact float images of the half values).
// counters (optional, 6 x u64): rays, sphere_tests, slab_tests, bucket_visits, draws(unused), samples
void orc_render(orc_scene* s, float* fb, int max_x, int max_y, int ns, void* states, int row0, int rows, int nthreads, uint64_t* counters) {
    with_world(s, [&](auto& W) {
        using R = decltype(W.cam.lens_radius);
        Xorwow* st = (Xorwow*)states;
        std::vector<Counters> cs(nthreads > 1 ? nthreads : 1);
        parallel_rows(rows, nthreads, [&](int jr, int tid) {
            const int j = row0 + jr;
            for (int i = 0; i < max_x; ++i) {
                const size_t li = (size_t)jr * max_x + i;
                const V3<R> c = render_pixel<R>(W, i, j, max_x, max_y, ns, st[li], counters ? &cs[tid] : nullptr);
                for (int k = 0; k < 3; ++k) fb[li * 3 + k] = to_f(c.e[k]);
            }
        });
        if (counters) {
            for (int k = 0; k < 6; ++k) counters[k] = 0;
            for (auto& c : cs) { counters[0] += c.rays; counters[1] += c.sphere_tests; counters[2] += c.slab_tests; counters[3] += c.bucket_visits; counters[5] += c.samples; }
        }
        return 0;
    });
}

// render_progressive (main.cu:119-142): one sample; fb = col if current_sample == 1 else fb += col (real_t adds).
void orc_render_progressive(orc_scene* s, float* fb, int max_x, int max_y, int current_sample, void* states, int nthreads) {
    with_world(s, [&](auto& W) {
        using R = decltype(W.cam.lens_radius);
        Xorwow* st = (Xorwow*)states;
        parallel_rows(max_y, nthreads, [&](int j, int) {
            for (int i = 0; i < max_x; ++i) {
                const size_t li = (size_t)j * max_x + i;
                const V3<R> c = sample_pixel<R>(W, i, j, max_x, max_y, st[li], nullptr);
                for (int k = 0; k < 3; ++k) {
                    if (current_sample == 1) fb[li * 3 + k] = to_f(c.e[k]);
                    else fb[li * 3 + k] = to_f(from_f<R>(fb[li * 3 + k]) + c.e[k]);
                }
            }
        });
        return 0;
    });
}

// output_to_stream (main.cu:321-333): ASCII P3, top row first, int(255.99*c) with a double multiply.
int64_t orc_ppm(const float* fb, int nx, int ny, char* out, int64_t cap) {
    std::string s = "P3\n" + std::to_string(nx) + " " + std::to_string(ny) + "\n255\n";
    s.reserve((size_t)nx * ny * 12 + 32);
    for (int j = ny - 1; j >= 0; --j)
        for (int i = 0; i < nx; ++i) {
            const size_t pi = (size_t)j * nx + i;
            const int ir = (int)(255.99 * (double)fb[pi * 3 + 0]);
            const int ig = (int)(255.99 * (double)fb[pi * 3 + 1]);
            const int ib = (int)(255.99 * (double)fb[pi * 3 + 2]);
            s += std::to_string(ir); s += ' '; s += std::to_string(ig); s += ' '; s += std::to_string(ib); s += '\n';
        }
    if ((int64_t)s.size() <= cap && out) std::memcpy(out, s.data(), s.size());
    return (int64_t)s.size();
}

void orc_xorwow_init(void* st, uint64_t seed) { xorwow_init(*(Xorwow*)st, seed); }
uint32_t orc_xorwow_next(void* st) { return xorwow_next(*(Xorwow*)st); }
float orc_uniform(void* st) { return uniform(*(Xorwow*)st); }
uint16_t orc_f32_to_f16(float f) { return f32_to_f16(f); }
float orc_f16_to_f32(uint16_t h) { return f16_to_f32(h); }
float orc_pow5(float x) { return pow5(x); }
int orc_hw_threads() { return (int)std::thread::hardware_concurrency(); }

} // extern "C"
