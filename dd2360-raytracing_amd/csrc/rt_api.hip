// rt_api.hip — the C-ABI of include/rt_amd.h: scene upload, octree flattening, kernel launches, PPM output.
// Host code only (compiled by hipcc together with rt_kernels.hip into librt_amd.so).  No CPU render path exists
// here: every compute entry point launches the gfx950 kernels or fails with the HIP error.
#include <hip/hip_runtime.h>
#include <atomic>
#include <vector>
#include <string>
#include <cstdio>
#include <cstring>
#include <new>
#include <limits>
#include <algorithm>
#include <mutex>
#include <cstdlib>
#include "rt_handles.h"
#include "rt_octgeom.h"
#include "../host/rt_image.hpp"

namespace rt {
hipError_t launch_render_init(rt_rand_state* rs, int max_x, int max_y, int part, int nparts, long long begin, long long end, hipStream_t st);
hipError_t launch_zero_counters(unsigned int* p, int n, hipStream_t st);
hipError_t launch_pilot(const RenderArgs& A, bool tree, int* cost, unsigned char* pilot, int* work, hipStream_t st);
hipError_t launch_pilot_h(const RenderArgs& A, bool tree, int* cost, hipStream_t st);
hipError_t launch_assemble_split(void* full, const void* parts, int max_x, int max_y, int nparts, const long long* starts, long long part_stride_px, bool half, hipStream_t st);
hipError_t launch_render(const RenderArgs& A, bool tree, int mode, hipStream_t st);
namespace fmac { hipError_t launch_render(const RenderArgs& A, bool tree, int mode, hipStream_t st); }      // rt_kernels_contract.hip
hipError_t launch_tile_order(const RenderArgs& A, bool tree, int* cost, unsigned int* order, unsigned char* flags, unsigned int* long_list, hipStream_t st);
hipError_t launch_render_h(const RenderArgs& A, bool tree, int mode, hipStream_t st);
hipError_t launch_tile_order_h(const RenderArgs& A, bool tree, int* cost, unsigned int* order, unsigned char* flags, unsigned int* long_list, hipStream_t st);
hipError_t launch_trace_h(const DevScene& S, const DevTree& T, bool tree, const float* rays, long long n, rt_hit_record* out, hipStream_t st);
hipError_t launch_assemble_h(void* full, const void* parts, int max_x, int max_y, int nparts, hipStream_t st);
hipError_t launch_trace(const DevScene& S, const DevTree& T, bool tree, const float* rays, long long n, rt_hit_record* out, hipStream_t st);
hipError_t launch_assemble(float* full, const float* parts, int max_x, int max_y, int nparts, hipStream_t st);
const char* render_kernel_name(bool tree, int mode, const DevAccel& acc);
namespace gpubuild { int build(rt_octree* O, const float4* d_geom, const int32_t* d_kind, int n, int spl, hipStream_t st); }
const char* render_kernel_name_h(bool tree, int mode);
#ifdef RT_H16_STATS
hipError_t read_h16_stats(unsigned long long* out, int reset);
#endif
#ifdef RT_STATS
hipError_t read_stats(unsigned long long* out, int reset);
hipError_t read_wave_dbg(unsigned long long* out);
hipError_t read_pilot_dbg(int* out, int n);
#endif
}

using namespace rt;

#define RT_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return (int)e_; } while (0)

// device copy of a host vector into *d.  A buffer that exists already (an earlier upload attempt got that far) is kept;
// on failure nothing allocated here is left behind.
template <class V> static int upload(const V& v, void** d) {
    if (*d) return 0;
    const size_t bytes = (v.empty() ? 1 : v.size()) * sizeof(typename V::value_type);
    void* p = nullptr;
    RT_TRY(hipMalloc(&p, bytes));
    if (!v.empty()) {
        const hipError_t e = hipMemcpy(p, v.data(), v.size() * sizeof(typename V::value_type), hipMemcpyHostToDevice);
        if (e != hipSuccess) { (void)hipFree(p); return (int)e; }
    }
    *d = p;
    return 0;
}
static int free_all(void** bufs, int n) {
    int rc = 0;
    for (int k = 0; k < n; ++k) if (bufs[k]) { const hipError_t e = hipFree(bufs[k]); if (e != hipSuccess && !rc) rc = (int)e; bufs[k] = nullptr; }
    return rc;
}

// The list as a tree of ONE node without bounds whose entries are the hittable spheres 1..n-1 in list order: hitTree on
// it (slot 0 first, then the entries in order, strict "<") is hitable_list::hit, so the fp32 octree kernels — and their
// candidate grid — serve the list path unchanged.  Null when the grid would not pay or cannot be used.
static rt_octree* build_list_tree(const rt_world* W) {
    rt_octree* O = new rt_octree();
    O->z = new rt_octree::Lazy();
    O->precision = RT_PRECISION_FP32;
    DevNode d; memset(&d, 0, sizeof(d));
    const float inf = std::numeric_limits<float>::infinity();
    d.lo[0] = d.lo[1] = d.lo[2] = -inf; d.hix = d.hiy = d.hiz = inf;
    d.skip = 1; d.first = 0; d.ref_index = 0;
    for (int i = 1; i < W->n; ++i) {
        if (W->h_kind[i] == RT_MAT_NONE) continue;
        const float4 g = W->h_geom[i];
        O->h_ent_hot.push_back(make_float4(g.x, g.y, g.z, g.w * g.w));          // radius*radius in float (sphere.h:21)
        O->h_ent_id.push_back(i);
    }
    d.count = (int32_t)O->h_ent_id.size();
    O->h_nodes.push_back(d);
    O->n_nodes = 1; O->n_entries = d.count;
    if (d.count > 0) build_accel(O->accel, O->h_nodes, O->h_ent_id, O->h_ent_hot, W->n, true);
    // every ray tests the spheres the grid cannot hold: with many of them the scan is the better list path
    if (d.count < 64 || !O->accel.p.enabled || O->accel.p.n_large > 64) { delete O->z; delete O; return nullptr; }
    return O;
}
// built on first use (a render or trace call without an octree, rt_world_list_accel_info): a world that is only ever
// rendered through its octree never pays for it
static rt_octree* ensure_list_tree(const rt_world* W) {
    rt_world::Lazy& Z = *W->z;
    if (!Z.list_tree_tried && W->precision == RT_PRECISION_FP32) {
        Z.list_tree_tried = true;
        try { Z.list_tree = build_list_tree(W); } catch (const std::bad_alloc&) { Z.list_tree = nullptr; }
    }
    return Z.list_tree;
}

static bool valid_partition(rt_partition p) {
    if (!(p.nparts >= 1 && p.part >= 0 && p.part < p.nparts)) return false;
    if (p.tile_begin == 0 && p.tile_end == 0) return true;                          // runs of RT_PART_RUN tiles
    return p.tile_begin >= 0 && p.tile_end > p.tile_begin;                          // a range of tiles (checked against the frame where the frame is known)
}
static bool range_in_frame(rt_partition p, int64_t tiles) { return p.tile_end <= p.tile_begin || p.tile_end <= tiles; }
static int64_t local_tiles_of(int64_t tiles, rt_partition p) { return part_local_tiles(tiles, p.part, p.nparts, p.tile_begin, p.tile_end); }
static bool hittable(const rt_sphere& s) { return s.material != RT_MAT_NONE; }
// radius*radius in real_t (sphere.h:21), as a float image
static float radius_squared(const rt_sphere& s, int precision) {
    if (precision == RT_PRECISION_FP16) { const half_t r(s.radius); return (r * r).f(); }
    return s.radius * s.radius;
}

template <class R> static int create_world_impl(rt_sphere* list, int num_spheres, float sphere_radius, rt_camera* cam, int nx, int ny, rt_rand_state* st, int* num_created) {
    const int created = create_world_pods<R>(list, num_spheres, sphere_radius, cam, nx, ny, st);
    if (num_created) *num_created = created;
    return 0;
}

template <class R> static void camera_impl(rt_camera* cam, const float* lf, const float* la, const float* up, float vfov, float aspect, float aperture, float focus) {
    auto v = [](const float* p) { return vec3_t<R>(real_from<R>(p[0]), real_from<R>(p[1]), real_from<R>(p[2])); };
    camera_t<R> c(v(lf), v(la), v(up), real_from<R>(vfov), real_from<R>(aspect), real_from<R>(aperture), real_from<R>(focus));
    c.serialise(*cam);
}

uint64_t rt_next_serial() { static std::atomic<uint64_t> n{0}; return ++n; }

extern "C" {

int rt_abi_version(void) { return RT_ABI_VERSION; }

int rt_device_check(int* device_count) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (device_count) *device_count = (e == hipSuccess) ? n : 0;
    if (e != hipSuccess) return (int)e;
    if (n <= 0) return (int)hipErrorNoDevice;
    return 0;
}

const char* rt_error_string(int code) {
    switch (code) {
        case 0: return "ok";
        case RT_EINVAL: return "invalid argument";
        case RT_ENOMEM: return "out of host memory";
        case RT_EIO: return "i/o error";
        case RT_ENOTSUP: return "not supported";
        case RT_ECOMM: return "multi-GPU exchange failed";
        default: return code > 0 ? hipGetErrorString((hipError_t)code) : "unknown error";
    }
}

// ------------------------------------------------------------------------------------------------ scene definition
int rt_rand_init(rt_rand_state* rand_state) {
    if (!rand_state) return RT_EINVAL;
    xorwow::init(*rand_state, 1984ull);
    return 0;
}

int rt_create_world(rt_sphere* list, int num_spheres, float sphere_radius, rt_camera* cam, int nx, int ny, rt_rand_state* rand_state, int precision, int* num_created) {
    if (!list || !cam || !rand_state || num_spheres < 5 || nx <= 0 || ny <= 0) return RT_EINVAL;   // NUM_SPHERES "just > 4" (main.cu:22)
    try {
        if (precision == RT_PRECISION_FP16) return create_world_impl<half_t>(list, num_spheres, sphere_radius, cam, nx, ny, rand_state, num_created);
        if (precision == RT_PRECISION_FP32) return create_world_impl<float>(list, num_spheres, sphere_radius, cam, nx, ny, rand_state, num_created);
    } catch (const std::bad_alloc&) { return RT_ENOMEM; }
    return RT_EINVAL;
}

int rt_camera_init(rt_camera* cam, const float lookfrom[3], const float lookat[3], const float vup[3], float vfov, float aspect, float aperture, float focus_dist, int precision) {
    if (!cam || !lookfrom || !lookat || !vup) return RT_EINVAL;
    if (precision == RT_PRECISION_FP16) camera_impl<half_t>(cam, lookfrom, lookat, vup, vfov, aspect, aperture, focus_dist);
    else if (precision == RT_PRECISION_FP32) camera_impl<float>(cam, lookfrom, lookat, vup, vfov, aspect, aperture, focus_dist);
    else return RT_EINVAL;
    return 0;
}

int rt_world_create(const rt_sphere* list, int num_spheres, const rt_camera* cam, int precision, rt_world** out) {
    if (!list || !cam || !out || num_spheres <= 0) return RT_EINVAL;
    if (precision != RT_PRECISION_FP32 && precision != RT_PRECISION_FP16) return RT_EINVAL;
    *out = nullptr;
    rt_world* W = new (std::nothrow) rt_world();
    if (!W) return RT_ENOMEM;
    W->z = new (std::nothrow) rt_world::Lazy();
    if (!W->z) { delete W; return RT_ENOMEM; }
    W->precision = precision; W->n = num_spheres;
    std::vector<float4>& hot = W->h_hot; std::vector<float4>& geom = W->h_geom; std::vector<float4>& mat = W->h_mat;
    std::vector<int32_t>& ids = W->h_ids; std::vector<int32_t>& kind = W->h_kind;
    geom.resize(num_spheres); mat.resize(num_spheres); kind.resize(num_spheres);
    for (int i = 0; i < num_spheres; ++i) {
        const rt_sphere& s = list[i];
        geom[i] = make_float4(s.center[0], s.center[1], s.center[2], s.radius);
        mat[i] = make_float4(s.albedo[0], s.albedo[1], s.albedo[2], s.param);
        kind[i] = s.material;
        if (hittable(s)) { hot.push_back(make_float4(s.center[0], s.center[1], s.center[2], radius_squared(s, precision))); ids.push_back(i); }
    }
    W->z->dev.n = num_spheres; W->z->dev.n_list = (int)hot.size();
    W->z->dev.ground_valid = hittable(list[0]) ? 1 : 0;
    W->z->dev.cam = *cam;
    *out = W;
    return 0;
}

int rt_world_set_list_traversal(rt_world* W, int mode) {
    if (!W || (mode != RT_TRAVERSAL_REFERENCE && mode != RT_TRAVERSAL_FAST)) return RT_EINVAL;
    W->list_traversal = mode;
    return 0;
}

int rt_world_set_arith(rt_world* W, int mode) {
    if (!W || (mode != RT_ARITH_IEEE && mode != RT_ARITH_CONTRACT)) return RT_EINVAL;
    if (mode == RT_ARITH_CONTRACT && W->precision == RT_PRECISION_FP16) return RT_ENOTSUP;
    W->arith = mode;
    return 0;
}
int rt_world_list_accel_info(const rt_world* W, int* enabled, int* grid_dim, float* cell_size, int* grid_entries, int* large_spheres) {
    if (!W) return RT_EINVAL;
    rt_octree* LT = ensure_list_tree(W);
    if (enabled) *enabled = LT != nullptr;
    if (!LT) { if (grid_dim) *grid_dim = 0; if (cell_size) *cell_size = 0.f; if (grid_entries) *grid_entries = 0; if (large_spheres) *large_spheres = 0; return 0; }
    return rt_octree_accel_info(LT, grid_dim, cell_size, grid_entries, large_spheres);
}

// ---- render contexts ---------------------------------------------------------------------------------------------
static int ctx_prepare(rt_render_ctx& C) {          // device counters and events; not inside a stream capture
    if (!C.d_queue) {
        void* q = nullptr;
        RT_TRY(hipMalloc(&q, kQueueSlots * kQueueStride * sizeof(unsigned int)));
        const hipError_t e = hipMemset(q, 0, kQueueSlots * kQueueStride * sizeof(unsigned int));
        if (e != hipSuccess) { (void)hipFree(q); return (int)e; }
        C.d_queue = (unsigned int*)q;
    }
    if (!C.ev_ready) {
        for (int k = 0; k < 64; ++k) {
            if (!C.ev0[k]) RT_TRY(hipEventCreate(&C.ev0[k]));
            if (!C.ev1[k]) RT_TRY(hipEventCreate(&C.ev1[k]));
        }
        if (!C.done) RT_TRY(hipEventCreateWithFlags(&C.done, hipEventDisableTiming));
        C.ev_ready = true;
    }
    return 0;
}
// the scheduling workspace for frames of up to `tiles` local tiles.  Growing it frees the old one: hipFree waits for the
// device, so launches still queued on it finish first.
static int ctx_reserve(rt_render_ctx& C, int64_t tiles) {
    if (C.sched_tiles >= tiles) return 0;
    void* old[5] = {C.d_cost, C.d_order, C.d_flags, C.d_long, C.d_work};
    C.d_cost = nullptr; C.d_order = nullptr; C.d_flags = nullptr; C.d_long = nullptr; C.d_work = nullptr; C.sched_tiles = 0;
    int rc = free_all(old, 5);
    if (rc) return rc;
    void* nw[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    // (flags: 64 + 16 + 16 bytes per tile; long chains: 64 entries per tile, then the tail list: up to 64 per tile, then the tail sort's 512 words)
    const size_t bytes[5] = {sizeof(int) * (size_t)tiles, sizeof(unsigned int) * (size_t)tiles, (size_t)tiles * 96, sizeof(unsigned int) * ((size_t)tiles * 128 + 512), sizeof(int) * (size_t)tiles * 2};
    for (int k = 0; k < 5; ++k) {
        const hipError_t e = hipMalloc(&nw[k], bytes[k]);
        if (e != hipSuccess) { (void)free_all(nw, 5); return (int)e; }
    }
    C.d_cost = (int*)nw[0]; C.d_order = (unsigned int*)nw[1]; C.d_flags = (unsigned char*)nw[2]; C.d_long = (unsigned int*)nw[3]; C.d_work = (int*)nw[4];
    C.sched_tiles = tiles;
    return 0;
}
// the same for the tile order a progressive sequence keeps (its own buffers: a render on the same context does not disturb it)
static int ctx_reserve_progressive(rt_render_ctx& C, int64_t tiles) {
    if (C.p_tiles >= tiles) return 0;
    if (C.p_pinned) return RT_EINVAL;        // a hipGraph replays passes that read these buffers: a larger frame needs a context of its own
    void* old[2] = {C.p_cost, C.p_order};
    C.p_cost = nullptr; C.p_order = nullptr; C.p_tiles = 0; C.p_valid = false;
    int rc = free_all(old, 2);
    if (rc) return rc;
    void* nw[2] = {nullptr, nullptr};
    const size_t bytes[2] = {sizeof(int) * (size_t)tiles, sizeof(unsigned int) * (size_t)tiles};
    for (int k = 0; k < 2; ++k) {
        const hipError_t e = hipMalloc(&nw[k], bytes[k]);
        if (e != hipSuccess) { (void)free_all(nw, 2); return (int)e; }
    }
    C.p_cost = (int*)nw[0]; C.p_order = (unsigned int*)nw[1];
    C.p_tiles = tiles;
    return 0;
}
static int ctx_release(rt_render_ctx& C) {
    void* bufs[8] = {C.d_queue, C.d_cost, C.d_order, C.d_flags, C.d_long, C.p_cost, C.p_order, C.d_work};
    const int rc = free_all(bufs, 8);
    C.d_queue = nullptr; C.d_cost = nullptr; C.d_order = nullptr; C.d_flags = nullptr; C.d_long = nullptr; C.d_work = nullptr; C.sched_tiles = 0;
    C.p_cost = nullptr; C.p_order = nullptr; C.p_tiles = 0; C.p_valid = false;
    for (int k = 0; k < 64; ++k) { if (C.ev0[k]) (void)hipEventDestroy(C.ev0[k]); if (C.ev1[k]) (void)hipEventDestroy(C.ev1[k]); C.ev0[k] = nullptr; C.ev1[k] = nullptr; }
    if (C.done) (void)hipEventDestroy(C.done);
    C.done = nullptr; C.ev_ready = false; C.has_done = false; C.ev_count = 0;
    return rc;
}

int rt_render_ctx_create(rt_render_ctx** out) {
    if (!out) return RT_EINVAL;
    *out = nullptr;
    rt_render_ctx* C = new (std::nothrow) rt_render_ctx();
    if (!C) return RT_ENOMEM;
    const int rc = ctx_prepare(*C);
    if (rc) { (void)ctx_release(*C); delete C; return rc; }
    *out = C;
    return 0;
}
int rt_render_ctx_reserve(rt_render_ctx* C, int max_x, int max_y, rt_partition part) {
    if (!C || max_x <= 0 || max_y <= 0 || !valid_partition(part)) return RT_EINVAL;
    const int64_t tiles = (int64_t)((max_x + 7) / 8) * ((max_y + 7) / 8);
    const int rc = ctx_prepare(*C);
    if (!range_in_frame(part, tiles)) return RT_EINVAL;
    return rc ? rc : ctx_reserve(*C, local_tiles_of(tiles, part));
}
int rt_render_ctx_destroy(rt_render_ctx* C) {
    if (!C) return 0;
    const int rc = ctx_release(*C);
    delete C;
    return rc;
}

static int world_upload(const rt_world* W) {
    rt_world::Lazy& Z = *W->z;
    if (Z.uploaded) return 0;
    int rc;
    std::vector<float4> shade(2 * (size_t)W->n);
    std::vector<uint8_t> kind8((size_t)W->n);
    for (int i = 0; i < W->n; ++i) { shade[2 * (size_t)i] = W->h_geom[i]; shade[2 * (size_t)i + 1] = W->h_mat[i]; kind8[i] = (uint8_t)W->h_kind[i]; }
    if ((rc = upload(W->h_hot, &Z.d_list_hot)) || (rc = upload(W->h_ids, &Z.d_list_id)) || (rc = upload(W->h_geom, &Z.d_geom)) ||
        (rc = upload(W->h_mat, &Z.d_mat)) || (rc = upload(W->h_kind, &Z.d_kind)) || (rc = upload(shade, &Z.d_shade)) || (rc = upload(kind8, &Z.d_kind8))) {
        void* bufs[7] = {Z.d_list_hot, Z.d_list_id, Z.d_geom, Z.d_mat, Z.d_kind, Z.d_shade, Z.d_kind8};       // nothing half-made stays behind
        (void)free_all(bufs, 7);
        Z.d_list_hot = Z.d_list_id = Z.d_geom = Z.d_mat = Z.d_kind = Z.d_shade = Z.d_kind8 = nullptr;
        return rc;
    }
    Z.dev.list_hot = (const float4*)Z.d_list_hot; Z.dev.list_id = (const int32_t*)Z.d_list_id;
    Z.dev.geom = (const float4*)Z.d_geom; Z.dev.mat = (const float4*)Z.d_mat; Z.dev.kind = (const int32_t*)Z.d_kind;
    Z.dev.shade = (const float4*)Z.d_shade; Z.dev.kind8 = (const uint8_t*)Z.d_kind8;
    if ((rc = ctx_prepare(Z.ctx))) return rc;
    if (Z.list_tree && (rc = rt_octree_upload(Z.list_tree))) return rc;      // (if it has been built already)
    Z.uploaded = true;
    return 0;
}
int rt_world_upload(rt_world* W) { return W ? world_upload(W) : RT_EINVAL; }

int rt_free_world(rt_world* W) {
    if (!W) return 0;
    int rc = 0;
    if (W->z) {
        rt_world::Lazy& Z = *W->z;
        if (Z.list_tree) { rc = rt_free_octree(Z.list_tree); Z.list_tree = nullptr; }
        void* bufs[7] = {Z.d_list_hot, Z.d_list_id, Z.d_geom, Z.d_mat, Z.d_kind, Z.d_shade, Z.d_kind8};
        const int r2 = free_all(bufs, 7); if (!rc) rc = r2;
        const int r3 = ctx_release(Z.ctx); if (!rc) rc = r3;
        delete W->z;
    }
    delete W;
    return rc;
}

// pre-order flattening of the reference-layout tree (children in index order = traverseTree's visit order)
static void flatten(const Octree& T, const rt_sphere* list, int precision, int node, std::vector<DevNode>& out, std::vector<float4>& ent_hot, std::vector<int32_t>& ent_id) {
    const rt_octnode& n = T.nodes[node];
    const size_t me = out.size();
    DevNode d; memset(&d, 0, sizeof(d));
    d.lo[0] = n.aabb[0]; d.lo[1] = n.aabb[1]; d.lo[2] = n.aabb[2]; d.hix = n.aabb[3]; d.hiy = n.aabb[4]; d.hiz = n.aabb[5];
    d.ref_index = node; d.first = (int32_t)ent_id.size(); d.count = 0;
    out.push_back(d);
    if (n.level == 3) {
        for (int i = 0; i < 8; ++i) {                            // buckets until the first empty child (:284-285)
            const int leaf = n.children[i];
            if (leaf == 0) break;
            for (int j = 0; j < T.leaf_count[leaf]; ++j) {
                const int si = T.leaf_indices[(size_t)leaf * T.spl + j];
                if (si == 0 || !hittable(list[si])) continue;   // processHit skips index 0 (:255); ghosts are never hittable
                const rt_sphere& s = list[si];
                ent_hot.push_back(make_float4(s.center[0], s.center[1], s.center[2], radius_squared(s, precision)));
                ent_id.push_back(si);
            }
        }
        out[me].count = (int32_t)ent_id.size() - out[me].first;
    } else {
        for (int i = 0; i < 8; ++i) if (n.children[i] != 0) flatten(T, list, precision, n.children[i], out, ent_hot, ent_id);
    }
    out[me].skip = (int32_t)out.size();
}

int rt_build_octree(const rt_sphere* list, int num_hitables, int spheres_per_leaf, int precision, rt_octree** out) {
    if (!list || !out || num_hitables <= 0 || spheres_per_leaf <= 0) return RT_EINVAL;
    if (precision != RT_PRECISION_FP32 && precision != RT_PRECISION_FP16) return RT_EINVAL;
    *out = nullptr;
    rt_octree* O = new (std::nothrow) rt_octree();
    if (!O) return RT_ENOMEM;
    O->z = new (std::nothrow) rt_octree::Lazy();
    if (!O->z) { delete O; return RT_ENOMEM; }
    O->precision = precision;
    try {
        O->host = (precision == RT_PRECISION_FP16) ? buildOctree<half_t>(list, num_hitables, spheres_per_leaf)
                                                   : buildOctree<float>(list, num_hitables, spheres_per_leaf);
        flatten(*O->host, list, precision, 0, O->h_nodes, O->h_ent_hot, O->h_ent_id);
        O->n_nodes = (int)O->h_nodes.size(); O->n_entries = (int)O->h_ent_id.size();
        build_accel(O->accel, O->h_nodes, O->h_ent_id, O->h_ent_hot, num_hitables);
        O->n_world = num_hitables; O->bit_rows = (int)(O->accel.cellbits.size() / 16);
        if (precision != RT_PRECISION_FP32) O->accel.p.enabled = 0;      // the error bounds behind the grid are binary32 bounds
    } catch (const std::bad_alloc&) { rt_free_octree(O); return RT_ENOMEM; }
    *out = O;
    return 0;
}

static int octree_upload(const rt_octree* O) {
    rt_octree::Lazy& Z = *O->z;
    if (Z.uploaded) return 0;
    int rc;
    if (O->precision == RT_PRECISION_FP16) {
        // binary16 trees (rt_kernels_fp16.hip tests two spheres per packed instruction): each node's entries as PAIRS,
        // 16 bytes a pair — (cx_A | cx_B << 16), (cy..), (cz..), (r^2..) — an odd count padded with a NaN sphere (its
        // discriminant is never > 0); the node's first/count become pair indices, the entry -> sphere table follows the pairs.
        // The values are binary16 numbers already (every float of an FP16 world holds an exact binary16 image).
        std::vector<uint4> pairs; std::vector<int32_t> pid; std::vector<DevNode> hn(O->h_nodes);
        auto hb = [](float v) { return (uint32_t)half_t(v).bits; };
        for (DevNode& d : hn) {
            const int first = d.first, cnt = d.count;
            d.first = (int32_t)pairs.size();
            for (int e = 0; e < cnt; e += 2) {
                const float4 A4 = O->h_ent_hot[(size_t)first + e];
                const bool two = e + 1 < cnt;
                const uint32_t nanb = 0x7e00u;
                const float4 B4 = two ? O->h_ent_hot[(size_t)first + e + 1] : A4;
                pairs.push_back(make_uint4(hb(A4.x) | ((two ? hb(B4.x) : nanb) << 16), hb(A4.y) | ((two ? hb(B4.y) : nanb) << 16),
                                           hb(A4.z) | ((two ? hb(B4.z) : nanb) << 16), hb(A4.w) | ((two ? hb(B4.w) : nanb) << 16)));
                pid.push_back(O->h_ent_id[(size_t)first + e]);
                pid.push_back(two ? O->h_ent_id[(size_t)first + e + 1] : -1);
            }
            d.count = (int32_t)pairs.size() - d.first;
        }
        if (pairs.size() >= (size_t)1 << 23) return RT_ENOTSUP;         // a pair index shares a dword with a count and the owner lane in the kernels' segment pools (rt_kernels_fp16.hip)
        // plane table: the distinct box coordinates per axis, and per node its six indices into the concatenated table
        std::vector<float> planes; int np[3] = {0, 0, 0};
        {
            // the tree's boxes are the root box halved three times: 9 planes per axis, whichever nodes exist (rt_octgeom.h)
            static float fbox[kFullNodes][6];
            full_tree_boxes<half_t>(fbox);
            std::vector<float> ax[3];
            for (int fr = 0; fr < kFullNodes; ++fr) for (int k = 0; k < 3; ++k) { ax[k].push_back(fbox[fr][k]); ax[k].push_back(fbox[fr][3 + k]); }
            bool ok = true;
            for (int k = 0; k < 3; ++k) { std::sort(ax[k].begin(), ax[k].end()); ax[k].erase(std::unique(ax[k].begin(), ax[k].end()), ax[k].end()); }
            for (const DevNode& d : hn) {                                // (a tree from elsewhere: every node box must lie on those planes)
                const float bx[6] = {d.lo[0], d.lo[1], d.lo[2], d.hix, d.hiy, d.hiz};
                for (int q = 0; q < 6 && ok; ++q) ok = std::binary_search(ax[q % 3].begin(), ax[q % 3].end(), bx[q]);
            }
            if (ok && ax[0].size() + ax[1].size() + ax[2].size() <= 30) {
                int off[3] = {0, (int)ax[0].size(), (int)(ax[0].size() + ax[1].size())};
                for (int k = 0; k < 3; ++k) { np[k] = (int)ax[k].size(); planes.insert(planes.end(), ax[k].begin(), ax[k].end()); }
                for (DevNode& d : hn) {
                    const float b[6] = {d.lo[0], d.hix, d.lo[1], d.hiy, d.lo[2], d.hiz};        // x_low, x_high, y_low, y_high, z_low, z_high
                    uint32_t w = 0;
                    for (int q = 0; q < 6; ++q) {
                        const int k = q / 2;
                        const int idx = off[k] + (int)(std::lower_bound(ax[k].begin(), ax[k].end(), b[q]) - ax[k].begin());
                        w |= (uint32_t)idx << (5 * q);
                    }
                    d.pad[0] = (int32_t)w;
                }
            }
        }
        if (planes.empty()) planes.assign(1, 0.f);
        Z.dev.h16_np[0] = np[0]; Z.dev.h16_np[1] = np[1]; Z.dev.h16_np[2] = np[2];
        rc = upload(pairs, &Z.d_ent_hot);
        if (!rc) rc = upload(hn, &Z.d_nodes);
        if (!rc) rc = upload(pid, &Z.d_ent_id);
        if (!rc) rc = upload(planes, &Z.d_acc[11]);
        Z.dev.h16_planes = (const float*)Z.d_acc[11];
        Z.dev.n_entries = (int32_t)(2 * pairs.size());
    } else rc = upload(O->h_ent_hot, &Z.d_ent_hot);
    const AccelHost& A = O->accel;
    if (rc || (rc = upload(O->h_nodes, &Z.d_nodes)) || (rc = upload(O->h_ent_id, &Z.d_ent_id)) ||
        (rc = upload(A.large_hot, &Z.d_acc[0])) || (rc = upload(A.large_brick, &Z.d_acc[1])) || (rc = upload(A.cs, &Z.d_acc[2])) ||
        (rc = upload(A.hot, &Z.d_acc[3])) || (rc = upload(A.brick, &Z.d_acc[4])) || (rc = upload(A.memb_start, &Z.d_acc[5])) ||
        (rc = upload(A.memb_cell, &Z.d_acc[6])) || (rc = upload(A.cellnode, &Z.d_acc[8])) ||
        (rc = upload(A.bits_index, &Z.d_acc[9])) || (rc = upload(A.cellbits, &Z.d_acc[10]))) {
        void* bufs[3] = {Z.d_nodes, Z.d_ent_hot, Z.d_ent_id};                            // nothing half-made stays behind
        (void)free_all(bufs, 3); (void)free_all(Z.d_acc, 12);
        Z.d_nodes = Z.d_ent_hot = Z.d_ent_id = nullptr;
        return rc;
    }
    Z.dev.n_nodes = O->n_nodes;
    if (O->precision != RT_PRECISION_FP16) Z.dev.n_entries = O->n_entries;      // (binary16: entry slots of the pair layout, set above)
    Z.dev.nodes4 = (const float4*)Z.d_nodes; Z.dev.ent_hot = (const float4*)Z.d_ent_hot; Z.dev.ent_id = (const int32_t*)Z.d_ent_id;
    DevAccel p = A.p;
    p.large_hot = (const float4*)Z.d_acc[0]; p.large_brick = (const float4*)Z.d_acc[1];
    p.cs = (const int32_t*)Z.d_acc[2]; p.hot = (const float4*)Z.d_acc[3]; p.brick = (const float4*)Z.d_acc[4];
    p.memb_start = (const int32_t*)Z.d_acc[5]; p.memb_cell = (const int32_t*)Z.d_acc[6]; p.cellnode = (const int32_t*)Z.d_acc[8];
    p.bits_index = (const int32_t*)Z.d_acc[9]; p.cellbits = (const uint32_t*)Z.d_acc[10];
    Z.dev.acc = p;
    Z.uploaded = true;
    return 0;
}
int rt_octree_upload(rt_octree* O) { return O ? octree_upload(O) : RT_EINVAL; }

// RT_TRAVERSAL_REFERENCE: scan every bucket of every visited level-3 node, exactly like traverseTree.
// RT_TRAVERSAL_FAST (default): same hit records through the candidate-culling grid (fp32 only; fp16 always scans).
int rt_octree_set_traversal(rt_octree* O, int mode) {
    if (!O || (mode != RT_TRAVERSAL_REFERENCE && mode != RT_TRAVERSAL_FAST)) return RT_EINVAL;
    O->traversal = mode;
    return 0;
}

int rt_octree_accel_info(const rt_octree* O, int* grid_dim, float* cell_size, int* grid_entries, int* large_spheres) {
    if (!O) return RT_EINVAL;
    if (grid_dim) *grid_dim = O->accel.p.G;
    if (cell_size) *cell_size = O->accel.p.h;
    if (grid_entries) *grid_entries = (int)O->accel.n_entries;
    if (large_spheres) *large_spheres = O->accel.p.n_large;
    return 0;
}

// traversal copy (pre-order nodes with skip links, entry -> sphere index) for inspection
int rt_octree_flat_info(const rt_octree* O, int* n_nodes, int* n_entries) {
    if (!O) return RT_EINVAL;
    if (n_nodes) *n_nodes = O->n_nodes;
    if (n_entries) *n_entries = O->n_entries;
    return 0;
}

int rt_free_octree(rt_octree* O) {
    if (!O) return 0;
    int rc = 0;
    if (O->z) {
        void* bufs[3] = {O->z->d_nodes, O->z->d_ent_hot, O->z->d_ent_id};
        rc = free_all(bufs, 3);
        const int r2 = free_all(O->z->d_acc, 12); if (!rc) rc = r2;
        if (O->z->d_arena) { const hipError_t e = hipFree(O->z->d_arena); if (e != hipSuccess && !rc) rc = (int)e; }
        delete O->z->host_view;
        delete O->z;
    }
    delete O->host;      // the reference frees a new'ed Octree with free() (main.cu:473); here new/delete match
    delete O;
    return rc;
}

// reference-layout view of a tree built on the device: downloaded when an inspection call first asks for it
static int ensure_host_view(const rt_octree* O, const Octree** view) {
    rt_octree::Lazy& Z = *O->z;
    if (O->host) { *view = O->host; return 0; }
    if (Z.host_view) { *view = Z.host_view; return 0; }
    if (!Z.d_ref_nodes) return RT_EINVAL;
    Octree* T = new (std::nothrow) Octree();
    if (!T) return RT_ENOMEM;
    try {
        T->spl = Z.ref_spl; T->nodeCount = Z.ref_node_count; T->leafCount = Z.ref_leaf_count;
        T->dropped_full = Z.ref_dropped_full; T->dropped_outside = Z.ref_dropped_outside;
        T->nodes.resize(RT_OCTREE_MAX_NODES); T->leaf_count.resize(T->leafCount); T->leaf_indices.resize((size_t)T->leafCount * T->spl);
    } catch (const std::bad_alloc&) { delete T; return RT_ENOMEM; }
    hipError_t e = hipMemcpy(T->nodes.data(), Z.d_ref_nodes, sizeof(rt_octnode) * RT_OCTREE_MAX_NODES, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(T->leaf_count.data(), Z.d_leaf_count, sizeof(int32_t) * (size_t)T->leafCount, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(T->leaf_indices.data(), Z.d_leaf_indices, sizeof(int32_t) * (size_t)T->leafCount * T->spl, hipMemcpyDeviceToHost);
    if (e != hipSuccess) { delete T; return (int)e; }
    Z.host_view = T; *view = T;
    return 0;
}

int rt_octree_info(const rt_octree* O, int* node_count, int* leaf_count, int* spheres_per_leaf, int* dropped_full, int* dropped_outside) {
    if (!O) return RT_EINVAL;
    if (!O->host && O->z && O->z->d_ref_nodes) {                      // device-built: the counts came back with the build
        const rt_octree::Lazy& Z = *O->z;
        if (node_count) *node_count = Z.ref_node_count;
        if (leaf_count) *leaf_count = Z.ref_leaf_count;
        if (spheres_per_leaf) *spheres_per_leaf = Z.ref_spl;
        if (dropped_full) *dropped_full = Z.ref_dropped_full;
        if (dropped_outside) *dropped_outside = Z.ref_dropped_outside;
        return 0;
    }
    if (!O->host) return RT_EINVAL;
    if (node_count) *node_count = O->host->nodeCount;
    if (leaf_count) *leaf_count = O->host->leafCount;
    if (spheres_per_leaf) *spheres_per_leaf = O->host->spl;
    if (dropped_full) *dropped_full = O->host->dropped_full;
    if (dropped_outside) *dropped_outside = O->host->dropped_outside;
    return 0;
}
int rt_octree_nodes(const rt_octree* O, rt_octnode* out_nodes) {
    if (!O || !out_nodes) return RT_EINVAL;
    const Octree* T = nullptr;
    const int rc = ensure_host_view(O, &T);
    if (rc) return rc;
    memcpy(out_nodes, T->nodes.data(), sizeof(rt_octnode) * RT_OCTREE_MAX_NODES);
    return 0;
}
int rt_octree_leaves(const rt_octree* O, int32_t* counts, int32_t* indices) {
    if (!O || !counts || !indices) return RT_EINVAL;
    const Octree* T = nullptr;
    const int rc = ensure_host_view(O, &T);
    if (rc) return rc;
    memcpy(counts, T->leaf_count.data(), sizeof(int32_t) * T->leafCount);
    memcpy(indices, T->leaf_indices.data(), sizeof(int32_t) * (size_t)T->leafCount * T->spl);
    return 0;
}

// One device array of an uploaded FP32 tree, copied to the host (parity checks of the device build against the host build):
// 0 nodes, 1 ent_hot, 2 ent_id, 3 large_hot, 4 large_brick, 5 cs, 6 hot, 7 brick, 8 memb_start, 9 memb_cell, 10 cellnode,
// 11 bits_index, 12 cellbits.  *bytes receives the array's size; the copy happens when cap suffices.
int rt_octree_debug_array(const rt_octree* O, int which, void* out, size_t cap, size_t* bytes) {
    if (!O || !bytes) return RT_EINVAL;
    if (O->precision != RT_PRECISION_FP32 && which > 2) return RT_EINVAL;      // a binary16 tree has no candidate grid
    const int rc = octree_upload(O);
    if (rc) return rc;
    const rt_octree::Lazy& Z = *O->z;
    if (O->precision == RT_PRECISION_FP16) {                                   // pair layout: 16 B per pair, two table entries per pair
        const void* src16 = which == 0 ? (const void*)Z.dev.nodes4 : which == 1 ? (const void*)Z.dev.ent_hot : (const void*)Z.dev.ent_id;
        const size_t sz16 = which == 0 ? (size_t)Z.dev.n_nodes * sizeof(DevNode) : which == 1 ? (size_t)(Z.dev.n_entries / 2) * 16 : (size_t)Z.dev.n_entries * 4;
        *bytes = sz16;
        if (!out || cap < sz16 || sz16 == 0) return 0;
        return (int)hipMemcpy(out, src16, sz16, hipMemcpyDeviceToHost);
    }
    const DevAccel& p = Z.dev.acc;
    const size_t total = O->accel.n_entries + O->accel.n_entries_z, nbin = (size_t)O->accel.p.G * O->accel.p.Gf, nl = (size_t)O->accel.p.n_large;
    const void* src = nullptr; size_t sz = 0;
    switch (which) {
        case 0: src = Z.dev.nodes4; sz = (size_t)O->n_nodes * sizeof(DevNode); break;
        case 1: src = Z.dev.ent_hot; sz = (size_t)O->n_entries * 16; break;
        case 2: src = Z.dev.ent_id; sz = (size_t)O->n_entries * 4; break;
        case 3: src = p.large_hot; sz = nl * 16; break;
        case 4: src = p.large_brick; sz = (nl ? nl : 1) * 32; break;
        case 5: src = p.cs; sz = O->accel.p.enabled || total ? 2 * (nbin + 1) * 4 : 0; break;
        case 6: src = p.hot; sz = O->accel.p.G ? (total + 16) * 16 : 0; break;
        case 7: src = p.brick; sz = O->accel.p.G ? (total + 16) * 32 : 0; break;
        case 8: src = p.memb_start; sz = ((size_t)O->n_world + 1) * 4; break;
        case 9: src = p.memb_cell; sz = (size_t)O->n_entries * 4; break;
        case 10: src = p.cellnode; sz = 512 * 4; break;
        case 11: src = p.bits_index; sz = (size_t)O->n_world * 4; break;
        case 12: src = p.cellbits; sz = (size_t)O->bit_rows * 64; break;
        default: return RT_EINVAL;
    }
    *bytes = sz;
    if (!out || cap < sz || sz == 0 || !src) return 0;
    return (int)hipMemcpy(out, src, sz, hipMemcpyDeviceToHost);
}

// buildOctree + traversal copy + candidate grid on the device, from the world's device-resident sphere list (rt_build.hip).
// Inputs the device build declines are built on the host from the world's own copy of the list.
int rt_build_octree_gpu(const rt_world* world, int spheres_per_leaf, rt_octree** out, void* stream) {
    if (!world || !out || spheres_per_leaf <= 0) return RT_EINVAL;
    *out = nullptr;
    {
        int rc = world_upload(world);
        if (rc) return rc;
        rt_octree* O = new (std::nothrow) rt_octree();
        if (!O) return RT_ENOMEM;
        O->z = new (std::nothrow) rt_octree::Lazy();
        if (!O->z) { delete O; return RT_ENOMEM; }
        O->precision = world->precision;
        rc = gpubuild::build(O, (const float4*)world->z->d_geom, (const int32_t*)world->z->d_kind, world->n, spheres_per_leaf, (hipStream_t)stream);
        if (rc == 0) { *out = O; return 0; }
        (void)rt_free_octree(O);
        if (rc != RT_ENOTSUP) return rc;
    }
    std::vector<rt_sphere> list;
    try {
        list.resize((size_t)world->n);
        for (int i = 0; i < world->n; ++i) {
            rt_sphere& s = list[i];
            const float4 g = world->h_geom[i], m = world->h_mat[i];
            s.center[0] = g.x; s.center[1] = g.y; s.center[2] = g.z; s.radius = g.w;
            s.material = world->h_kind[i]; s.albedo[0] = m.x; s.albedo[1] = m.y; s.albedo[2] = m.z; s.param = m.w;
        }
    } catch (const std::bad_alloc&) { return RT_ENOMEM; }
    return rt_build_octree(list.data(), world->n, spheres_per_leaf, world->precision, out);
}

// ------------------------------------------------------------------------------------------------ the hot path
int64_t rt_part_pixels(int max_x, int max_y, rt_partition part) {
    if (max_x <= 0 || max_y <= 0 || !valid_partition(part)) return RT_EINVAL;
    if (part_whole(part.nparts, part.tile_begin, part.tile_end)) return (int64_t)max_x * max_y;
    const int64_t tiles = (int64_t)((max_x + 7) / 8) * ((max_y + 7) / 8);
    if (!range_in_frame(part, tiles)) return RT_EINVAL;
    return local_tiles_of(tiles, part) * 64;
}

int rt_render_init(int max_x, int max_y, rt_rand_state* d_rand_state, rt_partition part, void* stream) {
    if (max_x <= 0 || max_y <= 0 || !valid_partition(part)) return RT_EINVAL;
    const int64_t npx = rt_part_pixels(max_x, max_y, part);
    if (npx < 0) return RT_EINVAL;
    if (npx == 0) return 0;                                              // a part without tiles (more parts than tiles): nothing to do
    if (!d_rand_state) return RT_EINVAL;
    return (int)launch_render_init(d_rand_state, max_x, max_y, part.part, part.nparts, part.tile_begin, part.tile_end, (hipStream_t)stream);
}

// no octree passed: the world's one-node list tree, if it has one and the fast list traversal is selected
static const rt_octree* list_tree_of(const rt_world* world) {
    if (world->list_traversal != RT_TRAVERSAL_FAST) return nullptr;
    return ensure_list_tree(world);
}

// what a render / trace call needs on the device: the world, and the tree it walks (the caller's octree, or the world's list tree)
static int ensure_on_device(const rt_world* world, const rt_octree*& d_octree) {
    if (!d_octree) d_octree = list_tree_of(world);                    // hitable_list::hit through the candidate grid
    int rc = world_upload(world);
    if (!rc && d_octree) rc = octree_upload(d_octree);
    return rc;
}
static DevTree tree_args(const rt_octree* d_octree) {
    DevTree T;
    if (d_octree) { T = d_octree->z->dev; T.acc.enabled = T.acc.enabled && d_octree->traversal == RT_TRAVERSAL_FAST; }
    else memset(&T, 0, sizeof(T));
    return T;
}
// a scheduling knob of rt_tuning.h, overridable from the environment for tuning sweeps on one build (read once per process and knob)
static float tune_value(const char* env, float dflt) {
    static std::mutex mu;
    static std::vector<std::pair<std::string, float>> seen;
    std::lock_guard<std::mutex> lock(mu);
    for (const auto& kv : seen) if (kv.first == env) return kv.second;
    float v = dflt;
    if (const char* e = getenv(env)) { char* end = nullptr; const float x = strtof(e, &end); if (end != e) v = x; }
    seen.emplace_back(env, v);
    return v;
}
static bool capturing(hipStream_t st) {
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cs) != hipSuccess) { (void)hipGetLastError(); return false; }
    return cs != hipStreamCaptureStatusNone;
}

static int render_common(rt_render_ctx* ctx, void* fb, int max_x, int max_y, int ns, const rt_world* world, rt_rand_state* d_rand_state, const rt_octree* d_octree, rt_partition part, void* stream, int mode) {
    if (!world || max_x <= 0 || max_y <= 0 || ns <= 0 || !valid_partition(part)) return RT_EINVAL;
    if (d_octree && d_octree->precision != world->precision) return RT_EINVAL;
    { const int64_t npx = rt_part_pixels(max_x, max_y, part); if (npx < 0) return RT_EINVAL; if (npx == 0) return 0; }   // (a part without tiles — more parts than tiles: nothing to do)
    if (!fb || !d_rand_state) return RT_EINVAL;
    const hipStream_t st = (hipStream_t)stream;
    const bool cap = capturing(st);
    int rc = ensure_on_device(world, d_octree);
    rt_render_ctx& C = ctx ? *ctx : world->z->ctx;
    if (!rc && !cap) rc = ctx_prepare(C);
    if (rc) return rc;
    if (!C.d_queue) return RT_EINVAL;                                  // a context first used inside a capture: prepare it before (rt_render_ctx_reserve)
    RenderArgs A;
    A.fb = fb; A.rand_state = d_rand_state; A.max_x = max_x; A.max_y = max_y; A.ns = ns;
    A.tiles_x = (max_x + 7) / 8; A.tiles_y = (max_y + 7) / 8;
    A.part = part.part; A.nparts = part.nparts; A.tile_begin = part.tile_begin; A.tile_end = part.tile_end;
    const int64_t tiles = (int64_t)A.tiles_x * A.tiles_y;
    A.n_local_tiles = local_tiles_of(tiles, part);
    A.scene = world->z->dev;
    A.tree = tree_args(d_octree);
    A.order = nullptr; A.long_flag = nullptr; A.long_list = nullptr;
    A.tail_list = nullptr; A.tail_ws = nullptr; A.f_tail = 0.f; A.head_sum = 0; A.head_sum_dense = 0; A.head_min_load = 0.f;
    A.n_lanes = 0; A.f_inflight = tune_value("RT_F_INFLIGHT", RT_F_INFLIGHT); A.f_inflight_dense = tune_value("RT_F_INFLIGHT_DENSE", RT_F_INFLIGHT_DENSE); A.f_static = tune_value("RT_F_STATIC", RT_F_STATIC);
    const bool sched = mode == 0 && ns >= 4;
    // expensive tiles first (k_tile_cost / k_tile_order).  The workspace grows on first use of a larger frame: call rt_render
    // (or rt_render_ctx_reserve) once before capturing it into a hipGraph.
    if (sched && C.sched_tiles < A.n_local_tiles) {
        if (cap) return RT_EINVAL;
        if ((rc = ctx_reserve(C, A.n_local_tiles))) return rc;
    }
    // launches sharing a context are ordered: the previous render kernel has finished before this call's kernels touch the
    // counters ring's neighbours and the workspace (a no-op on the same stream)
    if (!cap && C.has_done && C.last_stream != st) RT_TRY(hipStreamWaitEvent(st, C.done, 0));
    A.queue = C.d_queue + (size_t)(C.launches++ % kQueueSlots) * kQueueStride;
    C.last_queue = A.queue;
    RT_TRY(launch_zero_counters(A.queue, (int)kQueueStride, st));
    if (mode == 1) {
        // render_progressive is one sample per launch (main.cu:119-142, called once per displayed frame, :275).  The pass with
        // current_sample == 1 runs the pilot pass of rt_render and KEEPS the tile order (most expensive tiles first) in the context;
        // the following passes of the same frame reuse it at no cost (C3: 0.594 -> 0.567 ms per pass; the pilot pass itself is
        // 0.5 ms).  The long-chain list is not kept: a pass is one sample, and waves set aside for the crevice pixels' ~40 bounces cost
        // a pass more than they save (0.70 - 0.88 ms).  Scheduling only: which lane renders a pixel and when never changes the pixel.
        const uint64_t key[5] = {world->serial, d_octree ? d_octree->serial : 0, ((uint64_t)(uint32_t)max_x << 32) | (uint32_t)max_y,
                                 (((uint64_t)(uint32_t)part.part << 32) | (uint32_t)part.nparts) ^ ((uint64_t)part.tile_begin * 0x9e3779b97f4a7c15ull) ^ ((uint64_t)part.tile_end << 20),
                                 (uint64_t)(d_octree ? d_octree->traversal : 0)};
        if (ns == 1 && !cap) {
            C.p_valid = false;
            if ((rc = ctx_reserve_progressive(C, A.n_local_tiles))) return rc;
            if (world->precision == RT_PRECISION_FP16) RT_TRY(launch_tile_order_h(A, d_octree != nullptr, C.p_cost, C.p_order, nullptr, nullptr, st));
            else RT_TRY(launch_tile_order(A, d_octree != nullptr, C.p_cost, C.p_order, nullptr, nullptr, st));
            memcpy(C.p_key, key, sizeof(key)); C.p_valid = true;
        }
        if (C.p_valid && memcmp(C.p_key, key, sizeof(key)) == 0 && C.p_tiles >= A.n_local_tiles) { A.order = C.p_order; if (cap) C.p_pinned = true; }
    }
    if (sched) {
        const bool classify = ns >= 16;          // long-chain pre-classification pays only when chains are long
        if (classify && world->precision != RT_PRECISION_FP16) { A.tail_list = C.d_long + (size_t)A.n_local_tiles * 64; A.tail_ws = C.d_long + (size_t)C.sched_tiles * 128; A.f_tail = tune_value("RT_F_TAIL", RT_F_TAIL); A.head_sum = (int)tune_value("RT_HEAD_SUM_SPARSE", (float)RT_HEAD_SUM_SPARSE); A.head_sum_dense = (int)tune_value("RT_HEAD_SUM_DENSE", (float)RT_HEAD_SUM_DENSE); A.head_min_load = tune_value("RT_HEAD_LOAD_DENSE", (float)RT_HEAD_LOAD_DENSE); }
        if (world->precision == RT_PRECISION_FP16) RT_TRY(launch_tile_order_h(A, d_octree != nullptr, C.d_cost, C.d_order, classify ? C.d_flags : nullptr, classify ? C.d_long : nullptr, st));
        else RT_TRY(launch_tile_order(A, d_octree != nullptr, C.d_cost, C.d_order, classify ? C.d_flags : nullptr, classify ? C.d_long : nullptr, st));
        A.order = C.d_order;
        if (classify) { A.long_flag = C.d_flags; A.long_list = C.d_long; }
    }
    // timing events only outside a capture (recorded into a graph they would never be "recorded" for hipEventElapsedTime)
    const unsigned ek = C.ev_head % 64u;
    if (!cap) RT_TRY(hipEventRecord(C.ev0[ek], st));
    if (world->precision == RT_PRECISION_FP16) RT_TRY(launch_render_h(A, d_octree != nullptr, mode, st));
    else if (world->arith == RT_ARITH_CONTRACT) RT_TRY(fmac::launch_render(A, d_octree != nullptr, mode, st));
    else RT_TRY(launch_render(A, d_octree != nullptr, mode, st));
    if (!cap) {
        RT_TRY(hipEventRecord(C.ev1[ek], st));
        C.ev_head++; if (C.ev_count < 64) C.ev_count++;
        RT_TRY(hipEventRecord(C.done, st));
        C.has_done = true; C.last_stream = st;
    }
    return 0;
}

static int ctx_times(rt_render_ctx& C, float* ms_out, int max, int* count) {
    int n = 0;
    const unsigned have = C.ev_count;
    for (unsigned k = 0; k < have && n < max; ++k) {
        const unsigned slot = (C.ev_head - have + k) % 64u;
        RT_TRY(hipEventSynchronize(C.ev1[slot]));
        float ms = 0.f;
        RT_TRY(hipEventElapsedTime(&ms, C.ev0[slot], C.ev1[slot]));
        ms_out[n++] = ms;
    }
    *count = n;
    C.ev_count = 0;
    return 0;
}
// the scheduling counters of the context's latest launch, once it has finished
static int ctx_counters(rt_render_ctx& C, uint32_t* out4) {
    if (!C.last_queue) { out4[0] = out4[1] = out4[2] = out4[3] = 0u; return 0; }
    if (C.has_done) RT_TRY(hipEventSynchronize(C.done));
    uint32_t q[6];
    RT_TRY(hipMemcpy(q, C.last_queue, sizeof(q), hipMemcpyDeviceToHost));
    out4[0] = q[0]; out4[1] = q[1]; out4[2] = q[2] + q[4]; out4[3] = q[3] + q[5];      // (chains started alone in a wave are listed apart: [4], [5])
    return 0;
}
int rt_world_render_counters(rt_world* W, uint32_t* out4) {
    if (!W || !out4) return RT_EINVAL;
    return ctx_counters(W->z->ctx, out4);
}
int rt_render_ctx_counters(rt_render_ctx* C, uint32_t* out4) {
    if (!C || !out4) return RT_EINVAL;
    return ctx_counters(*C, out4);
}
int rt_world_render_times(rt_world* W, float* ms_out, int max, int* count) {
    if (!W || !ms_out || !count || max < 0) return RT_EINVAL;
    return ctx_times(W->z->ctx, ms_out, max, count);
}
int rt_render_ctx_times(rt_render_ctx* C, float* ms_out, int max, int* count) {
    if (!C || !ms_out || !count || max < 0) return RT_EINVAL;
    return ctx_times(*C, ms_out, max, count);
}

int rt_render(void* fb, int max_x, int max_y, int ns, const rt_world* world, rt_rand_state* d_rand_state, const rt_octree* d_octree, rt_partition part, void* stream) {
    return render_common(nullptr, fb, max_x, max_y, ns, world, d_rand_state, d_octree, part, stream, 0);
}
int rt_render_progressive(void* fb, int max_x, int max_y, int current_sample, const rt_world* world, rt_rand_state* d_rand_state, const rt_octree* d_octree, rt_partition part, void* stream) {
    return render_common(nullptr, fb, max_x, max_y, current_sample, world, d_rand_state, d_octree, part, stream, 1);
}
int rt_render_on(rt_render_ctx* ctx, void* fb, int max_x, int max_y, int ns, const rt_world* world, rt_rand_state* d_rand_state, const rt_octree* d_octree, rt_partition part, void* stream) {
    if (!ctx) return RT_EINVAL;
    return render_common(ctx, fb, max_x, max_y, ns, world, d_rand_state, d_octree, part, stream, 0);
}
int rt_render_progressive_on(rt_render_ctx* ctx, void* fb, int max_x, int max_y, int current_sample, const rt_world* world, rt_rand_state* d_rand_state, const rt_octree* d_octree, rt_partition part, void* stream) {
    if (!ctx) return RT_EINVAL;
    return render_common(ctx, fb, max_x, max_y, current_sample, world, d_rand_state, d_octree, part, stream, 1);
}

// the kernel rt_render (mode 0) / rt_render_progressive (mode 1) launches for this world and tree — the library's own
// selection rule, for profiles and bench lines (rocprofv3 shows the same name)
int rt_render_kernel_name(const rt_world* world, const rt_octree* d_octree, int mode, char* out, int cap) {
    if (!world || !out || cap <= 0 || (mode != 0 && mode != 1)) return RT_EINVAL;
    if (d_octree && d_octree->precision != world->precision) return RT_EINVAL;
    if (!d_octree) d_octree = list_tree_of(world);
    const bool tree = d_octree != nullptr;
    std::string name;
    if (world->precision == RT_PRECISION_FP16) name = render_kernel_name_h(tree, mode);
    else {
        DevAccel acc{};
        if (tree) { acc = d_octree->accel.p; acc.enabled = acc.enabled && d_octree->traversal == RT_TRAVERSAL_FAST; }
        name = render_kernel_name(tree, mode, acc);
    }
    if ((int)name.size() + 1 > cap) return RT_EINVAL;
    memcpy(out, name.c_str(), name.size() + 1);
    return 0;
}

int rt_assemble(void* fb_full, const void* fb_parts, int max_x, int max_y, int nparts, int precision, void* stream) {
    if (!fb_full || !fb_parts || max_x <= 0 || max_y <= 0 || nparts < 1) return RT_EINVAL;
    if (precision == RT_PRECISION_FP16) return (int)launch_assemble_h(fb_full, fb_parts, max_x, max_y, nparts, (hipStream_t)stream);
    if (precision != RT_PRECISION_FP32) return RT_EINVAL;
    return (int)launch_assemble((float*)fb_full, (const float*)fb_parts, max_x, max_y, nparts, (hipStream_t)stream);
}

int rt_assemble_split(void* fb_full, const void* fb_parts, int max_x, int max_y, int nparts, const int64_t* starts, int64_t part_stride_px, int precision, void* stream) {
    if (!fb_full || !fb_parts || !starts || max_x <= 0 || max_y <= 0 || nparts < 1 || nparts > rt::kMaxSplitParts || part_stride_px < 0) return RT_EINVAL;
    if (precision != RT_PRECISION_FP32 && precision != RT_PRECISION_FP16) return RT_EINVAL;
    const int64_t tiles = (int64_t)((max_x + 7) / 8) * ((max_y + 7) / 8);
    if (starts[0] != 0 || starts[nparts] != tiles) return RT_EINVAL;
    long long st64[rt::kMaxSplitParts + 1];
    for (int p = 0; p <= nparts; ++p) { if (p && starts[p] < starts[p - 1]) return RT_EINVAL; if (p && (starts[p] - starts[p - 1]) * 64 > part_stride_px && nparts > 1) return RT_EINVAL; st64[p] = starts[p]; }
    return (int)launch_assemble_split(fb_full, fb_parts, max_x, max_y, nparts, st64, part_stride_px, precision == RT_PRECISION_FP16, (hipStream_t)stream);
}

// rt_split_balanced: the pilot pass over the whole frame, the counts to the host, the cuts in integer arithmetic
int rt_split_balanced(rt_render_ctx* ctx, const rt_world* world, const rt_octree* d_octree, int max_x, int max_y, int nparts, int64_t* starts,
                      int32_t* tile_bounces, int32_t* tile_tests, int32_t* tile_columns, void* stream) {
    if (!world || !starts || max_x <= 0 || max_y <= 0 || nparts < 1 || nparts > rt::kMaxSplitParts) return RT_EINVAL;
    if (d_octree && d_octree->precision != world->precision) return RT_EINVAL;
    const int64_t tiles = (int64_t)((max_x + 7) / 8) * ((max_y + 7) / 8);
    if (tiles < nparts) return RT_EINVAL;
    const hipStream_t st = (hipStream_t)stream;
    if (capturing(st)) return RT_EINVAL;
    int rc = ensure_on_device(world, d_octree);
    rt_render_ctx& C = ctx ? *ctx : world->z->ctx;
    if (!rc) rc = ctx_prepare(C);
    if (!rc) rc = ctx_reserve(C, tiles);
    if (rc) return rc;
    if (C.has_done && C.last_stream != st) RT_TRY(hipStreamWaitEvent(st, C.done, 0));      // (the workspace is shared with the context's renders)
    RenderArgs A;
    memset(&A, 0, sizeof(A));
    A.max_x = max_x; A.max_y = max_y; A.ns = 1;
    A.tiles_x = (max_x + 7) / 8; A.tiles_y = (max_y + 7) / 8;
    A.part = 0; A.nparts = 1; A.tile_begin = 0; A.tile_end = 0;
    A.n_local_tiles = tiles;
    A.scene = world->z->dev;
    A.tree = tree_args(d_octree);
    const bool half = world->precision == RT_PRECISION_FP16;
    RT_TRY(hipMemsetAsync(C.d_work, 0, sizeof(int) * (size_t)tiles * 2, st));
    if (half) RT_TRY(launch_pilot_h(A, d_octree != nullptr, C.d_cost, st));
    else RT_TRY(launch_pilot(A, d_octree != nullptr, C.d_cost, nullptr, C.d_work, st));
    std::vector<int32_t> cost, work;
    try { cost.resize((size_t)tiles); work.resize((size_t)tiles * 2); } catch (const std::bad_alloc&) { return RT_ENOMEM; }
    RT_TRY(hipMemcpyAsync(cost.data(), C.d_cost, sizeof(int) * (size_t)tiles, hipMemcpyDeviceToHost, st));
    RT_TRY(hipMemcpyAsync(work.data(), C.d_work, sizeof(int) * (size_t)tiles * 2, hipMemcpyDeviceToHost, st));
    RT_TRY(hipEventRecord(C.done, st)); C.has_done = true; C.last_stream = st;
    RT_TRY(hipStreamSynchronize(st));
    // a tile's cost: wb x bounces + wt x tests + wc x columns (rt_tuning.h RT_SPLIT_W*: integer weights fitted to measured band times)
    const int64_t wb = (int64_t)tune_value("RT_SPLIT_WB", RT_SPLIT_WB), wt = (int64_t)tune_value("RT_SPLIT_WT", RT_SPLIT_WT), wc = (int64_t)tune_value("RT_SPLIT_WC", RT_SPLIT_WC);
    const int32_t* cols = work.data() + tiles;
    int64_t total = 0;
    for (int64_t t = 0; t < tiles; ++t) {
        cost[t] /= 4;                                                     // (k_tile_cost stores 4 x the bounces of the tile's pilot samples)
        if (tile_bounces) tile_bounces[t] = cost[t];
        if (tile_tests) tile_tests[t] = work[t];
        if (tile_columns) tile_columns[t] = cols[t];
        total += wb * cost[t] + wt * work[t] + wc * cols[t];
    }
    starts[0] = 0; starts[nparts] = tiles;
    int64_t run = 0, t = 0;
    for (int p = 1; p < nparts; ++p) {
        // the first tile at which the running cost has passed p/nparts of the total — but every band keeps at least one tile
        const int64_t want = (total / nparts) * p + (total % nparts) * p / nparts;
        while (t < tiles - (nparts - p) && (run < want || t < starts[p - 1] + 1)) { run += wb * cost[t] + wt * work[t] + wc * cols[t]; ++t; }
        starts[p] = t;
    }
    return 0;
}

int rt_trace_rays(const rt_world* world, const rt_octree* d_octree, const float* d_rays, int64_t n, rt_hit_record* d_out, void* stream) {
    if (!world || !d_rays || !d_out || n < 0) return RT_EINVAL;
    if (d_octree && d_octree->precision != world->precision) return RT_EINVAL;
    const int rc = ensure_on_device(world, d_octree);
    if (rc) return rc;
    const DevTree T = tree_args(d_octree);
    if (world->precision == RT_PRECISION_FP16) return (int)launch_trace_h(world->z->dev, T, d_octree != nullptr, d_rays, n, d_out, (hipStream_t)stream);
    return (int)launch_trace(world->z->dev, T, d_octree != nullptr, d_rays, n, d_out, (hipStream_t)stream);
}

#ifdef RT_H16_STATS
int rt_debug_h16(unsigned long long* out8, int reset) { return (int)rt::read_h16_stats(out8, reset); }      // diagnostic variant only
#endif
#ifdef RT_STATS
// diagnostic build only (librt_amd_stats.so): 16 work counters, see rt_kernels.hip
int rt_debug_stats(unsigned long long* out16, int reset) { return (int)read_stats(out16, reset); }
int rt_debug_waves(unsigned long long* out) { return (int)read_wave_dbg(out); }
int rt_debug_pilot(int* out, int n) { return (int)read_pilot_dbg(out, n); }
#endif

// ------------------------------------------------------------------------------------------------ output
// the formatters live in host/rt_image.hpp (host-only C++: also compiled under ASan / UBSan by tests/test_host_sanitizers.py)
using rt::ppm_text;

int64_t rt_format_ppm(int nx, int ny, const void* fb, int precision, char* out, int64_t cap) {
    if (nx <= 0 || ny <= 0 || !fb) return RT_EINVAL;
    std::string s;
    try { ppm_text(nx, ny, fb, precision, s); } catch (const std::bad_alloc&) { return RT_ENOMEM; }
    if (out && (int64_t)s.size() <= cap) memcpy(out, s.data(), s.size());
    return (int64_t)s.size();
}

int rt_write_ppm(const char* path, int nx, int ny, const void* fb, int precision) {
    if (nx <= 0 || ny <= 0 || !fb) return RT_EINVAL;
    std::string s;                                            // formatted once, written straight from the string
    try { ppm_text(nx, ny, fb, precision, s); } catch (const std::bad_alloc&) { return RT_ENOMEM; }
    FILE* f = path ? fopen(path, "wb") : stdout;
    if (!f) return RT_EIO;
    const size_t w = fwrite(s.data(), 1, s.size(), f);
    if (path) { if (fclose(f) != 0) return RT_EIO; } else fflush(f);
    return w == s.size() ? 0 : RT_EIO;
}

int rt_write_image(const char* path, int nx, int ny, const void* fb, int precision, int format) {
    if (!path || nx <= 0 || ny <= 0 || !fb) return RT_EINVAL;
    if (precision != RT_PRECISION_FP32 && precision != RT_PRECISION_FP16) return RT_EINVAL;
    if (format == RT_IMAGE_P3) return rt_write_ppm(path, nx, ny, fb, precision);
    if (format != RT_IMAGE_P6 && format != RT_IMAGE_PFM) return RT_EINVAL;
    FILE* f = fopen(path, "wb");
    if (!f) return RT_EIO;
    bool ok = true;
    try {
        ok = rt::write_binary_image(f, nx, ny, fb, precision, format);
    } catch (const std::bad_alloc&) { fclose(f); return RT_ENOMEM; }
    if (fclose(f) != 0) ok = false;
    return ok ? 0 : RT_EIO;
}

} // extern "C"
