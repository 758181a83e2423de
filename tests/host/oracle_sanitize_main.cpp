// oracle_sanitize_main.cpp — test program (tests/test_host_sanitizers.py compiles it TOGETHER WITH oracle/rt_oracle_capi.cpp under
// -fsanitize=address,undefined and runs it; not part of the product).  Drives the CPU oracle's C entry points the way tests/oracle_lib.py
// does — worlds, octrees (with full buckets too), rays, frames in both precisions, progressive passes, PPM text — so the checker
// itself has been under the sanitizers.  Prints a checksum per step.
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <vector>

struct orc_scene;
extern "C" {
orc_scene* orc_scene_create(int num_spheres, float radius, int nx, int ny, int fp16, int use_octree, int spl);
void orc_scene_destroy(orc_scene* s);
void orc_scene_info(orc_scene* s, int64_t* out);
void orc_scene_spheres(orc_scene* s, float* geom, float* mat, int32_t* kind);
void orc_scene_octree_nodes(orc_scene* s, int32_t* level, float* box, int32_t* children);
void orc_scene_octree_leaves(orc_scene* s, int32_t* counts, int32_t* indices);
void orc_trace(orc_scene* s, int64_t n, const float* rays, int mode, int32_t* hit, int32_t* sph, float* t, float* p, float* nrm);
void orc_render_init(int max_x, int max_y, int row0, int rows, void* states);
void orc_render(orc_scene* s, float* fb, int max_x, int max_y, int ns, void* states, int row0, int rows, int nthreads, uint64_t* counters);
void orc_render_progressive(orc_scene* s, float* fb, int max_x, int max_y, int current_sample, void* states, int nthreads);
int64_t orc_ppm(const float* fb, int nx, int ny, char* out, int64_t cap);
}

static uint64_t fnv(const void* p, size_t n) { uint64_t h = 1469598103934665603ull; const unsigned char* b = (const unsigned char*)p; for (size_t i = 0; i < n; ++i) { h ^= b[i]; h *= 1099511628211ull; } return h; }

static void scene(int n, int fp16, int use_octree, int spl, int nx, int ny, int ns) {
    orc_scene* S = orc_scene_create(n, 0.1f, nx, ny, fp16, use_octree, spl);
    int64_t info[8]; orc_scene_info(S, info);
    std::vector<float> geom((size_t)n * 4), mat((size_t)n * 4); std::vector<int32_t> kind((size_t)n);
    orc_scene_spheres(S, geom.data(), mat.data(), kind.data());
    if (use_octree) {
        std::vector<int32_t> level(585), children(585 * 8); std::vector<float> box(585 * 6);
        orc_scene_octree_nodes(S, level.data(), box.data(), children.data());
        std::vector<int32_t> counts((size_t)info[4]), idx((size_t)info[4] * spl);
        orc_scene_octree_leaves(S, counts.data(), idx.data());
    }
    // a few rays from the camera position towards the sphere field, plus awkward ones (zero components, far origin)
    const float rays[6 * 4] = {13, 2, 3, -13, -2, -3,   13, 2, 3, -13, -1.9f, -2,   0, 5, 0, 0, -1, 0,   100, 50, 100, -1, -0.5f, -1};
    int32_t hit[4], sph[4]; float t[4], p[12], nrm[12];
    for (int mode = 0; mode < (use_octree ? 3 : 2); ++mode) orc_trace(S, 4, rays, mode, hit, sph, t, p, nrm);
    std::vector<unsigned char> st((size_t)nx * ny * 48);
    std::vector<float> fb((size_t)nx * ny * 3);
    orc_render_init(nx, ny, 0, ny, st.data());
    uint64_t counters[6];
    orc_render(S, fb.data(), nx, ny, ns, st.data(), 0, ny, 2, counters);
    const uint64_t h_frame = fnv(fb.data(), fb.size() * 4);
    // a band of rows rendered on its own (compact buffers) — the form the GPU tests use at full size
    orc_render_init(nx, ny, 3, 2, st.data());
    orc_render(S, fb.data(), nx, ny, ns, st.data(), 3, 2, 1, nullptr);
    orc_render_init(nx, ny, 0, ny, st.data());
    for (int k = 1; k <= 2; ++k) orc_render_progressive(S, fb.data(), nx, ny, k, st.data(), 2);
    const int64_t need = orc_ppm(fb.data(), nx, ny, nullptr, 0);
    std::vector<char> txt((size_t)need);
    orc_ppm(fb.data(), nx, ny, txt.data(), need);
    printf("N=%d fp16=%d octree=%d spl=%d: real=%lld nodes=%lld leaves=%lld entries=%lld dropped_full=%lld frame=%016llx rays=%llu ppm=%lld\n", n, fp16, use_octree, spl,
           (long long)info[1], (long long)info[3], (long long)info[4], (long long)info[5], (long long)info[6], (unsigned long long)h_frame, (unsigned long long)counters[0], (long long)need);
    orc_scene_destroy(S);
}

int main() {
    scene(22, 0, 0, 30, 24, 14, 2);
    scene(22, 0, 1, 30, 24, 14, 2);
    scene(500, 0, 1, 30, 20, 12, 2);
    scene(2000, 0, 1, 3, 16, 10, 1);          // full buckets (dropped_full > 0)
    scene(22, 1, 0, 30, 16, 10, 2);
    scene(500, 1, 1, 30, 12, 8, 1);
    scene(10000, 0, 1, 32, 12, 8, 1);
    return 0;
}
