// host_sanitize_check.cpp — test program (tests/test_host_sanitizers.py builds it with -fsanitize=address,undefined and runs it;
// not part of the product).  Drives the product's HOST code — dd2360-raytracing_amd/host/rt_scene.hpp and host/rt_image.hpp, the
// code behind rt_create_world / rt_build_octree / rt_format_ppm / rt_write_image — through the sizes the BASELINE configs use, so
// that every allocation, index and cast on those paths runs under the sanitizers once.  Prints one summary line per step; the test
// compares them with the pinned counts.  The cautionary example is the reference itself: main.cu:410 allocates the
// octree with `new` and main.cu:473 releases it with free().
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cfloat>
#include <cmath>
#include <limits>
#include <memory>
#include <string>
#include <vector>
#include "../../dd2360-raytracing_amd/host/rt_scene.hpp"
#include "../../dd2360-raytracing_amd/host/rt_image.hpp"

using namespace rt;

template <class R> static void world_and_tree(const char* tag, int n, int spl, int nx, int ny) {
    rt_rand_state rs;
    xorwow::init(rs, 1984ull);                                            // rand_init, main.cu:78-82
    std::vector<rt_sphere> list((size_t)n);
    rt_camera cam;
    const int created = create_world_pods<R>(list.data(), n, 0.1f, &cam, nx, ny, &rs);
    std::unique_ptr<Octree> T(buildOctree<R>(list.data(), n, spl));       // (released with delete — not with free() as main.cu:473 does)
    long long entries = 0;
    for (int l = 1; l < T->leafCount; ++l) entries += T->leaf_count[l];
    printf("%s N=%d spl=%d created=%d nodes=%d leaves=%d entries=%lld dropped_full=%d dropped_outside=%d\n", tag, n, spl, created, T->nodeCount, T->leafCount, entries,
           T->dropped_full, T->dropped_outside);
}

// the object form of create_world (heap-allocated materials behind shared_ptr, the hitable_list over raw pointers) and one ray through it
template <class R> static void world_objects(const char* tag, int n) {
    rt_rand_state rs;
    xorwow::init(rs, 1984ull);
    world_t<R> W;
    create_world<R>(W, n, 0.1f, 1200, 800, &rs);
    rt_rand_state px;
    xorwow::init(px, 1984ull + 12345ull);
    int hits = 0;
    for (int k = 0; k < 2000; ++k) {
        const R u = real_from<R>(curand_uniform_of(&px)), v = real_from<R>(curand_uniform_of(&px));
        const ray_t<R> r = W.d_camera.get_ray(u, v, &px);
        hit_record_t<R> rec;
        if (W.d_world.hit(r, real_from<R>(0.001f), real_from<R>(FLT_MAX), rec)) {
            ++hits;
            vec3_t<R> att; ray_t<R> sc;
            (void)rec.mat_ptr->scatter(r, rec, att, sc, &px);
        }
    }
    printf("%s objects N=%d created=%d hits=%d\n", tag, n, W.created, hits);
}

static void images() {
    const int nx = 37, ny = 11;                                            // ragged on purpose
    std::vector<float> fb((size_t)nx * ny * 3);
    for (size_t k = 0; k < fb.size(); ++k) fb[k] = (float)(k % 257) / 256.0f;
    fb[5] = std::numeric_limits<float>::quiet_NaN();                       // the reference's dielectric produces NaN pixels
    fb[7] = 1.0e30f; fb[8] = -3.0f;
    std::string s;
    ppm_text(nx, ny, fb.data(), RT_PRECISION_FP32, s);
    std::vector<uint16_t> hb(fb.size());
    for (size_t k = 0; k < fb.size(); ++k) hb[k] = half_t(fb[k]).bits;
    std::string sh;
    ppm_text(nx, ny, hb.data(), RT_PRECISION_FP16, sh);
    size_t bytes = 0;
    for (int fmt = RT_IMAGE_P6; fmt <= RT_IMAGE_PFM; ++fmt) {
        FILE* f = tmpfile();
        if (!f) { printf("images tmpfile failed\n"); exit(3); }
        if (!write_binary_image(f, nx, ny, fb.data(), RT_PRECISION_FP32, fmt) || !write_binary_image(f, nx, ny, hb.data(), RT_PRECISION_FP16, fmt)) { printf("images short write\n"); exit(3); }
        bytes += (size_t)ftell(f);
        fclose(f);
    }
    printf("images p3=%zu p3_fp16=%zu binary=%zu\n", s.size(), sh.size(), bytes);
}

int main(int argc, char** argv) {
    const bool big = argc > 1 && !strcmp(argv[1], "big");
    world_and_tree<float>("fp32", 22, 30, 400, 225);
    world_and_tree<float>("fp32", 500, 30, 1200, 800);
    world_and_tree<float>("fp32", 10000, 32, 1200, 800);                   // C3
    world_and_tree<float>("fp32", 2000, 3, 1200, 800);                     // full buckets: the "leaf nodes are full" path
    world_and_tree<half_t>("fp16", 10000, 32, 1200, 800);                  // C4
    if (big) world_and_tree<float>("fp32", 100000, 320, 3840, 2160);       // C5
    world_objects<float>("fp32", 500);
    world_objects<half_t>("fp16", 500);
    images();
    return 0;
}
