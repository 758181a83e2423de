// main.cpp — host program of the MI355X render path; the counterpart of the reference's main() (main.cu:347-477).
//
// Same sequence, same stderr lines, same output modes (0 = PPM to stdout, 1 = none, 3 = output.ppm; mode 2, the
// OpenGL viewer, is out of scope), same error convention (message + exit code 99), but every step goes through the
// C-ABI of include/rt_amd.h.  The reference's compile-time knobs become optional trailing arguments:
//   rt_main [output_mode] [NUM_SPHERES] [nx] [ny] [ns] [USE_OCTREE 0|1] [SPHERES_PER_LEAF] [SPHERE_RADIUS] [USE_FP16 0|1] [BUILD_ON_GPU 0|1]
// with the reference's values as defaults (main.cu:22-24, :348-350; acceleration_structure.h:15).  BUILD_ON_GPU 1 replaces
// buildOctree + upload (main.cu:405-417) by rt_build_octree_gpu: the same tree, built on the device.
#include <iostream>
#include <string>
#include <vector>
#include <cstdlib>
#include <time.h>
#include <hip/hip_runtime.h>
#include "../../include/rt_amd.h"

// limited version of the reference's checkCudaErrors (main.cu:27-37)
#define checkHipErrors(val) check_hip((int)(val), #val, __FILE__, __LINE__)
static void check_hip(int result, char const* const func, const char* const file, int const line) {
    if (result) {
        std::cerr << "HIP error = " << static_cast<int>(result) << " (" << rt_error_string(result) << ") at " << file << ":" << line << " '" << func << "' \n";
        (void)hipDeviceReset();
        exit(99);
    }
}

int main(int argc, char** argv) {
    int output_mode = 0;                 // 0 = to stdout (default), 1 = disabled, 3 = to file
    int num_spheres = 8000;              // NUM_SPHERES
    int nx = 1200, ny = 800, ns = 10;
    int use_octree = 1;                  // USE_OCTREE
    int spheres_per_leaf = 30;           // SPHERES_PER_LEAF
    float sphere_radius = 0.1f;          // SPHERE_RADIUS
    int use_fp16 = 0;                    // USE_FP16
    int build_on_gpu = 0;
    const int tx = 8, ty = 8;
    if (argc > 1) output_mode = std::stoi(argv[1]);
    if (argc > 2) num_spheres = std::stoi(argv[2]);
    if (argc > 3) nx = std::stoi(argv[3]);
    if (argc > 4) ny = std::stoi(argv[4]);
    if (argc > 5) ns = std::stoi(argv[5]);
    if (argc > 6) use_octree = std::stoi(argv[6]);
    if (argc > 7) spheres_per_leaf = std::stoi(argv[7]);
    if (argc > 8) sphere_radius = std::stof(argv[8]);
    if (argc > 9) use_fp16 = std::stoi(argv[9]);
    if (argc > 10) build_on_gpu = std::stoi(argv[10]);
    const int precision = use_fp16 ? RT_PRECISION_FP16 : RT_PRECISION_FP32;

    std::cerr << "Rendering a " << nx << "x" << ny << " image with " << ns << " samples per pixel ";
    std::cerr << "in " << tx << "x" << ty << " blocks.\n";
    std::cerr << "Number of spheres: " << num_spheres << "\n";
    std::cerr << "Sphere radius: " << sphere_radius << "\n";
    std::cerr << (use_octree ? "Use octree: ON\n" : "Use octree: OFF\n");
    std::cerr << "Output mode: " << output_mode << "\n";

    checkHipErrors(rt_device_check(nullptr));
    const rt_partition whole = {0, 1};
    const size_t num_pixels = (size_t)nx * ny;
    const size_t fb_size = num_pixels * 3 * (use_fp16 ? 2 : 4);

    // allocate FB and random state
    void* fb = nullptr;
    checkHipErrors(hipMalloc(&fb, fb_size));
    rt_rand_state* d_rand_state = nullptr;
    checkHipErrors(hipMalloc(reinterpret_cast<void**>(&d_rand_state), num_pixels * sizeof(rt_rand_state)));

    // world RNG, world of hitables & the camera
    rt_rand_state rand_state2;
    checkHipErrors(rt_rand_init(&rand_state2));
    std::vector<rt_sphere> list(num_spheres);
    rt_camera camera;
    int created = 0;
    checkHipErrors(rt_create_world(list.data(), num_spheres, sphere_radius, &camera, nx, ny, &rand_state2, precision, &created));
    rt_world* d_world = nullptr;
    checkHipErrors(rt_world_create(list.data(), num_spheres, &camera, precision, &d_world));
    checkHipErrors(rt_world_upload(d_world));

    // build octree and upload it
    rt_octree* d_octree = nullptr;
    if (use_octree) {
        if (build_on_gpu) checkHipErrors(rt_build_octree_gpu(d_world, spheres_per_leaf, &d_octree, nullptr));
        else checkHipErrors(rt_build_octree(list.data(), num_spheres, spheres_per_leaf, precision, &d_octree));
        int dropped_full = 0, dropped_outside = 0;
        checkHipErrors(rt_octree_info(d_octree, nullptr, nullptr, nullptr, &dropped_full, &dropped_outside));
        if (dropped_full || dropped_outside)   // the reference prints one line per drop to stdout, corrupting mode 0; here: stderr, once
            std::cerr << "octree: " << dropped_full << " insertions dropped (leaf nodes full), " << dropped_outside << " outside the root box\n";
        checkHipErrors(rt_octree_upload(d_octree));
    }
    checkHipErrors(hipDeviceSynchronize());

    clock_t start, stop;
    start = clock();
    checkHipErrors(rt_render_init(nx, ny, d_rand_state, whole, nullptr));
    checkHipErrors(hipDeviceSynchronize());
    checkHipErrors(rt_render(fb, nx, ny, ns, d_world, d_rand_state, d_octree, whole, nullptr));
    checkHipErrors(hipDeviceSynchronize());
    stop = clock();
    const double timer_seconds = static_cast<double>(stop - start) / CLOCKS_PER_SEC;
    std::cerr << "took " << timer_seconds << " seconds.\n";

    if (output_mode == 0 || output_mode == 3) {
        std::vector<char> host(fb_size);
        checkHipErrors(hipMemcpy(host.data(), fb, fb_size, hipMemcpyDeviceToHost));
        checkHipErrors(rt_write_ppm(output_mode == 3 ? "output.ppm" : nullptr, nx, ny, host.data(), precision));
    }

    // clean up
    checkHipErrors(hipDeviceSynchronize());
    checkHipErrors(rt_free_octree(d_octree));
    checkHipErrors(rt_free_world(d_world));
    checkHipErrors(hipFree(d_rand_state));
    checkHipErrors(hipFree(fb));
    (void)hipDeviceReset();
    return 0;
}
