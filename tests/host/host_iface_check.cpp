// host_iface_check.cpp — test program (tests/test_host_interfaces.py compiles and runs it; not part of the product).
//
// Drives the callable host interfaces of dd2360-raytracing_amd/host/rt_scene.hpp exactly the way the reference's kernels
// drive theirs — render() / color() of main.cu:43-117 restated over camera::get_ray, hitable_list::hit and
// material::scatter — and writes the frame as raw floats, which the test compares with the CPU oracle bit for bit.
//   host_iface_check PRECISION(0 fp32 | 1 fp16) NUM_SPHERES NX NY NS  > frame.f32
#include <cstdio>
#include <cstdlib>
#include <cfloat>
#include <vector>
#include "../../dd2360-raytracing_amd/host/rt_scene.hpp"

using namespace rt;

template <class R> static vec3_t<R> color(const ray_t<R>& r, const hitable_t<R>& world, rt_rand_state* local_rand_state) {   // main.cu:43-75
    typedef vec3_t<R> vec3;
    ray_t<R> cur_ray = r;
    vec3 cur_attenuation(real_from_double<R>(1.0), real_from_double<R>(1.0), real_from_double<R>(1.0));
    for (int i = 0; i < 50; i++) {
        hit_record_t<R> rec;
        if (world.hit(cur_ray, real_from<R>(0.001f), real_from<R>(FLT_MAX), rec)) {
            ray_t<R> scattered;
            vec3 attenuation;
            if (rec.mat_ptr->scatter(cur_ray, rec, attenuation, scattered, local_rand_state)) {
                cur_attenuation = cur_attenuation * attenuation;
                cur_ray = scattered;
            } else {
                return vec3(real_from_double<R>(0.0), real_from_double<R>(0.0), real_from_double<R>(0.0));
            }
        } else {
            const vec3 unit_direction = unit_vector(cur_ray.direction());
            const R t = real_from<R>(0.5f) * (unit_direction.y() + real_from<R>(1.0f));
            const R omt = real_from<R>(1.0f - as_float(t));                                  // `1.0f - t` is a float subtraction
            const vec3 c = omt * vec3(real_from_double<R>(1.0), real_from_double<R>(1.0), real_from_double<R>(1.0)) + t * vec3(real_from_double<R>(0.5), real_from_double<R>(0.7), real_from_double<R>(1.0));
            return cur_attenuation * c;
        }
    }
    return vec3(real_from_double<R>(0.0), real_from_double<R>(0.0), real_from_double<R>(0.0));
}

template <class R> static int run(int n, int nx, int ny, int ns) {
    typedef vec3_t<R> vec3;
    rt_rand_state rs;
    xorwow::init(rs, 1984ull);
    world_t<R> W;
    create_world<R>(W, n, 0.1f, nx, ny, &rs);
    std::vector<float> fb((size_t)nx * ny * 3);
    for (int j = 0; j < ny; ++j)
        for (int i = 0; i < nx; ++i) {
            const int pixel_index = j * nx + i;
            rt_rand_state local_rand_state;
            xorwow::init(local_rand_state, 1984ull + (unsigned long long)pixel_index);          // render_init, main.cu:93
            vec3 col(real_from_int<R>(0), real_from_int<R>(0), real_from_int<R>(0));
            for (int s = 0; s < ns; s++) {
                const float du = curand_uniform_of(&local_rand_state);
                const R u = real_from<R>((float)i + du) / real_from_int<R>(nx);
                const float dv = curand_uniform_of(&local_rand_state);
                const R v = real_from<R>((float)j + dv) / real_from_int<R>(ny);
                const ray_t<R> r = W.d_camera.get_ray(u, v, &local_rand_state);
                col = col + color<R>(r, W.d_world, &local_rand_state);
            }
            const R k = real_from_double<R>(1.0 / (double)as_float(real_from_int<R>(ns)));     // vec3::operator/=: k = 1.0/t in double (vec3.h:137)
            col = k * col;
            for (int c = 0; c < 3; ++c) fb[(size_t)pixel_index * 3 + c] = as_float(sqrt_real(col.e[c]));
        }
    return fwrite(fb.data(), sizeof(float), fb.size(), stdout) == fb.size() ? 0 : 1;
}

int main(int argc, char** argv) {
    if (argc < 6) return 2;
    const int precision = atoi(argv[1]), n = atoi(argv[2]), nx = atoi(argv[3]), ny = atoi(argv[4]), ns = atoi(argv[5]);
    return precision ? run<half_t>(n, nx, ny, ns) : run<float>(n, nx, ny, ns);
}
