// Issue rate of packed binary16 arithmetic (v_pk_mul_f16 / v_pk_add_f16 / v_pk_fma_f16 / v_pk_min_f16) on gfx950, next to v_fma_f32.
// Build: hipcc --offload-arch=gfx950 -O3 -o pk16_rate tools/micro/pk16_rate.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h2 __attribute__((ext_vector_type(2)));

template <int MODE, int CH>
__global__ __launch_bounds__(256) void k(float* out, int iters, float s) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (MODE == 0) {
        float a[CH]; for (int i = 0; i < CH; ++i) a[i] = (float)(t + i);
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < CH; ++i) a[i] = __builtin_fmaf(a[i], s, 1.0f);
        }
        float r = 0; for (int i = 0; i < CH; ++i) r += a[i];
        out[t] = r;
    } else {
        h2 a[CH]; for (int i = 0; i < CH; ++i) a[i] = (h2){(_Float16)(t & 7), (_Float16)(i)};
        const h2 ss = {(_Float16)s, (_Float16)s}, one = {(_Float16)1.0f, (_Float16)1.0f};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < CH; ++i) {
                if (MODE == 1) { h2 m; asm volatile("v_pk_mul_f16 %0, %1, %2" : "=v"(m) : "v"(a[i]), "v"(ss)); asm volatile("v_pk_add_f16 %0, %1, %2" : "=v"(a[i]) : "v"(m), "v"(one)); }
                if (MODE == 2) { asm volatile("v_pk_fma_f16 %0, %1, %2, %3" : "=v"(a[i]) : "v"(a[i]), "v"(ss), "v"(one)); asm volatile("v_pk_fma_f16 %0, %1, %2, %3" : "=v"(a[i]) : "v"(a[i]), "v"(ss), "v"(one)); }
                if (MODE == 3) { h2 m; asm volatile("v_pk_min_f16 %0, %1, %2" : "=v"(m) : "v"(a[i]), "v"(ss)); asm volatile("v_pk_add_f16 %0, %1, %2" : "=v"(a[i]) : "v"(m), "v"(one)); }
            }
        }
        float r = 0; for (int i = 0; i < CH; ++i) r += (float)a[i].x + (float)a[i].y;
        out[t] = r;
    }
}

template <int MODE, int CH> void run(const char* name, int ops, int blocks) {
    float* out; hipMalloc(&out, 256 * 8192 * 4);
    const int iters = 4096;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MODE, CH>), dim3(blocks), dim3(256), 0, 0, out, iters, 0.999f);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, CH>), dim3(blocks), dim3(256), 0, 0, out, iters, 0.999f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double instrs = (double)blocks * 4 * iters * CH * ops;
    printf("%-44s %8.3f ms  %7.1f G wave-instr/s\n", name, ms, instrs / ms / 1e6);
    hipFree(out);
}

int main() {
    for (int w : {4, 8}) {
        char nm[96];
        snprintf(nm, 96, "v_fma_f32, 8 chains, %d waves/SIMD", w); run<0, 8>(nm, 1, 256 * w);
        snprintf(nm, 96, "v_pk_mul_f16+v_pk_add_f16, 8 chains, %d w/SIMD", w); run<1, 8>(nm, 2, 256 * w);
        snprintf(nm, 96, "v_pk_fma_f16 x2, 8 chains, %d w/SIMD", w); run<2, 8>(nm, 2, 256 * w);
        snprintf(nm, 96, "v_pk_min_f16+v_pk_add_f16, 8 chains, %d w/SIMD", w); run<3, 8>(nm, 2, 256 * w);
        snprintf(nm, 96, "v_pk_mul_f16+v_pk_add_f16, 1 chain, %d w/SIMD", w); run<1, 1>(nm, 2, 256 * w);
        snprintf(nm, 96, "v_pk_mul_f16+v_pk_add_f16, 2 chains, %d w/SIMD", w); run<1, 2>(nm, 2, 256 * w);
        snprintf(nm, 96, "v_pk_mul_f16+v_pk_add_f16, 4 chains, %d w/SIMD", w); run<1, 4>(nm, 2, 256 * w);
    }
    return 0;
}
