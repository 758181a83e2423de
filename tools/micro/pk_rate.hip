// Issue rate of packed fp32 (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32) against the scalar forms on gfx950.
// Build: hipcc --offload-arch=gfx950 -O3 -o pk_rate tools/micro/pk_rate.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float s) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (MODE == 0) {            // 8 independent scalar fma chains
        float a[8]; for (int i = 0; i < 8; ++i) a[i] = (float)(t + i);
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) a[i] = __builtin_fmaf(a[i], s, 1.0f);
        }
        float r = 0; for (int i = 0; i < 8; ++i) r += a[i];
        out[t] = r;
    } else if (MODE == 1) {     // 8 independent packed fma chains (16 floats)
        v2f a[8]; for (int i = 0; i < 8; ++i) a[i] = (v2f){(float)(t + i), (float)(t - i)};
        const v2f ss = {s, s}, one = {1.0f, 1.0f};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) a[i] = __builtin_elementwise_fma(a[i], ss, one);
        }
        float r = 0; for (int i = 0; i < 8; ++i) r += a[i].x + a[i].y;
        out[t] = r;
    } else if (MODE == 2) {     // scalar mul + add (no fma), 8 chains
        float a[8]; for (int i = 0; i < 8; ++i) a[i] = (float)(t + i);
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) { float m; asm volatile("v_mul_f32 %0, %1, %2" : "=v"(m) : "v"(a[i]), "v"(s)); asm volatile("v_add_f32 %0, %1, 1.0" : "=v"(a[i]) : "v"(m)); }
        }
        float r = 0; for (int i = 0; i < 8; ++i) r += a[i];
        out[t] = r;
    } else if (MODE == 4) {     // ONE dependent scalar fma chain per lane: issue-to-issue latency of dependent VALU
        float a = (float)t;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) a = __builtin_fmaf(a, s, 1.0f);
        }
        out[t] = a;
    } else if (MODE == 5) {     // two independent chains
        float a = (float)t, b = (float)(t + 1);
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 4; ++i) { a = __builtin_fmaf(a, s, 1.0f); b = __builtin_fmaf(b, s, 1.0f); }
        }
        out[t] = a + b;
    } else {                    // packed mul + add, 8 chains
        v2f a[8]; for (int i = 0; i < 8; ++i) a[i] = (v2f){(float)(t + i), (float)(t - i)};
        const v2f ss = {s, s}, one = {1.0f, 1.0f};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) { v2f m; asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(m) : "v"(a[i]), "v"(ss)); asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(a[i]) : "v"(m), "v"(one)); }
        }
        float r = 0; for (int i = 0; i < 8; ++i) r += a[i].x + a[i].y;
        out[t] = r;
    }
}

template <int MODE> void run(const char* name, int ops_per_chain_step, int floats, int blocks = 256 * 4 * 2 /* 8 waves per SIMD */) {
    float* out; hipMalloc(&out, 256 * 4096 * 4);
    const int iters = 4096;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters, 0.999f);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters, 0.999f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double instrs = (double)blocks * 4 /*waves*/ * iters * 8 * ops_per_chain_step;
    const double flt = instrs * 64 * floats;
    printf("%-28s %8.3f ms  %7.1f G wave-instr/s  %7.2f T float-ops/s\n", name, ms, instrs / ms / 1e6, flt / ms / 1e9);
    hipFree(out);
}

int main() {
    run<0>("v_fma_f32", 1, 1);
    run<1>("v_pk_fma_f32", 1, 2);
    run<2>("v_mul_f32+v_add_f32", 2, 1);
    run<3>("v_pk_mul+v_pk_add", 2, 2);
    // dependent chains, waves per SIMD = blocks / 256 (one 256-thread block = one wave on each SIMD of a CU)
    for (int w = 1; w <= 8; ++w) { char nm[64]; snprintf(nm, 64, "1 dep chain, %d waves/SIMD", w); run<4>(nm, 1, 1, 256 * w); }
    for (int w = 1; w <= 4; ++w) { char nm[64]; snprintf(nm, 64, "2 dep chains, %d waves/SIMD", w); run<5>(nm, 1, 1, 256 * w); }
    return 0;
}
