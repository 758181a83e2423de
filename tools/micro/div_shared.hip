// tools/micro/div_shared.hip — is a float division with a SHARED, once-refined reciprocal of the divisor bit-identical to the compiler's
// IEEE division for every pair of binary16 operands?  (The binary16 walk forms 27 quotients a ray, nine per divisor: rt_kernels_fp16.hip,
// plane table.)  The compiler's sequence scales numerator and denominator (v_div_scale_f32) before the same five multiply-adds; for
// operands that came from binary16 — magnitudes in [2^-24, 65504] or 0 / inf / NaN — no scaling ever happens, so the scaling
// instructions and the per-quotient reciprocal can go.  This program checks ALL 65536 x 65536 pairs on the GPU, bits for bits
// (including zeros, subnormal binary16 values, infinities and NaNs — v_div_fixup_f32 decides those in both forms).
// hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-fast-math -fno-gpu-flush-denormals-to-zero -o div_shared div_shared.hip && ./div_shared
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include "../../dd2360-raytracing_amd/csrc/rt_divshared.h"      // the kernels' own div_prepare / div_by — not a copy
using rt::DivBy; using rt::div_prepare; using rt::div_by;
__device__ __forceinline__ float h2f(uint32_t bits) { _Float16 h = __builtin_bit_cast(_Float16, (uint16_t)bits); return (float)h; }
__global__ void k(unsigned long long* bad, unsigned long long* bad16, uint32_t* first) {
    const uint32_t db = blockIdx.x;                               // divisor: every binary16 bit pattern
    const float d = h2f(db);
    const DivBy D = div_prepare(d);
    unsigned long long nbad = 0, nbad16 = 0;
    for (uint32_t nb = threadIdx.x; nb < 65536u; nb += blockDim.x) {
        const float n = h2f(nb);
        const float a = n / d, b = div_by(n, D);
        const uint32_t ab = __float_as_uint(a), bb = __float_as_uint(b);
        if (ab != bb) { ++nbad; if (atomicCAS(first, 0xffffffffu, (nb << 16) | db) == 0xffffffffu) {} }
        const _Float16 ha = (_Float16)a, hb = (_Float16)b;        // what the walk keeps: the quotient rounded to binary16
        if (__builtin_bit_cast(uint16_t, ha) != __builtin_bit_cast(uint16_t, hb)) ++nbad16;
    }
    if (nbad) atomicAdd(bad, nbad);
    if (nbad16) atomicAdd(bad16, nbad16);
}
int main() {
    unsigned long long *bad, *bad16; uint32_t* first;
    (void)hipMalloc(&bad, 8); (void)hipMalloc(&bad16, 8); (void)hipMalloc(&first, 4);
    (void)hipMemset(bad, 0, 8); (void)hipMemset(bad16, 0, 8); (void)hipMemset(first, 0xff, 4);
    hipLaunchKernelGGL(k, dim3(65536), dim3(256), 0, 0, bad, bad16, first);
    unsigned long long h = 0, h16 = 0; uint32_t f = 0;
    if (hipMemcpy(&h, bad, 8, hipMemcpyDeviceToHost) != hipSuccess) { printf("launch failed\n"); return 2; }
    (void)hipMemcpy(&h16, bad16, 8, hipMemcpyDeviceToHost); (void)hipMemcpy(&f, first, 4, hipMemcpyDeviceToHost);
    printf("all 2^32 pairs of binary16 operands: %llu float quotients differ from the compiler's division, %llu after rounding to binary16", h, h16);
    if (h) printf(" (first: numerator bits 0x%04x, divisor bits 0x%04x)", f >> 16, f & 0xffffu);
    printf("\n");
    return h ? 1 : 0;
}
