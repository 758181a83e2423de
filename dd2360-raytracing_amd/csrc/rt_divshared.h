// rt_divshared.h — float division by a divisor that serves several quotients (the binary16 walk's plane table: 27 quotients a
// ray, nine per divisor).  ONE definition, compiled into the kernels (rt_kernels_fp16.hip) and into the exhaustive check
// (tools/micro/div_shared.hip, run by tests/test_gpu_micro.py): the 2^32-pair proof covers the code the kernels run.
//
// The compiler's IEEE division is v_div_scale_f32 x 2, v_rcp_f32, two multiply-adds refining the reciprocal, five forming the
// quotient, v_div_fmas_f32, v_div_fixup_f32; for operands that came from binary16 (magnitudes in [2^-24, 65504], or 0 / inf /
// NaN) the scale factors are always 1, so the scaling drops out and the refined reciprocal can be shared: the same multiply-adds
// on the same values.  The check compares the two forms for all 2^32 pairs of binary16 operands on the GPU: no quotient
// differs, in float or rounded to binary16.
#pragma once
#include <hip/hip_runtime.h>

namespace rt {
struct DivBy { float d, r; };
static __device__ __forceinline__ DivBy div_prepare(float d) {
    const float r0 = __builtin_amdgcn_rcpf(d);
    const float e = __builtin_fmaf(-d, r0, 1.0f);
    return {d, __builtin_fmaf(e, r0, r0)};
}
static __device__ __forceinline__ float div_by(float n, const DivBy& D) {
    float q = n * D.r;
    float rem = __builtin_fmaf(-D.d, q, n);
    q = __builtin_fmaf(rem, D.r, q);
    rem = __builtin_fmaf(-D.d, q, n);
    return __builtin_amdgcn_div_fixupf(__builtin_fmaf(rem, D.r, q), D.d, n);
}
} // namespace rt
