// rt_real.h — real_t for the product: float, or binary16 with one rounding per operator.
//
// Mirrors precision_types.h:16-160 of the reference (USE_FP16): storage is binary16, every operator computes
// on the float images and rounds the result once (round-to-nearest-even).  For + - * / and sqrt this is the same
// value as a native binary16 operation (24 >= 2*11+2 bits), so device code may use either form.
// Host and device share this header; only the two conversion primitives differ (integer code on the host,
// v_cvt_f16_f32 / v_cvt_f32_f16 on gfx950).
#pragma once
#include <stdint.h>
#include <string.h>
#include <math.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define RT_HD __host__ __device__ __forceinline__
#else
#define RT_HD inline
#endif

namespace rt {

RT_HD uint32_t bits_of(float f) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __float_as_uint(f);
#else
    uint32_t u; memcpy(&u, &f, 4); return u;
#endif
}
RT_HD float float_of(uint32_t u) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __uint_as_float(u);
#else
    float f; memcpy(&f, &u, 4); return f;
#endif
}

// float -> binary16 bits, round to nearest even
RT_HD uint16_t float_to_half_bits(float f) {
#if defined(__HIP_DEVICE_COMPILE__)
    const _Float16 h = (_Float16)f;
    return __builtin_bit_cast(uint16_t, h);
#else
    const uint32_t x = bits_of(f);
    const uint16_t sign = (uint16_t)((x >> 16) & 0x8000u);
    const uint32_t mag = x & 0x7fffffffu;
    if (mag > 0x7f800000u) return (uint16_t)(sign | 0x7e00u | ((mag >> 13) & 0x3ffu));   // NaN
    if (mag == 0x7f800000u) return (uint16_t)(sign | 0x7c00u);
    const int e = (int)(mag >> 23) - 127;
    if (e > 15) return (uint16_t)(sign | 0x7c00u);
    if (e >= -14) {
        const uint32_t m = mag & 0x7fffffu;
        uint32_t h = ((uint32_t)(e + 15) << 10) | (m >> 13);
        const uint32_t rem = m & 0x1fffu;
        if (rem > 0x1000u || (rem == 0x1000u && (h & 1u))) h++;          // may carry up to 0x7c00 = inf
        return (uint16_t)(sign | h);
    }
    if (e >= -25) {                                                     // binary16 subnormal
        const uint32_t m = (mag & 0x7fffffu) | 0x800000u;
        const int shift = -1 - e;                                       // 14..24
        uint32_t h = m >> shift;
        const uint32_t rem = m & ((1u << shift) - 1u);
        const uint32_t halfway = 1u << (shift - 1);
        if (rem > halfway || (rem == halfway && (h & 1u))) h++;
        return (uint16_t)(sign | h);
    }
    return sign;
#endif
}

RT_HD float half_bits_to_float(uint16_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
    return (float)__builtin_bit_cast(_Float16, b);
#else
    const uint32_t sign = ((uint32_t)b & 0x8000u) << 16;
    const uint32_t e = (b >> 10) & 0x1fu;
    uint32_t m = b & 0x3ffu;
    if (e == 0x1fu) return float_of(sign | 0x7f800000u | (m << 13));
    if (e != 0) return float_of(sign | ((e + 112u) << 23) | (m << 13));
    if (m == 0) return float_of(sign);
    int sh = 0;
    while (!(m & 0x400u)) { m <<= 1; ++sh; }                             // normalise the subnormal
    return float_of(sign | ((uint32_t)(113 - sh) << 23) | ((m & 0x3ffu) << 13));
#endif
}

// real_t of USE_FP16
struct half_t {
    uint16_t bits;
    RT_HD half_t() : bits(0) {}
    RT_HD explicit half_t(float f) : bits(float_to_half_bits(f)) {}
    RT_HD float f() const { return half_bits_to_float(bits); }
};
RT_HD half_t operator+(half_t a, half_t b) { return half_t(a.f() + b.f()); }
RT_HD half_t operator-(half_t a, half_t b) { return half_t(a.f() - b.f()); }
RT_HD half_t operator*(half_t a, half_t b) { return half_t(a.f() * b.f()); }
RT_HD half_t operator/(half_t a, half_t b) { return half_t(a.f() / b.f()); }
RT_HD bool operator<(half_t a, half_t b) { return a.f() < b.f(); }
RT_HD bool operator>(half_t a, half_t b) { return a.f() > b.f(); }
RT_HD bool operator<=(half_t a, half_t b) { return a.f() <= b.f(); }
RT_HD bool operator>=(half_t a, half_t b) { return a.f() >= b.f(); }

// conversions shared by both real types (the constructors of precision_types.h:21-25)
template <class R> struct real_ops;
template <> struct real_ops<float> {
    static RT_HD float from_float(float f) { return f; }
    static RT_HD float to_float(float r) { return r; }
};
template <> struct real_ops<half_t> {
    static RT_HD half_t from_float(float f) { return half_t(f); }
    static RT_HD float to_float(half_t r) { return r.f(); }
};
template <class R> RT_HD R real_from(float f) { return real_ops<R>::from_float(f); }
template <class R> RT_HD R real_from_double(double d) { return real_ops<R>::from_float((float)d); }
template <class R> RT_HD R real_from_int(int i) { return real_ops<R>::from_float((float)i); }
template <class R> RT_HD float as_float(R r) { return real_ops<R>::to_float(r); }

} // namespace rt
