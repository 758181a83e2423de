// rt_image.hpp — output_to_stream (main.cu:321-333) and its binary companions, host-only C++ (no HIP): the one formatter behind
// rt_format_ppm / rt_write_ppm / rt_write_image (csrc/rt_api.hip) — and compiled on its own, with the host scene code, under
// AddressSanitizer / UBSan by tests/test_host_sanitizers.py.
#pragma once
#include <cstdio>
#include <cstdint>
#include <string>
#include <vector>
#include <new>
#include "../csrc/rt_real.h"
#include "../../include/rt_amd.h"

namespace rt {

inline float image_channel(const void* fb, size_t k, int precision) {
    if (precision == RT_PRECISION_FP16) return half_bits_to_float(((const uint16_t*)fb)[k]);
    return ((const float*)fb)[k];
}

// the reference's quantisation: int(255.99 * c), a double multiply truncated (main.cu:327-329).  A NaN or out-of-range channel
// (the reference's dielectric produces NaN pixels) is undefined behaviour in the reference's cast; x86's cvttsd2si gives INT_MIN,
// which is what the reference's binary prints — stated here instead of left to the compiler.
inline int image_level(float c) {
    const double v = 255.99 * (double)c;
    if (!(v > -2147483649.0 && v < 2147483648.0)) return INT32_MIN;
    return static_cast<int>(v);
}

// ASCII P3, top row first, "r g b\n" per pixel
inline void ppm_text(int nx, int ny, const void* fb, int precision, std::string& s) {
    s.reserve((size_t)nx * ny * 12 + 32);
    s += "P3\n"; s += std::to_string(nx); s += ' '; s += std::to_string(ny); s += "\n255\n";
    char line[48];
    for (int j = ny - 1; j >= 0; j--) {
        for (int i = 0; i < nx; i++) {
            const size_t pixel_index = (size_t)j * nx + i;
            const int ir = image_level(image_channel(fb, pixel_index * 3 + 0, precision));
            const int ig = image_level(image_channel(fb, pixel_index * 3 + 1, precision));
            const int ib = image_level(image_channel(fb, pixel_index * 3 + 2, precision));
            const int len = snprintf(line, sizeof(line), "%d %d %d\n", ir, ig, ib);
            s.append(line, (size_t)len);
        }
    }
}

// RT_IMAGE_P6 / RT_IMAGE_PFM into an open file; false on a short write
inline bool write_binary_image(FILE* f, int nx, int ny, const void* fb, int precision, int format) {
    bool ok = true;
    if (format == RT_IMAGE_P6) {
        ok = fprintf(f, "P6\n%d %d\n255\n", nx, ny) > 0;
        std::vector<unsigned char> row((size_t)nx * 3);
        for (int j = ny - 1; j >= 0 && ok; j--) {
            for (int i = 0; i < nx * 3; i++) {
                const int v = image_level(image_channel(fb, (size_t)j * nx * 3 + i, precision));
                row[i] = (unsigned char)(v < 0 ? 0 : (v > 255 ? 255 : v));
            }
            ok = fwrite(row.data(), 1, row.size(), f) == row.size();
        }
    } else {
        ok = fprintf(f, "PF\n%d %d\n-1.0\n", nx, ny) > 0;
        std::vector<float> row((size_t)nx * 3);
        for (int j = 0; j < ny && ok; j++) {
            for (int i = 0; i < nx * 3; i++) row[i] = image_channel(fb, (size_t)j * nx * 3 + i, precision);
            ok = fwrite(row.data(), sizeof(float), row.size(), f) == row.size();
        }
    }
    return ok;
}

} // namespace rt
