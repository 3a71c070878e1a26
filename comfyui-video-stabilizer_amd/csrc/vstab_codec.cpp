// vstab_codec.cpp -- host-side loops of the coded node-boundary transfers (vstab_xfer.hip): plain C++ for the host
// compiler, each loop in a baseline and an AVX2 build of the same source, chosen once at run time.
#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <cstring>

namespace {

// values -> bytes; false (output unspecified) unless every value has exactly the bits of float32(k) / 255.0f, k = 0 .. 255
#define VSTAB_ENCODE_BODY                                                                                       \
    uint32_t bad = 0;                                                                                           \
    for (size_t i = 0; i < n; i++) {                                                                            \
        const float v = src[i];                                                                                 \
        const bool in = (v >= 0.0f) & (v <= 1.0f);           /* NaN: false */                                   \
        const float c = in ? v : 0.0f;                                                                          \
        const int k = (int)(c * 255.0f + 0.5f);              /* 0 .. 255 */                                     \
        const float back = (float)k / 255.0f;                /* IEEE division, as numpy's `arr /= 255.0` */     \
        uint32_t vb, bb;                                                                                        \
        memcpy(&vb, &v, 4);                                                                                     \
        memcpy(&bb, &back, 4);                                                                                  \
        bad |= (vb ^ bb);                                    /* -0.0f, NaN, outside [0, 1], between two quotients */ \
        dst[i] = (unsigned char)k;                                                                              \
    }                                                                                                           \
    return bad == 0;

bool encode_base(const float* __restrict__ src, unsigned char* __restrict__ dst, size_t n) { VSTAB_ENCODE_BODY }
__attribute__((target("avx2"))) bool encode_avx2(const float* __restrict__ src, unsigned char* __restrict__ dst, size_t n) { VSTAB_ENCODE_BODY }

#define VSTAB_EXPAND_BODY \
    for (size_t i = 0; i < n; i++) dst[i] = src[i] ? 1.0f : 0.0f;

void expand_base(const unsigned char* __restrict__ src, float* __restrict__ dst, size_t n) { VSTAB_EXPAND_BODY }
__attribute__((target("avx2"))) void expand_avx2(const unsigned char* __restrict__ src, float* __restrict__ dst, size_t n) { VSTAB_EXPAND_BODY }

// VSTAB_CODEC_BASELINE=1 (tests): the baseline build of the loops on a machine that has AVX2
#define VSTAB_LEVELS_BODY \
    for (size_t i = 0; i < n; i++) dst[i] = lut[src[i]];

void levels_base(const unsigned char* __restrict__ src, float* __restrict__ dst, size_t n, const float* __restrict__ lut) { VSTAB_LEVELS_BODY }
__attribute__((target("avx2"))) void levels_avx2(const unsigned char* __restrict__ src, float* __restrict__ dst, size_t n, const float* __restrict__ lut) { VSTAB_LEVELS_BODY }

const bool have_avx2 = __builtin_cpu_supports("avx2") && !(getenv("VSTAB_CODEC_BASELINE") && atoi(getenv("VSTAB_CODEC_BASELINE")) != 0);

}  // namespace

// (C linkage so that the CPU test suite can call the two loops through ctypes; not part of include/vstab.h)
extern "C" bool vstab_host_encode_q8(const float* src, unsigned char* dst, size_t n) { return have_avx2 ? encode_avx2(src, dst, n) : encode_base(src, dst, n); }
extern "C" void vstab_host_expand_mask(const unsigned char* src, float* dst, size_t n) { have_avx2 ? expand_avx2(src, dst, n) : expand_base(src, dst, n); }
// bytes c -> lut[c] (a soft mask's levels: the float32 values 1 - c / S of the motion-blur warp, formed by the caller)
extern "C" void vstab_host_expand_levels(const unsigned char* src, float* dst, size_t n, const float* lut) { have_avx2 ? levels_avx2(src, dst, n, lut) : levels_base(src, dst, n, lut); }
