// TEST INFRASTRUCTURE -- NOT PRODUCT CODE (see hip_runtime.h next to this file): the few fp16 helpers planes.h uses.
#pragma once
#include "hip_runtime.h"
struct __half {
    uint16_t bits;
};
inline __half __ushort_as_half(uint16_t b) { return __half{b}; }
inline uint16_t __half_as_ushort(__half h) { return h.bits; }
inline float __half2float(__half h) { return hwmath::F16BitsToF32(h.bits); }
inline __half __float2half_rn(float f) { return __half{hwmath::F32ToF16Bits(f)}; }
