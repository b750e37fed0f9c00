// ORACLE -- TEST INFRASTRUCTURE, NOT PRODUCT CODE (see oracle/hlsl.h).
//
// Texture-unit semantics the NRD shaders rely on, restated on host memory planes (SURVEY.md section 7 "hard parts"):
//   Load / operator[]    : out-of-bounds reads return 0
//   typed UAV store      : out-of-bounds writes are dropped; fp32 -> fp16 is round-to-nearest-even,
//                          UNORM is floor(saturate(x) * max + 0.5)
//   gNearestClamp sample : texel = clamp(floor(uv * size), 0, size - 1)
//   gLinearClamp sample  : bilinear about uv * size - 0.5 with edge clamp, weights in fp32, fixed evaluation order
//   Gather* at an integer texel corner + offset: the 2x2 quad, returned here already in (0,0)(1,0)(0,1)(1,1) order
#pragma once

#include "hlsl.h"

namespace orc {

// numeric values of nrd::Format (include/NRDDescs.h)
enum Fmt : uint32_t {
    FMT_R8_UNORM = 0,
    FMT_R8_UINT = 2,
    FMT_RG8_UNORM = 4,
    FMT_RGBA8_UNORM = 8,
    FMT_RGBA8_SNORM = 9,
    FMT_R16_UNORM = 13,
    FMT_R16_UINT = 15,
    FMT_R16_SFLOAT = 17,
    FMT_RGBA16_UNORM = 23,
    FMT_RGBA16_SNORM = 24,
    FMT_RGBA16_SFLOAT = 27,
    FMT_R32_UINT = 28,
    FMT_R32_SFLOAT = 30,
    FMT_RGBA32_SFLOAT = 39,
    FMT_R10_G10_B10_A2_UNORM = 40,
};

// ---- software fp16 (IEEE binary16, RNE, denormals) -----------------------------------------------------------------
inline float f16tof32(uint32_t h) {
    uint32_t s = (h & 0x8000u) << 16, e = (h >> 10) & 0x1Fu, m = h & 0x3FFu;
    if (e == 0) {
        if (m == 0)
            return asfloat(s);
        float v = float(m) * (1.0f / 16777216.0f); // m * 2^-24
        return (h & 0x8000u) ? -v : v;
    }
    if (e == 31)
        return asfloat(s | 0x7F800000u | (m << 13));
    return asfloat(s | ((e + 112u) << 23) | (m << 13));
}

inline uint32_t f32tof16(float f) {
    uint32_t u = asuint(f), s = (u >> 16) & 0x8000u, a = u & 0x7FFFFFFFu;
    if (a >= 0x7F800000u)
        return s | 0x7C00u | ((a > 0x7F800000u) ? 0x200u : 0u);
    if (a >= 0x477FF000u) // >= 65520 rounds to infinity
        return s | 0x7C00u;
    if (a < 0x38800000u) { // below the smallest normal half: denormal (or zero)
        if (a < 0x33000000u) // < 2^-25
            return s;
        uint32_t m = (a & 0x007FFFFFu) | 0x00800000u;
        int shift = 113 - (int)(a >> 23) + 13; // bits to drop so that the result is in units of 2^-24
        uint32_t r = m >> shift, rem = m & ((1u << shift) - 1u), half = 1u << (shift - 1);
        if (rem > half || (rem == half && (r & 1u)))
            r++;
        return s | r;
    }
    uint32_t r = ((a >> 13) - (112u << 10)), rem = a & 0x1FFFu;
    if (rem > 0x1000u || (rem == 0x1000u && (r & 1u)))
        r++;
    return s | r;
}

inline uint32_t ToUnorm(float x, float maxValue) { return (uint32_t)floorf(saturate(x) * maxValue + 0.5f); }
// SNORM16: clamp to [-1, 1], scale by 32767, round half away from zero; decode = max(i / 32767, -1)
inline int32_t ToSnorm16(float x) {
    float c = min(max(x, -1.0f), 1.0f) * 32767.0f;
    return c >= 0.0f ? (int32_t)floorf(c + 0.5f) : -(int32_t)floorf(-c + 0.5f);
}
inline float FromSnorm16(int16_t v) { return max(float(v) / 32767.0f, -1.0f); }

// ---- plane view ----------------------------------------------------------------------------------------------------
struct Plane {
    uint8_t* data;
    uint32_t pitch;
    uint32_t format;
    uint16_t width, height;
};

struct Tex {
    Plane p;
    Tex() : p{} {}
    explicit Tex(const Plane& pl) : p(pl) {}

    int W() const { return p.width; }
    int H() const { return p.height; }
    bool In(int x, int y) const { return (unsigned)x < p.width && (unsigned)y < p.height; }
    const uint8_t* Row(int y) const { return p.data + (size_t)y * p.pitch; }
    uint8_t* Row(int y) { return p.data + (size_t)y * p.pitch; }

    // raw texel fetch, no bounds check
    float4 Fetch(int x, int y) const {
        const uint8_t* r = Row(y);
        switch (p.format) {
            case FMT_R32_SFLOAT:
                return float4(((const float*)r)[x], 0, 0, 0);
            case FMT_RGBA32_SFLOAT: {
                const float* f = (const float*)r + x * 4;
                return float4(f[0], f[1], f[2], f[3]);
            }
            case FMT_R16_SFLOAT:
                return float4(f16tof32(((const uint16_t*)r)[x]), 0, 0, 0);
            case FMT_R16_UNORM:
                return float4(float(((const uint16_t*)r)[x]) / 65535.0f, 0, 0, 0);
            case FMT_RGBA16_SFLOAT: {
                const uint16_t* h = (const uint16_t*)r + x * 4;
                return float4(f16tof32(h[0]), f16tof32(h[1]), f16tof32(h[2]), f16tof32(h[3]));
            }
            case FMT_RGBA16_SNORM: {
                const int16_t* h = (const int16_t*)r + x * 4;
                return float4(FromSnorm16(h[0]), FromSnorm16(h[1]), FromSnorm16(h[2]), FromSnorm16(h[3]));
            }
            case FMT_R8_UNORM:
                return float4(float(r[x]) / 255.0f, 0, 0, 0);
            case FMT_RG8_UNORM:
                return float4(float(r[x * 2]) / 255.0f, float(r[x * 2 + 1]) / 255.0f, 0, 0);
            case FMT_RGBA8_UNORM:
                return float4(float(r[x * 4]) / 255.0f, float(r[x * 4 + 1]) / 255.0f, float(r[x * 4 + 2]) / 255.0f, float(r[x * 4 + 3]) / 255.0f);
            case FMT_RGBA8_SNORM: { // (IN_NORMAL_ROUGHNESS / PREV_NORMAL_ROUGHNESS of NRD_NORMAL_ENCODING 1: ml.h)
                const int8_t* b = (const int8_t*)r + x * 4;
                return float4(max(float(b[0]) / 127.0f, -1.0f), max(float(b[1]) / 127.0f, -1.0f), max(float(b[2]) / 127.0f, -1.0f), max(float(b[3]) / 127.0f, -1.0f));
            }
            case FMT_RGBA16_UNORM: { // (NRD_NORMAL_ENCODING 3)
                const uint16_t* h = (const uint16_t*)r + x * 4;
                return float4(float(h[0]) / 65535.0f, float(h[1]) / 65535.0f, float(h[2]) / 65535.0f, float(h[3]) / 65535.0f);
            }
            case FMT_R10_G10_B10_A2_UNORM: {
                uint32_t v = ((const uint32_t*)r)[x];
                return float4(float(v & 0x3FFu) / 1023.0f, float((v >> 10) & 0x3FFu) / 1023.0f, float((v >> 20) & 0x3FFu) / 1023.0f, float(v >> 30) / 3.0f);
            }
            default:
                return float4(0.0f);
        }
    }
    uint32_t FetchUint(int x, int y) const {
        const uint8_t* r = Row(y);
        switch (p.format) {
            case FMT_R32_UINT:
                return ((const uint32_t*)r)[x];
            case FMT_R16_UINT:
                return ((const uint16_t*)r)[x];
            case FMT_R8_UINT:
                return r[x];
            default:
                return 0;
        }
    }

    // Texture2D::Load / operator[] : zero outside
    float4 Load(int x, int y) const { return In(x, y) ? Fetch(x, y) : float4(0.0f); }
    float4 Load(int2 q) const { return Load(q.x, q.y); }
    uint32_t LoadUint(int x, int y) const { return In(x, y) ? FetchUint(x, y) : 0u; }

    // clamp-addressed fetches (gather taps, nearest / linear samplers)
    float4 FetchClamped(int x, int y) const { return Fetch(clamp(x, 0, W() - 1), clamp(y, 0, H() - 1)); }
    uint32_t FetchUintClamped(int x, int y) const { return FetchUint(clamp(x, 0, W() - 1), clamp(y, 0, H() - 1)); }

    // SampleLevel( gNearestClamp, uv, 0 )
    float4 SampleNearest(float2 uv) const { return FetchClamped((int)floorf(uv.x * float(W())), (int)floorf(uv.y * float(H()))); }

    // SampleLevel( gLinearClamp, ... ) with the position already in TEXEL units (pos = uv * size)
    float4 SampleLinearTexel(float2 pos) const {
        float tx = pos.x - 0.5f, ty = pos.y - 0.5f;
        float fx0 = floorf(tx), fy0 = floorf(ty);
        float fx = tx - fx0, fy = ty - fy0;
        int x0 = (int)fx0, y0 = (int)fy0;
        float4 s00 = FetchClamped(x0, y0), s10 = FetchClamped(x0 + 1, y0), s01 = FetchClamped(x0, y0 + 1), s11 = FetchClamped(x0 + 1, y0 + 1);
        float w00 = (1.0f - fx) * (1.0f - fy), w10 = fx * (1.0f - fy), w01 = (1.0f - fx) * fy, w11 = fx * fy;
        return s00 * w00 + s10 * w10 + s01 * w01 + s11 * w11;
    }
    // the same sample of a single-channel plane, in scalar arithmetic: NOT the .x of the above under -ffp-contract=on (the scalar sum is a chain of
    // fmas, the overloaded vector operators round every product first -- hlsl.h); the device's SampleLinearR16F is scalar
    float SampleLinearTexelScalar(float2 pos) const {
        float tx = pos.x - 0.5f, ty = pos.y - 0.5f;
        float fx0 = floorf(tx), fy0 = floorf(ty);
        float fx = tx - fx0, fy = ty - fy0;
        int x0 = (int)fx0, y0 = (int)fy0;
        float s00 = FetchClamped(x0, y0).x, s10 = FetchClamped(x0 + 1, y0).x, s01 = FetchClamped(x0, y0 + 1).x, s11 = FetchClamped(x0 + 1, y0 + 1).x;
        float w00 = (1.0f - fx) * (1.0f - fy), w10 = fx * (1.0f - fy), w01 = (1.0f - fx) * fy, w11 = fx * fy;
        return s00 * w00 + s10 * w10 + s01 * w01 + s11 * w11;
    }

    // typed stores
    void Store(int x, int y, float4 v) {
        if (!In(x, y))
            return;
        uint8_t* r = Row(y);
        switch (p.format) {
            case FMT_R32_SFLOAT:
                ((float*)r)[x] = v.x;
                break;
            case FMT_RGBA32_SFLOAT: {
                float* f = (float*)r + x * 4;
                f[0] = v.x, f[1] = v.y, f[2] = v.z, f[3] = v.w;
                break;
            }
            case FMT_R16_SFLOAT:
                ((uint16_t*)r)[x] = (uint16_t)f32tof16(v.x);
                break;
            case FMT_R16_UNORM:
                ((uint16_t*)r)[x] = (uint16_t)ToUnorm(v.x, 65535.0f);
                break;
            case FMT_RGBA16_SFLOAT: {
                uint16_t* h = (uint16_t*)r + x * 4;
                h[0] = (uint16_t)f32tof16(v.x), h[1] = (uint16_t)f32tof16(v.y), h[2] = (uint16_t)f32tof16(v.z), h[3] = (uint16_t)f32tof16(v.w);
                break;
            }
            case FMT_RGBA16_SNORM: {
                int16_t* h = (int16_t*)r + x * 4;
                h[0] = (int16_t)ToSnorm16(v.x), h[1] = (int16_t)ToSnorm16(v.y), h[2] = (int16_t)ToSnorm16(v.z), h[3] = (int16_t)ToSnorm16(v.w);
                break;
            }
            case FMT_R8_UNORM:
                r[x] = (uint8_t)ToUnorm(v.x, 255.0f);
                break;
            case FMT_RG8_UNORM:
                r[x * 2] = (uint8_t)ToUnorm(v.x, 255.0f), r[x * 2 + 1] = (uint8_t)ToUnorm(v.y, 255.0f);
                break;
            case FMT_RGBA8_UNORM:
                r[x * 4] = (uint8_t)ToUnorm(v.x, 255.0f), r[x * 4 + 1] = (uint8_t)ToUnorm(v.y, 255.0f), r[x * 4 + 2] = (uint8_t)ToUnorm(v.z, 255.0f),
                r[x * 4 + 3] = (uint8_t)ToUnorm(v.w, 255.0f);
                break;
            case FMT_RGBA8_SNORM: {
                int8_t* b = (int8_t*)r + x * 4;
                const float q[4] = {v.x, v.y, v.z, v.w};
                for (int k = 0; k < 4; k++) { // clamp, scale, round half away from zero (as ToSnorm16)
                    float t = min(max(q[k], -1.0f), 1.0f) * 127.0f;
                    b[k] = (int8_t)(t >= 0.0f ? (int32_t)floorf(t + 0.5f) : -(int32_t)floorf(-t + 0.5f));
                }
                break;
            }
            case FMT_RGBA16_UNORM: {
                uint16_t* h = (uint16_t*)r + x * 4;
                h[0] = (uint16_t)ToUnorm(v.x, 65535.0f), h[1] = (uint16_t)ToUnorm(v.y, 65535.0f), h[2] = (uint16_t)ToUnorm(v.z, 65535.0f), h[3] = (uint16_t)ToUnorm(v.w, 65535.0f);
                break;
            }
            case FMT_R10_G10_B10_A2_UNORM:
                ((uint32_t*)r)[x] = ToUnorm(v.x, 1023.0f) | (ToUnorm(v.y, 1023.0f) << 10) | (ToUnorm(v.z, 1023.0f) << 20) | (ToUnorm(v.w, 3.0f) << 30);
                break;
            default:
                break;
        }
    }
    void Store(int x, int y, float v) { Store(x, y, float4(v, 0, 0, 0)); }
    void StoreUint(int x, int y, uint32_t v) {
        if (!In(x, y))
            return;
        uint8_t* r = Row(y);
        switch (p.format) {
            case FMT_R32_UINT:
                ((uint32_t*)r)[x] = v;
                break;
            case FMT_R16_UINT:
                ((uint16_t*)r)[x] = (uint16_t)v;
                break;
            case FMT_R8_UINT:
                r[x] = (uint8_t)v;
                break;
            default:
                break;
        }
    }
    // raw texel copy between planes of the same format (e.g. PREV_NORMAL_ROUGHNESS <- IN_NORMAL_ROUGHNESS)
    void CopyTexelFrom(const Tex& src, int x, int y, uint32_t bytesPerTexel) {
        if (!In(x, y) || !src.In(x, y))
            return;
        memcpy(Row(y) + (size_t)x * bytesPerTexel, src.Row(y) + (size_t)x * bytesPerTexel, bytesPerTexel);
    }
};

} // namespace orc
