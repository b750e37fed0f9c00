// ORACLE/_ref -- TEST INFRASTRUCTURE. Host-side stand-in for MathLib's dual-language ml.hlsli ("parity unpinned", see ml.h beside it): the three functions the reference's host
// calls (InstanceImpl.cpp:339-349), restated with the conventions the shaders pin (rotator = ( cos, sin, -sin, cos ), Common.hlsli:465).
#pragma once
namespace Sequence {
// additive recurrence on 24 bits with the golden-ratio increment
inline float Weyl1D(float p, uint32_t n) {
    float t = p + float((n * 10368889u) & 0x00FFFFFFu) / 16777216.0f;
    return t - std::floor(t);
}
inline uint32_t Bayer4x4ui(uint2 pos, uint32_t frameIndex) {
    uint32_t x = pos.x & 3u, y = pos.y & 3u;
    uint32_t a = 2068378560u * (1u - (x >> 1)) + 1500172770u * (x >> 1);
    uint32_t b = (y + ((x & 1u) << 2)) << 2;
    return ((a >> b) + frameIndex) & 0xFu;
}
inline float Bayer4x4(uint2 pos, uint32_t frameIndex) { return float(Bayer4x4ui(pos, frameIndex)) / 16.0f; } // RESULT: [0; 1) (round 5: i / 16, as csrc/host/hostmath.h)
} // namespace Sequence
namespace Geometry {
inline float4 GetRotator(float angle) {
    float ca = (float)std::cos((double)angle), sa = (float)std::sin((double)angle);
    return float4(ca, sa, -sa, ca);
}
// r1.xyxy * r2.xxzz + r1.zwzw * r2.yyww
inline float4 CombineRotators(const float4& r1, const float4& r2) {
    return float4(r1.x * r2.x + r1.z * r2.y, r1.y * r2.x + r1.w * r2.y, r1.x * r2.z + r1.z * r2.w, r1.y * r2.z + r1.w * r2.w);
}
} // namespace Geometry
