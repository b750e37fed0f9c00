// ORACLE/_ref -- TEST INFRASTRUCTURE. Host-side stand-in for MathLib's dual-language ml.hlsli ("parity unpinned", see ml.h beside it): the three functions the reference's host
// calls (InstanceImpl.cpp:339-349), restated with the conventions the shaders pin (rotator = ( cos, sin, -sin, cos ), Common.hlsli:465).
#pragma once
namespace Sequence {
// additive recurrence on 24 bits with the golden-ratio increment
inline float Weyl1D(float p, uint32_t n) {
    float t = p + float((n * 10368889u) & 0x00FFFFFFu) / 16777216.0f;
    return t - std::floor(t);
}
// NRD_MATHLIB_BAYER_REVERSEBITS (default 0): MathLib is not vendored in the reference tree, so Sequence::Bayer4x4ui is a restatement; the round-5 reviewer recalls a MathLib
// default ML_BAYER_REVERSEBITS that advances the dither index by ReverseBits4( frameIndex ) instead of frameIndex. Unverifiable here; 1 selects that alternative in every place the
// function is restated (product host + device, oracle, both MathLib stand-ins under oracle/ref) -- it changes the dither PHASE per frame, nothing else (INTEGRATION.md "MathLib").
#ifndef NRD_MATHLIB_BAYER_REVERSEBITS
#define NRD_MATHLIB_BAYER_REVERSEBITS 0
#endif
inline uint32_t BayerFrameOffset(uint32_t frameIndex) {
#if NRD_MATHLIB_BAYER_REVERSEBITS
    const uint32_t v = frameIndex & 0xFu;
    return ((v & 1u) << 3) | ((v & 2u) << 1) | ((v & 4u) >> 1) | ((v & 8u) >> 3); // ReverseBits4
#else
    return frameIndex;
#endif
}
inline uint32_t Bayer4x4ui(uint2 pos, uint32_t frameIndex) {
    uint32_t x = pos.x & 3u, y = pos.y & 3u;
    uint32_t a = 2068378560u * (1u - (x >> 1)) + 1500172770u * (x >> 1);
    uint32_t b = (y + ((x & 1u) << 2)) << 2;
    return ((a >> b) + BayerFrameOffset(frameIndex)) & 0xFu;
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
