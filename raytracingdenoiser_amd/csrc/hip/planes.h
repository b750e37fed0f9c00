// Pitched HBM planes ("textures" of the reference become linear surfaces with a 256-byte-aligned row pitch) and the
// texel codecs for every storage format the hot path uses. Quantisation on store is part of the contract
// (history is fp16, accumulation speeds are UNORM8, ...), so each codec is spelled out:
//   fp16   : round-to-nearest-even (v_cvt_f16_f32), denormals kept
//   UNORMn : store floor(saturate(x) * (2^n - 1) + 0.5), load u / (2^n - 1)
// Out-of-bounds semantics of the reference's texture units that the passes rely on are reproduced by the callers:
// "Load" outside the plane returns 0, stores outside are dropped, clamp-samplers clamp the texel index.
#pragma once

#include <hip/hip_fp16.h>
#include <hip/hip_runtime.h>

#include <cstdint>
#include <type_traits>

namespace nrdhip {

struct Plane {
    uint8_t* ptr;   // device pointer to texel (0, 0)
    uint32_t pitch; // bytes per row
    int32_t w, h;   // texels
};

#define NRD_D __device__ __forceinline__

// SGPR diet for kernels that bind dozens of planes (a Plane costs 5 SGPRs and the temporal passes bind 20-35 of them, which
// otherwise spills scalars into VGPR lanes): pool planes of one format share their pitch, and every full-resolution plane shares
// (w, h). The launcher verifies that with SameLayout / SameSize, the kernel overwrites the fields of its by-value Plane copies
// with those of one representative, and the redundant kernel-argument loads disappear.
NRD_D void ShareLayout(Plane& p, const Plane& ref) {
    p.pitch = ref.pitch;
    p.w = ref.w;
    p.h = ref.h;
}
NRD_D void ShareSize(Plane& p, const Plane& ref) {
    p.w = ref.w;
    p.h = ref.h;
}
inline bool SameSize(const Plane& a, const Plane& b) { return !a.ptr || !b.ptr || (a.w == b.w && a.h == b.h); }
inline bool SameLayout(const Plane& a, const Plane& b) { return !a.ptr || !b.ptr || (a.pitch == b.pitch && a.w == b.w && a.h == b.h); }

NRD_D bool InBounds(const Plane& p, int x, int y) { return (unsigned)x < (unsigned)p.w && (unsigned)y < (unsigned)p.h; }

// 32-bit byte offsets (planes are far below 4 GiB) from a 24-bit multiply (row index and pitch are both < 2^24): one full-rate
// v_mad_u32_u24 instead of a quarter-rate v_mul_lo_u32 or 64-bit multiply-adds per access
// NRD_EXPERIMENT_L1_RESIDENT (A/B builds only, tools/build_variant.py; never the product): every LOAD is redirected into a 128-byte window of two rows in the
// middle of its plane -- the same instruction stream plus two VALU operations per address, but every request is an L1 hit. Run on a scene whose planes
// are constant (bench.py --uniform) the redirected loads return the very values the real ones would, so the control flow is the same and the time of
// this build is the kernel's issue floor: arithmetic + address generation + L1-hit latency, no L2 / HBM.
#ifndef NRD_EXPERIMENT_L1_RESIDENT
#define NRD_EXPERIMENT_L1_RESIDENT 0
#endif
NRD_D uint32_t TexelOffset(const Plane& p, int x, int y, uint32_t bytesPerTexel, bool isLoad) {
    if (NRD_EXPERIMENT_L1_RESIDENT == 2 && isLoad)
        return 0u; // every load reads texel (0, 0): the compiler keeps ONE load per plane and the addressing disappears -- the arithmetic alone (a lower bound of the floor)
    if (NRD_EXPERIMENT_L1_RESIDENT && isLoad)
        return __umul24((uint32_t)(p.h >> 1) + ((uint32_t)y & 1u), p.pitch) + (((uint32_t)x * bytesPerTexel) & 127u);
    return __umul24((uint32_t)y, p.pitch) + (uint32_t)x * bytesPerTexel;
}
template <typename T>
NRD_D T* TexelPtr(const Plane& p, int x, int y) {
    return (T*)(p.ptr + TexelOffset(p, x, y, (uint32_t)sizeof(T), std::is_const<T>::value));
}

// k / c for a small non-negative integer k held in a float, bit-identical to the IEEE quotient but 3 VALU ops instead
// of the ~11 of a generic correctly rounded division: q0 = k * RN(1/c), r = fma(-q0, c, k) (exact), q = fma(r, RN(1/c), q0).
// Exhaustively verified for every numerator the codecs can produce (tests/test_numerics.py: c = 65535, 1023, 255, 63, 15, 3).
NRD_D float DivSmallIntByConst(float k, float c, float rcpC) {
    float q0 = k * rcpC;
    float r = __builtin_fmaf(-q0, c, k);
    return __builtin_fmaf(r, rcpC, q0);
}
#define NRD_DIV_1023(k) DivSmallIntByConst(k, 1023.0f, 0.0009775171056389809f)
#define NRD_DIV_255(k) DivSmallIntByConst(k, 255.0f, 0.003921568859368563f)
#define NRD_DIV_32767(k) DivSmallIntByConst(k, 32767.0f, 3.0518509447574615e-05f) // also exact for the negative numerators of SNORM16
#define NRD_DIV_65535(k) DivSmallIntByConst(k, 65535.0f, 1.5259021893143654e-05f)
#define NRD_DIV_127(k) DivSmallIntByConst(k, 127.0f, 0.007874015718698502f) // SNORM8: numerators -128..127
#define NRD_DIV_63(k) DivSmallIntByConst(k, 63.0f, 0.01587301678955555f)
#define NRD_DIV_15(k) DivSmallIntByConst(k, 15.0f, 0.06666667014360428f)
#define NRD_DIV_3(k) DivSmallIntByConst(k, 3.0f, 0.3333333432674408f)

// ---- LDS texels -------------------------------------------------------------------------------------------------
// A float4 LDS texel of which the code uses three components (or three here and one there) is read by the compiler as ds_read_b96 (+ ds_read_b32). On
// gfx950 that costs 8 LDS cycles per wave (b96: eight lane groups) plus 8 for the b32 -- a dword read at a 16-byte lane stride hits 8 of the 32 banks, a
// 4-way conflict -- where ONE ds_read_b128 costs 4 (MI355X_MICROARCH.md "LDS": banking is per instruction). RELAX HistoryClamping was bound by exactly this:
// 102 b96 + 51 b32 reads per wave = 1 224 LDS cycles = 0.26 of its 0.28 ms at 4K and SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 24 %
// (profiles/r03_e_relax_ds_sh_pmc5.txt). LdsFloat4 keeps all four components alive behind an empty asm, so the read stays one b128.
#ifndef NRD_LDS_WHOLE_TEXEL // (the CPU emulation of these sources under tests/emu defines it away)
#define NRD_LDS_WHOLE_TEXEL(v) asm("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w)) // not volatile: free to be scheduled, alive as soon as one component is used
#endif
NRD_D float4 LdsFloat4(const float4* p) {
    float4 v = *p;
    NRD_LDS_WHOLE_TEXEL(v);
    return v;
}

// ---- fp16 -------------------------------------------------------------------------------------------------------
NRD_D float HalfBitsToFloat(uint16_t h) { return __half2float(__ushort_as_half(h)); }
// The conversion is an opaque instruction on purpose: left to the compiler, "fp32 multiply -> convert" is fused into
// v_fma_mixlo_f16, which rounds the exact product ONCE (to fp16) instead of twice (fp32, then fp16). That is a different
// result whenever the fp32 product lands on an fp16 tie, and the numerics contract pins the two-step rounding.
#ifndef NRD_OPAQUE_CVT_F16 // (the CPU emulation of these sources under tests/emu substitutes the one instruction it cannot assemble)
#define NRD_OPAQUE_CVT_F16(h, f) asm("v_cvt_f16_f32 %0, %1" : "=v"(h) : "v"(f))
#endif
NRD_D uint16_t FloatToHalfBits(float f) {
    uint32_t h;
    NRD_OPAQUE_CVT_F16(h, f);
    return (uint16_t)h;
}

// ---- R32_SFLOAT / R32_UINT / R16_UINT / R8_UINT -------------------------------------------------------------------
NRD_D float LoadR32F(const Plane& p, int x, int y) { return *TexelPtr<const float>(p, x, y); }
NRD_D void StoreR32F(const Plane& p, int x, int y, float v) { *TexelPtr<float>(p, x, y) = v; }
NRD_D uint32_t LoadR32U(const Plane& p, int x, int y) { return *TexelPtr<const uint32_t>(p, x, y); }
NRD_D void StoreR32U(const Plane& p, int x, int y, uint32_t v) { *TexelPtr<uint32_t>(p, x, y) = v; }
NRD_D uint32_t LoadR16U(const Plane& p, int x, int y) { return *TexelPtr<const uint16_t>(p, x, y); }
NRD_D void StoreR16U(const Plane& p, int x, int y, uint32_t v) { *TexelPtr<uint16_t>(p, x, y) = (uint16_t)v; }
NRD_D uint32_t LoadR8U(const Plane& p, int x, int y) { return *TexelPtr<const uint8_t>(p, x, y); }
NRD_D void StoreR8U(const Plane& p, int x, int y, uint32_t v) { *TexelPtr<uint8_t>(p, x, y) = (uint8_t)v; }

// ---- R16_SFLOAT ---------------------------------------------------------------------------------------------------
NRD_D float LoadR16F(const Plane& p, int x, int y) { return HalfBitsToFloat(*TexelPtr<const uint16_t>(p, x, y)); }
NRD_D void StoreR16F(const Plane& p, int x, int y, float v) { *TexelPtr<uint16_t>(p, x, y) = FloatToHalfBits(v); }

// ---- RGBA16_SFLOAT (one 8-byte access per texel) ------------------------------------------------------------------
NRD_D float4 LoadRGBA16F(const Plane& p, int x, int y) {
    uint2 raw = *TexelPtr<const uint2>(p, x, y);
    float4 r;
    r.x = HalfBitsToFloat((uint16_t)(raw.x & 0xFFFFu));
    r.y = HalfBitsToFloat((uint16_t)(raw.x >> 16));
    r.z = HalfBitsToFloat((uint16_t)(raw.y & 0xFFFFu));
    r.w = HalfBitsToFloat((uint16_t)(raw.y >> 16));
    return r;
}
// two conversions and the packing in ONE instruction (gfx950's v_cvt_pk_f16_f32: each half rounded to nearest even exactly as v_cvt_f16_f32 does,
// tests/test_numerics.py); opaque for the same reason as above
#ifndef NRD_OPAQUE_CVT_PK_F16
#define NRD_OPAQUE_CVT_PK_F16(r, a, b) asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b))
#endif
NRD_D uint32_t FloatsToHalf2Bits(float lo, float hi) {
    uint32_t r;
    NRD_OPAQUE_CVT_PK_F16(r, lo, hi);
    return r;
}
NRD_D void StoreRGBA16F(const Plane& p, int x, int y, float4 v) {
    uint2 raw;
    raw.x = FloatsToHalf2Bits(v.x, v.y);
    raw.y = FloatsToHalf2Bits(v.z, v.w);
    *TexelPtr<uint2>(p, x, y) = raw;
}

// The same store with the non-temporal hint as a compile-time choice of the kernel (NT = true: "global_store_dwordx2 ... nt"). For passes whose output exceeds the caches -- a 4K
// frame writes 270-560 MB per pass, more than the L2 and the memory-side cache hold together -- a plain store only evicts what the gathers of the same pass re-read; below that size
// the next pass finds its input cached and the hint would throw it away, so the LAUNCHER picks the instantiation by frame size (NRD_NT_STORE_PIXELS; profiles/HISTORY.md, round 6:
// a run-time branch in the store helpers cost the kernels that never take it 4 %, and LLVM merges a hinted and a plain store in two arms of a branch into one plain store).
#ifndef NRD_NT_STORE_PIXELS
#define NRD_NT_STORE_PIXELS 6000000u
#endif
template <bool NT>
NRD_D void StoreRGBA16FHinted(const Plane& p, int x, int y, float4 v) {
    if constexpr (NT) {
        typedef uint32_t V2 __attribute__((ext_vector_type(2)));
        V2 raw;
        raw.x = FloatsToHalf2Bits(v.x, v.y);
        raw.y = FloatsToHalf2Bits(v.z, v.w);
        __builtin_nontemporal_store(raw, (V2*)TexelPtr<uint2>(p, x, y));
    } else {
        StoreRGBA16F(p, x, y, v);
    }
}

// ---- RGBA32_SFLOAT --------------------------------------------------------------------------------------------------
NRD_D float4 LoadRGBA32F(const Plane& p, int x, int y) { return *TexelPtr<const float4>(p, x, y); }
NRD_D void StoreRGBA32F(const Plane& p, int x, int y, float4 v) { *TexelPtr<float4>(p, x, y) = v; }

// ---- UNORM --------------------------------------------------------------------------------------------------------
NRD_D float Saturate(float x) { return fminf(fmaxf(x, 0.0f), 1.0f); }
NRD_D uint32_t ToUnorm(float x, float maxValue) { return (uint32_t)floorf(Saturate(x) * maxValue + 0.5f); }

NRD_D float LoadR8Unorm(const Plane& p, int x, int y) { return NRD_DIV_255(float(*TexelPtr<const uint8_t>(p, x, y))); }
NRD_D void StoreR8Unorm(const Plane& p, int x, int y, float v) { *TexelPtr<uint8_t>(p, x, y) = (uint8_t)ToUnorm(v, 255.0f); }

// R16_UNORM (signals and fast history of the REBLUR occlusion family)
NRD_D float LoadR16Unorm(const Plane& p, int x, int y) { return NRD_DIV_65535(float(*TexelPtr<const uint16_t>(p, x, y))); }
NRD_D void StoreR16Unorm(const Plane& p, int x, int y, float v) { *TexelPtr<uint16_t>(p, x, y) = (uint16_t)ToUnorm(v, 65535.0f); }

// RGBA16_SNORM (signal planes of REBLUR_DIFFUSE_DIRECTIONAL_OCCLUSION): decode max(i / 32767, -1); encode clamp, scale, round half away from 0
NRD_D float FromSnorm16(uint32_t bits16) { return fmaxf(NRD_DIV_32767(float((int32_t)(int16_t)bits16)), -1.0f); }
NRD_D uint32_t ToSnorm16(float x) {
    float c = fminf(fmaxf(x, -1.0f), 1.0f) * 32767.0f;
    int32_t i = c >= 0.0f ? (int32_t)floorf(c + 0.5f) : -(int32_t)floorf(-c + 0.5f);
    return (uint32_t)i & 0xFFFFu;
}
NRD_D float4 LoadRGBA16Snorm(const Plane& p, int x, int y) {
    uint2 raw = *TexelPtr<const uint2>(p, x, y);
    return make_float4(FromSnorm16(raw.x & 0xFFFFu), FromSnorm16(raw.x >> 16), FromSnorm16(raw.y & 0xFFFFu), FromSnorm16(raw.y >> 16));
}
NRD_D void StoreRGBA16Snorm(const Plane& p, int x, int y, float4 v) {
    uint2 raw;
    raw.x = ToSnorm16(v.x) | (ToSnorm16(v.y) << 16);
    raw.y = ToSnorm16(v.z) | (ToSnorm16(v.w) << 16);
    *TexelPtr<uint2>(p, x, y) = raw;
}

NRD_D float2 LoadRG8Unorm(const Plane& p, int x, int y) {
    uint32_t raw = *TexelPtr<const uint16_t>(p, x, y);
    return make_float2(NRD_DIV_255(float(raw & 0xFFu)), NRD_DIV_255(float(raw >> 8)));
}
NRD_D void StoreRG8Unorm(const Plane& p, int x, int y, float2 v) {
    *TexelPtr<uint16_t>(p, x, y) = (uint16_t)(ToUnorm(v.x, 255.0f) | (ToUnorm(v.y, 255.0f) << 8));
}

NRD_D float4 DecodeRGBA8Unorm(uint32_t raw) {
    return make_float4(NRD_DIV_255(float(raw & 0xFFu)), NRD_DIV_255(float((raw >> 8) & 0xFFu)), NRD_DIV_255(float((raw >> 16) & 0xFFu)), NRD_DIV_255(float(raw >> 24)));
}
NRD_D float4 LoadRGBA8Unorm(const Plane& p, int x, int y) { return DecodeRGBA8Unorm(*TexelPtr<const uint32_t>(p, x, y)); }
NRD_D void StoreRGBA8Unorm(const Plane& p, int x, int y, float4 v) {
    *TexelPtr<uint32_t>(p, x, y) = ToUnorm(v.x, 255.0f) | (ToUnorm(v.y, 255.0f) << 8) | (ToUnorm(v.z, 255.0f) << 16) | (ToUnorm(v.w, 255.0f) << 24);
}

NRD_D float4 DecodeR10G10B10A2(uint32_t raw) {
    float4 r;
    r.x = NRD_DIV_1023(float(raw & 0x3FFu));
    r.y = NRD_DIV_1023(float((raw >> 10) & 0x3FFu));
    r.z = NRD_DIV_1023(float((raw >> 20) & 0x3FFu));
    r.w = NRD_DIV_3(float(raw >> 30));
    return r;
}
NRD_D float4 LoadR10G10B10A2(const Plane& p, int x, int y) { return DecodeR10G10B10A2(*TexelPtr<const uint32_t>(p, x, y)); }

// the texel formats of the other normal encodings (nrdmath.h NRD_NORMAL_ENCODING): RGBA8_SNORM, RGBA16_UNORM, RGBA16_SNORM (decode max(i / (2^(n-1) - 1), -1)), RGBA16_SFLOAT
NRD_D float FromSnorm8(uint32_t bits8) { return fmaxf(NRD_DIV_127(float((int32_t)(int8_t)bits8)), -1.0f); }
NRD_D float4 DecodeRGBA8Snorm(uint32_t raw) { return make_float4(FromSnorm8(raw & 0xFFu), FromSnorm8((raw >> 8) & 0xFFu), FromSnorm8((raw >> 16) & 0xFFu), FromSnorm8(raw >> 24)); }
NRD_D float4 DecodeRGBA16Unorm(uint2 raw) {
    return make_float4(NRD_DIV_65535(float(raw.x & 0xFFFFu)), NRD_DIV_65535(float(raw.x >> 16)), NRD_DIV_65535(float(raw.y & 0xFFFFu)), NRD_DIV_65535(float(raw.y >> 16)));
}
NRD_D float4 DecodeRGBA16Snorm(uint2 raw) { return make_float4(FromSnorm16(raw.x & 0xFFFFu), FromSnorm16(raw.x >> 16), FromSnorm16(raw.y & 0xFFFFu), FromSnorm16(raw.y >> 16)); }
NRD_D float4 DecodeRGBA16Float(uint2 raw) {
    return make_float4(HalfBitsToFloat((uint16_t)(raw.x & 0xFFFFu)), HalfBitsToFloat((uint16_t)(raw.x >> 16)), HalfBitsToFloat((uint16_t)(raw.y & 0xFFFFu)), HalfBitsToFloat((uint16_t)(raw.y >> 16)));
}

} // namespace nrdhip
