// The library's G-buffer encoding: shared by the host dispatch compiler (pool formats, nrd::GetLibraryDesc) and the device sources (texel codecs).
#pragma once

// The G-buffer encoding is a BUILD configuration of the library, as in the reference (CMakeLists.txt:28-29 NRD_NORMAL_ENCODING 0..4 / NRD_ROUGHNESS_ENCODING 0..2, NRD.hlsli:298-309;
// nrd::GetLibraryDesc reports it): raytracingdenoiser_amd/build.py passes -DNRD_NORMAL_ENCODING / -DNRD_ROUGHNESS_ENCODING from the environment variables of the same names.
#define NRD_NORMAL_ENCODING_RGBA8_UNORM 0
#define NRD_NORMAL_ENCODING_RGBA8_SNORM 1
#define NRD_NORMAL_ENCODING_R10G10B10A2_UNORM 2 // supports material ID bits
#define NRD_NORMAL_ENCODING_RGBA16_UNORM 3
#define NRD_NORMAL_ENCODING_RGBA16_SNORM 4
#define NRD_ROUGHNESS_ENCODING_SQ_LINEAR 0   // linearRoughness * linearRoughness
#define NRD_ROUGHNESS_ENCODING_LINEAR 1      // linearRoughness
#define NRD_ROUGHNESS_ENCODING_SQRT_LINEAR 2 // sqrt( linearRoughness )
#ifndef NRD_NORMAL_ENCODING
#define NRD_NORMAL_ENCODING NRD_NORMAL_ENCODING_R10G10B10A2_UNORM
#endif
#ifndef NRD_ROUGHNESS_ENCODING
#define NRD_ROUGHNESS_ENCODING NRD_ROUGHNESS_ENCODING_LINEAR
#endif
#if NRD_NORMAL_ENCODING < 0 || NRD_NORMAL_ENCODING > 4 || NRD_ROUGHNESS_ENCODING < 0 || NRD_ROUGHNESS_ENCODING > 2
#error "NRD_NORMAL_ENCODING must be 0..4 and NRD_ROUGHNESS_ENCODING 0..2 (nrd::NormalEncoding / nrd::RoughnessEncoding)"
#endif
