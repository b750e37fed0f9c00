// Descriptor types of the NRD C-ABI, layout-compatible with the reference Include/NRDDescs.h (v4.14).
// Enumerator ORDER and struct field ORDER are the ABI, so they are identical to the reference; comments are ours.
// sizeof checks at the bottom pin the layout (values measured from the reference headers with g++ 11).
#pragma once

#define NRD_DESCS_VERSION_MAJOR 4
#define NRD_DESCS_VERSION_MINOR 14

static_assert(NRD_VERSION_MAJOR == NRD_DESCS_VERSION_MAJOR && NRD_VERSION_MINOR == NRD_DESCS_VERSION_MINOR, "NRD.h / NRDDescs.h version mismatch");

namespace nrd {

typedef uint32_t Identifier;
struct Instance; // opaque

enum class Result : uint32_t { SUCCESS, FAILURE, INVALID_ARGUMENT, UNSUPPORTED, NON_UNIQUE_IDENTIFIER, MAX_NUM };

// Slots the application binds (inputs/outputs) plus the two pool selectors. reference NRDDescs.h:37-154
enum class ResourceType : uint32_t {
    // guides (non-noisy)
    IN_MV,                         // RGBA16f+ (3D world / 2.5D) or RG16f+ (2D), mv = prev - cur, non-jittered
    IN_NORMAL_ROUGHNESS,           // NRD_FrontEnd_PackNormalAndRoughness encoding
    IN_VIEWZ,                      // linear view depth
    IN_DIFF_CONFIDENCE,            // optional, 0-1
    IN_SPEC_CONFIDENCE,            // optional, 0-1
    IN_DISOCCLUSION_THRESHOLD_MIX, // optional, 0-1
    IN_BASECOLOR_METALNESS,        // optional
    // noisy signals
    IN_DIFF_RADIANCE_HITDIST,
    IN_SPEC_RADIANCE_HITDIST,
    IN_DIFF_HITDIST,
    IN_SPEC_HITDIST,
    IN_DIFF_DIRECTION_HITDIST,
    IN_DIFF_SH0,
    IN_DIFF_SH1,
    IN_SPEC_SH0,
    IN_SPEC_SH1,
    IN_PENUMBRA,
    IN_TRANSLUCENCY,
    IN_SIGNAL,
    // outputs
    OUT_DIFF_RADIANCE_HITDIST,
    OUT_SPEC_RADIANCE_HITDIST,
    OUT_DIFF_SH0,
    OUT_DIFF_SH1,
    OUT_SPEC_SH0,
    OUT_SPEC_SH1,
    OUT_DIFF_HITDIST,
    OUT_SPEC_HITDIST,
    OUT_DIFF_DIRECTION_HITDIST,
    OUT_SHADOW_TRANSLUCENCY, // also the SIGMA history when stabilization is on
    OUT_SIGNAL,
    OUT_VALIDATION,
    // pools
    TRANSIENT_POOL, // reusable by the app after denoising
    PERMANENT_POOL, // history, owned by NRD between frames
    MAX_NUM,
};

// reference NRDDescs.h:156-259
enum class Denoiser : uint32_t {
    REBLUR_DIFFUSE,
    REBLUR_DIFFUSE_OCCLUSION,
    REBLUR_DIFFUSE_SH,
    REBLUR_SPECULAR,
    REBLUR_SPECULAR_OCCLUSION,
    REBLUR_SPECULAR_SH,
    REBLUR_DIFFUSE_SPECULAR,
    REBLUR_DIFFUSE_SPECULAR_OCCLUSION,
    REBLUR_DIFFUSE_SPECULAR_SH,
    REBLUR_DIFFUSE_DIRECTIONAL_OCCLUSION,
    RELAX_DIFFUSE,
    RELAX_DIFFUSE_SH,
    RELAX_SPECULAR,
    RELAX_SPECULAR_SH,
    RELAX_DIFFUSE_SPECULAR,
    RELAX_DIFFUSE_SPECULAR_SH,
    SIGMA_SHADOW,
    SIGMA_SHADOW_TRANSLUCENCY,
    REFERENCE,
    MAX_NUM
};

// reference NRDDescs.h:261-319
enum class Format : uint32_t {
    R8_UNORM, R8_SNORM, R8_UINT, R8_SINT,
    RG8_UNORM, RG8_SNORM, RG8_UINT, RG8_SINT,
    RGBA8_UNORM, RGBA8_SNORM, RGBA8_UINT, RGBA8_SINT, RGBA8_SRGB,
    R16_UNORM, R16_SNORM, R16_UINT, R16_SINT, R16_SFLOAT,
    RG16_UNORM, RG16_SNORM, RG16_UINT, RG16_SINT, RG16_SFLOAT,
    RGBA16_UNORM, RGBA16_SNORM, RGBA16_UINT, RGBA16_SINT, RGBA16_SFLOAT,
    R32_UINT, R32_SINT, R32_SFLOAT,
    RG32_UINT, RG32_SINT, RG32_SFLOAT,
    RGB32_UINT, RGB32_SINT, RGB32_SFLOAT,
    RGBA32_UINT, RGBA32_SINT, RGBA32_SFLOAT,
    R10_G10_B10_A2_UNORM, R10_G10_B10_A2_UINT, R11_G11_B10_UFLOAT, R9_G9_B9_E5_UFLOAT,
    MAX_NUM
};

enum class DescriptorType : uint32_t { TEXTURE /*read-only*/, STORAGE_TEXTURE /*read-write*/, MAX_NUM };
enum class Sampler : uint32_t { NEAREST_CLAMP, LINEAR_CLAMP, MAX_NUM };

// Compile-time encodings of IN_NORMAL_ROUGHNESS (this build: R10_G10_B10_A2_UNORM + LINEAR, the reference defaults,
// reference CMakeLists.txt:29-30).
enum class NormalEncoding : uint8_t { RGBA8_UNORM, RGBA8_SNORM, R10_G10_B10_A2_UNORM, RGBA16_UNORM, RGBA16_SNORM, MAX_NUM };
enum class RoughnessEncoding : uint8_t { SQ_LINEAR, LINEAR, SQRT_LINEAR, MAX_NUM };

struct AllocationCallbacks {
    void* (*Allocate)(void* userArg, size_t size, size_t alignment);
    void* (*Reallocate)(void* userArg, void* memory, size_t size, size_t alignment);
    void (*Free)(void* userArg, void* memory);
    void* userArg;
};

struct SPIRVBindingOffsets {
    uint32_t samplerOffset;
    uint32_t textureOffset;
    uint32_t constantBufferOffset;
    uint32_t storageTextureAndBufferOffset;
};

struct LibraryDesc {
    SPIRVBindingOffsets spirvBindingOffsets;
    const Denoiser* supportedDenoisers;
    uint32_t supportedDenoisersNum;
    uint8_t versionMajor;
    uint8_t versionMinor;
    uint8_t versionBuild;
    NormalEncoding normalEncoding;
    RoughnessEncoding roughnessEncoding;
};

struct DenoiserDesc {
    Identifier identifier;
    Denoiser denoiser;
};

struct InstanceCreationDesc {
    AllocationCallbacks allocationCallbacks;
    const DenoiserDesc* denoisers;
    uint32_t denoisersNum;
};

struct TextureDesc {
    Format format;
    uint16_t downsampleFactor;
};

struct ResourceDesc {
    DescriptorType descriptorType;
    ResourceType type;
    uint16_t indexInPool;
};

struct ResourceRangeDesc {
    DescriptorType descriptorType;
    uint32_t baseRegisterIndex;
    uint32_t descriptorsNum;
};

struct ComputeShaderDesc {
    const void* bytecode;
    uint64_t size;
};

struct PipelineDesc {
    ComputeShaderDesc computeShaderDXBC;  // always empty in this build
    ComputeShaderDesc computeShaderDXIL;  // always empty in this build
    ComputeShaderDesc computeShaderSPIRV; // always empty in this build
    const char* shaderFileName;           // pass name; the HIP executor keys its kernel table on it
    const char* shaderEntryPointName;
    const ResourceRangeDesc* resourceRanges; // <= 2: TEXTURE inputs then STORAGE_TEXTURE outputs
    uint32_t resourceRangesNum;
    bool hasConstantData;
};

struct DescriptorPoolDesc {
    uint32_t setsMaxNum;
    uint32_t constantBuffersMaxNum;
    uint32_t samplersMaxNum;
    uint32_t texturesMaxNum;
    uint32_t storageTexturesMaxNum;
};

struct InstanceDesc {
    uint32_t constantBufferMaxDataSize;
    uint32_t constantBufferSpaceIndex;
    uint32_t constantBufferRegisterIndex;

    const Sampler* samplers;
    uint32_t samplersNum;
    uint32_t samplersSpaceIndex;
    uint32_t samplersBaseRegisterIndex;

    const PipelineDesc* pipelines;
    uint32_t pipelinesNum;
    uint32_t resourcesSpaceIndex;

    const TextureDesc* permanentPool;
    uint32_t permanentPoolSize;
    const TextureDesc* transientPool;
    uint32_t transientPoolSize;

    DescriptorPoolDesc descriptorPoolDesc;
};

struct DispatchDesc {
    const char* name;
    Identifier identifier;

    const ResourceDesc* resources; // inputs then outputs, in the pass's binding order
    uint32_t resourcesNum;

    const uint8_t* constantBufferData;
    uint32_t constantBufferDataSize;
    bool constantBufferDataMatchesPreviousDispatch;

    uint16_t pipelineIndex;
    uint16_t gridWidth;
    uint16_t gridHeight;
};

static_assert(sizeof(DispatchDesc) == 56, "DispatchDesc ABI");
static_assert(sizeof(InstanceDesc) == 104, "InstanceDesc ABI");
static_assert(sizeof(PipelineDesc) == 80, "PipelineDesc ABI");
static_assert(sizeof(ResourceDesc) == 12, "ResourceDesc ABI");
static_assert(sizeof(TextureDesc) == 8, "TextureDesc ABI");
static_assert(sizeof(LibraryDesc) == 40, "LibraryDesc ABI");

} // namespace nrd
