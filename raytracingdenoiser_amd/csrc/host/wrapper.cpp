// C-ABI entry points (reference Source/Wrapper.cpp:125-303): thin forwards onto InstanceImpl.
#include "instance.h"

#include <new>

#define NRD_EXPORT extern "C" __attribute__((visibility("default")))

namespace {

// Denoisers this build implements end-to-end (host tables + HIP kernels). Everything else reports UNSUPPORTED from
// CreateInstance, exactly like a reference build asked for a denoiser it was compiled without.
const nrd::Denoiser g_Supported[] = { // in enum order, like the reference's list (Wrapper.cpp:23-44)
    nrd::Denoiser::REBLUR_DIFFUSE,
    nrd::Denoiser::REBLUR_DIFFUSE_OCCLUSION,
    nrd::Denoiser::REBLUR_DIFFUSE_SH,
    nrd::Denoiser::REBLUR_SPECULAR,
    nrd::Denoiser::REBLUR_SPECULAR_OCCLUSION,
    nrd::Denoiser::REBLUR_SPECULAR_SH,
    nrd::Denoiser::REBLUR_DIFFUSE_SPECULAR,
    nrd::Denoiser::REBLUR_DIFFUSE_SPECULAR_OCCLUSION,
    nrd::Denoiser::REBLUR_DIFFUSE_SPECULAR_SH,
    nrd::Denoiser::REBLUR_DIFFUSE_DIRECTIONAL_OCCLUSION,
    nrd::Denoiser::RELAX_DIFFUSE,
    nrd::Denoiser::RELAX_DIFFUSE_SH,
    nrd::Denoiser::RELAX_SPECULAR,
    nrd::Denoiser::RELAX_SPECULAR_SH,
    nrd::Denoiser::RELAX_DIFFUSE_SPECULAR,
    nrd::Denoiser::RELAX_DIFFUSE_SPECULAR_SH,
    nrd::Denoiser::SIGMA_SHADOW,
    nrd::Denoiser::SIGMA_SHADOW_TRANSLUCENCY,
    nrd::Denoiser::REFERENCE,
};

const nrd::LibraryDesc g_LibraryDesc = {
    {100, 200, 300, 400}, // SPIR-V binding offsets: meaningless here, kept at the reference values
    g_Supported,
    (uint32_t)(sizeof(g_Supported) / sizeof(g_Supported[0])),
    NRD_VERSION_MAJOR,
    NRD_VERSION_MINOR,
    NRD_VERSION_BUILD,
    (nrd::NormalEncoding)NRD_NORMAL_ENCODING, // the build's choice (csrc/common/encoding.h; reference Wrapper.cpp:54-55)
    (nrd::RoughnessEncoding)NRD_ROUGHNESS_ENCODING,
};
static_assert(NRD_NORMAL_ENCODING < (int)nrd::NormalEncoding::MAX_NUM && NRD_ROUGHNESS_ENCODING < (int)nrd::RoughnessEncoding::MAX_NUM, "encoding out of bounds");

// In ResourceType enum order (the reference table at Wrapper.cpp:58-95 is shifted for entries 3..15; ours is not)
const char* const g_ResourceTypeNames[] = {
    "IN_MV", "IN_NORMAL_ROUGHNESS", "IN_VIEWZ", "IN_DIFF_CONFIDENCE", "IN_SPEC_CONFIDENCE", "IN_DISOCCLUSION_THRESHOLD_MIX",
    "IN_BASECOLOR_METALNESS", "IN_DIFF_RADIANCE_HITDIST", "IN_SPEC_RADIANCE_HITDIST", "IN_DIFF_HITDIST", "IN_SPEC_HITDIST",
    "IN_DIFF_DIRECTION_HITDIST", "IN_DIFF_SH0", "IN_DIFF_SH1", "IN_SPEC_SH0", "IN_SPEC_SH1", "IN_PENUMBRA", "IN_TRANSLUCENCY",
    "IN_SIGNAL", "OUT_DIFF_RADIANCE_HITDIST", "OUT_SPEC_RADIANCE_HITDIST", "OUT_DIFF_SH0", "OUT_DIFF_SH1", "OUT_SPEC_SH0",
    "OUT_SPEC_SH1", "OUT_DIFF_HITDIST", "OUT_SPEC_HITDIST", "OUT_DIFF_DIRECTION_HITDIST", "OUT_SHADOW_TRANSLUCENCY", "OUT_SIGNAL",
    "OUT_VALIDATION", "TRANSIENT_POOL", "PERMANENT_POOL",
};
static_assert(sizeof(g_ResourceTypeNames) / sizeof(char*) == (size_t)nrd::ResourceType::MAX_NUM, "name table");

// the reference's own table (Source/Wrapper.cpp:58-95): entries 3..15 are out of step with the ResourceType enum (the confidence / threshold-mix / base-colour inputs were moved up in
// the enum, not in the table). Returned under NRD_HIP_REFERENCE_QUIRKS for callers that rely on what the reference prints.
const char* const g_ResourceTypeNamesOfTheReference[] = {
    "IN_MV", "IN_NORMAL_ROUGHNESS", "IN_VIEWZ", "IN_DIFF_RADIANCE_HITDIST", "IN_SPEC_RADIANCE_HITDIST", "IN_DIFF_HITDIST", "IN_SPEC_HITDIST", "IN_DIFF_DIRECTION_HITDIST",
    "IN_DIFF_SH0", "IN_DIFF_SH1", "IN_SPEC_SH0", "IN_SPEC_SH1", "IN_DIFF_CONFIDENCE", "IN_SPEC_CONFIDENCE", "IN_DISOCCLUSION_THRESHOLD_MIX", "IN_BASECOLOR_METALNESS",
    "IN_PENUMBRA", "IN_TRANSLUCENCY", "IN_SIGNAL", "OUT_DIFF_RADIANCE_HITDIST", "OUT_SPEC_RADIANCE_HITDIST", "OUT_DIFF_SH0", "OUT_DIFF_SH1", "OUT_SPEC_SH0",
    "OUT_SPEC_SH1", "OUT_DIFF_HITDIST", "OUT_SPEC_HITDIST", "OUT_DIFF_DIRECTION_HITDIST", "OUT_SHADOW_TRANSLUCENCY", "OUT_SIGNAL",
    "OUT_VALIDATION", "TRANSIENT_POOL", "PERMANENT_POOL",
};
static_assert(sizeof(g_ResourceTypeNamesOfTheReference) / sizeof(char*) == (size_t)nrd::ResourceType::MAX_NUM, "name table");

const char* const g_DenoiserNames[] = {
    "REBLUR_DIFFUSE", "REBLUR_DIFFUSE_OCCLUSION", "REBLUR_DIFFUSE_SH", "REBLUR_SPECULAR", "REBLUR_SPECULAR_OCCLUSION",
    "REBLUR_SPECULAR_SH", "REBLUR_DIFFUSE_SPECULAR", "REBLUR_DIFFUSE_SPECULAR_OCCLUSION", "REBLUR_DIFFUSE_SPECULAR_SH",
    "REBLUR_DIFFUSE_DIRECTIONAL_OCCLUSION", "RELAX_DIFFUSE", "RELAX_DIFFUSE_SH", "RELAX_SPECULAR", "RELAX_SPECULAR_SH",
    "RELAX_DIFFUSE_SPECULAR", "RELAX_DIFFUSE_SPECULAR_SH", "SIGMA_SHADOW", "SIGMA_SHADOW_TRANSLUCENCY", "REFERENCE",
};
static_assert(sizeof(g_DenoiserNames) / sizeof(char*) == (size_t)nrd::Denoiser::MAX_NUM, "name table");

} // namespace

NRD_EXPORT const nrd::LibraryDesc& NRD_CALL nrd::GetLibraryDesc() { return g_LibraryDesc; }

NRD_EXPORT nrd::Result NRD_CALL nrd::CreateInstance(const InstanceCreationDesc& instanceCreationDesc, Instance*& instance) {
    AllocationCallbacks cb = instanceCreationDesc.allocationCallbacks;
    CheckAndSetDefaultAllocator(cb);

    void* memory = cb.Allocate(cb.userArg, sizeof(InstanceImpl), alignof(InstanceImpl));
    if (!memory)
        return Result::FAILURE;

    InstanceImpl* impl = new (memory) InstanceImpl(cb);
    Result result = impl->Create(instanceCreationDesc);
    if (result == Result::SUCCESS) {
        instance = (Instance*)impl;
        return Result::SUCCESS;
    }

    impl->~InstanceImpl();
    cb.Free(cb.userArg, memory);
    return result;
}

NRD_EXPORT void NRD_CALL nrd::DestroyInstance(Instance& instance) {
    InstanceImpl* impl = (InstanceImpl*)&instance;
    AllocationCallbacks cb = impl->GetAllocationCallbacks();
    impl->~InstanceImpl();
    cb.Free(cb.userArg, impl);
}

NRD_EXPORT const nrd::InstanceDesc& NRD_CALL nrd::GetInstanceDesc(const Instance& instance) { return ((const InstanceImpl&)instance).GetDesc(); }

NRD_EXPORT nrd::Result NRD_CALL nrd::SetCommonSettings(Instance& instance, const CommonSettings& commonSettings) {
    return ((InstanceImpl&)instance).SetCommonSettings(commonSettings);
}

NRD_EXPORT nrd::Result NRD_CALL nrd::SetDenoiserSettings(Instance& instance, Identifier identifier, const void* denoiserSettings) {
    return ((InstanceImpl&)instance).SetDenoiserSettings(identifier, denoiserSettings);
}

NRD_EXPORT nrd::Result NRD_CALL nrd::GetComputeDispatches(Instance& instance, const Identifier* identifiers, uint32_t identifiersNum,
    const DispatchDesc*& dispatchDescs, uint32_t& dispatchDescsNum) {
    return ((InstanceImpl&)instance).GetComputeDispatches(identifiers, identifiersNum, dispatchDescs, dispatchDescsNum);
}

bool nrd::ReferenceQuirksEnabled() {
    const char* v = getenv("NRD_HIP_REFERENCE_QUIRKS");
    return v && atoi(v) != 0;
}

uint16_t nrd::TransientAliasOf(const Instance& instance, Identifier identifier, uint16_t indexInPool) { return ((const InstanceImpl&)instance).TransientAlias(identifier, indexInPool); }

NRD_EXPORT const char* nrd::GetResourceTypeString(ResourceType resourceType) {
    uint32_t i = (uint32_t)resourceType;
    if (i >= (uint32_t)ResourceType::MAX_NUM)
        return nullptr;
    return ReferenceQuirksEnabled() ? g_ResourceTypeNamesOfTheReference[i] : g_ResourceTypeNames[i];
}

NRD_EXPORT const char* nrd::GetDenoiserString(Denoiser denoiser) {
    uint32_t i = (uint32_t)denoiser;
    return i < (uint32_t)Denoiser::MAX_NUM ? g_DenoiserNames[i] : nullptr;
}
