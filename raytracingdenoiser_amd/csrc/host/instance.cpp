// Instance core of the host dispatch compiler. Behavioural contract: reference Source/InstanceImpl.cpp
// (Create :100-267, SetCommonSettings :269-473, SetDenoiserSettings :475-488, GetComputeDispatches :490-578,
// pipeline de-duplication :580-647, descriptor accounting :649-725, ping-pong :727-736, transient aliasing
// :773-803, grid sizing :805-862).
#include "instance.h"

#include <algorithm>
#include <cassert>
#include <new>

using namespace nrdhost;

namespace nrd {

// ------------------------------------------------------------------------------------------------ allocation
static void* DefaultAllocate(void*, size_t size, size_t alignment) {
    void* p = nullptr;
    if (alignment < sizeof(void*))
        alignment = sizeof(void*);
    return posix_memalign(&p, alignment, size ? size : 1) == 0 ? p : nullptr;
}
static void* DefaultReallocate(void*, void* memory, size_t size, size_t) { return realloc(memory, size); }
static void DefaultFree(void*, void* memory) { free(memory); }

void CheckAndSetDefaultAllocator(AllocationCallbacks& cb) {
    if (cb.Allocate != nullptr && cb.Free != nullptr)
        return;
    cb.Allocate = DefaultAllocate;
    cb.Reallocate = DefaultReallocate;
    cb.Free = DefaultFree;
    cb.userArg = nullptr;
}

static bool IsIntegerFormat(Format f) {
    switch (f) {
        case Format::R8_UINT:
        case Format::RG8_UINT:
        case Format::RGBA8_UINT:
        case Format::R16_UINT:
        case Format::RG16_UINT:
        case Format::RGBA16_UINT:
        case Format::R32_UINT:
        case Format::RG32_UINT:
        case Format::RGB32_UINT:
        case Format::RGBA32_UINT:
        case Format::R10_G10_B10_A2_UINT:
            return true;
        default:
            return false;
    }
}

static bool Contains(Identifier id, const Identifier* ids, uint32_t n) {
    for (uint32_t i = 0; i < n; i++)
        if (ids[i] == id)
            return true;
    return false;
}

// ------------------------------------------------------------------------------------------------ ctor / dtor
InstanceImpl::InstanceImpl(const AllocationCallbacks& cb)
    : m_Callbacks(cb)
    , m_DenoiserData(HostAllocator<DenoiserData>(cb))
    , m_PermanentPool(HostAllocator<TextureDesc>(cb))
    , m_TransientPool(HostAllocator<TextureDesc>(cb))
    , m_Resources(HostAllocator<ResourceDesc>(cb))
    , m_ClearResources(HostAllocator<ClearResource>(cb))
    , m_PingPongs(HostAllocator<PingPong>(cb))
    , m_ResourceRanges(HostAllocator<ResourceRangeDesc>(cb))
    , m_PipelineRangeOffset(HostAllocator<size_t>(cb))
    , m_Pipelines(HostAllocator<PipelineDesc>(cb))
    , m_Passes(HostAllocator<PassTemplate>(cb))
    , m_ActiveDispatches(HostAllocator<DispatchDesc>(cb))
    , m_IndexRemap(HostAllocator<uint16_t>(cb))
    , m_TransientAliases(HostAllocator<uint32_t>(cb))
    , m_Strings(HostAllocator<char*>(cb)) {
    m_ReferenceQuirks = ReferenceQuirksEnabled();
    // the arena is followed by one scratch block: a dispatch that no longer fits writes its constants there (so the fillers never see null) and the
    // whole GetComputeDispatches call fails
    m_ConstantDataUnaligned = (uint8_t*)cb.Allocate(cb.userArg, CONSTANT_DATA_SIZE + CONSTANT_SCRATCH_SIZE + 64, 64);
    m_ConstantData = (uint8_t*)(((uintptr_t)m_ConstantDataUnaligned + 63) & ~(uintptr_t)63);
    memset(m_ConstantData, 0, CONSTANT_DATA_SIZE + CONSTANT_SCRATCH_SIZE);
}

InstanceImpl::~InstanceImpl() {
    m_Callbacks.Free(m_Callbacks.userArg, m_ConstantDataUnaligned);
    for (char* s : m_Strings)
        m_Callbacks.Free(m_Callbacks.userArg, s);
}

const char* InstanceImpl::InternString(const char* s) {
    for (char* t : m_Strings)
        if (!strcmp(t, s))
            return t;
    size_t n = strlen(s) + 1;
    char* t = (char*)m_Callbacks.Allocate(m_Callbacks.userArg, n, 1);
    memcpy(t, s, n);
    m_Strings.push_back(t);
    return t;
}

// ------------------------------------------------------------------------------------------------ creation
Result InstanceImpl::Create(const InstanceCreationDesc& desc) {
    const LibraryDesc& lib = GetLibraryDesc();

    for (uint32_t i = 0; i < desc.denoisersNum; i++) {
        const DenoiserDesc& dd = desc.denoisers[i];

        bool supported = false;
        for (uint32_t j = 0; j < lib.supportedDenoisersNum; j++)
            supported |= lib.supportedDenoisers[j] == dd.denoiser;
        if (!supported)
            return Result::UNSUPPORTED;

        for (uint32_t j = 0; j < desc.denoisersNum; j++)
            if (i != j && desc.denoisers[j].identifier == dd.identifier)
                return Result::NON_UNIQUE_IDENTIFIER;

        m_PermanentPoolOffset = (uint16_t)m_PermanentPool.size();
        m_TransientPoolOffset = (uint16_t)m_TransientPool.size();
        m_IndexRemap.clear();

        DenoiserData data;
        data.desc = dd;
        data.dispatchOffset = m_Passes.size();
        data.pingPongOffset = m_PingPongs.size();

        const size_t firstResource = m_Resources.size();

        switch (dd.denoiser) {
            case Denoiser::REBLUR_DIFFUSE:
                Add_Reblur(data, true, false, false);
                break;
            case Denoiser::REBLUR_SPECULAR:
                Add_Reblur(data, false, true, false);
                break;
            case Denoiser::REBLUR_DIFFUSE_SPECULAR:
                Add_Reblur(data, true, true, false);
                break;
            case Denoiser::REBLUR_DIFFUSE_SH:
                Add_Reblur(data, true, false, true);
                break;
            case Denoiser::REBLUR_DIFFUSE_DIRECTIONAL_OCCLUSION:
                Add_Reblur(data, true, false, false, true);
                break;
            case Denoiser::REBLUR_SPECULAR_SH:
                Add_Reblur(data, false, true, true);
                break;
            case Denoiser::REBLUR_DIFFUSE_SPECULAR_SH:
                Add_Reblur(data, true, true, true);
                break;
            case Denoiser::REBLUR_DIFFUSE_OCCLUSION:
                Add_ReblurOcclusion(data, true, false);
                break;
            case Denoiser::REBLUR_SPECULAR_OCCLUSION:
                Add_ReblurOcclusion(data, false, true);
                break;
            case Denoiser::REBLUR_DIFFUSE_SPECULAR_OCCLUSION:
                Add_ReblurOcclusion(data, true, true);
                break;
            case Denoiser::RELAX_DIFFUSE:
            case Denoiser::RELAX_DIFFUSE_SH:
            case Denoiser::RELAX_SPECULAR:
            case Denoiser::RELAX_SPECULAR_SH:
            case Denoiser::RELAX_DIFFUSE_SPECULAR:
            case Denoiser::RELAX_DIFFUSE_SPECULAR_SH:
                Add_RelaxVariant(data);
                break;
            case Denoiser::SIGMA_SHADOW:
                Add_SigmaShadow(data, false);
                break;
            case Denoiser::SIGMA_SHADOW_TRANSLUCENCY:
                Add_SigmaShadow(data, true);
                break;
            case Denoiser::REFERENCE:
                Add_Reference(data);
                break;
            default:
                return Result::INVALID_ARGUMENT;
        }

        data.pingPongNum = m_PingPongs.size() - data.pingPongOffset;

        for (size_t p = data.dispatchOffset; p < m_Passes.size(); p++)
            m_Passes[p].identifier = dd.identifier;

        // Every plane some pass writes gets cleared on CLEAR_AND_RESTART -- each (type, index) once, plus the
        // ping-pong partner of a swapped plane (reference InstanceImpl.cpp:189-242).
        for (size_t r = firstResource; r < m_Resources.size(); r++) {
            const ResourceDesc& res = m_Resources[r];
            if (res.descriptorType != DescriptorType::STORAGE_TEXTURE || res.type == ResourceType::OUT_VALIDATION)
                continue;

            bool known = false;
            for (const ClearResource& c : m_ClearResources)
                known |= c.resource.descriptorType == res.descriptorType && c.resource.type == res.type && c.resource.indexInPool == res.indexInPool;
            if (known)
                continue;

            bool isInteger = false;
            uint16_t downsample = 1;
            if (res.type == ResourceType::PERMANENT_POOL || res.type == ResourceType::TRANSIENT_POOL) {
                const TextureDesc& t = res.type == ResourceType::PERMANENT_POOL ? m_PermanentPool[res.indexInPool] : m_TransientPool[res.indexInPool];
                isInteger = IsIntegerFormat(t.format);
                downsample = t.downsampleFactor;
            }
            m_ClearResources.push_back({dd.identifier, res, downsample, isInteger});

            for (size_t p = 0; p < data.pingPongNum; p++) {
                const PingPong& pp = m_PingPongs[data.pingPongOffset + p];
                if (pp.resourceIndex == r) {
                    ResourceDesc pong = {res.descriptorType, res.type, pp.indexInPoolToSwapWith};
                    m_ClearResources.push_back({dd.identifier, pong, downsample, isInteger});
                    break;
                }
            }
        }

        m_DenoiserData.push_back(data);
    }

    // The two clear passes come last (their single resource slot is patched per dispatch)
    m_ClearPassIndex[0] = m_Passes.size();
    BeginPass("Clear (f)");
    Out((uint16_t)0);
    EndPass("Clear_Float.cs", 16, 16, 0);

    m_ClearPassIndex[1] = m_Passes.size();
    BeginPass("Clear (ui)");
    Out((uint16_t)0);
    EndPass("Clear_Uint.cs", 16, 16, 0);

    FinalizeDesc();
    return Result::SUCCESS;
}

void InstanceImpl::AddTransient(Format format, uint16_t downsample) {
    // Reuse a compatible transient plane of an earlier denoiser that this denoiser has not claimed yet
    for (uint16_t i = 0; i < m_TransientPoolOffset; i++) {
        const TextureDesc& t = m_TransientPool[i];
        if (t.format != format || t.downsampleFactor != downsample)
            continue;
        if (std::find(m_IndexRemap.begin(), m_IndexRemap.end(), i) == m_IndexRemap.end()) {
            m_IndexRemap.push_back(i);
            return;
        }
    }
    m_IndexRemap.push_back((uint16_t)m_TransientPool.size());
    m_TransientPool.push_back({format, downsample});
}

void InstanceImpl::PushPlane(DescriptorType descriptorType, uint16_t localIndex, uint16_t swapWith) {
    ResourceType type = (ResourceType)localIndex;
    uint16_t globalIndex = 0;

    if (localIndex >= TRANSIENT_POOL_START) {
        type = ResourceType::TRANSIENT_POOL;
        globalIndex = m_IndexRemap[localIndex - TRANSIENT_POOL_START];
        if (swapWith != NO_SWAP) {
            assert(swapWith >= TRANSIENT_POOL_START);
            m_PingPongs.push_back({m_Resources.size(), m_IndexRemap[swapWith - TRANSIENT_POOL_START]});
        }
    } else if (localIndex >= PERMANENT_POOL_START) {
        type = ResourceType::PERMANENT_POOL;
        globalIndex = uint16_t(m_PermanentPoolOffset + localIndex - PERMANENT_POOL_START);
        if (swapWith != NO_SWAP) {
            assert(swapWith >= PERMANENT_POOL_START && swapWith < TRANSIENT_POOL_START);
            m_PingPongs.push_back({m_Resources.size(), uint16_t(m_PermanentPoolOffset + swapWith - PERMANENT_POOL_START)});
        }
    }

    m_Resources.push_back({descriptorType, type, globalIndex});
}

void InstanceImpl::EndPass(const char* pipeline, uint8_t groupW, uint8_t groupH, uint32_t constantSize, uint16_t downsampleFactor, uint16_t maxRepeats) {
    size_t pipelineIndex = 0;
    for (; pipelineIndex < m_Pipelines.size(); pipelineIndex++)
        if (!strcmp(m_Pipelines[pipelineIndex].shaderFileName, pipeline))
            break;

    if (pipelineIndex == m_Pipelines.size()) {
        PipelineDesc p = {};
        p.shaderFileName = InternString(pipeline);
        p.shaderEntryPointName = "main";
        p.hasConstantData = constantSize != 0;
        m_PipelineRangeOffset.push_back(m_ResourceRanges.size());

        for (int r = 0; r < 2; r++) {
            ResourceRangeDesc range = {};
            range.descriptorType = r == 0 ? DescriptorType::TEXTURE : DescriptorType::STORAGE_TEXTURE;
            for (size_t i = m_ResourceOffset; i < m_Resources.size(); i++)
                if (m_Resources[i].descriptorType == range.descriptorType)
                    range.descriptorsNum++;
            if (range.descriptorsNum) {
                m_ResourceRanges.push_back(range);
                p.resourceRangesNum++;
            }
        }
        m_Pipelines.push_back(p);
    }

    PassTemplate t = {};
    t.name = m_PassName;
    t.resourceOffset = m_ResourceOffset;
    t.resourcesNum = uint32_t(m_Resources.size() - m_ResourceOffset);
    t.constantBufferDataSize = constantSize;
    t.pipelineIndex = (uint16_t)pipelineIndex;
    t.downsampleFactor = downsampleFactor;
    t.maxRepeatsNum = maxRepeats;
    t.groupW = groupW;
    t.groupH = groupH;
    m_Passes.push_back(t);
}

void InstanceImpl::FinalizeDesc() {
    static const Sampler kSamplers[] = {Sampler::NEAREST_CLAMP, Sampler::LINEAR_CLAMP};

    m_Desc = {};
    m_Desc.constantBufferRegisterIndex = 0;
    m_Desc.constantBufferSpaceIndex = 0;
    m_Desc.samplers = kSamplers;
    m_Desc.samplersNum = 2;
    m_Desc.samplersSpaceIndex = 0;
    m_Desc.samplersBaseRegisterIndex = 0;
    m_Desc.resourcesSpaceIndex = 0;

    for (size_t i = 0; i < m_Pipelines.size(); i++)
        m_Pipelines[i].resourceRanges = m_ResourceRanges.data() + m_PipelineRangeOffset[i];

    m_Desc.pipelines = m_Pipelines.data();
    m_Desc.pipelinesNum = (uint32_t)m_Pipelines.size();
    m_Desc.permanentPool = m_PermanentPool.data();
    m_Desc.permanentPoolSize = (uint32_t)m_PermanentPool.size();
    m_Desc.transientPool = m_TransientPool.data();
    m_Desc.transientPoolSize = (uint32_t)m_TransientPool.size();

    // Descriptor accounting, kept so a graphics-API caller sizing heaps from it sees the reference numbers
    // (all three "spaces" are 0, so samplers are counted per set)
    DescriptorPoolDesc& dp = m_Desc.descriptorPoolDesc;
    for (const PassTemplate& t : m_Passes) {
        for (uint32_t i = 0; i < t.resourcesNum; i++) {
            if (m_Resources[t.resourceOffset + i].descriptorType == DescriptorType::TEXTURE)
                dp.texturesMaxNum += t.maxRepeatsNum;
            else
                dp.storageTexturesMaxNum += t.maxRepeatsNum;
        }
        dp.setsMaxNum += t.maxRepeatsNum;
        dp.samplersMaxNum += t.maxRepeatsNum * m_Desc.samplersNum;
        if (t.constantBufferDataSize) {
            dp.constantBuffersMaxNum += t.maxRepeatsNum;
            m_Desc.constantBufferMaxDataSize = std::max(m_Desc.constantBufferMaxDataSize, t.constantBufferDataSize);
        }
    }
    uint32_t clearNum = (uint32_t)m_ClearResources.size();
    dp.storageTexturesMaxNum += clearNum;
    dp.setsMaxNum += clearNum;
    dp.samplersMaxNum += clearNum * m_Desc.samplersNum;
}

// ------------------------------------------------------------------------------------------------ settings
Result InstanceImpl::SetCommonSettings(const CommonSettings& commonSettings) {
    m_SplitScreenPrev = m_CommonSettings.splitScreen;
    memcpy(&m_CommonSettings, &commonSettings, sizeof(commonSettings));
    CommonSettings& cs = m_CommonSettings;

    // The very first frame always clears (reference :276-280)
    if (m_IsFirstUse) {
        cs.accumulationMode = AccumulationMode::CLEAR_AND_RESTART;
        m_IsFirstUse = false;
    }

    if (cs.accumulationMode != AccumulationMode::CONTINUE) {
        m_SplitScreenPrev = 0.0f;
        // NB: as in the reference, "prev" takes the values stored by the PREVIOUS SetCommonSettings call
        m_WorldToViewPrev = m_WorldToView;
        m_ViewToClipPrev = m_ViewToClip;
        for (int i = 0; i < 2; i++) {
            cs.resourceSizePrev[i] = cs.resourceSize[i];
            cs.rectSizePrev[i] = cs.rectSize[i];
            cs.cameraJitterPrev[i] = cs.cameraJitter[i];
        }
    }

    // Validation: same predicates as the reference asserts (:300-337); we report instead of aborting
    bool ok = cs.viewZScale > 0.0f;
    ok &= cs.resourceSize[0] != 0 && cs.resourceSize[1] != 0;
    ok &= cs.resourceSizePrev[0] != 0 && cs.resourceSizePrev[1] != 0;
    ok &= cs.rectSize[0] != 0 && cs.rectSize[1] != 0;
    ok &= cs.rectSizePrev[0] != 0 && cs.rectSizePrev[1] != 0;
    ok &= (cs.motionVectorScale[0] != 0.0f && cs.motionVectorScale[1] != 0.0f) || cs.isMotionVectorInWorldSpace;
    for (int i = 0; i < 2; i++) {
        ok &= cs.cameraJitter[i] >= -0.5f && cs.cameraJitter[i] <= 0.5f;
        ok &= cs.cameraJitterPrev[i] >= -0.5f && cs.cameraJitterPrev[i] <= 0.5f;
    }
    ok &= cs.denoisingRange > 0.0f;
    ok &= cs.disocclusionThreshold > 0.0f;
    ok &= cs.disocclusionThresholdAlternate > 0.0f;
    // encodings without material bits decode material 0 everywhere: a special material 0 would claim every pixel (reference InstanceImpl.cpp:333-337)
    ok &= cs.strandMaterialID != 0.0f || NRD_NORMAL_ENCODING == NRD_NORMAL_ENCODING_R10G10B10A2_UNORM;
    ok &= cs.cameraAttachedReflectionMaterialID != 0.0f || NRD_NORMAL_ENCODING == NRD_NORMAL_ENCODING_R10G10B10A2_UNORM;
    // (strand / camera-attached material id 0 is legal with the R10G10B10A2 encoding this build uses)

    // Per-frame kernel rotators (reference :339-349)
    float angle1 = Weyl1D(0.5f, cs.frameIndex) * Radians(90.0f);
    m_RotatorPre = GetRotator(angle1);

    float a0 = Weyl1D(0.0f, cs.frameIndex * 2) * Radians(90.0f);
    float a1 = Bayer4x4(0, 0, cs.frameIndex * 2) * Radians(360.0f);
    m_Rotator = CombineRotators(GetRotator(a0), GetRotator(a1));

    float a2 = Weyl1D(0.0f, cs.frameIndex * 2 + 1) * Radians(90.0f);
    float a3 = Bayer4x4(0, 0, cs.frameIndex * 2 + 1) * Radians(360.0f);
    m_RotatorPost = CombineRotators(GetRotator(a2), GetRotator(a3));

    // Matrices
    m_ViewToClip = Mat4::FromColumnMajor(cs.viewToClipMatrix);
    m_ViewToClipPrev = Mat4::FromColumnMajor(cs.viewToClipMatrixPrev);
    m_WorldToView = Mat4::FromColumnMajor(cs.worldToViewMatrix);
    m_WorldToViewPrev = Mat4::FromColumnMajor(cs.worldToViewMatrixPrev);
    m_WorldPrevToWorld = Mat4::FromColumnMajor(cs.worldPrevToWorldMatrix);

    // Everything downstream is left-handed: flip a right-handed camera (reference :392-408)
    ProjectionInfo info = DecomposeProjection(m_ViewToClip);
    if (!info.isLeftHanded) {
        for (int i = 0; i < 4; i++) {
            m_ViewToClip.c[2].v[i] = -m_ViewToClip.c[2].v[i];
            m_ViewToClipPrev.c[2].v[i] = -m_ViewToClipPrev.c[2].v[i];
        }
        for (int j = 0; j < 4; j++) { // negate ROW 2 of the view matrices (view-space z)
            m_WorldToView.at(2, j) = -m_WorldToView.at(2, j);
            m_WorldToViewPrev.at(2, j) = -m_WorldToViewPrev.at(2, j);
        }
    }

    m_ViewToWorld = InvertRigid(m_WorldToView);
    m_ViewToWorldPrev = InvertRigid(m_WorldToViewPrev);

    Vec3 camPos = {m_ViewToWorld.at(0, 3), m_ViewToWorld.at(1, 3), m_ViewToWorld.at(2, 3)};
    Vec3 camPosPrev = {m_ViewToWorldPrev.at(0, 3), m_ViewToWorldPrev.at(1, 3), m_ViewToWorldPrev.at(2, 3)};
    Vec3 delta = camPosPrev - camPos;

    // Camera-relative frames: current camera sits at the origin, the previous one at "delta" (reference :417-428)
    m_ViewToWorld.at(0, 3) = m_ViewToWorld.at(1, 3) = m_ViewToWorld.at(2, 3) = 0.0f;
    m_WorldToView = InvertRigid(m_ViewToWorld);

    m_ViewToWorldPrev.at(0, 3) = delta.x;
    m_ViewToWorldPrev.at(1, 3) = delta.y;
    m_ViewToWorldPrev.at(2, 3) = delta.z;
    m_WorldToViewPrev = InvertRigid(m_ViewToWorldPrev);

    m_WorldToClip = Mul(m_ViewToClip, m_WorldToView);
    m_WorldToClipPrev = Mul(m_ViewToClipPrev, m_WorldToViewPrev);
    m_ClipToWorldPrev = Invert(m_WorldToClipPrev);
    m_ClipToView = Invert(m_ViewToClip);
    m_ClipToViewPrev = Invert(m_ViewToClipPrev);
    m_ClipToWorld = Invert(m_WorldToClip);

    info = DecomposeProjection(m_ViewToClip);
    for (int i = 0; i < 4; i++)
        m_Frustum.v[i] = info.frustum[i];
    m_ProjectY = info.projectY;
    m_OrthoMode = info.isOrtho ? -1.0f : 0.0f;

    ProjectionInfo infoPrev = DecomposeProjection(m_ViewToClipPrev);
    for (int i = 0; i < 4; i++)
        m_FrustumPrev.v[i] = infoPrev.frustum[i];

    m_ViewDirection = {-m_ViewToWorld.at(0, 2), -m_ViewToWorld.at(1, 2), -m_ViewToWorld.at(2, 2)};
    m_ViewDirectionPrev = {-m_ViewToWorldPrev.at(0, 2), -m_ViewToWorldPrev.at(1, 2), -m_ViewToWorldPrev.at(2, 2)};
    m_CameraDelta = delta;

    // Frame time: explicit, else a smoothed wall clock (reference Source/Timer.cpp; non-deterministic, avoid)
    auto now = std::chrono::steady_clock::now();
    if (m_HasLastTime) {
        float ms = std::chrono::duration<float, std::milli>(now - m_LastTime).count();
        m_SmoothedDeltaMs = m_SmoothedDeltaMs + (ms - m_SmoothedDeltaMs) * 0.1f;
    }
    m_LastTime = now;
    m_HasLastTime = true;

    m_TimeDelta = cs.timeDeltaBetweenFrames > 0.0f ? cs.timeDeltaBetweenFrames : m_SmoothedDeltaMs;
    m_FrameRateScale = std::max(33.333f / m_TimeDelta, 1.0f);

    float dx = std::fabs(cs.cameraJitter[0] - cs.cameraJitterPrev[0]);
    float dy = std::fabs(cs.cameraJitter[1] - cs.cameraJitterPrev[1]);
    m_JitterDelta = std::max(dx, dy);

    float fps = m_FrameRateScale * 30.0f;
    float nonLinearAccumSpeed = fps * 0.25f / (1.0f + fps * 0.25f);
    m_CheckerboardResolveAccumSpeed = nonLinearAccumSpeed + (0.5f - nonLinearAccumSpeed) * m_JitterDelta;

    return ok ? Result::SUCCESS : Result::INVALID_ARGUMENT;
}

Result InstanceImpl::SetDenoiserSettings(Identifier identifier, const void* denoiserSettings) {
    for (DenoiserData& d : m_DenoiserData) {
        if (d.desc.identifier == identifier) {
            memcpy((void*)&d.settings, denoiserSettings, d.settingsSize);
            return Result::SUCCESS;
        }
    }
    return Result::INVALID_ARGUMENT;
}

// ------------------------------------------------------------------------------------------------ per-frame list
Result InstanceImpl::GetComputeDispatches(const Identifier* identifiers, uint32_t identifiersNum, const DispatchDesc*& dispatchDescs, uint32_t& dispatchDescsNum) {
    m_ConstantDataOffset = 0;
    m_ConstantOverflow = false;
    m_ActiveDispatches.clear();

    if (!identifiers || !identifiersNum) {
        dispatchDescs = nullptr;
        dispatchDescsNum = 0;
        return !identifiersNum ? Result::SUCCESS : Result::INVALID_ARGUMENT;
    }

    if (m_CommonSettings.accumulationMode == AccumulationMode::CLEAR_AND_RESTART) {
        for (const ClearResource& c : m_ClearResources) {
            if (!Contains(c.identifier, identifiers, identifiersNum))
                continue;

            const PassTemplate& t = m_Passes[m_ClearPassIndex[c.isInteger ? 1 : 0]];
            uint16_t w = DivideUp(m_CommonSettings.resourceSize[0], c.downsampleFactor);
            uint16_t h = DivideUp(m_CommonSettings.resourceSize[1], c.downsampleFactor);

            DispatchDesc d = {};
            d.name = t.name;
            d.identifier = c.identifier;
            d.resources = &c.resource;
            d.resourcesNum = 1;
            d.pipelineIndex = t.pipelineIndex;
            d.gridWidth = DivideUp(w, t.groupW);
            d.gridHeight = DivideUp(h, t.groupH);
            m_ActiveDispatches.push_back(d);
        }
    }

    for (const DenoiserData& d : m_DenoiserData) {
        if (!Contains(d.desc.identifier, identifiers, identifiersNum))
            continue;

        SwapPingPong(d);

        switch (d.desc.denoiser) {
            case Denoiser::REBLUR_DIFFUSE:
            case Denoiser::REBLUR_SPECULAR:
            case Denoiser::REBLUR_DIFFUSE_SPECULAR:
            case Denoiser::REBLUR_DIFFUSE_SH:
            case Denoiser::REBLUR_SPECULAR_SH:
            case Denoiser::REBLUR_DIFFUSE_SPECULAR_SH:
            case Denoiser::REBLUR_DIFFUSE_DIRECTIONAL_OCCLUSION:
                Update_Reblur(d);
                break;
            case Denoiser::REBLUR_DIFFUSE_OCCLUSION:
            case Denoiser::REBLUR_SPECULAR_OCCLUSION:
            case Denoiser::REBLUR_DIFFUSE_SPECULAR_OCCLUSION:
                Update_ReblurOcclusion(d);
                break;
            case Denoiser::RELAX_DIFFUSE:
            case Denoiser::RELAX_DIFFUSE_SH:
            case Denoiser::RELAX_SPECULAR:
            case Denoiser::RELAX_SPECULAR_SH:
            case Denoiser::RELAX_DIFFUSE_SPECULAR:
            case Denoiser::RELAX_DIFFUSE_SPECULAR_SH:
                Update_Relax(d);
                break;
            case Denoiser::SIGMA_SHADOW:
            case Denoiser::SIGMA_SHADOW_TRANSLUCENCY:
                Update_SigmaShadow(d);
                break;
            case Denoiser::REFERENCE:
                Update_Reference(d);
                break;
            default:
                break;
        }
    }

    for (size_t i = 1; i < m_ActiveDispatches.size(); i++) {
        const DispatchDesc& prev = m_ActiveDispatches[i - 1];
        DispatchDesc& cur = m_ActiveDispatches[i];
        if (prev.constantBufferDataSize == cur.constantBufferDataSize && cur.constantBufferDataSize != 0 && prev.constantBufferData && cur.constantBufferData) {
            if (!memcmp(prev.constantBufferData, cur.constantBufferData, cur.constantBufferDataSize))
                cur.constantBufferDataMatchesPreviousDispatch = true;
        } else if (prev.constantBufferDataSize == 0 && cur.constantBufferDataSize == 0)
            cur.constantBufferDataMatchesPreviousDispatch = true; // memcmp of 0 bytes "matches" in the reference too
    }

    if (m_ConstantOverflow) { // more constant data than the arena holds: hand out nothing rather than dispatches sharing the scratch block
        m_ActiveDispatches.clear();
        dispatchDescs = nullptr;
        dispatchDescsNum = 0;
        return Result::FAILURE;
    }

    dispatchDescs = m_ActiveDispatches.data();
    dispatchDescsNum = (uint32_t)m_ActiveDispatches.size();
    return dispatchDescsNum ? Result::SUCCESS : Result::INVALID_ARGUMENT;
}

void InstanceImpl::SwapPingPong(const DenoiserData& d) {
    for (size_t i = 0; i < d.pingPongNum; i++) {
        PingPong& pp = m_PingPongs[d.pingPongOffset + i];
        std::swap(m_Resources[pp.resourceIndex].indexInPool, pp.indexInPoolToSwapWith);
    }
}

void* InstanceImpl::PushDispatch(const DenoiserData& d, uint32_t localIndex) {
    const PassTemplate& t = m_Passes[d.dispatchOffset + localIndex];

    DispatchDesc desc = {};
    desc.name = t.name;
    desc.identifier = t.identifier;
    desc.resources = m_Resources.data() + t.resourceOffset;
    desc.resourcesNum = t.resourcesNum;
    desc.pipelineIndex = t.pipelineIndex;

    if (m_ConstantDataOffset + t.constantBufferDataSize > CONSTANT_DATA_SIZE || t.constantBufferDataSize > CONSTANT_SCRATCH_SIZE) {
        m_ConstantOverflow = true;
        desc.constantBufferData = m_ConstantData + CONSTANT_DATA_SIZE; // scratch
    } else {
        desc.constantBufferData = m_ConstantData + m_ConstantDataOffset;
        m_ConstantDataOffset += (t.constantBufferDataSize + 15u) & ~15u;
    }
    desc.constantBufferDataSize = std::min<uint32_t>(t.constantBufferDataSize, (uint32_t)CONSTANT_SCRATCH_SIZE);
    memset((void*)desc.constantBufferData, 0, desc.constantBufferDataSize);

    uint16_t w = m_CommonSettings.rectSize[0];
    uint16_t h = m_CommonSettings.rectSize[1];
    uint16_t ds = t.downsampleFactor;
    if (ds == USE_MAX_DIMS) {
        w = std::max(w, m_CommonSettings.rectSizePrev[0]);
        h = std::max(h, m_CommonSettings.rectSizePrev[1]);
        ds = 1;
    } else if (ds == IGNORE_RS) {
        w = m_CommonSettings.resourceSize[0];
        h = m_CommonSettings.resourceSize[1];
        ds = 1;
    }
    w = DivideUp(w, ds);
    h = DivideUp(h, ds);
    desc.gridWidth = DivideUp(w, t.groupW);
    desc.gridHeight = DivideUp(h, t.groupH);

    m_ActiveDispatches.push_back(desc);
    return (void*)desc.constantBufferData;
}

} // namespace nrd
