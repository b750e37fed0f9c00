// Host "dispatch compiler": turns (denoiser list, per-frame settings) into an ordered list of pass dispatches,
// each naming a pass (pipeline), its input/output planes and its constant block. Same data model as the reference
// (reference Source/InstanceImpl.h:98-352, InstanceImpl.cpp:100-862): two plane pools (permanent = history,
// transient = scratch, aliased between denoisers), ping-pong plane swapping, injected clears on CLEAR_AND_RESTART,
// permutation selection per frame. The list is consumed by the HIP executor (csrc/hip/executor.cpp) -- or by any
// other backend, which is how the test-suite drives the CPU oracle.
#pragma once

#include "NRD.h"

#include "../common/encoding.h"
#include "../common/pass_constants.h"
#include "hostmath.h"

#include <chrono>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace nrd {

// NRD_HIP_REFERENCE_QUIRKS=1 (environment, read when asked): reproduce what callers of the reference observe where this library deliberately differs -- the (shifted)
// GetResourceTypeString table of Source/Wrapper.cpp:58-95 and the 11 transient textures of REBLUR_DIFFUSE_SPECULAR_SH (INTEGRATION.md "Reference quirks")
bool ReferenceQuirksEnabled();
class InstanceImpl;
// index of the transient plane the HIP executor binds where a dispatch of denoiser `identifier` names `indexInPool` (the identity without the quirks switch)
uint16_t TransientAliasOf(const Instance& instance, Identifier identifier, uint16_t indexInPool);

// ---- allocation through the user callbacks (reference Source/StdAllocator.h) ------------------------------------
void CheckAndSetDefaultAllocator(AllocationCallbacks& cb);

template <typename T>
struct HostAllocator {
    typedef T value_type;
    AllocationCallbacks cb;

    explicit HostAllocator(const AllocationCallbacks& c) : cb(c) {}
    template <typename U>
    HostAllocator(const HostAllocator<U>& o) : cb(o.cb) {}

    T* allocate(size_t n) { return (T*)cb.Allocate(cb.userArg, n * sizeof(T), alignof(T) < 16 ? 16 : alignof(T)); }
    void deallocate(T* p, size_t) { cb.Free(cb.userArg, p); }
    template <typename U>
    bool operator==(const HostAllocator<U>&) const { return true; }
    template <typename U>
    bool operator!=(const HostAllocator<U>&) const { return false; }
};

template <typename T>
using Vector = std::vector<T, HostAllocator<T>>;

// ---- internal records -------------------------------------------------------------------------------------------
constexpr uint16_t PERMANENT_POOL_START = 1000; // local plane ids >= 1000 index the denoiser's permanent planes
constexpr uint16_t TRANSIENT_POOL_START = 2000; // local plane ids >= 2000 index the denoiser's transient planes
constexpr size_t CONSTANT_DATA_SIZE = 128 * 1024;
constexpr size_t CONSTANT_SCRATCH_SIZE = 4096; // >= the largest constant block (REBLUR: 832 bytes)
constexpr uint16_t USE_MAX_DIMS = 0xFFFF; // grid from max(rect, rectPrev)
constexpr uint16_t IGNORE_RS = 0xFFFE;    // grid from resourceSize
constexpr uint16_t NO_SWAP = 0xFFFF;

inline uint16_t DivideUp(uint32_t x, uint16_t y) { return uint16_t((x + y - 1) / y); }

union Settings {
    ReblurSettings reblur;
    RelaxSettings relax;
    SigmaSettings sigma;
    ReferenceSettings reference;
    Settings() { memset((void*)this, 0, sizeof(*this)); }
};

struct DenoiserData {
    DenoiserDesc desc = {};
    Settings settings;
    size_t settingsSize = 0;
    size_t dispatchOffset = 0;
    size_t pingPongOffset = 0;
    size_t pingPongNum = 0;
};

struct PingPong {
    size_t resourceIndex;
    uint16_t indexInPoolToSwapWith;
};

// One entry of the static pass table (all permutations are materialised at creation)
struct PassTemplate {
    const char* name;       // "<DENOISER> - <pass>"
    size_t resourceOffset;  // into m_Resources
    uint32_t resourcesNum;
    uint32_t constantBufferDataSize;
    Identifier identifier;
    uint16_t pipelineIndex;
    uint16_t downsampleFactor;
    uint16_t maxRepeatsNum;
    uint8_t groupW, groupH; // the reference shader's thread-group size: defines DispatchDesc::gridWidth/Height
};

struct ClearResource {
    Identifier identifier;
    ResourceDesc resource;
    uint16_t downsampleFactor;
    bool isInteger;
};

bool IsRelax(Denoiser d);

class InstanceImpl {
public:
    explicit InstanceImpl(const AllocationCallbacks& cb);
    ~InstanceImpl();

    Result Create(const InstanceCreationDesc& desc);
    Result SetCommonSettings(const CommonSettings& commonSettings);
    Result SetDenoiserSettings(Identifier identifier, const void* denoiserSettings);
    Result GetComputeDispatches(const Identifier* identifiers, uint32_t identifiersNum, const DispatchDesc*& dispatchDescs, uint32_t& dispatchDescsNum);

    const InstanceDesc& GetDesc() const { return m_Desc; }
    // NRD_HIP_REFERENCE_QUIRKS (wrapper.cpp ReferenceQuirksEnabled, read at creation): the instance reproduces what a caller of the reference observes where this library
    // otherwise corrects it -- today: REBLUR_DIFFUSE_SPECULAR_SH describes the reference's 11 transient textures, whose Transient::TILES index names a full-resolution
    // RGBA16F texture (Reblur_DiffuseSpecularSh.hpp:61-85). The HIP executor then binds the real tile plane wherever a dispatch names that index: TransientAlias.
    bool ReferenceQuirks() const { return m_ReferenceQuirks; }
    uint16_t TransientAlias(Identifier identifier, uint16_t indexInPool) const {
        for (size_t i = 0; i + 2 < m_TransientAliases.size(); i += 3)
            if (m_TransientAliases[i] == identifier && m_TransientAliases[i + 1] == indexInPool)
                return (uint16_t)m_TransientAliases[i + 2];
        return indexInPool;
    }
    const AllocationCallbacks& GetAllocationCallbacks() const { return m_Callbacks; }

private:
    // ---- denoiser tables (one translation unit per family)
    void Add_Reference(DenoiserData& d);
    void Update_Reference(const DenoiserData& d);

    void Add_Reblur(DenoiserData& d, bool hasDiff, bool hasSpec, bool sh, bool directionalOcclusion = false);
    void Update_Reblur(const DenoiserData& d);
    void Add_ReblurOcclusion(DenoiserData& d, bool hasDiff, bool hasSpec);
    void Update_ReblurOcclusion(const DenoiserData& d);
    void FillReblurConstants(const ReblurSettings& settings, void* data);

    void Add_Relax(DenoiserData& d, bool hasDiff, bool hasSpec, bool sh);
    void Add_RelaxVariant(DenoiserData& d);
    void Update_Relax(const DenoiserData& d);
    void FillRelaxConstants(const RelaxSettings& settings, void* data);

    void Add_SigmaShadow(DenoiserData& d, bool translucent);
    void Update_SigmaShadow(const DenoiserData& d);

    // ---- table building helpers
    void AddPermanent(Format format, uint16_t downsample = 1) { m_PermanentPool.push_back({format, downsample}); }
    void AddTransient(Format format, uint16_t downsample = 1);
    void BeginPass(const char* name) {
        m_PassName = name;
        m_ResourceOffset = m_Resources.size();
    }
    void In(uint16_t localIndex, uint16_t swapWith = NO_SWAP) { PushPlane(DescriptorType::TEXTURE, localIndex, swapWith); }
    void Out(uint16_t localIndex, uint16_t swapWith = NO_SWAP) { PushPlane(DescriptorType::STORAGE_TEXTURE, localIndex, swapWith); }
    void In(ResourceType t) { In((uint16_t)t); }
    void Out(ResourceType t) { Out((uint16_t)t); }
    void PushPlane(DescriptorType descriptorType, uint16_t localIndex, uint16_t swapWith);
    // registers the pass just described under "pipeline" (pipelines are unique by name)
    void EndPass(const char* pipeline, uint8_t groupW, uint8_t groupH, uint32_t constantSize, uint16_t downsampleFactor = 1, uint16_t maxRepeats = 1);

    // ---- per-frame
    void SwapPingPong(const DenoiserData& d);
    void* PushDispatch(const DenoiserData& d, uint32_t localIndex);
    void FinalizeDesc();
    const char* InternString(const char* s);

    friend struct ReblurTableBuilder;

private:
    AllocationCallbacks m_Callbacks;
    Vector<DenoiserData> m_DenoiserData;
    Vector<TextureDesc> m_PermanentPool;
    Vector<TextureDesc> m_TransientPool;
    Vector<ResourceDesc> m_Resources;
    Vector<ClearResource> m_ClearResources;
    Vector<PingPong> m_PingPongs;
    Vector<ResourceRangeDesc> m_ResourceRanges;
    Vector<size_t> m_PipelineRangeOffset;
    Vector<PipelineDesc> m_Pipelines;
    Vector<PassTemplate> m_Passes;
    Vector<DispatchDesc> m_ActiveDispatches;
    Vector<uint16_t> m_IndexRemap;
    Vector<uint32_t> m_TransientAliases; // triples (denoiser identifier, index its dispatches name, index the executor binds): reference quirks only
    Vector<char*> m_Strings;

    InstanceDesc m_Desc = {};
    CommonSettings m_CommonSettings = {};

    nrdhost::Mat4 m_ViewToClip = nrdhost::Mat4::Identity();
    nrdhost::Mat4 m_ViewToClipPrev = nrdhost::Mat4::Identity();
    nrdhost::Mat4 m_ClipToView = nrdhost::Mat4::Identity();
    nrdhost::Mat4 m_ClipToViewPrev = nrdhost::Mat4::Identity();
    nrdhost::Mat4 m_WorldToView = nrdhost::Mat4::Identity();
    nrdhost::Mat4 m_WorldToViewPrev = nrdhost::Mat4::Identity();
    nrdhost::Mat4 m_ViewToWorld = nrdhost::Mat4::Identity();
    nrdhost::Mat4 m_ViewToWorldPrev = nrdhost::Mat4::Identity();
    nrdhost::Mat4 m_WorldToClip = nrdhost::Mat4::Identity();
    nrdhost::Mat4 m_WorldToClipPrev = nrdhost::Mat4::Identity();
    nrdhost::Mat4 m_ClipToWorld = nrdhost::Mat4::Identity();
    nrdhost::Mat4 m_ClipToWorldPrev = nrdhost::Mat4::Identity();
    nrdhost::Mat4 m_WorldPrevToWorld = nrdhost::Mat4::Identity();
    nrdhost::Vec4 m_RotatorPre, m_Rotator, m_RotatorPost;
    nrdhost::Vec4 m_Frustum, m_FrustumPrev;
    nrdhost::Vec3 m_CameraDelta, m_ViewDirection, m_ViewDirectionPrev;

    float m_SplitScreenPrev = 0.0f;
    const char* m_PassName = nullptr;
    uint8_t* m_ConstantDataUnaligned = nullptr;
    uint8_t* m_ConstantData = nullptr;
    size_t m_ConstantDataOffset = 0;
    bool m_ConstantOverflow = false;
    size_t m_ResourceOffset = 0;
    size_t m_ClearPassIndex[2] = {};
    float m_OrthoMode = 0.0f;
    float m_CheckerboardResolveAccumSpeed = 0.0f;
    float m_JitterDelta = 0.0f;
    float m_TimeDelta = 0.0f;
    float m_FrameRateScale = 0.0f;
    float m_ProjectY = 0.0f;
    uint32_t m_AccumulatedFrameNum = 0;
    uint16_t m_TransientPoolOffset = 0;
    uint16_t m_PermanentPoolOffset = 0;
    bool m_IsFirstUse = true;
    bool m_ReferenceQuirks = false;

    // wall-clock fallback when CommonSettings::timeDeltaBetweenFrames == 0 (reference Source/Timer.cpp)
    std::chrono::steady_clock::time_point m_LastTime;
    bool m_HasLastTime = false;
    float m_SmoothedDeltaMs = 16.667f;
};

} // namespace nrd
