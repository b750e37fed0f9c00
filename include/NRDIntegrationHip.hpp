// nrd::IntegrationHip -- header-only C++ convenience layer over the HIP back-end's C-ABI (include/NRDHip.h), shaped like the
// reference integration class so that an application written against it changes types, not structure:
//
//   reference Integration/NRDIntegration.h:37-48   UserPool + Integration_SetResource        -> nrd::UserPoolHip + IntegrationHip_SetResource
//   reference Integration/NRDIntegration.h:50-81   IntegrationCreationDesc                    -> nrd::IntegrationHipCreationDesc
//   reference Integration/NRDIntegration.h:83-127  Integration::{Initialize, NewFrame, SetCommonSettings, SetDenoiserSettings,
//                                                  Denoise, Destroy, Get*MemoryUsageInMb}     -> same member names
//
// What differs, and why: textures are pitched device-memory planes (NrdHipPlaneDesc) instead of nri::TextureBarrierDesc, the
// "command buffer" is the hipStream_t given at Initialize (launches are asynchronous on it, in order), there are no barriers,
// descriptor pools or buffered-frame constants to manage (constants travel as kernel arguments), and pipelines need no creation step.
// Include after NRD.h and NRDHip.h. No HIP headers are needed: the stream is passed as void*.
#pragma once

#include <array>
#include <stdint.h>
#include <stdio.h>

#define NRD_INTEGRATION_HIP_MAJOR 1
#define NRD_INTEGRATION_HIP_MINOR 0

#ifndef NRD_INTEGRATION_ASSERT
#    include <assert.h>
#    define NRD_INTEGRATION_ASSERT(expr, msg) assert(msg && expr)
#endif

namespace nrd {

// One entry per ResourceType slot (the two pool pseudo-slots excluded, as in the reference). Zero-initialise; fill the slots the
// requested denoisers need (NRDDescs.h lists them per denoiser).
typedef std::array<NrdHipPlaneDesc, (size_t)ResourceType::MAX_NUM - 2> UserPoolHip;

inline void IntegrationHip_SetResource(UserPoolHip& pool, ResourceType slot, const NrdHipPlaneDesc& plane) {
    NRD_INTEGRATION_ASSERT(plane.data != nullptr, "Invalid plane!");
    pool[(size_t)slot] = plane;
}

struct IntegrationHipCreationDesc {
    const char* name = "";
    uint16_t resourceWidth = 0;
    uint16_t resourceHeight = 0;
    void* hipStream = nullptr; // hipStream_t; nullptr = the default stream
    // Optional caller-owned pool memory ("NRD allocates no GPU memory"): at least nrdHipGetArenaSize() bytes, 256-byte aligned.
    void* arena = nullptr;
    uint64_t arenaSize = 0;
};

class IntegrationHip {
public:
    inline IntegrationHip() {}
    inline ~IntegrationHip() { NRD_INTEGRATION_ASSERT(m_Executor == nullptr, "Destroy() must be called before the destructor!"); }

    // There is no "Resize": recreate (Destroy + Initialize), exactly as with the reference integration.
    inline bool Initialize(const IntegrationHipCreationDesc& desc, const InstanceCreationDesc& instanceCreationDesc) {
        NRD_INTEGRATION_ASSERT(m_Instance == nullptr, "Already initialized! Did you forget to call 'Destroy'?");
        if (CreateInstance(instanceCreationDesc, m_Instance) != Result::SUCCESS)
            return false;
        uint32_t r = desc.arena ? nrdHipCreateExecutorWithArena(m_Instance, desc.resourceWidth, desc.resourceHeight, desc.hipStream, desc.arena, desc.arenaSize, &m_Executor)
                                : nrdHipCreateExecutor(m_Instance, desc.resourceWidth, desc.resourceHeight, desc.hipStream, &m_Executor);
        if (r != (uint32_t)Result::SUCCESS) {
            DestroyInstance(*m_Instance);
            m_Instance = nullptr;
            m_Executor = nullptr;
            return false;
        }
        m_Name = desc.name;
        m_FrameIndex = 0;
        nrdHipGetPoolMemoryUsage(m_Executor, &m_PermanentPoolSize, &m_TransientPoolSize);
        return true;
    }

    // Must be called once on a frame start (kept for call-order compatibility; nothing is buffered per frame here)
    inline void NewFrame() {
        NRD_INTEGRATION_ASSERT(m_Instance != nullptr, "Uninitialized! Did you forget to call 'Initialize'?");
        m_FrameIndex++;
    }

    // Explicitly call the eponymous NRD API functions
    inline bool SetCommonSettings(const CommonSettings& commonSettings) {
        NRD_INTEGRATION_ASSERT(m_Instance != nullptr, "Uninitialized! Did you forget to call 'Initialize'?");
        return nrd::SetCommonSettings(*m_Instance, commonSettings) == Result::SUCCESS;
    }
    inline bool SetDenoiserSettings(Identifier denoiser, const void* denoiserSettings) {
        NRD_INTEGRATION_ASSERT(m_Instance != nullptr, "Uninitialized! Did you forget to call 'Initialize'?");
        return nrd::SetDenoiserSettings(*m_Instance, denoiser, denoiserSettings) == Result::SUCCESS;
    }

    // Enqueues the denoising passes of the given denoisers on the stream. Every non-null entry of "userPool" is (re)bound first;
    // entries with data == nullptr are left as they are. Returns false (and GetLastError() says why) if a permutation is not
    // supported by this build or a required slot is not bound -- nothing is launched in that case.
    inline bool Denoise(const Identifier* denoisers, uint32_t denoisersNum, const UserPoolHip& userPool) {
        NRD_INTEGRATION_ASSERT(m_Executor != nullptr, "Uninitialized! Did you forget to call 'Initialize'?");
        for (size_t slot = 0; slot < userPool.size(); slot++) {
            if (userPool[slot].data && nrdHipBindResource(m_Executor, (uint32_t)slot, &userPool[slot]) != (uint32_t)Result::SUCCESS)
                return false;
        }
        return nrdHipDenoise(m_Executor, denoisers, denoisersNum) == (uint32_t)Result::SUCCESS;
    }

    // Assumes that no work of this integration is in flight on the stream
    inline void Destroy() {
        if (m_Executor)
            nrdHipDestroyExecutor(m_Executor);
        if (m_Instance)
            DestroyInstance(*m_Instance);
        m_Executor = nullptr;
        m_Instance = nullptr;
        m_PermanentPoolSize = m_TransientPoolSize = 0;
    }

    // Helpers
    inline double GetTotalMemoryUsageInMb() const { return double(m_PermanentPoolSize + m_TransientPoolSize) / (1024.0 * 1024.0); }
    inline double GetPersistentMemoryUsageInMb() const { return double(m_PermanentPoolSize) / (1024.0 * 1024.0); }
    inline double GetAliasableMemoryUsageInMb() const { return double(m_TransientPoolSize) / (1024.0 * 1024.0); }
    inline const char* GetLastError() const { return m_Executor ? nrdHipGetLastError(m_Executor) : "not initialized"; }
    inline Instance* GetInstance() const { return m_Instance; }
    inline NrdHipExecutor* GetExecutor() const { return m_Executor; }

private:
    IntegrationHip(const IntegrationHip&) = delete;

    Instance* m_Instance = nullptr;
    NrdHipExecutor* m_Executor = nullptr;
    const char* m_Name = "";
    uint64_t m_PermanentPoolSize = 0, m_TransientPoolSize = 0;
    uint32_t m_FrameIndex = 0;
};

} // namespace nrd
