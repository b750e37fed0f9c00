// MI355X-native NRD hot path -- public C-ABI (drop-in boundary).
//
// This header declares the SAME nine entry points, with the same names, argument meaning and
// error behaviour, as the reference library's Include/NRD.h:51-70 (NRD v4.14.0). A caller that was
// linked against the reference libNRD can be relinked against libNRD_hip.so unchanged; the only
// difference is that PipelineDesc::computeShader{DXBC,DXIL,SPIRV} are empty (exactly as in a
// reference build without NRD_EMBEDS_* -- reference Source/InstanceImpl.h:27-43), because the
// shader pass chain is executed by hand-written HIP kernels through include/NRDHip.h instead.
//
//   reference Include/NRD.h:51  CreateInstance          -> nrd::CreateInstance
//   reference Include/NRD.h:52  DestroyInstance         -> nrd::DestroyInstance
//   reference Include/NRD.h:55  GetLibraryDesc          -> nrd::GetLibraryDesc
//   reference Include/NRD.h:56  GetInstanceDesc         -> nrd::GetInstanceDesc
//   reference Include/NRD.h:59  SetCommonSettings       -> nrd::SetCommonSettings
//   reference Include/NRD.h:62  SetDenoiserSettings     -> nrd::SetDenoiserSettings
//   reference Include/NRD.h:66  GetComputeDispatches    -> nrd::GetComputeDispatches
//   reference Include/NRD.h:69  GetResourceTypeString   -> nrd::GetResourceTypeString
//   reference Include/NRD.h:70  GetDenoiserString       -> nrd::GetDenoiserString
#pragma once

#include <cstddef>
#include <cstdint>

#define NRD_VERSION_MAJOR 4
#define NRD_VERSION_MINOR 14
#define NRD_VERSION_BUILD 0
#define NRD_VERSION_DATE "19 February 2025"

#if defined(_WIN32)
#    define NRD_CALL __stdcall
#else
#    define NRD_CALL
#endif

#ifndef NRD_API
#    define NRD_API extern "C"
#endif

#include "NRDDescs.h"
#include "NRDSettings.h"

namespace nrd {

// Lifetime. "instance" memory comes from InstanceCreationDesc::allocationCallbacks (malloc family if null).
NRD_API Result NRD_CALL CreateInstance(const InstanceCreationDesc& instanceCreationDesc, Instance*& instance);
NRD_API void NRD_CALL DestroyInstance(Instance& instance);

// Queries. Returned references stay valid for the lifetime of the library / the instance.
NRD_API const LibraryDesc& NRD_CALL GetLibraryDesc();
NRD_API const InstanceDesc& NRD_CALL GetInstanceDesc(const Instance& instance);

// Once per frame, before GetComputeDispatches.
NRD_API Result NRD_CALL SetCommonSettings(Instance& instance, const CommonSettings& commonSettings);

// At least once per denoiser; "denoiserSettings" points to the matching *Settings struct.
NRD_API Result NRD_CALL SetDenoiserSettings(Instance& instance, Identifier identifier, const void* denoiserSettings);

// Builds this frame's pass list for the given denoisers. The returned array is owned by the instance
// and is overwritten by the next call.
NRD_API Result NRD_CALL GetComputeDispatches(Instance& instance, const Identifier* identifiers, uint32_t identifiersNum,
    const DispatchDesc*& dispatchDescs, uint32_t& dispatchDescsNum);

// Debug names.
NRD_API const char* GetResourceTypeString(ResourceType resourceType);
NRD_API const char* GetDenoiserString(Denoiser denoiser);

} // namespace nrd
