// Pass registry of the HIP executor: one launcher per NRD pass (= per reference shader file). The executor resolves
// DispatchDesc::pipelineIndex -> PipelineDesc::shaderFileName -> launcher once at creation.
#pragma once

#include "planes.h"

#include <hip/hip_runtime.h>

namespace nrdhip {

struct PassArgs {
    const Plane* planes;      // DispatchDesc::resources resolved to planes, same order (inputs then outputs)
    uint32_t planesNum;
    const void* constants;    // DispatchDesc::constantBufferData
    uint32_t constantsSize;
    hipStream_t stream;
};

// returns nullptr on success, or a static message if the dispatch cannot be executed by this build (nothing is launched then)
typedef const char* (*PassLauncher)(const PassArgs& args);

struct PassEntry {
    const char* shaderFileName;
    PassLauncher launch;
};

// each kernels_*.hip exports its table
const PassEntry* GetCommonPasses(uint32_t& num);
const PassEntry* GetReblurPasses(uint32_t& num);
const PassEntry* GetSigmaPasses(uint32_t& num);

inline dim3 GridFor(int w, int h, int tileW, int tileH) { return dim3((unsigned)((w + tileW - 1) / tileW), (unsigned)((h + tileH - 1) / tileH), 1); }

} // namespace nrdhip
