// ORACLE -- TEST INFRASTRUCTURE, NOT PRODUCT CODE (see oracle/hlsl.h).
// Pass entry points: one function per reference shader file, taking the planes in the pass's binding order
// (inputs then outputs, exactly DispatchDesc::resources) and the raw constant block.
#pragma once

#include "tex.h"

namespace orc {

struct PassIO {
    Tex* t;            // planes in binding order
    uint32_t num;
    const void* constants;
    uint32_t constantsSize;
};

typedef void (*PassFn)(const PassIO& io);

struct PassEntry {
    const char* shaderFileName;
    PassFn fn;
};

const PassEntry* GetCommonPasses(uint32_t& n);
const PassEntry* GetReblurPasses(uint32_t& n);
const PassEntry* GetSigmaPasses(uint32_t& n);
const PassEntry* GetRelaxPasses(uint32_t& n);

} // namespace orc
