// ORACLE -- TEST INFRASTRUCTURE, NOT PRODUCT CODE (see oracle/hlsl.h).
// Clear and REFERENCE passes.
#include "passes.h"

namespace orc {

// reference Shaders/Source/Clear_Float.cs.hlsl:16-23 / Clear_Uint.cs.hlsl: gOut[ pixelPos ] = 0
static void Clear(const PassIO& io) {
    Tex& out = io.t[0];
    for (int y = 0; y < out.H(); y++)
        memset(out.Row(y), 0, out.p.pitch);
}

struct ReferenceAccumulateConstants { // reference Shaders/Resources/REFERENCE_TemporalAccumulation.resources.hlsli:11-16
    uint32_t gRectOrigin[2];
    float gAccumSpeed;
    float gDebug;
    float gViewZScale;
};

// reference Shaders/Source/REFERENCE_TemporalAccumulation.cs.hlsl:18-27
static void ReferenceAccumulate(const PassIO& io) {
    const auto& c = *(const ReferenceAccumulateConstants*)io.constants;
    const Tex& gIn_Input = io.t[0];
    Tex& gInOut_History = io.t[1];
#pragma omp parallel for schedule(static)
    for (int y = 0; y < gInOut_History.H(); y++)
        for (int x = 0; x < gInOut_History.W(); x++) {
            float4 input = gIn_Input.Load(x, y);
            float4 history = gInOut_History.Load(x, y);
            // the sequential fp32 running mean with three roundings per component (BASELINE.json: bit-exact; tests/test_reference.py holds it against
            // numpy): no fused multiply-add here, whatever the contraction mode of the build
            float4 result;
            {
#pragma clang fp contract(off)
                result.x = history.x + (input.x - history.x) * c.gAccumSpeed;
                result.y = history.y + (input.y - history.y) * c.gAccumSpeed;
                result.z = history.z + (input.z - history.z) * c.gAccumSpeed;
                result.w = history.w + (input.w - history.w) * c.gAccumSpeed;
            }
            gInOut_History.Store(x, y, result);
        }
}

struct ReferenceCopyConstants { // reference Shaders/Resources/REFERENCE_Copy.resources.hlsli:11-16
    float gRectSizeInv[2];
    float gSplitScreen;
    float gDebug;
    float gViewZScale;
};

// reference Shaders/Source/REFERENCE_Copy.cs.hlsl:18-26
static void ReferenceCopy(const PassIO& io) {
    const auto& c = *(const ReferenceCopyConstants*)io.constants;
    const Tex& gIn_Input = io.t[0];
    Tex& gOut_Output = io.t[1];
#pragma omp parallel for schedule(static)
    for (int y = 0; y < gOut_Output.H(); y++)
        for (int x = 0; x < gOut_Output.W(); x++) {
            float pixelUvX = (float(x) + 0.5f) * c.gRectSizeInv[0];
            if (pixelUvX > c.gSplitScreen && gIn_Input.In(x, y))
                gOut_Output.Store(x, y, gIn_Input.Load(x, y));
        }
}

const PassEntry* GetCommonPasses(uint32_t& n) {
    static const PassEntry k[] = {
        {"Clear_Float.cs", Clear},
        {"Clear_Uint.cs", Clear},
        {"REFERENCE_TemporalAccumulation.cs", ReferenceAccumulate},
        {"REFERENCE_Copy.cs", ReferenceCopy},
    };
    n = sizeof(k) / sizeof(k[0]);
    return k;
}

} // namespace orc
