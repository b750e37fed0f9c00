// C++ smoke test of the drop-in boundary, written the way an application of the reference uses NRD: include/NRD.h +
// include/NRDHip.h + include/NRDIntegrationHip.hpp, linked against libNRD_hip.so. BASELINE.json configs[0]: the REFERENCE
// denoiser (running mean of the input signal) on a 256x256 RGBA32F signal, checked on the host against the sequential fp32
// lerp( history, input, 1 / (1 + N) ) the shader performs (reference Shaders/Source/REFERENCE_TemporalAccumulation.cs.hlsl,
// Source/Reference.cpp) -- bit for bit.
// usage: integration_reference [--no-gpu]   (--no-gpu: stop after the host-only part: instance creation + dispatch list)
#include "NRD.h"
#include "NRDHip.h"
#include "NRDIntegrationHip.hpp"

#include <hip/hip_runtime_api.h>

#include <cstdio>
#include <cstring>
#include <vector>

#define CHECK(x)                                                  \
    do {                                                          \
        if (!(x)) {                                               \
            printf("FAILED: %s (line %d)\n", #x, __LINE__);       \
            return 1;                                             \
        }                                                         \
    } while (0)

static uint32_t Pcg(uint32_t v) {
    uint32_t s = v * 747796405u + 2891336453u;
    uint32_t w = ((s >> ((s >> 28u) + 4u)) ^ s) * 277803737u;
    return (w >> 22u) ^ w;
}

int main(int argc, char** argv) {
    const bool noGpu = argc > 1 && !strcmp(argv[1], "--no-gpu");
    const uint16_t W = 256, H = 256;
    const int frames = 8;

    const nrd::LibraryDesc& lib = nrd::GetLibraryDesc();
    printf("NRD %u.%u.%u, normal encoding %u\n", lib.versionMajor, lib.versionMinor, lib.versionBuild, (unsigned)lib.normalEncoding);

    nrd::DenoiserDesc denoisers[] = {{7, nrd::Denoiser::REFERENCE}};
    nrd::InstanceCreationDesc icd = {};
    icd.denoisers = denoisers;
    icd.denoisersNum = 1;

    if (noGpu) { // host-only: the dispatch compiler works without a device
        nrd::Instance* instance = nullptr;
        CHECK(nrd::CreateInstance(icd, instance) == nrd::Result::SUCCESS);
        nrd::CommonSettings cs = {};
        cs.resourceSize[0] = cs.resourceSizePrev[0] = cs.rectSize[0] = cs.rectSizePrev[0] = W;
        cs.resourceSize[1] = cs.resourceSizePrev[1] = cs.rectSize[1] = cs.rectSizePrev[1] = H;
        CHECK(nrd::SetCommonSettings(*instance, cs) == nrd::Result::SUCCESS);
        const nrd::DispatchDesc* dispatches = nullptr;
        uint32_t num = 0;
        nrd::Identifier id = 7;
        CHECK(nrd::GetComputeDispatches(*instance, &id, 1, dispatches, num) == nrd::Result::SUCCESS);
        CHECK(num >= 1);
        printf("host-only OK: %u dispatches, first = %s\n", num, dispatches[0].name);
        nrd::DestroyInstance(*instance);
        return 0;
    }

    hipStream_t stream = nullptr;
    CHECK(hipStreamCreate(&stream) == hipSuccess);

    nrd::IntegrationHipCreationDesc desc = {};
    desc.name = "smoke";
    desc.resourceWidth = W;
    desc.resourceHeight = H;
    desc.hipStream = stream;
    nrd::IntegrationHip nrdi;
    CHECK(nrdi.Initialize(desc, icd));
    printf("pools: %.2f MB persistent, %.2f MB aliasable\n", nrdi.GetPersistentMemoryUsageInMb(), nrdi.GetAliasableMemoryUsageInMb());

    const size_t texels = (size_t)W * H, bytes = texels * 16;
    void *dIn = nullptr, *dOut = nullptr;
    CHECK(hipMalloc(&dIn, bytes) == hipSuccess && hipMalloc(&dOut, bytes) == hipSuccess);
    CHECK(hipMemsetAsync(dOut, 0, bytes, stream) == hipSuccess);

    nrd::UserPoolHip pool = {};
    nrd::IntegrationHip_SetResource(pool, nrd::ResourceType::IN_SIGNAL, NrdHipPlaneDesc{dIn, (uint32_t)W * 16, (uint32_t)nrd::Format::RGBA32_SFLOAT, W, H});
    nrd::IntegrationHip_SetResource(pool, nrd::ResourceType::OUT_SIGNAL, NrdHipPlaneDesc{dOut, (uint32_t)W * 16, (uint32_t)nrd::Format::RGBA32_SFLOAT, W, H});

    std::vector<float> in(texels * 4), mean(texels * 4, 0.0f), out(texels * 4);
    for (int f = 0; f < frames; f++) {
        for (size_t i = 0; i < texels * 4; i++)
            in[i] = float(i % 97) * 0.01f + float(Pcg((uint32_t)i * 31u + (uint32_t)f) >> 8) * (1.0f / 16777216.0f);
        CHECK(hipMemcpyAsync(dIn, in.data(), bytes, hipMemcpyHostToDevice, stream) == hipSuccess);

        nrd::CommonSettings cs = {}; // identity matrices: worldToClip == worldToClipPrev, so the accumulation continues
        cs.resourceSize[0] = cs.resourceSizePrev[0] = cs.rectSize[0] = cs.rectSizePrev[0] = W;
        cs.resourceSize[1] = cs.resourceSizePrev[1] = cs.rectSize[1] = cs.rectSizePrev[1] = H;
        cs.frameIndex = (uint32_t)f;
        cs.accumulationMode = f == 0 ? nrd::AccumulationMode::CLEAR_AND_RESTART : nrd::AccumulationMode::CONTINUE;
        nrd::ReferenceSettings rs = {};
        rs.maxAccumulatedFrameNum = 1024;

        nrdi.NewFrame();
        CHECK(nrdi.SetCommonSettings(cs));
        CHECK(nrdi.SetDenoiserSettings(7, &rs));
        nrd::Identifier id = 7;
        if (!nrdi.Denoise(&id, 1, pool)) {
            printf("Denoise failed: %s\n", nrdi.GetLastError());
            return 1;
        }
        CHECK(hipStreamSynchronize(stream) == hipSuccess); // "in" is reused by the host next frame

        // host model: history = lerp( history, input, 1 / (1 + N) ), N = accumulated frames so far (0 on the restart frame)
        const float a = 1.0f / (1.0f + float(f));
        for (size_t i = 0; i < texels * 4; i++)
            mean[i] = mean[i] + (in[i] - mean[i]) * a;
    }
    CHECK(hipMemcpy(out.data(), dOut, bytes, hipMemcpyDeviceToHost) == hipSuccess);
    size_t mismatches = 0;
    for (size_t i = 0; i < texels * 4; i++)
        mismatches += memcmp(&out[i], &mean[i], 4) != 0;
    printf("REFERENCE %ux%u, %d frames: %zu mismatching values\n", W, H, frames, mismatches);

    nrdi.Destroy();
    (void)hipFree(dIn);
    (void)hipFree(dOut);
    (void)hipStreamDestroy(stream);
    if (mismatches)
        return 1;
    printf("integration smoke OK\n");
    return 0;
}
