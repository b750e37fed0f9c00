// nrd::ShardedIntegrationHip (include/NRDShardedIntegrationHip.hpp) driven the way a C++ renderer would, on ONE GPU: N virtual ranks, each with its
// own instance + executor + planes, connected by a loop-back HaloTransport (a "receive" is a device-to-device copy out of the peer's plane). The
// ranks advance in lock-step (all exchanges of a step, then all segments of that step), which is what the real transports guarantee through stream
// ordering. Every rank's owned rows of every output must equal a plain single-GPU nrd::IntegrationHip run, bit for bit, every frame -- including a
// frame that cannot be sharded in mid-sequence (hit-distance reconstruction on) and the completion of the history planes in front of it.
// REBLUR_DIFFUSE_SPECULAR on a procedural G-buffer packed with include/NRD.hip.h (the front-end an application's own kernels would use).
// usage: sharded_virtual_ranks [world = 3] [measure]   --compile-only check: build with -DNRD_SHARDED_WITH_RCCL to also compile the RCCL transport
// measure: ShardedIntegrationHipCreationDesc::measureMotion -- the motion bound comes from nrdHipMeasureMotionRows on every rank's strip (PrepareFrame), the test takes the
// maximum (what HaloTransport::MaxOverRanks does between real ranks) and hands it to PlanFrame; in frame 4 ONE pixel of the last strip moves 30 rows (2D motion vectors):
// every rank has to run that frame unsharded, and the next one sharded again.
#include "NRD.h"
#include "NRDHip.h"
#include "NRDIntegrationHip.hpp"
#include "NRDShardedIntegrationHip.hpp"
#include "NRD.hip.h"

#include <hip/hip_fp16.h>
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CHECK(x)                                            \
    do {                                                    \
        if (!(x)) {                                         \
            printf("FAILED: %s (line %d)\n", #x, __LINE__); \
            fflush(stdout);                                 \
            return 1;                                       \
        }                                                   \
    } while (0)

static uint32_t Pcg(uint32_t v) {
    uint32_t s = v * 747796405u + 2891336453u;
    uint32_t w = ((s >> ((s >> 28u) + 4u)) ^ s) * 277803737u;
    return (w >> 22u) ^ w;
}
static float U(uint32_t a, uint32_t b, uint32_t c) { return float(Pcg(Pcg(a * 9781u + b) ^ (c * 6271u)) >> 8) * (1.0f / 16777216.0f); }

class LoopbackTransport : public nrd::HaloTransport {
public:
    std::vector<nrd::ShardedIntegrationHip*>* ranks = nullptr;
    uint32_t self = 0;
    size_t receivedBytes = 0;
    bool Exchange(const nrd::HaloTransfer* t, uint32_t n, void* stream) override {
        for (uint32_t i = 0; i < n; i++)
            if (!t[i].send && !CopyFrom(t[i], t[i].peer, stream))
                return false;
        return true;
    }
    bool Wait(void*) override { return true; } // the copies were enqueued on the compute stream itself
    bool Broadcast(const nrd::HaloTransfer& band, uint32_t root, void* stream) override { return root == self || CopyFrom(band, root, stream); }

private:
    bool CopyFrom(const nrd::HaloTransfer& t, uint32_t peer, void* stream) {
        NrdHipPlaneDesc p = {};
        if (!(*ranks)[peer]->GetPlane(t.resourceType, t.indexInPool, p))
            return false;
        receivedBytes += t.bytes;
        return hipMemcpyAsync(t.data, (const uint8_t*)p.data + (size_t)t.rowBegin * p.rowPitchBytes, t.bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream) == hipSuccess;
    }
};

struct Planes { // the user planes of one executor
    void *mv, *normalRoughness, *viewZ, *inDiff, *inSpec, *outDiff, *outSpec;
};

int main(int argc, char** argv) {
    const uint32_t world = argc > 1 ? (uint32_t)atoi(argv[1]) : 3;
    const bool measure = argc > 2 && !strcmp(argv[2], "measure");
    const int fastFrame = 4;
    const uint16_t W = 192, H = 288;
    const int frames = 6, unshardedFrame = 3;
    const size_t texels = (size_t)W * H;

    hipStream_t stream = nullptr;
    CHECK(hipStreamCreate(&stream) == hipSuccess);

    nrd::DenoiserDesc denoisers[] = {{1, nrd::Denoiser::REBLUR_DIFFUSE_SPECULAR}};
    nrd::InstanceCreationDesc icd = {};
    icd.denoisers = denoisers;
    icd.denoisersNum = 1;

    // executors: [0] = the single-GPU run, [1 + r] = virtual rank r
    nrd::IntegrationHip single;
    nrd::IntegrationHipCreationDesc idesc = {};
    idesc.resourceWidth = W;
    idesc.resourceHeight = H;
    idesc.hipStream = stream;
    CHECK(single.Initialize(idesc, icd));
    std::vector<nrd::ShardedIntegrationHip*> ranks(world);
    std::vector<LoopbackTransport> transports(world);
    for (uint32_t r = 0; r < world; r++) {
        ranks[r] = new nrd::ShardedIntegrationHip();
        transports[r].ranks = &ranks;
        transports[r].self = r;
        nrd::ShardedIntegrationHipCreationDesc sdesc = {};
        sdesc.integration = idesc;
        sdesc.transport = &transports[r];
        sdesc.rank = r;
        sdesc.world = world;
        sdesc.maxMotionRows = 8;
        sdesc.measureMotion = measure;
        CHECK(ranks[r]->Initialize(sdesc, icd));
    }

    // one device word per rank for the history reach its temporal kernels report (nrdHipSetHistoryReachWord; the measure mode reduces them by hand, like the motion)
    float* reachWords = nullptr;
    CHECK(hipMalloc((void**)&reachWords, sizeof(float) * world) == hipSuccess && hipMemset(reachWords, 0, sizeof(float) * world) == hipSuccess);
    if (measure)
        for (uint32_t r = 0; r < world; r++)
            CHECK(nrdHipSetHistoryReachWord(ranks[r]->GetIntegration().GetExecutor(), reachWords + r) == (uint32_t)nrd::Result::SUCCESS);

    std::vector<Planes> planes(world + 1);
    for (Planes& p : planes) {
        CHECK(hipMalloc(&p.mv, texels * 8) == hipSuccess && hipMalloc(&p.normalRoughness, texels * 4) == hipSuccess && hipMalloc(&p.viewZ, texels * 4) == hipSuccess);
        CHECK(hipMalloc(&p.inDiff, texels * 8) == hipSuccess && hipMalloc(&p.inSpec, texels * 8) == hipSuccess && hipMalloc(&p.outDiff, texels * 8) == hipSuccess && hipMalloc(&p.outSpec, texels * 8) == hipSuccess);
        CHECK(hipMemset(p.mv, 0, texels * 8) == hipSuccess && hipMemset(p.outDiff, 0, texels * 8) == hipSuccess && hipMemset(p.outSpec, 0, texels * 8) == hipSuccess);
    }
    auto pool = [&](const Planes& p) {
        nrd::UserPoolHip u = {};
        nrd::IntegrationHip_SetResource(u, nrd::ResourceType::IN_MV, NrdHipPlaneDesc{p.mv, (uint32_t)W * 8, (uint32_t)nrd::Format::RGBA16_SFLOAT, W, H});
        nrd::IntegrationHip_SetResource(u, nrd::ResourceType::IN_NORMAL_ROUGHNESS, NrdHipPlaneDesc{p.normalRoughness, (uint32_t)W * 4, (uint32_t)nrd::Format::R10_G10_B10_A2_UNORM, W, H});
        nrd::IntegrationHip_SetResource(u, nrd::ResourceType::IN_VIEWZ, NrdHipPlaneDesc{p.viewZ, (uint32_t)W * 4, (uint32_t)nrd::Format::R32_SFLOAT, W, H});
        nrd::IntegrationHip_SetResource(u, nrd::ResourceType::IN_DIFF_RADIANCE_HITDIST, NrdHipPlaneDesc{p.inDiff, (uint32_t)W * 8, (uint32_t)nrd::Format::RGBA16_SFLOAT, W, H});
        nrd::IntegrationHip_SetResource(u, nrd::ResourceType::IN_SPEC_RADIANCE_HITDIST, NrdHipPlaneDesc{p.inSpec, (uint32_t)W * 8, (uint32_t)nrd::Format::RGBA16_SFLOAT, W, H});
        nrd::IntegrationHip_SetResource(u, nrd::ResourceType::OUT_DIFF_RADIANCE_HITDIST, NrdHipPlaneDesc{p.outDiff, (uint32_t)W * 8, (uint32_t)nrd::Format::RGBA16_SFLOAT, W, H});
        nrd::IntegrationHip_SetResource(u, nrd::ResourceType::OUT_SPEC_RADIANCE_HITDIST, NrdHipPlaneDesc{p.outSpec, (uint32_t)W * 8, (uint32_t)nrd::Format::RGBA16_SFLOAT, W, H});
        return u;
    };

    // a floor receding from the camera under a wavy wall, sky at the top; static camera, noise changing every frame
    const float fy = 1.0f / tanf(0.5f * 1.0471975f), fx = fy * float(H) / float(W), zn = 0.1f, zf = 1000.0f;
    std::vector<uint32_t> hNR(texels);
    std::vector<float> hZ(texels);
    std::vector<__half> hDiff(texels * 4), hSpec(texels * 4);
    for (uint32_t y = 0; y < H; y++)
        for (uint32_t x = 0; x < W; x++) {
            const size_t i = (size_t)y * W + x;
            const bool sky = y < 40;
            const float t = float(y) / float(H);
            hZ[i] = sky ? 1.0e6f : 4.0f + 30.0f * (1.0f - t) * (1.0f - t) + 0.3f * sinf(float(x) * 0.07f);
            float3 N = nrd_hip_detail::normalize(make_float3(0.25f * sinf(float(x) * 0.05f), 0.6f + 0.3f * t, -0.7f));
            hNR[i] = NRD_StoreR10G10B10A2(NRD_FrontEnd_PackNormalAndRoughness(N, 0.15f + 0.7f * float((x / 24 + y / 24) & 1), 0.0f));
        }

    size_t mismatches = 0, shardedFrames = 0, gatherMismatches = 0;
    for (int f = 0; f < frames; f++) {
        const float4 hitDistParams = make_float4(3.0f, 0.1f, 20.0f, -25.0f);
        for (uint32_t y = 0; y < H; y++)
            for (uint32_t x = 0; x < W; x++) {
                const size_t i = (size_t)y * W + x;
                const bool sky = hZ[i] > 5.0e5f;
                const float base = 0.2f + 0.6f * float(y) / float(H);
                for (int k = 0; k < 2; k++) {
                    const float e = -logf(fmaxf(U(x, y, (uint32_t)f * 2u + (uint32_t)k), 1e-6f)); // exponential noise, mean 1
                    float3 radiance = sky ? make_float3(0.0f, 0.0f, 0.0f) : make_float3(base * e, 0.8f * base * e, 0.6f * base * e);
                    const float hitDist = 0.5f + 8.0f * U(x + 7u, y + 3u, (uint32_t)f * 2u + (uint32_t)k);
                    const float4 packed = REBLUR_FrontEnd_PackRadianceAndNormHitDist(radiance, sky ? 0.0f : REBLUR_FrontEnd_GetNormHitDist(hitDist, hZ[i], hitDistParams, k ? 0.5f : 1.0f));
                    __half* dst = (k ? hSpec.data() : hDiff.data()) + i * 4;
                    dst[0] = __float2half(packed.x), dst[1] = __float2half(packed.y), dst[2] = __float2half(packed.z), dst[3] = __float2half(packed.w);
                }
            }
        for (Planes& p : planes) {
            CHECK(hipMemcpyAsync(p.normalRoughness, hNR.data(), texels * 4, hipMemcpyHostToDevice, stream) == hipSuccess);
            CHECK(hipMemcpyAsync(p.viewZ, hZ.data(), texels * 4, hipMemcpyHostToDevice, stream) == hipSuccess);
            CHECK(hipMemcpyAsync(p.inDiff, hDiff.data(), texels * 8, hipMemcpyHostToDevice, stream) == hipSuccess);
            CHECK(hipMemcpyAsync(p.inSpec, hSpec.data(), texels * 8, hipMemcpyHostToDevice, stream) == hipSuccess);
        }

        nrd::CommonSettings cs = {};
        const float proj[16] = {fx, 0, 0, 0, 0, fy, 0, 0, 0, 0, zf / (zf - zn), 1.0f, 0, 0, -zn * zf / (zf - zn), 0};
        memcpy(cs.viewToClipMatrix, proj, sizeof(proj));
        memcpy(cs.viewToClipMatrixPrev, proj, sizeof(proj));
        for (int k = 0; k < 4; k++)
            cs.worldToViewMatrix[k * 5] = cs.worldToViewMatrixPrev[k * 5] = 1.0f;
        cs.resourceSize[0] = cs.resourceSizePrev[0] = cs.rectSize[0] = cs.rectSizePrev[0] = W;
        cs.resourceSize[1] = cs.resourceSizePrev[1] = cs.rectSize[1] = cs.rectSizePrev[1] = H;
        cs.motionVectorScale[0] = cs.motionVectorScale[1] = cs.motionVectorScale[2] = 0.0f;
        cs.isMotionVectorInWorldSpace = true;
        if (measure) { // 2D motion vectors in pixels: zero everywhere (static scene, static camera), except one pixel of the last strip in the fast frame
            cs.isMotionVectorInWorldSpace = false;
            cs.motionVectorScale[0] = 1.0f / float(W), cs.motionVectorScale[1] = 1.0f / float(H);
            std::vector<__half> hMv(texels * 4, __float2half(0.0f));
            if (f == fastFrame)
                hMv[((size_t)(H - 9) * W + 17) * 4 + 1] = __float2half(-30.0f);
            for (Planes& p : planes)
                CHECK(hipMemcpyAsync(p.mv, hMv.data(), texels * 8, hipMemcpyHostToDevice, stream) == hipSuccess);
            CHECK(hipStreamSynchronize(stream) == hipSuccess); // hMv goes out of scope
        }
        cs.frameIndex = (uint32_t)f;
        cs.timeDeltaBetweenFrames = 16.667f; // 0 would make every instance measure its own wall-clock frame time (frame-rate dependent constants)
        cs.accumulationMode = f == 0 ? nrd::AccumulationMode::CLEAR_AND_RESTART : nrd::AccumulationMode::CONTINUE;
        nrd::ReblurSettings rs = {};
        // halos that fit a 96-row strip -- except on one frame, whose blur rings are as large as their distance from the camera and therefore have no bounded reach
        // (executor.hip ReblurBlurReachRows): that frame runs unsharded on every rank. (Until round 5 a hit-distance reconstruction pass served as the trigger; it is sharded now.)
        rs.maxBlurRadius = f == unshardedFrame ? 400.0f : 10.0f;
        rs.diffusePrepassBlurRadius = rs.specularPrepassBlurRadius = 12.0f;
        rs.hitDistanceReconstructionMode = nrd::HitDistanceReconstructionMode::OFF;

        nrd::Identifier id = 1;
        single.NewFrame();
        CHECK(single.SetCommonSettings(cs) && single.SetDenoiserSettings(1, &rs));
        if (!single.Denoise(&id, 1, pool(planes[0]))) {
            printf("single-GPU Denoise failed: %s\n", single.GetLastError());
            return 1;
        }
        uint32_t steps = 0;
        for (uint32_t r = 0; r < world; r++) {
            ranks[r]->NewFrame();
            CHECK(ranks[r]->SetCommonSettings(cs) && ranks[r]->SetDenoiserSettings(1, &rs));
            if (!measure && !ranks[r]->BeginFrame(&id, 1, pool(planes[1 + r]))) {
                printf("rank %u BeginFrame failed: %s\n", r, ranks[r]->GetLastError());
                return 1;
            }
        }
        if (measure) { // PrepareFrame on every rank, the maximum of the measured rows, PlanFrame with it
            // ... and with what the temporal kernels of every rank reported for the PREVIOUS frame (nrdHipSetHistoryReachWord; round 6): the frame behind the fast one still runs
            // unsharded, because 1.25 x the 30 rows the fast frame's kernels read does not fit the 8-row halo either
            std::vector<float> reach(world, 0.0f);
            CHECK(hipMemcpy(reach.data(), reachWords, sizeof(float) * world, hipMemcpyDeviceToHost) == hipSuccess && hipMemset(reachWords, 0, sizeof(float) * world) == hipSuccess);
            float reachMax = 0.0f;
            for (uint32_t r = 0; r < world; r++)
                reachMax = fmaxf(reachMax, reach[r]);
            CHECK(world == 1 || f != fastFrame + 1 || fabsf(reachMax - 30.0f) < 0.05f);
            CHECK(world == 1 || f == fastFrame + 1 || reachMax < 0.5f); // a static scene under a static camera reads last frame's planes where it stands (up to the rounding of uv * size)
            float rowsMax = -1.0f;
            for (uint32_t r = 0; r < world; r++) {
                float rows = -1.0f;
                if (!ranks[r]->PrepareFrame(&id, 1, pool(planes[1 + r]), &rows)) {
                    printf("rank %u PrepareFrame failed: %s\n", r, ranks[r]->GetLastError());
                    return 1;
                }
                if (world > 1) {
                    CHECK(rows >= 0.0f);
                    CHECK(f != fastFrame || (r + 1 == world ? fabsf(rows - 30.0f) < 0.05f : rows == 0.0f)); // only the last strip sees the moving pixel
                }
                rowsMax = fmaxf(rowsMax, rows);
            }
            for (uint32_t r = 0; r < world; r++)
                CHECK(ranks[r]->PlanFrame(rowsMax, reachMax));
        }
        for (uint32_t r = 0; r < world; r++) {
            CHECK(r == 0 || ranks[r]->GetStepsNum() == steps);
            steps = ranks[r]->GetStepsNum();
        }
        const bool expectSharded = world > 1 && f != unshardedFrame && !(measure && (f == fastFrame || f == fastFrame + 1)); // (the restart frame too since round 6: clears are texel-local)
        CHECK((steps > 1) == expectSharded);
        shardedFrames += steps > 1;
        for (uint32_t s = 0; s < steps; s++) { // lock-step: every rank's transfers of step s see the peers' rows of step s - 1
            for (uint32_t r = 0; r < world; r++)
                if (!ranks[r]->ExchangeStep(s)) {
                    printf("rank %u ExchangeStep(%u) failed: %s\n", r, s, ranks[r]->GetLastError());
                    return 1;
                }
            for (uint32_t r = 0; r < world; r++)
                if (!ranks[r]->RunStep(s)) {
                    printf("rank %u RunStep(%u) failed: %s\n", r, s, ranks[r]->GetLastError());
                    return 1;
                }
        }
        for (uint32_t r = 0; r < world; r++)
            ranks[r]->EndFrame();
        CHECK(hipStreamSynchronize(stream) == hipSuccess);
        // the owned rows are compared below BEFORE the reassembly could paper over anything; the gather itself is checked after that comparison

        std::vector<uint16_t> ref(texels * 4), got(texels * 4);
        for (int k = 0; k < 2; k++) {
            CHECK(hipMemcpy(ref.data(), k ? planes[0].outSpec : planes[0].outDiff, texels * 8, hipMemcpyDeviceToHost) == hipSuccess);
            for (uint32_t r = 0; r < world; r++) {
                CHECK(hipMemcpy(got.data(), k ? planes[1 + r].outSpec : planes[1 + r].outDiff, texels * 8, hipMemcpyDeviceToHost) == hipSuccess);
                const size_t b = (size_t)ranks[r]->GetOwnedRowBegin() * W * 4, e = (size_t)ranks[r]->GetOwnedRowEnd() * W * 4;
                size_t bad = 0, firstRow = 0, lastRow = 0;
                for (size_t i = b; i < e; i++)
                    if (ref[i] != got[i]) {
                        if (!bad)
                            firstRow = i / ((size_t)W * 4);
                        lastRow = i / ((size_t)W * 4);
                        bad++;
                    }
                if (bad && getenv("NRD_TEST_VERBOSE"))
                    printf("  frame %d %s rank %u (rows %u..%u): %zu mismatching values in rows %zu..%zu\n", f, k ? "spec" : "diff", r, ranks[r]->GetOwnedRowBegin(), ranks[r]->GetOwnedRowEnd(), bad, firstRow,
                        lastRow);
                mismatches += bad;
            }
        }
        // the reassembly (ShardedIntegrationHip::GatherOutputs, what Denoise() ends with): afterwards EVERY rank holds the complete output planes, bit for bit
        for (uint32_t r = 0; r < world; r++)
            CHECK(ranks[r]->GatherOutputs());
        CHECK(hipStreamSynchronize(stream) == hipSuccess);
        for (int k = 0; k < 2; k++) {
            CHECK(hipMemcpy(ref.data(), k ? planes[0].outSpec : planes[0].outDiff, texels * 8, hipMemcpyDeviceToHost) == hipSuccess);
            for (uint32_t r = 0; r < world; r++) {
                CHECK(hipMemcpy(got.data(), k ? planes[1 + r].outSpec : planes[1 + r].outDiff, texels * 8, hipMemcpyDeviceToHost) == hipSuccess);
                size_t bad = 0;
                for (size_t i = 0; i < texels * 4; i++)
                    bad += ref[i] != got[i];
                if (bad && getenv("NRD_TEST_VERBOSE"))
                    printf("  frame %d %s rank %u: %zu mismatching values in the COMPLETE plane after the gather\n", f, k ? "spec" : "diff", r, bad);
                gatherMismatches += bad;
            }
        }
        size_t nonZero = 0;
        for (size_t i = 0; i < texels * 4; i++)
            nonZero += ref[i] != 0;
        printf("frame %d: %u step(s), %zu mismatching values in the owned rows so far, %zu non-zero output values\n", f, steps, mismatches, nonZero);
        CHECK(nonZero > texels); // the denoiser did produce something
    }
    size_t received = 0;
    for (const LoopbackTransport& t : transports)
        received += t.receivedBytes;
    printf("%u virtual ranks, %zu sharded frames, %zu bytes received through the transport, %zu mismatching values\n", world, shardedFrames, received, mismatches);

    single.Destroy();
    if (measure && world > 1)
        for (uint32_t r = 0; r < world; r++)
            CHECK(ranks[r]->GetMotionFallbacksNum() == 2 && ranks[r]->GetHistoryHaloViolationsNum() == 0 && fabsf(ranks[r]->GetLastHistoryReachRows() - 30.0f) < 0.05f);
    for (uint32_t r = 0; r < world; r++) {
        (void)nrdHipSetHistoryReachWord(ranks[r]->GetIntegration().GetExecutor(), nullptr);
        ranks[r]->Destroy();
        delete ranks[r];
    }
    (void)hipFree(reachWords);
    printf("%zu mismatching values in the complete planes after the output gather\n", gatherMismatches);
    if (gatherMismatches)
        return 1;
    if (mismatches || (world > 1 && (shardedFrames != (size_t)frames - 1 - (measure ? 2 : 0) || received == 0)))
        return 1;
    printf("sharded integration OK\n");
    return 0;
}
