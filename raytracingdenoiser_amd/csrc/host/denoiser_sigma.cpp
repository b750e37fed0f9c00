// SIGMA_SHADOW and SIGMA_SHADOW_TRANSLUCENCY host tables (one parametrised description). Pool layout, pass order and binding
// order: reference Source/Denoisers/Sigma_Shadow.hpp:11-157 and Sigma_ShadowTranslucency.hpp:11-160 (RGBA8 instead of R8 shadow
// planes, IN_TRANSLUCENCY appended to ClassifyTiles / Blur / SplitScreen);
// per-frame selection: reference Source/Sigma.cpp:25-90; shared constants: Sigma.cpp:92-145.
#include "instance.h"

#include <algorithm>
#include <cstdio>

namespace nrd {

namespace {
enum : uint16_t { P_HISTORY_LENGTH = PERMANENT_POOL_START };
enum : uint16_t { T_DATA_1 = TRANSIENT_POOL_START, T_DATA_2, T_TEMP_1, T_TEMP_2, T_HISTORY, T_HISTORY_LENGTH, T_TILES, T_SMOOTHED_TILES };
enum : uint32_t { PASS_CLASSIFY_TILES, PASS_SMOOTH_TILES, PASS_COPY, PASS_BLUR, PASS_POST_BLUR, PASS_TEMPORAL_STABILIZATION = PASS_POST_BLUR + 2, PASS_SPLIT_SCREEN };
} // namespace

static void FillSigmaConstants(const SigmaSettings& s, const CommonSettings& cs, void* data, const nrdhost::Mat4& worldToView,
    const nrdhost::Mat4& viewToClip, const nrdhost::Mat4& worldToClipPrev, const nrdhost::Mat4& worldToViewPrev, const nrdhost::Vec4& rotator,
    const nrdhost::Vec4& rotatorPost, const nrdhost::Vec3& viewDir, const nrdhost::Vec4& frustum, const nrdhost::Vec4& frustumPrev,
    const nrdhost::Vec3& cameraDelta, float orthoMode, float projectY) {
    if (!data)
        return;
    const float resourceW = cs.resourceSize[0], resourceH = cs.resourceSize[1];
    const float resourceWprev = cs.resourceSizePrev[0], resourceHprev = cs.resourceSizePrev[1];
    const float rectW = cs.rectSize[0], rectH = cs.rectSize[1];
    const float rectWprev = cs.rectSizePrev[0], rectHprev = cs.rectSizePrev[1];

    const float unproject = 1.0f / (0.5f * rectH * projectY);
    const uint16_t tilesW = DivideUp(cs.rectSize[0], 16), tilesH = DivideUp(cs.rectSize[1], 16);
    const bool isRectChanged = cs.rectSize[0] != cs.rectSizePrev[0] || cs.rectSize[1] != cs.rectSizePrev[1];
    const uint32_t frameNum = std::min(s.maxStabilizedFrameNum, SIGMA_MAX_HISTORY_FRAME_NUM);
    const float stabilizationStrength = float(frameNum) / (1.0f + float(frameNum));

    // light direction rotated into view space (3x3 part only)
    float l[3];
    for (int i = 0; i < 3; i++)
        l[i] = worldToView.at(i, 0) * s.lightDirection[0] + worldToView.at(i, 1) * s.lightDirection[1] + worldToView.at(i, 2) * s.lightDirection[2];

    nrdc::SigmaConstants& c = *(nrdc::SigmaConstants*)data;
    memcpy(c.gWorldToView, &worldToView, 64);
    memcpy(c.gViewToClip, &viewToClip, 64);
    memcpy(c.gWorldToClipPrev, &worldToClipPrev, 64);
    memcpy(c.gWorldToViewPrev, &worldToViewPrev, 64);
    c.gRotator = {rotator[0], rotator[1], rotator[2], rotator[3]};
    c.gRotatorPost = {rotatorPost[0], rotatorPost[1], rotatorPost[2], rotatorPost[3]};
    c.gViewVectorWorld = {viewDir.x, viewDir.y, viewDir.z, 0.0f};
    c.gLightDirectionView = {l[0], l[1], l[2], 0.0f};
    c.gFrustum = {frustum[0], frustum[1], frustum[2], frustum[3]};
    c.gFrustumPrev = {frustumPrev[0], frustumPrev[1], frustumPrev[2], frustumPrev[3]};
    c.gCameraDelta = {cameraDelta.x, cameraDelta.y, cameraDelta.z, 0.0f};
    c.gMvScale = {cs.motionVectorScale[0], cs.motionVectorScale[1], cs.motionVectorScale[2], cs.isMotionVectorInWorldSpace ? 1.0f : 0.0f};
    c.gResourceSizeInv = {1.0f / resourceW, 1.0f / resourceH};
    c.gResourceSizeInvPrev = {1.0f / resourceWprev, 1.0f / resourceHprev};
    c.gRectSize = {rectW, rectH};
    c.gRectSizeInv = {1.0f / rectW, 1.0f / rectH};
    c.gRectSizePrev = {rectWprev, rectHprev};
    c.gResolutionScale = {rectW / resourceW, rectH / resourceH};
    c.gRectOffset = {float(cs.rectOrigin[0]) / resourceW, float(cs.rectOrigin[1]) / resourceH};
    c.gPrintfAt = {cs.printfAt[0], cs.printfAt[1]};
    c.gRectOrigin = {cs.rectOrigin[0], cs.rectOrigin[1]};
    c.gRectSizeMinusOne = {int32_t(cs.rectSize[0]) - 1, int32_t(cs.rectSize[1]) - 1};
    c.gTilesSizeMinusOne = {int32_t(tilesW) - 1, int32_t(tilesH) - 1};
    c.gOrthoMode = orthoMode;
    c.gUnproject = unproject;
    c.gDenoisingRange = cs.denoisingRange;
    c.gPlaneDistSensitivity = s.planeDistanceSensitivity;
    c.gStabilizationStrength = cs.accumulationMode == AccumulationMode::CONTINUE ? stabilizationStrength : 0.0f;
    c.gDebug = cs.debug;
    c.gSplitScreen = cs.splitScreen;
    c.gViewZScale = cs.viewZScale;
    c.gMinRectDimMulUnproject = std::min(rectW, rectH) * unproject;
    c.gFrameIndex = cs.frameIndex;
    c.gIsRectChanged = isRectChanged ? 1 : 0;
}

void InstanceImpl::Add_SigmaShadow(DenoiserData& d, bool translucent) {
    d.settings.sigma = SigmaSettings();
    d.settingsSize = sizeof(SigmaSettings);
    // the reference's host-side struct of the shared constants holds 16-byte aligned float4 members, so its sizeof -- the constantBufferDataSize of every SIGMA dispatch --
    // is the 516 bytes of fields rounded up to 528 (Sigma.cpp:93-96 SharedConstants; what a D3D constant buffer of this block occupies too); the tail is zero
    const uint32_t constSize = (uint32_t(sizeof(nrdc::SigmaConstants)) + 15u) & ~15u;
    const Format shadowFormat = translucent ? Format::RGBA8_UNORM : Format::R8_UNORM;
    char family[40], passName[96], shader[96];
    snprintf(family, sizeof(family), "%s", translucent ? "SIGMA_ShadowTranslucency" : "SIGMA_Shadow");
    auto Pass = [&](const char* what) {
        snprintf(passName, sizeof(passName), "%s - %s", family, what);
        BeginPass(InternString(passName));
    };
    auto Shader = [&](const char* stage) {
        snprintf(shader, sizeof(shader), "%s_%s.cs", family, stage);
        return (const char*)shader;
    };

    AddPermanent(Format::R32_UINT); // viewZ (29 bits) packed with 3 bits of history length

    AddTransient(Format::R16_SFLOAT);      // DATA_1: penumbra after pass 1
    AddTransient(Format::R16_SFLOAT);      // DATA_2: penumbra after pass 2
    AddTransient(shadowFormat);            // TEMP_1: shadow (+ translucency) after pass 1
    AddTransient(shadowFormat);            // TEMP_2: shadow after pass 2
    AddTransient(shadowFormat);            // HISTORY: copy of the previous output
    AddTransient(Format::R32_UINT);        // HISTORY_LENGTH copy
    AddTransient(Format::RGBA8_UNORM, 16); // TILES
    AddTransient(Format::RG8_UNORM, 16);   // SMOOTHED_TILES

    Pass("Classify tiles");
    In(ResourceType::IN_VIEWZ);
    In(ResourceType::IN_PENUMBRA);
    if (translucent) In(ResourceType::IN_TRANSLUCENCY);
    Out(T_TILES);
    EndPass(Shader("ClassifyTiles"), 16, 16, constSize);

    Pass("Smooth tiles");
    In(T_TILES);
    Out(T_SMOOTHED_TILES);
    EndPass("SIGMA_SmoothTiles.cs", 16, 16, constSize, 16);

    Pass("Copy");
    In(T_SMOOTHED_TILES);
    In(ResourceType::OUT_SHADOW_TRANSLUCENCY);
    In(P_HISTORY_LENGTH);
    Out(T_HISTORY);
    Out(T_HISTORY_LENGTH);
    EndPass("SIGMA_Copy.cs", 8, 16, constSize, USE_MAX_DIMS);

    Pass("Blur");
    In(ResourceType::IN_VIEWZ);
    In(ResourceType::IN_NORMAL_ROUGHNESS);
    In(ResourceType::IN_PENUMBRA);
    In(T_SMOOTHED_TILES);
    if (translucent) In(ResourceType::IN_TRANSLUCENCY);
    Out(T_DATA_1);
    Out(T_TEMP_1);
    EndPass(Shader("Blur"), 8, 16, constSize, translucent ? 1 : USE_MAX_DIMS); // sic: the two reference tables differ here

    for (int i = 0; i < 2; i++) {
        bool isStabilizationEnabled = i & 1;
        Pass("Post-blur");
        In(ResourceType::IN_VIEWZ);
        In(ResourceType::IN_NORMAL_ROUGHNESS);
        In(T_DATA_1);
        In(T_SMOOTHED_TILES);
        In(T_TEMP_1);
        Out(T_DATA_2);
        Out(isStabilizationEnabled ? (uint16_t)T_TEMP_2 : (uint16_t)ResourceType::OUT_SHADOW_TRANSLUCENCY);
        EndPass(Shader("PostBlur"), 8, 16, constSize);
    }

    Pass("Temporal stabilization");
    In(ResourceType::IN_VIEWZ);
    In(ResourceType::IN_MV);
    In(T_DATA_2);
    In(T_TEMP_2);
    In(T_HISTORY);
    In(T_HISTORY_LENGTH);
    In(T_SMOOTHED_TILES);
    Out(ResourceType::OUT_SHADOW_TRANSLUCENCY);
    Out(P_HISTORY_LENGTH);
    EndPass(Shader("TemporalStabilization"), 8, 16, constSize);

    Pass("Split screen");
    In(ResourceType::IN_VIEWZ);
    In(ResourceType::IN_PENUMBRA);
    if (translucent) In(ResourceType::IN_TRANSLUCENCY);
    Out(ResourceType::OUT_SHADOW_TRANSLUCENCY);
    EndPass(Shader("SplitScreen"), 8, 16, constSize);
}

void InstanceImpl::Update_SigmaShadow(const DenoiserData& d) {
    const SigmaSettings& s = d.settings.sigma;
    auto Emit = [&](uint32_t localIndex) {
        FillSigmaConstants(s, m_CommonSettings, PushDispatch(d, localIndex), m_WorldToView, m_ViewToClip, m_WorldToClipPrev, m_WorldToViewPrev, m_Rotator,
            m_RotatorPost, m_ViewDirection, m_Frustum, m_FrustumPrev, m_CameraDelta, m_OrthoMode, m_ProjectY);
    };

    if (m_CommonSettings.splitScreen >= 1.0f) {
        Emit(PASS_SPLIT_SCREEN);
        return;
    }

    Emit(PASS_CLASSIFY_TILES);
    Emit(PASS_SMOOTH_TILES);
    if (s.maxStabilizedFrameNum)
        Emit(PASS_COPY);
    Emit(PASS_BLUR);
    Emit(PASS_POST_BLUR + (s.maxStabilizedFrameNum ? 1 : 0));
    if (s.maxStabilizedFrameNum)
        Emit(PASS_TEMPORAL_STABILIZATION);
    if (m_CommonSettings.splitScreen > 0.0f)
        Emit(PASS_SPLIT_SCREEN);
}

} // namespace nrd
