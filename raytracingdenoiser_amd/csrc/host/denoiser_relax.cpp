// RELAX host tables for all six variants (RELAX_DIFFUSE / _SPECULAR / _DIFFUSE_SPECULAR, each with and without SH),
// generated from ONE parametrised description: the reference spells them out in six files
// (reference Source/Denoisers/Relax_{Diffuse,Specular,DiffuseSpecular}{,Sh}.hpp) that differ only by dropping the
// planes of the absent signal and by appending the SH1 planes at the end of every input / output list.
// Per-frame pass selection: reference Source/Relax.cpp:179-301; shared constants: Relax.cpp:58-177.
#include "instance.h"

#include <algorithm>
#include <cmath>
#include <cstdio>

namespace nrd {

namespace {

constexpr uint32_t HITDIST_RECONSTRUCTION_PERMUTATIONS = 2;
constexpr uint32_t PREPASS_PERMUTATIONS = 2;
constexpr uint32_t TEMPORAL_ACCUMULATION_PERMUTATIONS = 4;
constexpr uint32_t ATROUS_PERMUTATIONS = 2;
constexpr uint32_t ATROUS_BINDING_VARIANTS = 5; // smem first iteration, odd, even, odd-last, even-last
constexpr uint32_t MAX_ATROUS_PASS_NUM = 8;

enum : uint32_t {
    PASS_CLASSIFY_TILES = 0,
    PASS_HITDIST_RECONSTRUCTION = PASS_CLASSIFY_TILES + 1,
    PASS_PREPASS = PASS_HITDIST_RECONSTRUCTION + HITDIST_RECONSTRUCTION_PERMUTATIONS,
    PASS_TEMPORAL_ACCUMULATION = PASS_PREPASS + PREPASS_PERMUTATIONS,
    PASS_HISTORY_FIX = PASS_TEMPORAL_ACCUMULATION + TEMPORAL_ACCUMULATION_PERMUTATIONS,
    PASS_HISTORY_CLAMPING,
    PASS_COPY,
    PASS_ANTI_FIREFLY,
    PASS_ATROUS,
    PASS_SPLIT_SCREEN = PASS_ATROUS + ATROUS_PERMUTATIONS * ATROUS_BINDING_VARIANTS,
    PASS_VALIDATION,
};

const uint16_t DUMMY = (uint16_t)ResourceType::IN_VIEWZ;

struct Variant {
    bool hasDiff, hasSpec, sh;
};

Variant VariantOf(Denoiser d) {
    switch (d) {
        case Denoiser::RELAX_DIFFUSE: return {true, false, false};
        case Denoiser::RELAX_DIFFUSE_SH: return {true, false, true};
        case Denoiser::RELAX_SPECULAR: return {false, true, false};
        case Denoiser::RELAX_SPECULAR_SH: return {false, true, true};
        case Denoiser::RELAX_DIFFUSE_SPECULAR: return {true, true, false};
        default: return {true, true, true};
    }
}

} // namespace

void InstanceImpl::Add_Relax(DenoiserData& d, bool hasDiff, bool hasSpec, bool sh) {
    d.settings.relax = RelaxSettings();
    d.settingsSize = sizeof(RelaxSettings);
    const uint32_t constSize = sizeof(nrdc::RelaxConstants);
    // (the reference's host struct holds 16-byte aligned float4 members: its sizeof -- the dispatch's constantBufferDataSize -- is the 712 bytes of fields rounded up to 720)
    const uint32_t atrousConstSize = (uint32_t(sizeof(nrdc::RelaxAtrousConstants)) + 15u) & ~15u;

    // ---- pools; "spec" planes precede "diff" planes, an SH1 plane directly follows its SH0 plane
    uint16_t nextPermanent = PERMANENT_POOL_START, nextTransient = TRANSIENT_POOL_START;
    auto Permanent = [&](Format f) {
        AddPermanent(f);
        return nextPermanent++;
    };
    auto Transient = [&](Format f, uint16_t downsample = 1) {
        AddTransient(f, downsample);
        return nextTransient++;
    };
    struct Signal { // one radiance signal: plane ids of SH0 and (optionally) SH1
        uint16_t prev = 0, prevSh = 0, responsivePrev = 0, responsivePrevSh = 0; // permanent
        uint16_t ping = 0, pingSh = 0, pong = 0, pongSh = 0;                       // transient
        uint16_t in0 = 0, in1 = 0, out0 = 0, out1 = 0, confidence = 0;            // user planes
    } spec, diff;

    auto AddSignalHistory = [&](Signal& s) {
        s.prev = Permanent(Format::RGBA16_SFLOAT);
        if (sh) s.prevSh = Permanent(Format::RGBA16_SFLOAT);
        s.responsivePrev = Permanent(Format::RGBA16_SFLOAT);
        if (sh) s.responsivePrevSh = Permanent(Format::RGBA16_SFLOAT);
    };
    if (hasSpec && hasDiff && !sh) { // sic: the one variant whose table interleaves the two signals (Relax_DiffuseSpecular.hpp:19-22; found by tests/test_ref_host.py)
        spec.prev = Permanent(Format::RGBA16_SFLOAT);
        diff.prev = Permanent(Format::RGBA16_SFLOAT);
        spec.responsivePrev = Permanent(Format::RGBA16_SFLOAT);
        diff.responsivePrev = Permanent(Format::RGBA16_SFLOAT);
    } else {
        if (hasSpec) AddSignalHistory(spec);
        if (hasDiff) AddSignalHistory(diff);
    }
    uint16_t P_HIT_T_CURR = 0, P_HIT_T_PREV = 0;
    if (hasSpec) {
        P_HIT_T_CURR = Permanent(Format::R16_SFLOAT);
        P_HIT_T_PREV = Permanent(Format::R16_SFLOAT);
    }
    const uint16_t P_HISTORY_LENGTH_PREV = Permanent(Format::R8_UNORM);
    const uint16_t P_NORMAL_ROUGHNESS_PREV = Permanent(Format::RGBA8_UNORM);
    const uint16_t P_MATERIAL_ID_PREV = Permanent(Format::R8_UNORM);
    const uint16_t P_VIEWZ_PREV = Permanent(Format::R32_SFLOAT);

    auto AddSignalScratch = [&](Signal& s) {
        s.ping = Transient(Format::RGBA16_SFLOAT);
        if (sh) s.pingSh = Transient(Format::RGBA16_SFLOAT);
        s.pong = Transient(Format::RGBA16_SFLOAT);
        if (sh) s.pongSh = Transient(Format::RGBA16_SFLOAT);
    };
    if (hasSpec) AddSignalScratch(spec);
    if (hasDiff) AddSignalScratch(diff);
    uint16_t T_SPEC_REPROJECTION_CONFIDENCE = 0;
    if (hasSpec) T_SPEC_REPROJECTION_CONFIDENCE = Transient(Format::R8_UNORM);
    const uint16_t T_TILES = Transient(Format::R8_UNORM, 16);
    const uint16_t T_HISTORY_LENGTH = Transient(Format::R8_UNORM);

    spec.in0 = (uint16_t)(sh ? ResourceType::IN_SPEC_SH0 : ResourceType::IN_SPEC_RADIANCE_HITDIST);
    spec.in1 = (uint16_t)ResourceType::IN_SPEC_SH1;
    spec.out0 = (uint16_t)(sh ? ResourceType::OUT_SPEC_SH0 : ResourceType::OUT_SPEC_RADIANCE_HITDIST);
    spec.out1 = (uint16_t)ResourceType::OUT_SPEC_SH1;
    spec.confidence = (uint16_t)ResourceType::IN_SPEC_CONFIDENCE;
    diff.in0 = (uint16_t)(sh ? ResourceType::IN_DIFF_SH0 : ResourceType::IN_DIFF_RADIANCE_HITDIST);
    diff.in1 = (uint16_t)ResourceType::IN_DIFF_SH1;
    diff.out0 = (uint16_t)(sh ? ResourceType::OUT_DIFF_SH0 : ResourceType::OUT_DIFF_RADIANCE_HITDIST);
    diff.out1 = (uint16_t)ResourceType::OUT_DIFF_SH1;
    diff.confidence = (uint16_t)ResourceType::IN_DIFF_CONFIDENCE;

    // ---- naming
    const char* signals = hasDiff && hasSpec ? "DiffuseSpecular" : (hasDiff ? "Diffuse" : "Specular");
    char family[32], familyNoSh[32], passName[96], shader[96];
    snprintf(family, sizeof(family), "%s%s", signals, sh ? "Sh" : "");
    snprintf(familyNoSh, sizeof(familyNoSh), "%s", signals);
    auto Pass = [&](const char* what) {
        snprintf(passName, sizeof(passName), "RELAX_%s - %s", family, what);
        BeginPass(InternString(passName));
    };
    auto Shader = [&](const char* fam, const char* stage) {
        snprintf(shader, sizeof(shader), "RELAX_%s_%s.cs", fam, stage);
        return (const char*)shader;
    };
    // binds "what" of spec, then of diff (present signals only)
    auto InBoth = [&](uint16_t Signal::*what) {
        if (hasSpec) In(spec.*what);
        if (hasDiff) In(diff.*what);
    };
    auto OutBoth = [&](uint16_t Signal::*what) {
        if (hasSpec) Out(spec.*what);
        if (hasDiff) Out(diff.*what);
    };

    // ---- passes (order defines the local pass indices used by Update_Relax)
    Pass("Classify tiles");
    In(ResourceType::IN_VIEWZ);
    Out(T_TILES);
    EndPass("RELAX_ClassifyTiles.cs", 16, 16, constSize);

    for (uint32_t i = 0; i < HITDIST_RECONSTRUCTION_PERMUTATIONS; i++) {
        bool is5x5 = i & 1;
        Pass("Hit distance reconstruction");
        In(T_TILES);
        InBoth(&Signal::in0);
        In(ResourceType::IN_NORMAL_ROUGHNESS);
        In(ResourceType::IN_VIEWZ);
        OutBoth(&Signal::ping);
        EndPass(Shader(familyNoSh, is5x5 ? "HitDistReconstruction_5x5" : "HitDistReconstruction"), 8, 8, constSize);
    }

    for (uint32_t i = 0; i < PREPASS_PERMUTATIONS; i++) {
        bool isAfterReconstruction = i & 1;
        Pass("Pre-pass");
        In(T_TILES);
        InBoth(isAfterReconstruction ? &Signal::ping : &Signal::in0);
        In(ResourceType::IN_NORMAL_ROUGHNESS);
        In(ResourceType::IN_VIEWZ);
        if (sh) InBoth(&Signal::in1);
        OutBoth(&Signal::out0);
        if (sh) OutBoth(&Signal::out1);
        EndPass(Shader(family, "PrePass"), 16, 16, constSize);
    }

    for (uint32_t i = 0; i < TEMPORAL_ACCUMULATION_PERMUTATIONS; i++) {
        bool hasDisocclusionThresholdMix = (i >> 1) & 1, hasConfidenceInputs = i & 1;
        Pass("Temporal accumulation");
        In(T_TILES);
        InBoth(&Signal::out0);
        In(ResourceType::IN_MV);
        In(ResourceType::IN_NORMAL_ROUGHNESS);
        In(ResourceType::IN_VIEWZ);
        InBoth(&Signal::responsivePrev);
        InBoth(&Signal::prev);
        In(P_NORMAL_ROUGHNESS_PREV);
        In(P_VIEWZ_PREV);
        if (hasSpec) In(P_HIT_T_PREV, P_HIT_T_CURR);
        In(P_HISTORY_LENGTH_PREV);
        In(P_MATERIAL_ID_PREV);
        if (hasSpec) In(hasConfidenceInputs ? spec.confidence : DUMMY);
        if (hasDiff) In(hasConfidenceInputs ? diff.confidence : DUMMY);
        In(hasDisocclusionThresholdMix ? (uint16_t)ResourceType::IN_DISOCCLUSION_THRESHOLD_MIX : DUMMY);
        if (sh) {
            InBoth(&Signal::out1);
            InBoth(&Signal::responsivePrevSh);
            InBoth(&Signal::prevSh);
        }
        OutBoth(&Signal::ping);
        OutBoth(&Signal::pong);
        if (hasSpec) Out(P_HIT_T_CURR, P_HIT_T_PREV);
        Out(T_HISTORY_LENGTH);
        if (hasSpec) Out(T_SPEC_REPROJECTION_CONFIDENCE);
        if (sh) {
            OutBoth(&Signal::pingSh);
            OutBoth(&Signal::pongSh);
        }
        EndPass(Shader(family, "TemporalAccumulation"), 8, 16, constSize);
    }

    Pass("History fix");
    In(T_TILES);
    InBoth(&Signal::ping); // normal history
    In(T_HISTORY_LENGTH);
    In(ResourceType::IN_NORMAL_ROUGHNESS);
    In(ResourceType::IN_VIEWZ);
    if (sh) InBoth(&Signal::pingSh);
    OutBoth(&Signal::pong); // responsive history
    if (sh) OutBoth(&Signal::pongSh);
    EndPass(Shader(family, "HistoryFix"), 8, 8, constSize);

    Pass("History clamping");
    In(T_TILES);
    In(ResourceType::IN_VIEWZ);
    InBoth(&Signal::out0); // noisy input after the pre-pass
    InBoth(&Signal::ping);
    InBoth(&Signal::pong);
    In(T_HISTORY_LENGTH);
    if (sh) {
        InBoth(&Signal::pingSh);
        InBoth(&Signal::pongSh);
    }
    OutBoth(&Signal::prev);
    OutBoth(&Signal::responsivePrev);
    Out(P_HISTORY_LENGTH_PREV);
    if (sh) {
        OutBoth(&Signal::prevSh);
        OutBoth(&Signal::responsivePrevSh);
    }
    EndPass(Shader(family, "HistoryClamping"), 8, 8, constSize);

    Pass("Copy");
    InBoth(&Signal::prev);
    OutBoth(&Signal::out0);
    EndPass(Shader(family, "Copy"), 8, 8, constSize);

    Pass("Anti-firefly");
    In(T_TILES);
    InBoth(&Signal::out0);
    In(ResourceType::IN_NORMAL_ROUGHNESS);
    In(ResourceType::IN_VIEWZ);
    OutBoth(&Signal::prev);
    EndPass(Shader(family, "AntiFirefly"), 8, 8, constSize);

    for (uint32_t i = 0; i < ATROUS_PERMUTATIONS; i++) {
        bool hasConfidenceInputs = i & 1;
        for (uint32_t j = 0; j < ATROUS_BINDING_VARIANTS; j++) {
            bool isSmem = j == 0, isEven = j % 2 == 0, isLast = j > 2;
            Pass(isSmem ? "A-trous (SMEM)" : "A-trous");
            In(T_TILES);
            InBoth(isSmem ? &Signal::prev : (isEven ? &Signal::pong : &Signal::ping));
            In(T_HISTORY_LENGTH);
            if (hasSpec) In(T_SPEC_REPROJECTION_CONFIDENCE);
            In(ResourceType::IN_NORMAL_ROUGHNESS);
            In(ResourceType::IN_VIEWZ);
            if (hasSpec) In(hasConfidenceInputs ? spec.confidence : DUMMY);
            if (hasDiff) In(hasConfidenceInputs ? diff.confidence : DUMMY);
            if (sh) InBoth(isSmem ? &Signal::prevSh : (isEven ? &Signal::pongSh : &Signal::pingSh));
            OutBoth(isLast ? &Signal::out0 : (isEven ? &Signal::ping : &Signal::pong));
            if (isSmem) {
                Out(P_NORMAL_ROUGHNESS_PREV);
                Out(P_MATERIAL_ID_PREV);
                Out(P_VIEWZ_PREV);
            }
            if (sh) OutBoth(isLast ? &Signal::out1 : (isEven ? &Signal::pingSh : &Signal::pongSh));
            if (isSmem)
                EndPass(Shader(family, "AtrousSmem"), 8, 8, atrousConstSize);
            else
                EndPass(Shader(family, "Atrous"), 16, 16, atrousConstSize, 1, isLast ? 1 : (MAX_ATROUS_PASS_NUM - 2 + 1) / 2);
        }
    }

    Pass("Split screen");
    In(ResourceType::IN_VIEWZ);
    if (hasDiff) In(diff.in0);
    if (hasSpec) In(spec.in0);
    if (sh && hasDiff) In(diff.in1);
    if (sh && hasSpec) In(spec.in1);
    if (hasDiff) Out(diff.out0);
    if (hasSpec) Out(spec.out0);
    if (sh && hasDiff) Out(diff.out1);
    if (sh && hasSpec) Out(spec.out1);
    EndPass(Shader(family, "SplitScreen"), 8, 16, constSize);

    Pass("Validation");
    In(ResourceType::IN_NORMAL_ROUGHNESS);
    In(ResourceType::IN_VIEWZ);
    In(ResourceType::IN_MV);
    In(T_HISTORY_LENGTH);
    Out(ResourceType::OUT_VALIDATION);
    EndPass("RELAX_Validation.cs", 8, 16, constSize, IGNORE_RS);
}

void InstanceImpl::FillRelaxConstants(const RelaxSettings& s, void* data) {
    if (!data)
        return;
    const CommonSettings& cs = m_CommonSettings;
    const float resourceW = cs.resourceSize[0], resourceH = cs.resourceSize[1];
    const float resourceWprev = cs.resourceSizePrev[0], resourceHprev = cs.resourceSizePrev[1];
    const float rectW = cs.rectSize[0], rectH = cs.rectSize[1];
    const float rectWprev = cs.rectSizePrev[0], rectHprev = cs.rectSizePrev[1];

    // World-space frustum basis: X(world) = viewZ * (forward + right * clipX - up * clipY), camera-relative
    auto Basis = [](const nrdhost::Mat4& viewToClip, const nrdhost::Mat4& worldToView, const nrdhost::Mat4& viewToWorld, const nrdhost::Vec4& frustum,
                     nrdc::F4& right, nrdc::F4& up, nrdc::F4& forward) {
        const float tanHalfFov = 1.0f / viewToClip.at(0, 0);
        const float aspect = viewToClip.at(0, 0) / viewToClip.at(1, 1);
        right = {worldToView.at(0, 0) * tanHalfFov, worldToView.at(0, 1) * tanHalfFov, worldToView.at(0, 2) * tanHalfFov, 0.0f};
        up = {worldToView.at(1, 0) * tanHalfFov * aspect, worldToView.at(1, 1) * tanHalfFov * aspect, worldToView.at(1, 2) * tanHalfFov * aspect, 0.0f};
        // view-space direction through the centre of the (possibly asymmetric) frustum, z = 1, rotated to world
        const float fv[4] = {0.5f * frustum[2] + frustum[0], 0.5f * frustum[3] + frustum[1], 1.0f, 0.0f};
        float f[3];
        for (int i = 0; i < 3; i++)
            f[i] = viewToWorld.at(i, 0) * fv[0] + viewToWorld.at(i, 1) * fv[1] + viewToWorld.at(i, 2) * fv[2] + viewToWorld.at(i, 3) * fv[3];
        forward = {f[0], f[1], f[2], 0.0f};
    };

    auto Saturate = [](float x) { return std::min(std::max(x, 0.0f), 1.0f); };
    const float disocclusionThresholdBonus = (1.0f + m_JitterDelta) / rectH;
    const bool isHistoryReset = cs.accumulationMode != AccumulationMode::CONTINUE;
    auto FrameNum = [&](uint32_t n) { return isHistoryReset ? 0.0f : float(std::min(n, RELAX_MAX_HISTORY_FRAME_NUM)); };

    uint32_t specCheckerboard = 2, diffCheckerboard = 2;
    if (s.checkerboardMode == CheckerboardMode::BLACK) {
        diffCheckerboard = 0;
        specCheckerboard = 1;
    } else if (s.checkerboardMode == CheckerboardMode::WHITE) {
        diffCheckerboard = 1;
        specCheckerboard = 0;
    }

    nrdc::RelaxConstants& c = *(nrdc::RelaxConstants*)data;
    memcpy(c.gWorldToClip, &m_WorldToClip, 64);
    memcpy(c.gWorldToClipPrev, &m_WorldToClipPrev, 64);
    memcpy(c.gWorldToViewPrev, &m_WorldToViewPrev, 64);
    memcpy(c.gWorldPrevToWorld, &m_WorldPrevToWorld, 64);
    c.gRotatorPre = {m_RotatorPre[0], m_RotatorPre[1], m_RotatorPre[2], m_RotatorPre[3]};
    Basis(m_ViewToClip, m_WorldToView, m_ViewToWorld, m_Frustum, c.gFrustumRight, c.gFrustumUp, c.gFrustumForward);
    Basis(m_ViewToClipPrev, m_WorldToViewPrev, m_ViewToWorldPrev, m_FrustumPrev, c.gPrevFrustumRight, c.gPrevFrustumUp, c.gPrevFrustumForward);
    c.gCameraDelta = {m_CameraDelta.x, m_CameraDelta.y, m_CameraDelta.z, 0.0f};
    c.gMvScale = {cs.motionVectorScale[0], cs.motionVectorScale[1], cs.motionVectorScale[2], cs.isMotionVectorInWorldSpace ? 1.0f : 0.0f};
    c.gJitter = {cs.cameraJitter[0], cs.cameraJitter[1]};
    c.gResolutionScale = {rectW / resourceW, rectH / resourceH};
    c.gRectOffset = {float(cs.rectOrigin[0]) / resourceW, float(cs.rectOrigin[1]) / resourceH};
    c.gResourceSizeInv = {1.0f / resourceW, 1.0f / resourceH};
    c.gResourceSize = {resourceW, resourceH};
    c.gRectSizeInv = {1.0f / rectW, 1.0f / rectH};
    c.gRectSizePrev = {rectWprev, rectHprev};
    c.gResourceSizeInvPrev = {1.0f / resourceWprev, 1.0f / resourceHprev};
    c.gPrintfAt = {cs.printfAt[0], cs.printfAt[1]};
    c.gRectOrigin = {cs.rectOrigin[0], cs.rectOrigin[1]};
    c.gRectSize = {int32_t(cs.rectSize[0]), int32_t(cs.rectSize[1])};
    c.gSpecMaxAccumulatedFrameNum = FrameNum(s.specularMaxAccumulatedFrameNum);
    c.gSpecMaxFastAccumulatedFrameNum = FrameNum(s.specularMaxFastAccumulatedFrameNum);
    c.gDiffMaxAccumulatedFrameNum = FrameNum(s.diffuseMaxAccumulatedFrameNum);
    c.gDiffMaxFastAccumulatedFrameNum = FrameNum(s.diffuseMaxFastAccumulatedFrameNum);
    c.gDisocclusionThreshold = cs.disocclusionThreshold + disocclusionThresholdBonus;
    c.gDisocclusionThresholdAlternate = cs.disocclusionThresholdAlternate + disocclusionThresholdBonus;
    c.gCameraAttachedReflectionMaterialID = cs.cameraAttachedReflectionMaterialID;
    c.gStrandMaterialID = cs.strandMaterialID;
    c.gStrandThickness = cs.strandThickness;
    c.gRoughnessFraction = s.roughnessFraction;
    c.gSpecVarianceBoost = s.specularVarianceBoost;
    c.gSplitScreen = cs.splitScreen;
    c.gDiffBlurRadius = s.diffusePrepassBlurRadius;
    c.gSpecBlurRadius = s.specularPrepassBlurRadius;
    c.gDepthThreshold = s.depthThreshold;
    c.gLobeAngleFraction = s.lobeAngleFraction;
    c.gSpecLobeAngleSlack = s.specularLobeAngleSlack * (3.14159265358979f / 180.0f);
    c.gHistoryFixEdgeStoppingNormalPower = s.historyFixEdgeStoppingNormalPower;
    c.gRoughnessEdgeStoppingRelaxation = s.roughnessEdgeStoppingRelaxation;
    c.gNormalEdgeStoppingRelaxation = s.normalEdgeStoppingRelaxation;
    c.gColorBoxSigmaScale = s.historyClampingColorBoxSigmaScale;
    c.gHistoryAccelerationAmount = s.antilagSettings.accelerationAmount;
    c.gHistoryResetTemporalSigmaScale = s.antilagSettings.temporalSigmaScale;
    c.gHistoryResetSpatialSigmaScale = s.antilagSettings.spatialSigmaScale;
    c.gHistoryResetAmount = s.antilagSettings.resetAmount;
    c.gDenoisingRange = cs.denoisingRange;
    c.gSpecPhiLuminance = s.specularPhiLuminance;
    c.gDiffPhiLuminance = s.diffusePhiLuminance;
    c.gDiffMaxLuminanceRelativeDifference = -logf(Saturate(s.diffuseMinLuminanceWeight));
    c.gSpecMaxLuminanceRelativeDifference = -logf(Saturate(s.specularMinLuminanceWeight));
    c.gLuminanceEdgeStoppingRelaxation = s.roughnessEdgeStoppingRelaxation; // sic: the reference feeds the roughness setting here (Relax.cpp:149)
    c.gConfidenceDrivenRelaxationMultiplier = s.confidenceDrivenRelaxationMultiplier;
    c.gConfidenceDrivenLuminanceEdgeStoppingRelaxation = s.confidenceDrivenLuminanceEdgeStoppingRelaxation;
    c.gConfidenceDrivenNormalEdgeStoppingRelaxation = s.confidenceDrivenNormalEdgeStoppingRelaxation;
    c.gDebug = cs.debug;
    c.gOrthoMode = m_OrthoMode;
    c.gUnproject = 1.0f / (0.5f * rectH * m_ProjectY);
    c.gFramerateScale = std::min(std::max(16.66f / m_TimeDelta, 0.25f), 4.0f);
    c.gCheckerboardResolveAccumSpeed = m_CheckerboardResolveAccumSpeed;
    c.gJitterDelta = m_JitterDelta;
    c.gHistoryFixFrameNum = float(s.historyFixFrameNum) + 1.0f;
    c.gHistoryFixBasePixelStride = float(s.historyFixBasePixelStride);
    c.gHistoryThreshold = float(s.spatialVarianceEstimationHistoryThreshold);
    c.gViewZScale = cs.viewZScale;
    c.gMinHitDistanceWeight = s.minHitDistanceWeight * 2.0f;
    c.gDiffMinMaterial = s.minMaterialForDiffuse;
    c.gSpecMinMaterial = s.minMaterialForSpecular;
    c.gRoughnessEdgeStoppingEnabled = s.enableRoughnessEdgeStopping ? 1 : 0;
    c.gFrameIndex = cs.frameIndex;
    c.gDiffCheckerboard = diffCheckerboard;
    c.gSpecCheckerboard = specCheckerboard;
    c.gHasHistoryConfidence = cs.isHistoryConfidenceAvailable ? 1 : 0;
    c.gHasDisocclusionThresholdMix = cs.isDisocclusionThresholdMixAvailable ? 1 : 0;
    c.gResetHistory = isHistoryReset ? 1 : 0;
}

void InstanceImpl::Update_Relax(const DenoiserData& d) {
    const RelaxSettings& s = d.settings.relax;
    const CommonSettings& cs = m_CommonSettings;
    const bool enableHitDistanceReconstruction = s.hitDistanceReconstructionMode != HitDistanceReconstructionMode::OFF && s.checkerboardMode == CheckerboardMode::OFF;
    const uint32_t iterationNum = std::min(std::max(s.atrousIterationNum, 2u), MAX_ATROUS_PASS_NUM);

    auto Emit = [&](uint32_t localIndex) { FillRelaxConstants(s, PushDispatch(d, localIndex)); };

    if (cs.splitScreen >= 1.0f) { // pure passthrough
        Emit(PASS_SPLIT_SCREEN);
        return;
    }

    Emit(PASS_CLASSIFY_TILES);
    if (enableHitDistanceReconstruction)
        Emit(PASS_HITDIST_RECONSTRUCTION + (s.hitDistanceReconstructionMode == HitDistanceReconstructionMode::AREA_5X5 ? 1 : 0));
    Emit(PASS_PREPASS + (enableHitDistanceReconstruction ? 1 : 0));
    Emit(PASS_TEMPORAL_ACCUMULATION + (cs.isDisocclusionThresholdMixAvailable ? 2 : 0) + (cs.isHistoryConfidenceAvailable ? 1 : 0));
    Emit(PASS_HISTORY_FIX);
    Emit(PASS_HISTORY_CLAMPING);

    if (s.enableAntiFirefly) {
        Emit(PASS_COPY);
        Emit(PASS_ANTI_FIREFLY);
    }

    // a-trous chain: history -> ping (smem), then ping <-> pong, the last iteration writes the user outputs
    for (uint32_t i = 0; i < iterationNum; i++) {
        uint32_t passIndex = PASS_ATROUS + (cs.isHistoryConfidenceAvailable ? ATROUS_BINDING_VARIANTS : 0);
        if (i != 0)
            passIndex += 2 - (i & 1);
        if (i == iterationNum - 1)
            passIndex += 2;

        void* data = PushDispatch(d, passIndex);
        FillRelaxConstants(s, data);
        if (data) {
            nrdc::RelaxAtrousConstants& c = *(nrdc::RelaxAtrousConstants*)data;
            c.gStepSize = 1u << i;
            c.gIsLastPass = i == iterationNum - 1 ? 1 : 0;
        }
    }

    if (cs.splitScreen > 0.0f)
        Emit(PASS_SPLIT_SCREEN);
    if (cs.enableValidation)
        Emit(PASS_VALIDATION);
}

bool IsRelax(Denoiser d) { return d >= Denoiser::RELAX_DIFFUSE && d <= Denoiser::RELAX_DIFFUSE_SPECULAR_SH; }

void InstanceImpl::Add_RelaxVariant(DenoiserData& d) {
    Variant v = VariantOf(d.desc.denoiser);
    Add_Relax(d, v.hasDiff, v.hasSpec, v.sh);
}

} // namespace nrd
