// REBLUR host tables for the radiance+hit-distance family: REBLUR_DIFFUSE, REBLUR_SPECULAR and
// REBLUR_DIFFUSE_SPECULAR, generated from ONE parametrised description (the reference spells the three out in
// Source/Denoisers/Reblur_{Diffuse,Specular,DiffuseSpecular}.hpp). Pool layouts, pass order, resource binding
// order, permutation order (=> local pass indices and pipeline indices) and the per-frame permutation selection
// are the reference's: Reblur.cpp:38-64 (formats), :104-210 (Update_Reblur), :297-406 (shared constants).
#include "instance.h"

#include <algorithm>
#include <cstdio>

namespace nrd {

namespace {

// Permutation counts: reference Reblur.cpp:26-30
constexpr uint32_t HITDIST_RECONSTRUCTION_PERMUTATIONS = 4;
constexpr uint32_t PREPASS_PERMUTATIONS = 2;
constexpr uint32_t TEMPORAL_ACCUMULATION_PERMUTATIONS = 8;
constexpr uint32_t POST_BLUR_PERMUTATIONS = 2;
constexpr uint32_t TEMPORAL_STABILIZATION_PERMUTATIONS = 2;

// Local pass indices; every pass except the first and the last two exists as {quality, performance} pair
enum : uint32_t {
    PASS_CLASSIFY_TILES = 0,
    PASS_HITDIST_RECONSTRUCTION = PASS_CLASSIFY_TILES + 1,
    PASS_PREPASS = PASS_HITDIST_RECONSTRUCTION + HITDIST_RECONSTRUCTION_PERMUTATIONS * 2,
    PASS_TEMPORAL_ACCUMULATION = PASS_PREPASS + PREPASS_PERMUTATIONS * 2,
    PASS_HISTORY_FIX = PASS_TEMPORAL_ACCUMULATION + TEMPORAL_ACCUMULATION_PERMUTATIONS * 2,
    PASS_BLUR = PASS_HISTORY_FIX + 2,
    PASS_POST_BLUR = PASS_BLUR + 2,
    PASS_TEMPORAL_STABILIZATION = PASS_POST_BLUR + POST_BLUR_PERMUTATIONS * 2,
    PASS_SPLIT_SCREEN = PASS_TEMPORAL_STABILIZATION + TEMPORAL_STABILIZATION_PERMUTATIONS * 2,
    PASS_VALIDATION = PASS_SPLIT_SCREEN + 1,
};

constexpr Format FMT_SIGNAL = Format::RGBA16_SFLOAT;           // YCoCg radiance + normalised hit distance
constexpr Format FMT_FAST = Format::R16_SFLOAT;                // fast-history luma
constexpr Format FMT_PREV_VIEWZ = Format::R32_SFLOAT;
// follows the library's normal encoding (reference Reblur.cpp:52-62)
constexpr Format FMT_PREV_NORMAL_ROUGHNESS = NRD_NORMAL_ENCODING == 0 ? Format::RGBA8_UNORM : NRD_NORMAL_ENCODING == 1 ? Format::RGBA8_SNORM : NRD_NORMAL_ENCODING == 2 ? Format::R10_G10_B10_A2_UNORM
    : NRD_NORMAL_ENCODING == 3 ? Format::RGBA16_UNORM : Format::RGBA16_SFLOAT;
constexpr Format FMT_PREV_INTERNAL_DATA = Format::R16_UINT;    // 6+6 bits of accumulated frames, 4 bits material id
constexpr Format FMT_TILES = Format::R8_UNORM;
constexpr Format FMT_HITDIST_FOR_TRACKING = Format::R16_SFLOAT;

const uint16_t DUMMY = (uint16_t)ResourceType::IN_VIEWZ; // placeholder bound to optional, unused inputs

} // namespace

// sh: the REBLUR_*_SH denoisers (reference Source/Denoisers/Reblur_{Diffuse,Specular,DiffuseSpecular}Sh.hpp): every signal carries a second
// RGBA16F plane (SH1) through all passes; inputs / outputs are the IN_/OUT_*_SH0 and _SH1 slots
// directionalOcclusion: REBLUR_DIFFUSE_DIRECTIONAL_OCCLUSION (reference Denoisers/Reblur_DiffuseDirectionalOcclusion.hpp) = the diffuse tables
// with RGBA16_SNORM signal planes, an R16_UNORM fast history and the IN_/OUT_DIFF_DIRECTION_HITDIST slots
void InstanceImpl::Add_Reblur(DenoiserData& d, bool hasDiff, bool hasSpec, bool sh, bool directionalOcclusion) {
    d.settings.reblur = ReblurSettings();
    d.settingsSize = sizeof(ReblurSettings);

    const char* baseFamily = hasDiff && hasSpec ? "DiffuseSpecular" : (hasDiff ? "Diffuse" : "Specular");
    const char* family = !sh ? baseFamily : (hasDiff && hasSpec ? "DiffuseSpecularSh" : (hasDiff ? "DiffuseSh" : "SpecularSh"));
    if (directionalOcclusion)
        family = "DiffuseDirectionalOcclusion";
    const Format FMT_SIGNAL = directionalOcclusion ? Format::RGBA16_SNORM : nrd::FMT_SIGNAL;
    const Format FMT_FAST = directionalOcclusion ? Format::R16_UNORM : nrd::FMT_FAST;
    const uint32_t constSize = sizeof(nrdc::ReblurConstants);

    // ---- permanent planes (history)
    uint16_t next = PERMANENT_POOL_START;
    const uint16_t P_PREV_VIEWZ = next++;
    const uint16_t P_PREV_NORMAL_ROUGHNESS = next++;
    const uint16_t P_PREV_INTERNAL_DATA = next++;
    AddPermanent(FMT_PREV_VIEWZ);
    AddPermanent(FMT_PREV_NORMAL_ROUGHNESS);
    AddPermanent(FMT_PREV_INTERNAL_DATA);

    uint16_t P_DIFF_HISTORY = 0, P_DIFF_FAST = 0, P_DIFF_STAB_PING = 0, P_DIFF_STAB_PONG = 0;
    if (hasDiff) {
        P_DIFF_HISTORY = next++;
        P_DIFF_FAST = next++;
        P_DIFF_STAB_PING = next++;
        P_DIFF_STAB_PONG = next++;
        AddPermanent(FMT_SIGNAL);
        AddPermanent(FMT_FAST);
        AddPermanent(Format::R16_SFLOAT);
        AddPermanent(Format::R16_SFLOAT);
    }
    uint16_t P_DIFF_SH_HISTORY = 0, P_SPEC_SH_HISTORY = 0;
    if (hasDiff && sh) {
        P_DIFF_SH_HISTORY = next++;
        AddPermanent(FMT_SIGNAL);
    }
    uint16_t P_SPEC_HISTORY = 0, P_SPEC_FAST = 0, P_SPEC_STAB_PING = 0, P_SPEC_STAB_PONG = 0, P_SPEC_HDT_PING = 0, P_SPEC_HDT_PONG = 0;
    if (hasSpec) {
        P_SPEC_HISTORY = next++;
        P_SPEC_FAST = next++;
        P_SPEC_STAB_PING = next++;
        P_SPEC_STAB_PONG = next++;
        if (sh)
            P_SPEC_SH_HISTORY = next++;
        P_SPEC_HDT_PING = next++;
        P_SPEC_HDT_PONG = next++;
        AddPermanent(FMT_SIGNAL);
        AddPermanent(FMT_FAST);
        AddPermanent(Format::R16_SFLOAT);
        AddPermanent(Format::R16_SFLOAT);
        if (sh)
            AddPermanent(FMT_SIGNAL);
        AddPermanent(FMT_HITDIST_FOR_TRACKING);
        AddPermanent(FMT_HITDIST_FOR_TRACKING);
    }

    // ---- transient planes (scratch)
    next = TRANSIENT_POOL_START;
    const uint16_t T_DATA1 = next++;
    const uint16_t T_DATA2 = next++;
    AddTransient(hasDiff && hasSpec ? Format::RG8_UNORM : Format::R8_UNORM);
    AddTransient(hasSpec ? Format::R32_UINT : Format::R8_UINT); // diffuse-only keeps just the 4 occlusion bits
    uint16_t T_SPEC_HDT = 0;
    if (hasSpec) {
        T_SPEC_HDT = next++;
        AddTransient(FMT_HITDIST_FOR_TRACKING);
    }
    uint16_t T_DIFF_TMP2 = 0, T_DIFF_FAST = 0, T_SPEC_TMP2 = 0, T_SPEC_FAST = 0, T_DIFF_SH_TMP2 = 0, T_SPEC_SH_TMP2 = 0;
    if (hasDiff) {
        T_DIFF_TMP2 = next++;
        T_DIFF_FAST = next++;
        AddTransient(FMT_SIGNAL);
        AddTransient(FMT_FAST);
        if (sh) {
            T_DIFF_SH_TMP2 = next++;
            AddTransient(FMT_SIGNAL);
        }
    }
    if (hasSpec) {
        T_SPEC_TMP2 = next++;
        T_SPEC_FAST = next++;
        AddTransient(FMT_SIGNAL);
        AddTransient(FMT_FAST);
        if (sh) {
            T_SPEC_SH_TMP2 = next++;
            AddTransient(FMT_SIGNAL);
        }
    }
    // NRD_HIP_REFERENCE_QUIRKS: REBLUR_DIFFUSE_SPECULAR_SH as the reference describes it (Reblur_DiffuseSpecularSh.hpp:61-85) -- its Transient enum lists 10 planes, the code adds
    // 11 textures: a full-resolution REBLUR_FORMAT texture sits where the enum says TILES and the tile texture behind it is never named. The dispatches keep naming index TILES;
    // the HIP executor binds the real tile plane there (InstanceImpl::TransientAlias).
    const bool extraTextureOfTheReference = m_ReferenceQuirks && sh && hasDiff && hasSpec && !directionalOcclusion;
    const uint16_t T_TILES = next++;
    if (extraTextureOfTheReference) {
        AddTransient(FMT_SIGNAL);
        next++;
    }
    AddTransient(FMT_TILES, 16);
    if (extraTextureOfTheReference) { // (global pool indices: planes may be shared with the other denoisers of the instance, hence the alias is this denoiser's alone)
        m_TransientAliases.push_back(d.desc.identifier);
        m_TransientAliases.push_back(m_IndexRemap[m_IndexRemap.size() - 2]);
        m_TransientAliases.push_back(m_IndexRemap[m_IndexRemap.size() - 1]);
    }

    // The user-visible outputs double as scratch ("TEMP1")
    const uint16_t OUT_DIFF = (uint16_t)(directionalOcclusion ? ResourceType::OUT_DIFF_DIRECTION_HITDIST : (sh ? ResourceType::OUT_DIFF_SH0 : ResourceType::OUT_DIFF_RADIANCE_HITDIST));
    const uint16_t OUT_SPEC = (uint16_t)(sh ? ResourceType::OUT_SPEC_SH0 : ResourceType::OUT_SPEC_RADIANCE_HITDIST);
    const uint16_t DIFF_TEMP1 = OUT_DIFF, DIFF_TEMP2 = T_DIFF_TMP2;
    const uint16_t SPEC_TEMP1 = OUT_SPEC, SPEC_TEMP2 = T_SPEC_TMP2;
    const uint16_t IN_DIFF = (uint16_t)(directionalOcclusion ? ResourceType::IN_DIFF_DIRECTION_HITDIST : (sh ? ResourceType::IN_DIFF_SH0 : ResourceType::IN_DIFF_RADIANCE_HITDIST));
    const uint16_t IN_SPEC = (uint16_t)(sh ? ResourceType::IN_SPEC_SH0 : ResourceType::IN_SPEC_RADIANCE_HITDIST);
    // SH1 planes: the user outputs double as scratch here too
    const uint16_t IN_DIFF_SH = (uint16_t)ResourceType::IN_DIFF_SH1, IN_SPEC_SH = (uint16_t)ResourceType::IN_SPEC_SH1;
    const uint16_t DIFF_SH_TEMP1 = (uint16_t)ResourceType::OUT_DIFF_SH1, DIFF_SH_TEMP2 = T_DIFF_SH_TMP2;
    const uint16_t SPEC_SH_TEMP1 = (uint16_t)ResourceType::OUT_SPEC_SH1, SPEC_SH_TEMP2 = T_SPEC_SH_TMP2;
    const bool diffSh = hasDiff && sh, specSh = hasSpec && sh;

    char passName[96], shader[128];
    auto Pass = [&](const char* what) {
        snprintf(passName, sizeof(passName), "REBLUR_%s - %s", directionalOcclusion ? "DirectionalOcclusion" : family, what); // (sic: Reblur_DiffuseDirectionalOcclusion.hpp:13 DENOISER_NAME)
        BeginPass(InternString(passName));
    };
    // registers the {quality, performance} pair of a pass
    auto EndPair = [&](const char* pass, const char* suffix) {
        snprintf(shader, sizeof(shader), "REBLUR_%s_%s%s.cs", family, pass, suffix);
        EndPass(shader, 8, 16, constSize);
        snprintf(shader, sizeof(shader), "REBLUR_Perf_%s_%s%s.cs", family, pass, suffix);
        EndPass(shader, 8, 16, constSize);
    };

    Pass("Classify tiles");
    In(ResourceType::IN_VIEWZ);
    Out(T_TILES);
    EndPass("REBLUR_ClassifyTiles.cs", 16, 16, constSize);

    for (uint32_t i = 0; i < HITDIST_RECONSTRUCTION_PERMUTATIONS; i++) {
        bool is5x5 = (i >> 1) & 1, isPrepassEnabled = i & 1;
        Pass("Hit distance reconstruction");
        In(T_TILES);
        In(ResourceType::IN_NORMAL_ROUGHNESS);
        In(ResourceType::IN_VIEWZ);
        if (hasDiff) In(IN_DIFF);
        if (hasSpec) In(IN_SPEC);
        if (hasDiff) Out(isPrepassEnabled ? DIFF_TEMP2 : DIFF_TEMP1);
        if (hasSpec) Out(isPrepassEnabled ? SPEC_TEMP2 : SPEC_TEMP1);
        { // the SH family reuses the radiance family's reconstruction shaders (the hit distance lives in .w of SH0)
            const char* keep = family;
            family = baseFamily;
            EndPair("HitDistReconstruction", is5x5 ? "_5x5" : "");
            family = keep;
        }
    }

    for (uint32_t i = 0; i < PREPASS_PERMUTATIONS; i++) {
        bool isAfterReconstruction = i & 1;
        Pass("Pre-pass");
        In(T_TILES);
        In(ResourceType::IN_NORMAL_ROUGHNESS);
        In(ResourceType::IN_VIEWZ);
        if (hasDiff) In(isAfterReconstruction ? DIFF_TEMP2 : IN_DIFF);
        if (hasSpec) In(isAfterReconstruction ? SPEC_TEMP2 : IN_SPEC);
        if (diffSh) In(IN_DIFF_SH);
        if (specSh) In(IN_SPEC_SH);
        if (hasDiff) Out(DIFF_TEMP1);
        if (hasSpec) Out(SPEC_TEMP1);
        if (hasSpec) Out(T_SPEC_HDT);
        if (diffSh) Out(DIFF_SH_TEMP1);
        if (specSh) Out(SPEC_SH_TEMP1);
        EndPair("PrePass", "");
    }

    for (uint32_t i = 0; i < TEMPORAL_ACCUMULATION_PERMUTATIONS; i++) {
        bool hasDisocclusionThresholdMix = (i >> 2) & 1, hasConfidenceInputs = (i >> 1) & 1, isAfterPrepass = i & 1;
        Pass("Temporal accumulation");
        In(T_TILES);
        In(ResourceType::IN_NORMAL_ROUGHNESS);
        In(ResourceType::IN_VIEWZ);
        In(ResourceType::IN_MV);
        In(P_PREV_VIEWZ);
        In(P_PREV_NORMAL_ROUGHNESS);
        In(P_PREV_INTERNAL_DATA);
        In(hasDisocclusionThresholdMix ? (uint16_t)ResourceType::IN_DISOCCLUSION_THRESHOLD_MIX : DUMMY);
        if (hasDiff) In(hasConfidenceInputs ? (uint16_t)ResourceType::IN_DIFF_CONFIDENCE : DUMMY);
        if (hasSpec) In(hasConfidenceInputs ? (uint16_t)ResourceType::IN_SPEC_CONFIDENCE : DUMMY);
        if (hasDiff) In(isAfterPrepass ? DIFF_TEMP1 : IN_DIFF);
        if (hasSpec) In(isAfterPrepass ? SPEC_TEMP1 : IN_SPEC);
        if (hasDiff) In(P_DIFF_HISTORY);
        if (hasSpec) In(P_SPEC_HISTORY);
        if (hasDiff) In(P_DIFF_FAST);
        if (hasSpec) In(P_SPEC_FAST);
        if (hasSpec) In(P_SPEC_HDT_PING, P_SPEC_HDT_PONG);
        if (hasSpec) In(T_SPEC_HDT);
        if (diffSh) In(isAfterPrepass ? DIFF_SH_TEMP1 : IN_DIFF_SH);
        if (specSh) In(isAfterPrepass ? SPEC_SH_TEMP1 : IN_SPEC_SH);
        if (diffSh) In(P_DIFF_SH_HISTORY);
        if (specSh) In(P_SPEC_SH_HISTORY);
        if (hasDiff) Out(DIFF_TEMP2);
        if (hasSpec) Out(SPEC_TEMP2);
        if (hasDiff) Out(T_DIFF_FAST);
        if (hasSpec) Out(T_SPEC_FAST);
        if (hasSpec) Out(P_SPEC_HDT_PONG, P_SPEC_HDT_PING);
        Out(T_DATA1);
        Out(T_DATA2);
        if (diffSh) Out(DIFF_SH_TEMP2);
        if (specSh) Out(SPEC_SH_TEMP2);
        EndPair("TemporalAccumulation", "");
    }

    Pass("History fix");
    In(T_TILES);
    In(ResourceType::IN_NORMAL_ROUGHNESS);
    In(T_DATA1);
    In(ResourceType::IN_VIEWZ);
    if (hasDiff) In(DIFF_TEMP2);
    if (hasSpec) In(SPEC_TEMP2);
    if (hasDiff) In(T_DIFF_FAST);
    if (hasSpec) In(T_SPEC_FAST);
    if (diffSh) In(DIFF_SH_TEMP2);
    if (specSh) In(SPEC_SH_TEMP2);
    if (hasDiff) Out(DIFF_TEMP1);
    if (hasSpec) Out(SPEC_TEMP1);
    if (hasDiff) Out(P_DIFF_FAST);
    if (hasSpec) Out(P_SPEC_FAST);
    if (diffSh) Out(DIFF_SH_TEMP1);
    if (specSh) Out(SPEC_SH_TEMP1);
    EndPair("HistoryFix", "");

    Pass("Blur");
    In(T_TILES);
    In(ResourceType::IN_NORMAL_ROUGHNESS);
    In(T_DATA1);
    if (hasDiff) In(DIFF_TEMP1);
    if (hasSpec) In(SPEC_TEMP1);
    In(ResourceType::IN_VIEWZ);
    if (diffSh) In(DIFF_SH_TEMP1);
    if (specSh) In(SPEC_SH_TEMP1);
    if (hasDiff) Out(DIFF_TEMP2);
    if (hasSpec) Out(SPEC_TEMP2);
    Out(P_PREV_VIEWZ);
    if (diffSh) Out(DIFF_SH_TEMP2);
    if (specSh) Out(SPEC_SH_TEMP2);
    EndPair("Blur", "");

    for (uint32_t i = 0; i < POST_BLUR_PERMUTATIONS; i++) {
        bool isTemporalStabilization = i & 1;
        Pass("Post-blur");
        In(T_TILES);
        In(ResourceType::IN_NORMAL_ROUGHNESS);
        In(T_DATA1);
        if (hasDiff) In(DIFF_TEMP2);
        if (hasSpec) In(SPEC_TEMP2);
        In(P_PREV_VIEWZ);
        if (diffSh) In(DIFF_SH_TEMP2);
        if (specSh) In(SPEC_SH_TEMP2);
        Out(P_PREV_NORMAL_ROUGHNESS);
        if (hasDiff) Out(P_DIFF_HISTORY);
        if (hasSpec) Out(P_SPEC_HISTORY);
        if (!isTemporalStabilization) {
            Out(P_PREV_INTERNAL_DATA);
            if (hasDiff) Out(OUT_DIFF);
            if (hasSpec) Out(OUT_SPEC);
            if (diffSh) Out(ResourceType::OUT_DIFF_SH1);
            if (specSh) Out(ResourceType::OUT_SPEC_SH1);
        }
        if (diffSh) Out(P_DIFF_SH_HISTORY);
        if (specSh) Out(P_SPEC_SH_HISTORY);
        EndPair("PostBlur", isTemporalStabilization ? "" : "_NoTemporalStabilization");
    }

    for (uint32_t i = 0; i < TEMPORAL_STABILIZATION_PERMUTATIONS; i++) {
        bool hasRf0AndMetalness = i & 1;
        Pass("Temporal stabilization");
        In(T_TILES);
        In(ResourceType::IN_NORMAL_ROUGHNESS);
        if (hasSpec) In(hasRf0AndMetalness ? (uint16_t)ResourceType::IN_BASECOLOR_METALNESS : DUMMY);
        In(P_PREV_VIEWZ);
        In(T_DATA1);
        In(T_DATA2);
        if (hasDiff) In(P_DIFF_HISTORY);
        if (hasSpec) In(P_SPEC_HISTORY);
        if (hasDiff) In(P_DIFF_STAB_PING, P_DIFF_STAB_PONG);
        if (hasSpec) In(P_SPEC_STAB_PING, P_SPEC_STAB_PONG);
        if (hasSpec) In(P_SPEC_HDT_PONG, P_SPEC_HDT_PING);
        if (diffSh) In(P_DIFF_SH_HISTORY);
        if (specSh) In(P_SPEC_SH_HISTORY);
        Out(ResourceType::IN_MV); // optionally patched in place (specular MV modification)
        Out(P_PREV_INTERNAL_DATA);
        if (hasDiff) Out(OUT_DIFF);
        if (hasSpec) Out(OUT_SPEC);
        if (hasDiff) Out(P_DIFF_STAB_PONG, P_DIFF_STAB_PING);
        if (hasSpec) Out(P_SPEC_STAB_PONG, P_SPEC_STAB_PING);
        if (diffSh) Out(ResourceType::OUT_DIFF_SH1);
        if (specSh) Out(ResourceType::OUT_SPEC_SH1);
        EndPair("TemporalStabilization", "");
    }

    Pass("Split screen");
    In(ResourceType::IN_VIEWZ);
    if (hasDiff) In(IN_DIFF);
    if (hasSpec) In(IN_SPEC);
    if (diffSh) In(IN_DIFF_SH);
    if (specSh) In(IN_SPEC_SH);
    if (hasDiff) Out(OUT_DIFF);
    if (hasSpec) Out(OUT_SPEC);
    if (diffSh) Out(ResourceType::OUT_DIFF_SH1);
    if (specSh) Out(ResourceType::OUT_SPEC_SH1);
    snprintf(shader, sizeof(shader), "REBLUR_%s_SplitScreen.cs", directionalOcclusion ? baseFamily : family); // sic: the radiance family's shader
    EndPass(shader, 8, 16, constSize);

    Pass("Validation");
    In(ResourceType::IN_NORMAL_ROUGHNESS);
    In(ResourceType::IN_VIEWZ);
    In(ResourceType::IN_MV);
    In(T_DATA1);
    In(T_DATA2);
    In(hasDiff ? IN_DIFF : IN_SPEC); // a single-signal denoiser binds its input twice (REBLUR_ADD_VALIDATION_DISPATCH call sites)
    In(hasSpec ? IN_SPEC : IN_DIFF);
    Out(ResourceType::OUT_VALIDATION);
    EndPass("REBLUR_Validation.cs", 8, 16, (uint32_t(sizeof(nrdc::ReblurValidationConstants)) + 15u) & ~15u, IGNORE_RS); // (rounded like the reference host struct)
}

void InstanceImpl::Update_Reblur(const DenoiserData& d) {
    const ReblurSettings& s = d.settings.reblur;
    const CommonSettings& cs = m_CommonSettings;
    const bool hasDiff = d.desc.denoiser != Denoiser::REBLUR_SPECULAR && d.desc.denoiser != Denoiser::REBLUR_SPECULAR_SH; // incl. REBLUR_DIFFUSE_DIRECTIONAL_OCCLUSION
    const bool hasSpec = d.desc.denoiser != Denoiser::REBLUR_DIFFUSE && d.desc.denoiser != Denoiser::REBLUR_DIFFUSE_SH && d.desc.denoiser != Denoiser::REBLUR_DIFFUSE_DIRECTIONAL_OCCLUSION;

    const bool enableHitDistanceReconstruction = s.hitDistanceReconstructionMode != HitDistanceReconstructionMode::OFF && s.checkerboardMode == CheckerboardMode::OFF;
    const bool skipTemporalStabilization = s.maxStabilizedFrameNum == 0;
    const bool skipPrePass = (s.diffusePrepassBlurRadius == 0.0f || !hasDiff) && (s.specularPrepassBlurRadius == 0.0f || !hasSpec) && s.checkerboardMode == CheckerboardMode::OFF;
    const uint32_t perf = s.enablePerformanceMode ? 1 : 0;

    auto Emit = [&](uint32_t localIndex) { FillReblurConstants(s, PushDispatch(d, localIndex)); };

    if (cs.splitScreen >= 1.0f) { // pure passthrough
        Emit(PASS_SPLIT_SCREEN);
        return;
    }

    Emit(PASS_CLASSIFY_TILES);

    if (enableHitDistanceReconstruction)
        Emit(PASS_HITDIST_RECONSTRUCTION + (s.hitDistanceReconstructionMode == HitDistanceReconstructionMode::AREA_5X5 ? 4 : 0) + (!skipPrePass ? 2 : 0) + perf);

    if (!skipPrePass)
        Emit(PASS_PREPASS + (enableHitDistanceReconstruction ? 2 : 0) + perf);

    Emit(PASS_TEMPORAL_ACCUMULATION + (cs.isDisocclusionThresholdMixAvailable ? 8 : 0) + (cs.isHistoryConfidenceAvailable ? 4 : 0) +
         ((!skipPrePass || enableHitDistanceReconstruction) ? 2 : 0) + perf);
    Emit(PASS_HISTORY_FIX + perf);
    Emit(PASS_BLUR + perf);
    Emit(PASS_POST_BLUR + (skipTemporalStabilization ? 0 : 2) + perf);

    if (!skipTemporalStabilization)
        Emit(PASS_TEMPORAL_STABILIZATION + (cs.isBaseColorMetalnessAvailable ? 2 : 0) + perf);

    if (cs.splitScreen > 0.0f)
        Emit(PASS_SPLIT_SCREEN);

    if (cs.enableValidation) {
        auto* c = (nrdc::ReblurValidationConstants*)PushDispatch(d, PASS_VALIDATION);
        FillReblurConstants(s, c);
        c->gHasDiffuse = hasDiff ? 1 : 0; // PushDispatch never returns null (arena overflow lands in the scratch block and fails the whole call)
        c->gHasSpecular = hasSpec ? 1 : 0;
    }
}

// ================================================================================================ occlusion-only family
// REBLUR_DIFFUSE_OCCLUSION / _SPECULAR_OCCLUSION / _DIFFUSE_SPECULAR_OCCLUSION: the signal is the normalised hit distance alone
// (R16_UNORM planes), there is no pre-pass and no temporal stabilisation. Tables: reference
// Source/Denoisers/Reblur_{Diffuse,Specular,DiffuseSpecular}Occlusion.hpp; per-frame selection: Reblur.cpp:212-296.
namespace {
constexpr uint32_t OCC_HITDIST_RECONSTRUCTION_PERMUTATIONS = 2;
constexpr uint32_t OCC_TEMPORAL_ACCUMULATION_PERMUTATIONS = 8;
enum : uint32_t {
    OCC_PASS_CLASSIFY_TILES = 0,
    OCC_PASS_HITDIST_RECONSTRUCTION = OCC_PASS_CLASSIFY_TILES + 1,
    OCC_PASS_TEMPORAL_ACCUMULATION = OCC_PASS_HITDIST_RECONSTRUCTION + OCC_HITDIST_RECONSTRUCTION_PERMUTATIONS * 2,
    OCC_PASS_HISTORY_FIX = OCC_PASS_TEMPORAL_ACCUMULATION + OCC_TEMPORAL_ACCUMULATION_PERMUTATIONS * 2,
    OCC_PASS_BLUR = OCC_PASS_HISTORY_FIX + 2,
    OCC_PASS_POST_BLUR = OCC_PASS_BLUR + 2,
    OCC_PASS_SPLIT_SCREEN = OCC_PASS_POST_BLUR + 2,
    OCC_PASS_VALIDATION = OCC_PASS_SPLIT_SCREEN + 1,
};
constexpr Format FMT_OCCLUSION = Format::R16_UNORM;
constexpr Format FMT_OCCLUSION_FAST = Format::R16_UNORM;
} // namespace

void InstanceImpl::Add_ReblurOcclusion(DenoiserData& d, bool hasDiff, bool hasSpec) {
    d.settings.reblur = ReblurSettings();
    d.settingsSize = sizeof(ReblurSettings);

    const char* family = hasDiff && hasSpec ? "DiffuseSpecularOcclusion" : (hasDiff ? "DiffuseOcclusion" : "SpecularOcclusion");
    const char* splitScreenFamily = hasDiff && hasSpec ? "DiffuseSpecular" : (hasDiff ? "Diffuse" : "Specular"); // sic: the radiance family's shader
    const uint32_t constSize = sizeof(nrdc::ReblurConstants);

    uint16_t next = PERMANENT_POOL_START;
    const uint16_t P_PREV_VIEWZ = next++;
    const uint16_t P_PREV_NORMAL_ROUGHNESS = next++;
    const uint16_t P_PREV_INTERNAL_DATA = next++;
    AddPermanent(FMT_PREV_VIEWZ);
    AddPermanent(FMT_PREV_NORMAL_ROUGHNESS);
    AddPermanent(FMT_PREV_INTERNAL_DATA);
    uint16_t P_DIFF_FAST = 0, P_SPEC_FAST = 0, P_SPEC_HDT_PING = 0, P_SPEC_HDT_PONG = 0;
    if (hasDiff) {
        P_DIFF_FAST = next++;
        AddPermanent(FMT_OCCLUSION_FAST);
    }
    if (hasSpec) {
        P_SPEC_FAST = next++;
        P_SPEC_HDT_PING = next++;
        P_SPEC_HDT_PONG = next++;
        AddPermanent(FMT_OCCLUSION_FAST);
        AddPermanent(FMT_HITDIST_FOR_TRACKING);
        AddPermanent(FMT_HITDIST_FOR_TRACKING);
    }

    next = TRANSIENT_POOL_START;
    const uint16_t T_DATA1 = next++;
    AddTransient(hasDiff && hasSpec ? Format::RG8_UNORM : Format::R8_UNORM);
    uint16_t T_DIFF_TMP2 = 0, T_DIFF_FAST = 0, T_SPEC_TMP2 = 0, T_SPEC_FAST = 0;
    if (hasDiff) {
        T_DIFF_TMP2 = next++;
        T_DIFF_FAST = next++;
        AddTransient(FMT_OCCLUSION);
        AddTransient(FMT_OCCLUSION_FAST);
    }
    if (hasSpec) {
        T_SPEC_TMP2 = next++;
        T_SPEC_FAST = next++;
        AddTransient(FMT_OCCLUSION);
        AddTransient(FMT_OCCLUSION_FAST);
    }
    const uint16_t T_TILES = next++;
    AddTransient(FMT_TILES, 16);

    const uint16_t DIFF_TEMP1 = (uint16_t)ResourceType::OUT_DIFF_HITDIST, DIFF_TEMP2 = T_DIFF_TMP2;
    const uint16_t SPEC_TEMP1 = (uint16_t)ResourceType::OUT_SPEC_HITDIST, SPEC_TEMP2 = T_SPEC_TMP2;
    const uint16_t IN_DIFF = (uint16_t)ResourceType::IN_DIFF_HITDIST, IN_SPEC = (uint16_t)ResourceType::IN_SPEC_HITDIST;

    char passName[96], shader[128];
    auto Pass = [&](const char* what) {
        snprintf(passName, sizeof(passName), "REBLUR_%s - %s", family, what);
        BeginPass(InternString(passName));
    };
    auto EndPair = [&](const char* pass, const char* suffix) {
        snprintf(shader, sizeof(shader), "REBLUR_%s_%s%s.cs", family, pass, suffix);
        EndPass(shader, 8, 16, constSize);
        snprintf(shader, sizeof(shader), "REBLUR_Perf_%s_%s%s.cs", family, pass, suffix);
        EndPass(shader, 8, 16, constSize);
    };

    Pass("Classify tiles");
    In(ResourceType::IN_VIEWZ);
    Out(T_TILES);
    EndPass("REBLUR_ClassifyTiles.cs", 16, 16, constSize);

    for (uint32_t i = 0; i < OCC_HITDIST_RECONSTRUCTION_PERMUTATIONS; i++) {
        bool is5x5 = i & 1;
        Pass("Hit distance reconstruction");
        In(T_TILES);
        In(ResourceType::IN_NORMAL_ROUGHNESS);
        In(ResourceType::IN_VIEWZ);
        if (hasDiff) In(IN_DIFF);
        if (hasSpec) In(IN_SPEC);
        if (hasDiff) Out(DIFF_TEMP1);
        if (hasSpec) Out(SPEC_TEMP1);
        EndPair("HitDistReconstruction", is5x5 ? "_5x5" : "");
    }

    for (uint32_t i = 0; i < OCC_TEMPORAL_ACCUMULATION_PERMUTATIONS; i++) {
        bool hasDisocclusionThresholdMix = (i >> 2) & 1, hasConfidenceInputs = (i >> 1) & 1, isAfterReconstruction = i & 1;
        Pass("Temporal accumulation");
        In(T_TILES);
        In(ResourceType::IN_NORMAL_ROUGHNESS);
        In(ResourceType::IN_VIEWZ);
        In(ResourceType::IN_MV);
        In(P_PREV_VIEWZ);
        In(P_PREV_NORMAL_ROUGHNESS);
        In(P_PREV_INTERNAL_DATA);
        In(hasDisocclusionThresholdMix ? (uint16_t)ResourceType::IN_DISOCCLUSION_THRESHOLD_MIX : DUMMY);
        if (hasDiff) In(hasConfidenceInputs ? (uint16_t)ResourceType::IN_DIFF_CONFIDENCE : DUMMY);
        if (hasSpec) In(hasConfidenceInputs ? (uint16_t)ResourceType::IN_SPEC_CONFIDENCE : DUMMY);
        if (hasDiff) In(isAfterReconstruction ? DIFF_TEMP1 : IN_DIFF);
        if (hasSpec) In(isAfterReconstruction ? SPEC_TEMP1 : IN_SPEC);
        if (hasDiff) In(ResourceType::OUT_DIFF_HITDIST); // the previous output is the history
        if (hasSpec) In(ResourceType::OUT_SPEC_HITDIST);
        if (hasDiff) In(P_DIFF_FAST);
        if (hasSpec) In(P_SPEC_FAST);
        if (hasSpec) In(P_SPEC_HDT_PING, P_SPEC_HDT_PONG);
        if (hasDiff) Out(DIFF_TEMP2);
        if (hasSpec) Out(SPEC_TEMP2);
        if (hasDiff) Out(T_DIFF_FAST);
        if (hasSpec) Out(T_SPEC_FAST);
        if (hasSpec) Out(P_SPEC_HDT_PONG, P_SPEC_HDT_PING);
        Out(T_DATA1);
        EndPair("TemporalAccumulation", "");
    }

    Pass("History fix");
    In(T_TILES);
    In(ResourceType::IN_NORMAL_ROUGHNESS);
    In(T_DATA1);
    In(ResourceType::IN_VIEWZ);
    if (hasDiff) In(DIFF_TEMP2);
    if (hasSpec) In(SPEC_TEMP2);
    if (hasDiff) In(T_DIFF_FAST);
    if (hasSpec) In(T_SPEC_FAST);
    if (hasDiff) Out(DIFF_TEMP1);
    if (hasSpec) Out(SPEC_TEMP1);
    if (hasDiff) Out(P_DIFF_FAST);
    if (hasSpec) Out(P_SPEC_FAST);
    EndPair("HistoryFix", "");

    Pass("Blur");
    In(T_TILES);
    In(ResourceType::IN_NORMAL_ROUGHNESS);
    In(T_DATA1);
    if (hasDiff) In(DIFF_TEMP1);
    if (hasSpec) In(SPEC_TEMP1);
    In(ResourceType::IN_VIEWZ);
    if (hasDiff) Out(DIFF_TEMP2);
    if (hasSpec) Out(SPEC_TEMP2);
    Out(P_PREV_VIEWZ);
    EndPair("Blur", "");

    Pass("Post-blur");
    In(T_TILES);
    In(ResourceType::IN_NORMAL_ROUGHNESS);
    In(T_DATA1);
    if (hasDiff) In(DIFF_TEMP2);
    if (hasSpec) In(SPEC_TEMP2);
    In(P_PREV_VIEWZ);
    Out(P_PREV_NORMAL_ROUGHNESS);
    if (hasDiff) Out(ResourceType::OUT_DIFF_HITDIST);
    if (hasSpec) Out(ResourceType::OUT_SPEC_HITDIST);
    Out(P_PREV_INTERNAL_DATA);
    EndPair("PostBlur", "_NoTemporalStabilization");

    Pass("Split screen");
    In(ResourceType::IN_VIEWZ);
    if (hasDiff) In(IN_DIFF);
    if (hasSpec) In(IN_SPEC);
    if (hasDiff) Out(ResourceType::OUT_DIFF_HITDIST);
    if (hasSpec) Out(ResourceType::OUT_SPEC_HITDIST);
    snprintf(shader, sizeof(shader), "REBLUR_%s_SplitScreen.cs", splitScreenFamily);
    EndPass(shader, 8, 16, constSize);

    Pass("Validation");
    In(ResourceType::IN_NORMAL_ROUGHNESS);
    In(ResourceType::IN_VIEWZ);
    In(ResourceType::IN_MV);
    In(T_DATA1);
    In(T_DATA1); // sic: no DATA2 in the occlusion family
    In(hasDiff ? IN_DIFF : IN_SPEC);
    In(hasSpec ? IN_SPEC : IN_DIFF);
    Out(ResourceType::OUT_VALIDATION);
    EndPass("REBLUR_Validation.cs", 8, 16, (uint32_t(sizeof(nrdc::ReblurValidationConstants)) + 15u) & ~15u, IGNORE_RS); // (rounded like the reference host struct)
}

void InstanceImpl::Update_ReblurOcclusion(const DenoiserData& d) {
    const ReblurSettings& s = d.settings.reblur;
    const CommonSettings& cs = m_CommonSettings;
    const bool hasDiff = d.desc.denoiser != Denoiser::REBLUR_SPECULAR_OCCLUSION;
    const bool hasSpec = d.desc.denoiser != Denoiser::REBLUR_DIFFUSE_OCCLUSION;
    const bool enableHitDistanceReconstruction = s.hitDistanceReconstructionMode != HitDistanceReconstructionMode::OFF && s.checkerboardMode == CheckerboardMode::OFF;
    const uint32_t perf = s.enablePerformanceMode ? 1 : 0;

    auto Emit = [&](uint32_t localIndex) { FillReblurConstants(s, PushDispatch(d, localIndex)); };

    if (cs.splitScreen >= 1.0f) {
        Emit(OCC_PASS_SPLIT_SCREEN);
        return;
    }
    Emit(OCC_PASS_CLASSIFY_TILES);
    if (enableHitDistanceReconstruction)
        Emit(OCC_PASS_HITDIST_RECONSTRUCTION + (s.hitDistanceReconstructionMode == HitDistanceReconstructionMode::AREA_5X5 ? 2 : 0) + perf);
    Emit(OCC_PASS_TEMPORAL_ACCUMULATION + (cs.isDisocclusionThresholdMixAvailable ? 8 : 0) + (cs.isHistoryConfidenceAvailable ? 4 : 0) + (enableHitDistanceReconstruction ? 2 : 0) + perf);
    Emit(OCC_PASS_HISTORY_FIX + (!s.enableAntiFirefly ? 1 : 0)); // sic (reference Reblur.cpp:264): the permutation follows enableAntiFirefly, not the performance mode
    Emit(OCC_PASS_BLUR + perf);
    Emit(OCC_PASS_POST_BLUR + perf);
    if (cs.splitScreen > 0.0f)
        Emit(OCC_PASS_SPLIT_SCREEN);
    if (cs.enableValidation) {
        auto* c = (nrdc::ReblurValidationConstants*)PushDispatch(d, OCC_PASS_VALIDATION);
        FillReblurConstants(s, c);
        c->gHasDiffuse = hasDiff ? 1 : 0; // PushDispatch never returns null (arena overflow lands in the scratch block and fails the whole call)
        c->gHasSpecular = hasSpec ? 1 : 0;
    }
}

static void StoreMatrix(float* dst, const nrdhost::Mat4& m) { memcpy(dst, &m, sizeof(float) * 16); }

void InstanceImpl::FillReblurConstants(const ReblurSettings& s, void* data) {
    if (!data)
        return;
    const CommonSettings& cs = m_CommonSettings;

    const float resourceW = cs.resourceSize[0], resourceH = cs.resourceSize[1];
    const float resourceWprev = cs.resourceSizePrev[0], resourceHprev = cs.resourceSizePrev[1];
    const float rectW = cs.rectSize[0], rectH = cs.rectSize[1];
    const float rectWprev = cs.rectSizePrev[0], rectHprev = cs.rectSizePrev[1];

    const bool isRectChanged = cs.rectSize[0] != cs.rectSizePrev[0] || cs.rectSize[1] != cs.rectSizePrev[1];
    const bool isHistoryReset = cs.accumulationMode != AccumulationMode::CONTINUE;
    const float unproject = 1.0f / (0.5f * rectH * m_ProjectY);
    const float worstResolutionScale = std::min(rectW / resourceW, rectH / resourceH);
    const float maxBlurRadius = s.maxBlurRadius * worstResolutionScale;
    const float disocclusionThresholdBonus = (1.0f + m_JitterDelta) / rectH;
    const float stabilizationStrength = float(s.maxStabilizedFrameNum) / (1.0f + float(s.maxStabilizedFrameNum));
    const float hitDistStabilizationStrength = float(s.maxStabilizedFrameNumForHitDistance) / (1.0f + float(s.maxStabilizedFrameNumForHitDistance));
    const uint32_t maxAccumulatedFrameNum = std::min(s.maxAccumulatedFrameNum, REBLUR_MAX_HISTORY_FRAME_NUM);

    uint32_t diffCheckerboard = 2, specCheckerboard = 2;
    if (s.checkerboardMode == CheckerboardMode::BLACK) {
        diffCheckerboard = 0;
        specCheckerboard = 1;
    } else if (s.checkerboardMode == CheckerboardMode::WHITE) {
        diffCheckerboard = 1;
        specCheckerboard = 0;
    }

    nrdc::ReblurConstants& c = *(nrdc::ReblurConstants*)data;
    StoreMatrix(c.gWorldToClip, m_WorldToClip);
    StoreMatrix(c.gViewToClip, m_ViewToClip);
    StoreMatrix(c.gViewToWorld, m_ViewToWorld);
    StoreMatrix(c.gWorldToViewPrev, m_WorldToViewPrev);
    StoreMatrix(c.gWorldToClipPrev, m_WorldToClipPrev);
    StoreMatrix(c.gWorldPrevToWorld, m_WorldPrevToWorld);
    c.gRotatorPre = {m_RotatorPre[0], m_RotatorPre[1], m_RotatorPre[2], m_RotatorPre[3]};
    c.gRotator = {m_Rotator[0], m_Rotator[1], m_Rotator[2], m_Rotator[3]};
    c.gRotatorPost = {m_RotatorPost[0], m_RotatorPost[1], m_RotatorPost[2], m_RotatorPost[3]};
    c.gFrustum = {m_Frustum[0], m_Frustum[1], m_Frustum[2], m_Frustum[3]};
    c.gFrustumPrev = {m_FrustumPrev[0], m_FrustumPrev[1], m_FrustumPrev[2], m_FrustumPrev[3]};
    c.gCameraDelta = {m_CameraDelta.x, m_CameraDelta.y, m_CameraDelta.z, 0.0f};
    c.gHitDistParams = {s.hitDistanceParameters.A, s.hitDistanceParameters.B, s.hitDistanceParameters.C, s.hitDistanceParameters.D};
    c.gViewVectorWorld = {m_ViewDirection.x, m_ViewDirection.y, m_ViewDirection.z, 0.0f};
    c.gViewVectorWorldPrev = {m_ViewDirectionPrev.x, m_ViewDirectionPrev.y, m_ViewDirectionPrev.z, 0.0f};
    c.gMvScale = {cs.motionVectorScale[0], cs.motionVectorScale[1], cs.motionVectorScale[2], cs.isMotionVectorInWorldSpace ? 1.0f : 0.0f};
    c.gAntilagParams = {s.antilagSettings.luminanceSigmaScale, s.antilagSettings.luminanceSensitivity};
    c.gResourceSize = {resourceW, resourceH};
    c.gResourceSizeInv = {1.0f / resourceW, 1.0f / resourceH};
    c.gResourceSizeInvPrev = {1.0f / resourceWprev, 1.0f / resourceHprev};
    c.gRectSize = {rectW, rectH};
    c.gRectSizeInv = {1.0f / rectW, 1.0f / rectH};
    c.gRectSizePrev = {rectWprev, rectHprev};
    c.gResolutionScale = {rectW / resourceW, rectH / resourceH};
    c.gResolutionScalePrev = {rectWprev / resourceWprev, rectHprev / resourceHprev};
    c.gRectOffset = {float(cs.rectOrigin[0]) / resourceW, float(cs.rectOrigin[1]) / resourceH};
    c.gSpecProbabilityThresholdsForMvModification = {cs.isBaseColorMetalnessAvailable ? s.specularProbabilityThresholdsForMvModification[0] : 2.0f,
        cs.isBaseColorMetalnessAvailable ? s.specularProbabilityThresholdsForMvModification[1] : 3.0f};
    c.gJitter = {cs.cameraJitter[0], cs.cameraJitter[1]};
    c.gPrintfAt = {cs.printfAt[0], cs.printfAt[1]};
    c.gRectOrigin = {cs.rectOrigin[0], cs.rectOrigin[1]};
    c.gRectSizeMinusOne = {int32_t(cs.rectSize[0]) - 1, int32_t(cs.rectSize[1]) - 1};
    c.gDisocclusionThreshold = cs.disocclusionThreshold + disocclusionThresholdBonus;
    c.gDisocclusionThresholdAlternate = cs.disocclusionThresholdAlternate + disocclusionThresholdBonus;
    c.gCameraAttachedReflectionMaterialID = cs.cameraAttachedReflectionMaterialID;
    c.gStrandMaterialID = cs.strandMaterialID;
    c.gStrandThickness = cs.strandThickness;
    c.gStabilizationStrength = isHistoryReset ? 0.0f : stabilizationStrength;
    c.gHitDistStabilizationStrength = isHistoryReset ? 0.0f : hitDistStabilizationStrength; // set but unused by any pass (as in the reference)
    c.gDebug = cs.debug;
    c.gOrthoMode = m_OrthoMode;
    c.gUnproject = unproject;
    c.gDenoisingRange = cs.denoisingRange;
    c.gPlaneDistSensitivity = s.planeDistanceSensitivity;
    c.gFramerateScale = m_FrameRateScale;
    c.gMaxBlurRadius = std::max(maxBlurRadius, s.minBlurRadius);
    c.gMinBlurRadius = s.minBlurRadius;
    c.gDiffPrepassBlurRadius = s.diffusePrepassBlurRadius * worstResolutionScale;
    c.gSpecPrepassBlurRadius = s.specularPrepassBlurRadius * worstResolutionScale;
    c.gMaxAccumulatedFrameNum = isHistoryReset ? 0.0f : float(maxAccumulatedFrameNum);
    c.gMaxFastAccumulatedFrameNum = isHistoryReset ? 0.0f : float(s.maxFastAccumulatedFrameNum);
    c.gAntiFirefly = s.enableAntiFirefly ? 1.0f : 0.0f;
    c.gLobeAngleFraction = s.lobeAngleFraction * s.lobeAngleFraction; // squared: reference Reblur.cpp:384
    c.gRoughnessFraction = s.roughnessFraction;
    c.gResponsiveAccumulationRoughnessThreshold = s.responsiveAccumulationRoughnessThreshold;
    c.gHistoryFixFrameNum = float(s.historyFixFrameNum);
    c.gHistoryFixBasePixelStride = float(s.historyFixBasePixelStride);
    c.gMinRectDimMulUnproject = std::min(rectW, rectH) * unproject;
    c.gUsePrepassNotOnlyForSpecularMotionEstimation = s.usePrepassOnlyForSpecularMotionEstimation ? 0.0f : 1.0f;
    c.gSplitScreen = cs.splitScreen;
    c.gSplitScreenPrev = m_SplitScreenPrev;
    c.gCheckerboardResolveAccumSpeed = m_CheckerboardResolveAccumSpeed;
    c.gViewZScale = cs.viewZScale;
    c.gFireflySuppressorMinRelativeScale = s.fireflySuppressorMinRelativeScale;
    c.gMinHitDistanceWeight = s.minHitDistanceWeight;
    c.gDiffMinMaterial = s.minMaterialForDiffuse;
    c.gSpecMinMaterial = s.minMaterialForSpecular;
    c.gHasHistoryConfidence = cs.isHistoryConfidenceAvailable ? 1 : 0;
    c.gHasDisocclusionThresholdMix = cs.isDisocclusionThresholdMixAvailable ? 1 : 0;
    c.gDiffCheckerboard = diffCheckerboard;
    c.gSpecCheckerboard = specCheckerboard;
    c.gFrameIndex = cs.frameIndex;
    c.gIsRectChanged = isRectChanged ? 1 : 0;
    c.gResetHistory = isHistoryReset ? 1 : 0;
}

} // namespace nrd
