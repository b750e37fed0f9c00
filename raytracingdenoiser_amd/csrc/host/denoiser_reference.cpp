// REFERENCE denoiser: a running mean of IN_SIGNAL with an optional split-screen copy to OUT_SIGNAL.
// Pass table and per-frame logic follow reference Source/Denoisers/Reference.hpp:13-90.
#include "instance.h"

#include <algorithm>

namespace nrd {

namespace {
enum : uint16_t { P_HISTORY = PERMANENT_POOL_START };
enum : uint32_t { PASS_ACCUMULATE, PASS_COPY };
} // namespace

void InstanceImpl::Add_Reference(DenoiserData& d) {
    d.settings.reference = ReferenceSettings();
    d.settingsSize = sizeof(ReferenceSettings);

    AddPermanent(Format::RGBA32_SFLOAT);

    BeginPass("Reference - Temporal accumulation");
    In(ResourceType::IN_SIGNAL);
    Out(P_HISTORY); // read-modify-write of the same pixel
    EndPass("REFERENCE_TemporalAccumulation.cs", 16, 16, sizeof(nrdc::ReferenceAccumulateConstants));

    BeginPass("Reference - Copy");
    In(P_HISTORY);
    Out(ResourceType::OUT_SIGNAL);
    EndPass("REFERENCE_Copy.cs", 16, 16, sizeof(nrdc::ReferenceCopyConstants));
}

void InstanceImpl::Update_Reference(const DenoiserData& d) {
    const ReferenceSettings& s = d.settings.reference;
    const CommonSettings& cs = m_CommonSettings;

    // History restarts whenever the camera moved, the rect changed or a restart was requested (Reference.hpp:64-74)
    bool restart = m_WorldToClip != m_WorldToClipPrev || cs.accumulationMode != AccumulationMode::CONTINUE;
    restart |= cs.rectSize[0] != cs.rectSizePrev[0] || cs.rectSize[1] != cs.rectSizePrev[1];
    if (restart)
        m_AccumulatedFrameNum = 0;
    else
        m_AccumulatedFrameNum = std::min(m_AccumulatedFrameNum + 1, std::min(s.maxAccumulatedFrameNum, REFERENCE_MAX_HISTORY_FRAME_NUM));

    auto* acc = (nrdc::ReferenceAccumulateConstants*)PushDispatch(d, PASS_ACCUMULATE);
    acc->gRectOrigin = {cs.rectOrigin[0], cs.rectOrigin[1]};
    acc->gAccumSpeed = 1.0f / (1.0f + float(m_AccumulatedFrameNum));
    acc->gDebug = cs.debug;

    auto* copy = (nrdc::ReferenceCopyConstants*)PushDispatch(d, PASS_COPY);
    copy->gRectSizeInv = {1.0f / float(cs.rectSize[0]), 1.0f / float(cs.rectSize[1])};
    copy->gSplitScreen = cs.splitScreen;
}

} // namespace nrd
