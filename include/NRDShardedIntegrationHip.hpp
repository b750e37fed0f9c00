// nrd::ShardedIntegrationHip -- one frame denoised by several GPUs, one rank (process or thread + GPU) per row strip, for a C++ host.
// The C++ counterpart of raytracingdenoiser_amd/sharding.py (HaloSharder): the dispatch list is cut into pass segments by
// nrdHipPlanHaloExchange (include/NRDHip.h); before a segment the rank swaps the boundary bands of the planes the segment reads with
// its two neighbours; a list that cannot be bounded (restart frame, hit-distance reconstruction, SIGMA, a dynamic-resolution step) runs
// unsharded on every rank, after the carried-over planes have been completed everywhere. Owned rows are bit-identical to a single-GPU
// run as long as the vertical motion stays below maxMotionRows (promised by the application, or measured every frame: measureMotion).
//
// The transfers go through an nrd::HaloTransport. nrd::RcclHaloTransport (below, compiled when NRD_SHARDED_WITH_RCCL is defined: needs
// <rccl/rccl.h> and the HIP runtime) issues them as grouped ncclSend / ncclRecv on its own stream, so the passes that do not touch the
// exchanged planes overlap with the transfers; tests/cpp/sharded_virtual_ranks.cpp plugs in a loop-back transport that connects several
// virtual ranks on one GPU, which is how the scheduling logic is verified on a single-GPU machine.
//
// Include after NRD.h, NRDHip.h and NRDIntegrationHip.hpp. The reference has no counterpart (its integration layer is single-GPU).
#pragma once

#include <algorithm>
#include <set>
#include <utility>
#include <vector>

namespace nrd {

struct HaloTransfer {
    bool send;            // false = receive
    uint32_t peer;        // rank on the other side
    uint32_t resourceType, indexInPool; // which plane (for transports that resolve the remote side themselves)
    uint32_t rowBegin, rowEnd;          // rows of that plane
    void* data;           // local device pointer of row rowBegin
    size_t bytes;         // (rowEnd - rowBegin) * rowPitchBytes: bands are contiguous
};

class HaloTransport {
public:
    virtual ~HaloTransport() {}
    // Issues one batch of transfers. They may only start once the work enqueued so far on computeStream has finished, and Wait() makes
    // computeStream wait for their completion (no host blocking required).
    virtual bool Exchange(const HaloTransfer* transfers, uint32_t transfersNum, void* computeStream) = 0;
    virtual bool Wait(void* computeStream) = 0;
    // rows [rowBegin, rowEnd) of a plane travel from rank root to every other rank (used before an unsharded frame that follows sharded ones)
    virtual bool Broadcast(const HaloTransfer& band, uint32_t root, void* computeStream) = 0;
    // value = the maximum of the ranks' values (one float per frame: the measured motion bound, ShardedIntegrationHipCreationDesc::measureMotion). A host
    // that drives the ranks itself (PrepareFrame / PlanFrame) never calls it; the default suits a single rank.
    virtual bool MaxOverRanks(float& value, void* computeStream) {
        (void)value, (void)computeStream;
        return true;
    }
    // The same reduction without the host in the middle (round 6): DeviceScalar() = 4 bytes of device memory the transport can all-reduce in place (nullptr: this transport
    // has only the host form above). BeginFrame then lets nrdHipMeasureMotionRowsAsync write the strip's value there in stream order, MaxOverRanksInPlace reduces it on the
    // device and synchronises ONCE to hand the result back -- instead of synchronise, read back, upload, reduce, read back.
    // The buffer holds TWO floats: [0] this frame's surface motion (nrdHipMeasureMotionRowsAsync), [1] the history reach the temporal kernels reported for the PREVIOUS frame
    // (nrdHipSetHistoryReachWord: virtual motion and look-back taps of the specular signal, which no measurement of the inputs bounds). MaxOverRanksInPlace reduces both with one
    // collective, hands them back and clears [1] behind the reduction (in stream order) for this frame's kernels.
    virtual void* DeviceScalar() { return nullptr; }
    virtual bool MaxOverRanksInPlace(float& value, float& historyReach, void* computeStream) {
        (void)value, (void)historyReach, (void)computeStream;
        return false;
    }
};

struct ShardedIntegrationHipCreationDesc {
    IntegrationHipCreationDesc integration;
    HaloTransport* transport = nullptr;
    uint32_t rank = 0, world = 1;
    uint32_t maxMotionRows = 32;     // the largest vertical motion (rows per frame) the history halos cover
    uint32_t exchangeThreshold = 24; // passes reaching further than this start a new segment (their inputs are exchanged, not recomputed)
    // false: maxMotionRows is the application's promise about every frame. true: it is CHECKED every frame -- nrdHipMeasureMotionRows reduces the frame's own
    // IN_VIEWZ / IN_MV over this rank's strip (the temporal passes' surface-motion reprojection, moving objects included), the ranks take the maximum
    // (HaloTransport::MaxOverRanks) and a frame with 2 x motion + 2 >= maxMotionRows (virtual motion of specular reflections, bicubic footprint) runs unsharded.
    // NOTE: the measurement bounds the SURFACE-motion reprojection. The virtual-motion (specular) reprojection is ASSUMED to reach at most twice as far (the factor 2 above):
    // a heuristic, not a check -- a strongly curved reflector or a very long hit distance can exceed it, and stale history-halo rows would then be read without notice.
    bool measureMotion = false;
    // Denoise() ends with GatherOutputs(): every rank receives the other ranks' rows of the OUT_* planes the frame wrote (BASELINE configs[3]: "screen tiled across the GPUs
    // with RCCL all-gather"; the reference's nrd::Integration::Denoise hands back complete outputs, NRDIntegration.hpp:516-623)
    bool gatherOutputs = true;
};

class ShardedIntegrationHip {
public:
    inline bool Initialize(const ShardedIntegrationHipCreationDesc& desc, const InstanceCreationDesc& instanceCreationDesc) {
        if (!desc.transport || desc.world == 0 || desc.rank >= desc.world || !m_Integration.Initialize(desc.integration, instanceCreationDesc))
            return false;
        m_Desc = desc;
        m_Bounds.resize(desc.world + 1);
        for (uint32_t r = 0; r <= desc.world; r++)
            m_Bounds[r] = r * (uint32_t)desc.integration.resourceHeight / desc.world; // uniform strips; SetStripBounds() re-cuts them
        m_Complete = true;
        return true;
    }
    inline void Destroy() { m_Integration.Destroy(); }
    inline void NewFrame() { m_Integration.NewFrame(); }
    inline bool SetCommonSettings(const CommonSettings& s) { return m_Integration.SetCommonSettings(s); }
    inline bool SetDenoiserSettings(Identifier id, const void* s) { return m_Integration.SetDenoiserSettings(id, s); }
    inline const char* GetLastError() const { return m_Error ? m_Error : m_Integration.GetLastError(); }
    inline IntegrationHip& GetIntegration() { return m_Integration; }
    inline const std::vector<uint32_t>& GetStripBounds() const { return m_Bounds; }
    inline uint32_t GetOwnedRowBegin() const { return m_Bounds[m_Desc.rank]; }
    inline uint32_t GetOwnedRowEnd() const { return m_Bounds[m_Desc.rank + 1]; }
    inline bool IsComplete() const { return m_Complete; } // every plane is complete on this rank (the last frame ran unsharded)

    // Strips may be re-cut (e.g. by the amount of non-sky tiles) whenever IsComplete(): all ranks must pass the same bounds.
    inline bool SetStripBounds(const uint32_t* bounds) {
        if (!m_Complete)
            return false;
        m_Bounds.assign(bounds, bounds + m_Desc.world + 1);
        return true;
    }

    // One frame = BeginFrame, then for every step ExchangeStep + RunStep, then EndFrame; Denoise() does exactly that. The step-wise form
    // exists for hosts that interleave their own work and for the virtual-rank test.
    inline bool BeginFrame(const Identifier* denoisers, uint32_t denoisersNum, const UserPoolHip& userPool) {
        float rows = -1.0f;
        void* deviceScalar = (m_Desc.measureMotion && m_Desc.world > 1) ? m_Desc.transport->DeviceScalar() : nullptr;
        if (deviceScalar) { // the measurement stays on the device until it is reduced
            if (!PrepareFrame(denoisers, denoisersNum, userPool, nullptr, false))
                return false;
            if (nrdHipMeasureMotionRowsAsync(m_Integration.GetExecutor(), m_Dispatches, m_DispatchesNum, m_Bounds[m_Desc.rank], m_Bounds[m_Desc.rank + 1], deviceScalar) != (uint32_t)Result::SUCCESS)
                return false;
            if (!m_ReachRegistered) { // the second float of the transport's buffer is where this rank's temporal passes report their history reach from now on
                if (nrdHipSetHistoryReachWord(m_Integration.GetExecutor(), (float*)deviceScalar + 1) != (uint32_t)Result::SUCCESS)
                    return false;
                m_ReachRegistered = true;
            }
            float reach = 0.0f;
            if (!m_Desc.transport->MaxOverRanksInPlace(rows, reach, m_Desc.integration.hipStream))
                return Fail("transport max-reduction failed");
            return PlanFrame(rows, reach);
        }
        if (!PrepareFrame(denoisers, denoisersNum, userPool, &rows))
            return false;
        if (rows >= 0.0f && !m_Desc.transport->MaxOverRanks(rows, m_Desc.integration.hipStream))
            return Fail("transport max-reduction failed");
        return PlanFrame(rows);
    }
    // BeginFrame in two halves, for hosts (and the virtual-rank test) that reduce the measured motion over the ranks themselves:
    //   PrepareFrame   binds the user planes, asks the instance for the frame's dispatch list and -- with measureMotion -- measures this strip's motion
    //                  (*localMotionRows; -1 when nothing was measured)
    //   PlanFrame      plans the halo exchange; motionRowsOverRanks >= 0 is held against maxMotionRows (the SAME value on every rank), < 0 = no check
    inline bool PrepareFrame(const Identifier* denoisers, uint32_t denoisersNum, const UserPoolHip& userPool, float* localMotionRows = nullptr, bool measure = true) {
        m_Error = nullptr;
        NrdHipExecutor* ex = m_Integration.GetExecutor();
        for (size_t slot = 0; slot < userPool.size(); slot++)
            if (userPool[slot].data) {
                if (nrdHipBindResource(ex, (uint32_t)slot, &userPool[slot]) != (uint32_t)Result::SUCCESS)
                    return false;
                m_UserPool[slot] = userPool[slot];
            }
        if (GetComputeDispatches(*m_Integration.GetInstance(), denoisers, denoisersNum, m_Dispatches, m_DispatchesNum) != Result::SUCCESS)
            return Fail("GetComputeDispatches failed");
        float rows = -1.0f;
        if (measure && m_Desc.measureMotion && m_Desc.world > 1 &&
            nrdHipMeasureMotionRows(ex, m_Dispatches, m_DispatchesNum, m_Bounds[m_Desc.rank], m_Bounds[m_Desc.rank + 1], &rows) != (uint32_t)Result::SUCCESS)
            return false;
        if (localMotionRows)
            *localMotionRows = rows;
        return true;
    }
    // historyReachOverRanks >= 0: what the temporal kernels reported LAST frame (rows, MAX over the ranks): this frame also runs unsharded when 1.25 x that + 3 rows (bicubic
    // footprint) does not fit the history halo, and a sharded last frame whose reach + 3 exceeded the halo is counted (GetHistoryHaloViolationsNum) -- the rule of the Python host
    inline bool PlanFrame(float motionRowsOverRanks = -1.0f, float historyReachOverRanks = -1.0f) {
        m_RowBegin.assign(m_DispatchesNum, -1);
        m_RowEnd.assign(m_DispatchesNum, 0);
        m_Steps.resize(64);
        m_Items.resize(1024);
        NrdHipHaloPlanInfo info = {};
        if (nrdHipPlanHaloExchange(m_Integration.GetInstance(), m_Dispatches, m_DispatchesNum, m_Bounds.data(), m_Desc.world, m_Desc.rank, m_Desc.integration.resourceHeight, m_Desc.maxMotionRows,
                m_Desc.exchangeThreshold, m_RowBegin.data(), m_RowEnd.data(), m_Steps.data(), (uint32_t)m_Steps.size(), m_Items.data(), (uint32_t)m_Items.size(), &info) != (uint32_t)Result::SUCCESS)
            return Fail("nrdHipPlanHaloExchange failed");
        m_LastMotionRows = motionRowsOverRanks;
        m_LastHistoryReach = historyReachOverRanks;
        if (historyReachOverRanks >= 0.0f && m_LastFrameSharded && !(historyReachOverRanks + 3.0f <= float(m_Desc.maxMotionRows)))
            m_HistoryHaloViolations++;
        const bool motionExceedsHalo = (motionRowsOverRanks >= 0.0f && !(2.0f * motionRowsOverRanks + 2.0f < float(m_Desc.maxMotionRows))) || // (a NaN exceeds)
                                       (historyReachOverRanks >= 0.0f && !(1.25f * historyReachOverRanks + 3.0f < float(m_Desc.maxMotionRows)));
        if (motionExceedsHalo && !info.fallback && m_Desc.world > 1)
            m_MotionFallbacks++;
        m_Fallback = info.fallback != 0 || motionExceedsHalo || m_Desc.world == 1;
        m_Steps.resize(m_Fallback ? 1 : info.stepsNum);
        if (m_Fallback)
            m_Steps[0] = NrdHipHaloStep{0, m_DispatchesNum, 0, 0, 0};
        return true;
    }
    inline float GetLastMotionRows() const { return m_LastMotionRows; }        // what the last PlanFrame was given (-1: nothing)
    inline float GetLastHistoryReachRows() const { return m_LastHistoryReach; } // the previous frame's history reach as the last PlanFrame was given it (-1: nothing)
    inline uint32_t GetHistoryHaloViolationsNum() const { return m_HistoryHaloViolations; } // sharded frames whose temporal passes read beyond the halo they ran with
    inline uint32_t GetMotionFallbacksNum() const { return m_MotionFallbacks; } // frames run unsharded because the measured motion did not fit the history halo
    inline uint32_t GetStepsNum() const { return (uint32_t)m_Steps.size(); }

    // Issues the transfers of a step (for an unsharded frame after sharded ones: the completion of the carried-over planes)
    inline bool ExchangeStep(uint32_t step) {
        void* stream = m_Desc.integration.hipStream;
        if (m_Fallback) {
            if (m_Complete || m_Desc.world == 1)
                return true;
            for (const std::pair<uint32_t, uint32_t>& key : CarriedOverPlanes())
                for (uint32_t root = 0; root < m_Desc.world; root++) {
                    HaloTransfer band = {};
                    if (m_Bounds[root + 1] > m_Bounds[root] && !MakeTransfer(key.first, key.second, m_Bounds[root], m_Bounds[root + 1], root == m_Desc.rank, root, band))
                        return false;
                    if (m_Bounds[root + 1] > m_Bounds[root] && !m_Desc.transport->Broadcast(band, root, stream))
                        return Fail("transport broadcast failed");
                }
            return true;
        }
        const NrdHipHaloStep& st = m_Steps[step];
        std::vector<HaloTransfer> transfers;
        const uint32_t rb = m_Bounds[m_Desc.rank], re = m_Bounds[m_Desc.rank + 1];
        for (uint32_t k = st.firstItem; k < st.firstItem + st.itemCount; k++) {
            const NrdHipHaloItem& it = m_Items[k];
            const uint32_t w = it.widthRows;
            HaloTransfer t = {};
            if (m_Desc.rank > 0) { // my top rows go up, the rows above my strip come down
                if (!MakeTransfer(it.resourceType, it.indexInPool, rb, rb + w, true, m_Desc.rank - 1, t)) return false;
                transfers.push_back(t);
                if (!MakeTransfer(it.resourceType, it.indexInPool, rb - w, rb, false, m_Desc.rank - 1, t)) return false;
                transfers.push_back(t);
            }
            if (m_Desc.rank + 1 < m_Desc.world) {
                if (!MakeTransfer(it.resourceType, it.indexInPool, re - w, re, true, m_Desc.rank + 1, t)) return false;
                transfers.push_back(t);
                if (!MakeTransfer(it.resourceType, it.indexInPool, re, re + w, false, m_Desc.rank + 1, t)) return false;
                transfers.push_back(t);
            }
        }
        m_Pending = !transfers.empty();
        if (m_Pending && !m_Desc.transport->Exchange(transfers.data(), (uint32_t)transfers.size(), stream))
            return Fail("transport exchange failed");
        return true;
    }

    // Runs the passes of a step: those that do not touch the exchanged planes first (they overlap with the transfers), the rest after the wait
    inline bool RunStep(uint32_t step) {
        NrdHipExecutor* ex = m_Integration.GetExecutor();
        void* stream = m_Desc.integration.hipStream;
        const NrdHipHaloStep& st = m_Steps[step];
        const int32_t *rowBegin = m_Fallback ? nullptr : m_RowBegin.data(), *rowEnd = m_Fallback ? nullptr : m_RowEnd.data();
        uint32_t early = (!m_Fallback && m_Pending) ? st.earlyCount : 0;
        if (early && nrdHipExecuteDispatchRange(ex, m_Dispatches, m_DispatchesNum, st.firstDispatch, early, rowBegin, rowEnd) != (uint32_t)Result::SUCCESS)
            return false;
        if (m_Pending && !m_Desc.transport->Wait(stream))
            return Fail("transport wait failed");
        m_Pending = false;
        return nrdHipExecuteDispatchRange(ex, m_Dispatches, m_DispatchesNum, st.firstDispatch + early, st.dispatchCount - early, rowBegin, rowEnd) == (uint32_t)Result::SUCCESS;
    }
    inline void EndFrame() {
        m_Complete = m_Fallback;
        m_LastFrameSharded = !m_Fallback && m_Desc.world > 1;
    }

    // The reassembly of the outputs: after a sharded frame every rank receives, IN PLACE in its bound OUT_* planes, the rows of every other rank -- one band per source rank
    // and plane through HaloTransport::Broadcast (over RCCL: ncclBroadcast, i.e. an all-gather spelled as the group of its broadcasts; strips re-cut by a load balancer
    // are unequal). The bound OUT_* planes are working planes of the pass chain: the application consumes them before its next Denoise(), as with the reference's
    // nrd::Integration. (The Python host stages its rows into separate complete planes and lets the all-gather overlap the next frame: sharding.py HaloSharder.)
    inline bool GatherOutputs() {
        if (m_Fallback || m_Desc.world == 1)
            return true; // every rank computed every row
        NrdHipExecutor* ex = m_Integration.GetExecutor();
        (void)ex;
        std::set<uint32_t> outputs;
        for (uint32_t i = 0; i < m_DispatchesNum; i++)
            for (uint32_t r = 0; r < m_Dispatches[i].resourcesNum; r++) {
                const ResourceDesc& res = m_Dispatches[i].resources[r];
                if (res.descriptorType == DescriptorType::STORAGE_TEXTURE && (uint32_t)res.type >= (uint32_t)ResourceType::OUT_DIFF_RADIANCE_HITDIST && (uint32_t)res.type < (uint32_t)ResourceType::TRANSIENT_POOL)
                    outputs.insert((uint32_t)res.type);
            }
        for (uint32_t type : outputs)
            for (uint32_t src = 0; src < m_Desc.world; src++) {
                HaloTransfer band;
                if (m_Bounds[src + 1] == m_Bounds[src])
                    continue;
                if (!MakeTransfer(type, 0, m_Bounds[src], m_Bounds[src + 1], src == m_Desc.rank, src, band))
                    return false;
                if (!m_Desc.transport->Broadcast(band, src, m_Desc.integration.hipStream))
                    return Fail("transport broadcast failed while gathering the outputs");
                m_GatheredBytes += src == m_Desc.rank ? 0 : band.bytes;
            }
        return true;
    }
    inline size_t GetGatheredBytes() const { return m_GatheredBytes; }

    inline bool Denoise(const Identifier* denoisers, uint32_t denoisersNum, const UserPoolHip& userPool) {
        if (!BeginFrame(denoisers, denoisersNum, userPool))
            return false;
        for (uint32_t s = 0; s < GetStepsNum(); s++)
            if (!ExchangeStep(s) || !RunStep(s))
                return false;
        EndFrame();
        return !m_Desc.gatherOutputs || GatherOutputs();
    }

    // The plane behind a (resourceType, indexInPool) key: a pool plane of the executor or the user plane bound to that slot
    inline bool GetPlane(uint32_t resourceType, uint32_t indexInPool, NrdHipPlaneDesc& plane) const {
        if (resourceType == (uint32_t)ResourceType::TRANSIENT_POOL || resourceType == (uint32_t)ResourceType::PERMANENT_POOL)
            return nrdHipGetPoolPlane(m_Integration.GetExecutor(), resourceType, indexInPool, &plane) == (uint32_t)Result::SUCCESS;
        if (resourceType >= m_UserPool.size() || !m_UserPool[resourceType].data)
            return false;
        plane = m_UserPool[resourceType];
        return true;
    }

private:
    inline bool Fail(const char* what) {
        m_Error = what;
        return false;
    }
    inline bool MakeTransfer(uint32_t type, uint32_t index, uint32_t row0, uint32_t row1, bool send, uint32_t peer, HaloTransfer& t) {
        NrdHipPlaneDesc p = {};
        if (!GetPlane(type, index, p) || row1 > p.height || row0 > row1)
            return Fail("halo plane not available (is the OUT_* plane it refers to bound?)");
        t = HaloTransfer{send, peer, type, index, row0, row1, (uint8_t*)p.data + (size_t)row0 * p.rowPitchBytes, (size_t)(row1 - row0) * p.rowPitchBytes};
        return true;
    }
    // planes an unsharded frame needs complete on every rank before it runs: those it reads before (or without) writing them -- the history it inherits -- and those a pass with a
    // neighbourhood (nrdHipGetDispatchReach != 0) reads after an earlier pass of the frame wrote them: every pass skips the sky, so such a plane keeps in its sky texels what the last
    // frame left there, and a rank is only complete inside its own strip (sharding.py carried_over_planes; round 6). Tile maps (down-sampled planes) are complete everywhere.
    inline std::vector<std::pair<uint32_t, uint32_t>> CarriedOverPlanes() const {
        std::set<std::pair<uint32_t, uint32_t>> written;
        std::vector<std::pair<uint32_t, uint32_t>> carried;
        std::vector<int32_t> reach(m_DispatchesNum, -1);
        (void)nrdHipGetDispatchReach(m_Integration.GetInstance(), m_Dispatches, m_DispatchesNum, reach.data());
        for (uint32_t i = 0; i < m_DispatchesNum; i++) {
            const DispatchDesc& d = m_Dispatches[i];
            const bool neighbourhood = reach[i] != 0; // (-1 = unbounded)
            for (uint32_t r = 0; r < d.resourcesNum; r++) {
                const ResourceDesc& res = d.resources[r];
                const std::pair<uint32_t, uint32_t> key((uint32_t)res.type, res.indexInPool);
                NrdHipPlaneDesc p = {};
                if (res.descriptorType != DescriptorType::TEXTURE || (uint32_t)res.type < (uint32_t)ResourceType::OUT_DIFF_RADIANCE_HITDIST || (written.count(key) && !neighbourhood) ||
                    std::find(carried.begin(), carried.end(), key) != carried.end() || !GetPlane(key.first, key.second, p) || p.height != m_Desc.integration.resourceHeight)
                    continue;
                carried.push_back(key);
            }
            for (uint32_t r = 0; r < d.resourcesNum; r++)
                if (d.resources[r].descriptorType == DescriptorType::STORAGE_TEXTURE)
                    written.insert(std::make_pair((uint32_t)d.resources[r].type, (uint32_t)d.resources[r].indexInPool));
        }
        return carried;
    }

    IntegrationHip m_Integration;
    ShardedIntegrationHipCreationDesc m_Desc;
    UserPoolHip m_UserPool = {};
    std::vector<uint32_t> m_Bounds;
    const DispatchDesc* m_Dispatches = nullptr;
    uint32_t m_DispatchesNum = 0;
    std::vector<int32_t> m_RowBegin, m_RowEnd;
    std::vector<NrdHipHaloStep> m_Steps;
    std::vector<NrdHipHaloItem> m_Items;
    bool m_Fallback = true, m_Complete = true, m_Pending = false;
    size_t m_GatheredBytes = 0;
    float m_LastMotionRows = -1.0f, m_LastHistoryReach = -1.0f;
    uint32_t m_MotionFallbacks = 0, m_HistoryHaloViolations = 0;
    bool m_ReachRegistered = false, m_LastFrameSharded = false;
    const char* m_Error = nullptr;
};

} // namespace nrd

#ifdef NRD_SHARDED_WITH_RCCL
#    include <hip/hip_runtime_api.h>
#    include <rccl/rccl.h>

namespace nrd {

// Point-to-point transfers over RCCL (xGMI between the GPUs of a node): one ncclGroup per batch on a stream of its own, ordered against
// the compute stream with events -- so passes enqueued on the compute stream between Exchange() and Wait() run during the transfers.
class RcclHaloTransport : public HaloTransport {
public:
    inline bool Initialize(ncclComm_t comm) {
        m_Comm = comm;
        return hipStreamCreateWithFlags(&m_Stream, hipStreamNonBlocking) == hipSuccess && hipEventCreateWithFlags(&m_Ready, hipEventDisableTiming) == hipSuccess &&
               hipEventCreateWithFlags(&m_Done, hipEventDisableTiming) == hipSuccess;
    }
    inline void Destroy() {
        if (m_Stream) (void)hipStreamDestroy(m_Stream);
        if (m_Ready) (void)hipEventDestroy(m_Ready);
        if (m_Done) (void)hipEventDestroy(m_Done);
        if (m_Scalar) (void)hipFree(m_Scalar);
        m_Scalar = nullptr;
        m_Stream = nullptr;
        m_Ready = m_Done = nullptr;
    }
    inline bool Exchange(const HaloTransfer* t, uint32_t n, void* computeStream) override {
        if (hipEventRecord(m_Ready, (hipStream_t)computeStream) != hipSuccess || hipStreamWaitEvent(m_Stream, m_Ready, 0) != hipSuccess)
            return false;
        bool ok = ncclGroupStart() == ncclSuccess;
        for (uint32_t i = 0; i < n && ok; i++)
            ok = (t[i].send ? ncclSend(t[i].data, t[i].bytes, ncclUint8, (int)t[i].peer, m_Comm, m_Stream) : ncclRecv(t[i].data, t[i].bytes, ncclUint8, (int)t[i].peer, m_Comm, m_Stream)) == ncclSuccess;
        ok = (ncclGroupEnd() == ncclSuccess) && ok;
        return ok && hipEventRecord(m_Done, m_Stream) == hipSuccess;
    }
    inline bool Wait(void* computeStream) override { return hipStreamWaitEvent((hipStream_t)computeStream, m_Done, 0) == hipSuccess; }
    inline bool Broadcast(const HaloTransfer& band, uint32_t root, void* computeStream) override {
        return ncclBroadcast(band.data, band.data, band.bytes, ncclUint8, (int)root, m_Comm, (hipStream_t)computeStream) == ncclSuccess;
    }
    inline bool MaxOverRanks(float& value, void* computeStream) override { // 4 bytes through an all-reduce on the compute stream, then read back
        hipStream_t s = (hipStream_t)computeStream;
        if (!m_Scalar && hipMalloc((void**)&m_Scalar, sizeof(float)) != hipSuccess)
            return false;
        return hipMemcpyAsync(m_Scalar, &value, sizeof(float), hipMemcpyHostToDevice, s) == hipSuccess && ncclAllReduce(m_Scalar, m_Scalar, 1, ncclFloat, ncclMax, m_Comm, s) == ncclSuccess &&
               hipMemcpyAsync(&value, m_Scalar, sizeof(float), hipMemcpyDeviceToHost, s) == hipSuccess && hipStreamSynchronize(s) == hipSuccess;
    }

    inline void* DeviceScalar() override {
        if (!m_Scalar && (hipMalloc((void**)&m_Scalar, 2 * sizeof(float)) != hipSuccess || hipMemset(m_Scalar, 0, 2 * sizeof(float)) != hipSuccess))
            m_Scalar = nullptr;
        return m_Scalar;
    }
    inline bool MaxOverRanksInPlace(float& value, float& historyReach, void* computeStream) override { // both values are on the device already: one collective, one read-back
        hipStream_t s = (hipStream_t)computeStream;
        float host[2] = {0.0f, 0.0f};
        const bool ok = m_Scalar && ncclAllReduce(m_Scalar, m_Scalar, 2, ncclFloat, ncclMax, m_Comm, s) == ncclSuccess &&
                        hipMemcpyAsync(host, m_Scalar, sizeof(host), hipMemcpyDeviceToHost, s) == hipSuccess && hipMemsetAsync(m_Scalar + 1, 0, sizeof(float), s) == hipSuccess &&
                        hipStreamSynchronize(s) == hipSuccess;
        value = host[0], historyReach = host[1];
        return ok;
    }

private:
    ncclComm_t m_Comm = nullptr;
    hipStream_t m_Stream = nullptr;
    hipEvent_t m_Ready = nullptr, m_Done = nullptr;
    float* m_Scalar = nullptr;
};

} // namespace nrd
#endif
