// TEST INFRASTRUCTURE -- NOT PRODUCT CODE (see shim/hip/hip_runtime.h). The execution model of the CPU emulation of the device sources:
// a kernel launch runs its blocks on the OpenMP threads of the process, one block at a time per OS thread, every GPU thread of the block as a
// ucontext fiber. A fiber runs until it finishes or reaches a block barrier / wave operation; the block scheduler releases a barrier when every
// fiber that is still running has arrived (threads that returned early do not take part, as on the device), and a wave operation when every
// running lane of that 64-lane wave has arrived. A block in which the running fibers wait for different things never makes progress on the GPU
// either; here it aborts with a message.
#include <hip/hip_runtime.h>

#include <sys/mman.h>
#include <ucontext.h>

#include <chrono>
#include <cstdio>
#include <vector>

#ifdef _OPENMP
#include <omp.h>
#endif

namespace hwmath {
const signed char* g_RcpDelta = nullptr;
const signed char* g_SqrtDelta = nullptr;
const signed char* g_RsqDelta = nullptr;
const signed char* g_Exp2Delta = nullptr;
const signed char* g_Exp2NegDelta = nullptr;
const signed char* g_Log2Delta = nullptr;
int g_IeeeMode = 0;
void TablesMissing(const char* which) {
    fprintf(stderr, "emu: the hardware delta table for %s is not loaded (oracle/hw_*.i8.z through tests/emu/emu_run.py)\n", which);
    abort();
}
} // namespace hwmath

namespace emu {

long g_Med3Violations = 0;
thread_local ThreadCtx* t_cur = nullptr;
thread_local const void* t_kernarg = nullptr;

namespace {

enum State : uint8_t { READY, WAIT_BLOCK, WAIT_WAVE, DONE };
enum WaveOp : uint8_t { OP_NONE, OP_ALL, OP_ANY, OP_SHFL_XOR };

constexpr size_t STACK_BYTES = 1u << 20; // per fiber; mapped lazily

struct Fiber {
    ucontext_t ctx;
    ThreadCtx tc;
    State state;
    WaveOp op;
    uint32_t operand, result;
    int arg0, arg1;
    void* stack;
};

struct BlockRunner {
    std::vector<Fiber> fibers;
    ucontext_t scheduler;
    const std::function<void()>* fn = nullptr;
    Fiber* running = nullptr;
    uint32_t count = 0;

    void EnsureFibers(uint32_t n) {
        while (fibers.size() < n) {
            Fiber f;
            f.stack = mmap(nullptr, STACK_BYTES, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
            if (f.stack == MAP_FAILED) {
                perror("emu: mmap of a fiber stack");
                abort();
            }
            fibers.push_back(f);
        }
    }
};

thread_local BlockRunner* t_runner = nullptr;

void FiberEntry() {
    BlockRunner* r = t_runner;
    (*r->fn)();
    r->running->state = DONE;
    // returns to uc_link = the scheduler
}

void Yield(State s) {
    BlockRunner* r = t_runner;
    Fiber* me = r->running;
    me->state = s;
    swapcontext(&me->ctx, &r->scheduler);
}

void RunBlock(BlockRunner& r, dim3 grid, dim3 block, dim3 bid, const std::function<void()>& fn) {
    const uint32_t n = block.x * block.y * block.z;
    r.EnsureFibers(n);
    r.fn = &fn;
    r.count = n;
    t_runner = &r;
    for (uint32_t i = 0; i < n; i++) {
        Fiber& f = r.fibers[i];
        f.tc.tid = dim3(i % block.x, (i / block.x) % block.y, i / (block.x * block.y));
        f.tc.bid = bid;
        f.tc.bdim = block;
        f.tc.gdim = grid;
        f.state = READY;
        f.op = OP_NONE;
        getcontext(&f.ctx);
        f.ctx.uc_stack.ss_sp = f.stack;
        f.ctx.uc_stack.ss_size = STACK_BYTES;
        f.ctx.uc_link = &r.scheduler;
        makecontext(&f.ctx, FiberEntry, 0);
    }
    for (;;) {
        bool ran = false;
        for (uint32_t i = 0; i < n; i++) {
            Fiber& f = r.fibers[i];
            if (f.state != READY)
                continue;
            r.running = &f;
            t_cur = &f.tc;
            swapcontext(&r.scheduler, &f.ctx);
            ran = true;
        }
        // release what can be released
        bool released = false, allDone = true, allAtBarrier = true;
        for (uint32_t i = 0; i < n; i++) {
            const State s = r.fibers[i].state;
            if (s != DONE)
                allDone = false;
            if (s != DONE && s != WAIT_BLOCK)
                allAtBarrier = false;
        }
        if (allDone)
            break;
        if (allAtBarrier) {
            for (uint32_t i = 0; i < n; i++)
                if (r.fibers[i].state == WAIT_BLOCK)
                    r.fibers[i].state = READY;
            released = true;
        }
        for (uint32_t w0 = 0; w0 < n; w0 += 64) {
            const uint32_t w1 = std::min(n, w0 + 64);
            bool any = false, all = true;
            WaveOp op = OP_NONE;
            for (uint32_t i = w0; i < w1; i++) {
                const Fiber& f = r.fibers[i];
                if (f.state == DONE)
                    continue;
                if (f.state != WAIT_WAVE) {
                    all = false;
                    break;
                }
                if (any && f.op != op)
                    all = false;
                op = f.op;
                any = true;
            }
            if (!any || !all)
                continue;
            uint32_t andAll = 1, orAll = 0;
            for (uint32_t i = w0; i < w1; i++)
                if (r.fibers[i].state == WAIT_WAVE) {
                    andAll &= r.fibers[i].operand ? 1u : 0u;
                    orAll |= r.fibers[i].operand ? 1u : 0u;
                }
            for (uint32_t i = w0; i < w1; i++) {
                Fiber& f = r.fibers[i];
                if (f.state != WAIT_WAVE)
                    continue;
                if (op == OP_ALL)
                    f.result = andAll;
                else if (op == OP_ANY)
                    f.result = orAll;
                else { // OP_SHFL_XOR: lanes are grouped in segments of `width`
                    const uint32_t lane = i - w0, width = (uint32_t)f.arg1;
                    const uint32_t src = ((lane ^ (uint32_t)f.arg0) & (width - 1)) | (lane & ~(width - 1));
                    const Fiber& s = r.fibers[w0 + src];
                    f.result = (w0 + src < w1 && s.state == WAIT_WAVE) ? s.operand : f.operand;
                }
            }
            for (uint32_t i = w0; i < w1; i++)
                if (r.fibers[i].state == WAIT_WAVE)
                    r.fibers[i].state = READY;
            released = true;
        }
        if (!ran && !released) {
            fprintf(stderr, "emu: block (%u, %u, %u) cannot make progress: its threads wait at different barriers / wave operations\n", bid.x, bid.y, bid.z);
            abort();
        }
    }
    t_cur = nullptr;
    t_runner = nullptr;
}

uint32_t WaveCall(WaveOp op, uint32_t operand, int arg0, int arg1) {
    Fiber* me = t_runner->running;
    me->op = op;
    me->operand = operand;
    me->arg0 = arg0;
    me->arg1 = arg1;
    Yield(WAIT_WAVE);
    return me->result;
}

} // namespace

void SyncThreads() { Yield(WAIT_BLOCK); }
int All(int p) { return (int)WaveCall(OP_ALL, p ? 1u : 0u, 0, 0); }
int Any(int p) { return (int)WaveCall(OP_ANY, p ? 1u : 0u, 0, 0); }
uint32_t ShflXorBits(uint32_t v, int laneMask, int width) { return WaveCall(OP_SHFL_XOR, v, laneMask, width); }

void Launch(dim3 grid, dim3 block, const void* kernarg, const std::function<void()>& thread) {
    const long blocks = (long)grid.x * grid.y * grid.z;
#pragma omp parallel
    {
        static thread_local BlockRunner runner; // keeps its fiber stacks for the life of the OS thread
        t_kernarg = kernarg;
#pragma omp for schedule(dynamic, 1)
        for (long b = 0; b < blocks; b++) {
            dim3 bid((unsigned)(b % grid.x), (unsigned)((b / grid.x) % grid.y), (unsigned)(b / ((long)grid.x * grid.y)));
            RunBlock(runner, grid, block, bid, thread);
        }
        t_kernarg = nullptr;
    }
}

} // namespace emu

// ------------------------------------------------------------------------------------------------ runtime API stubs
struct emuEvent {
    std::chrono::steady_clock::time_point t;
};

hipError_t hipGetDeviceCount(int* n) {
    *n = 1;
    return hipSuccess;
}
hipError_t hipMalloc(void** p, size_t bytes) {
    void* q = nullptr;
    if (posix_memalign(&q, 4096, bytes ? bytes : 1) != 0)
        return hipErrorOutOfMemory;
    memset(q, 0xCD, bytes); // device memory is not zeroed; a recognisable pattern shows reads of never-written planes
    *p = q;
    return hipSuccess;
}
hipError_t hipFree(void* p) {
    free(p);
    return hipSuccess;
}
hipError_t hipMemsetAsync(void* p, int v, size_t bytes, hipStream_t) {
    memset(p, v, bytes);
    return hipSuccess;
}
hipError_t hipMemcpy(void* dst, const void* src, size_t bytes, hipMemcpyKind) {
    memcpy(dst, src, bytes);
    return hipSuccess;
}
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
hipError_t hipGetLastError() { return hipSuccess; }
const char* hipGetErrorString(hipError_t e) { return e == hipSuccess ? "hipSuccess" : (e == hipErrorNotSupported ? "not supported by the CPU emulation" : "emulated HIP error"); }
hipError_t hipEventCreate(hipEvent_t* e) {
    *e = new emuEvent();
    return hipSuccess;
}
hipError_t hipEventDestroy(hipEvent_t e) {
    delete e;
    return hipSuccess;
}
hipError_t hipEventRecord(hipEvent_t e, hipStream_t) {
    e->t = std::chrono::steady_clock::now();
    return hipSuccess;
}
hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) {
    *ms = std::chrono::duration<float, std::milli>(b->t - a->t).count();
    return hipSuccess;
}
hipError_t hipGraphCreate(hipGraph_t*, unsigned) { return hipErrorNotSupported; }
hipError_t hipGraphDestroy(hipGraph_t) { return hipSuccess; }
hipError_t hipGraphAddKernelNode(hipGraphNode_t*, hipGraph_t, const hipGraphNode_t*, size_t, const hipKernelNodeParams*) { return hipErrorNotSupported; }
hipError_t hipGraphInstantiate(hipGraphExec_t*, hipGraph_t, hipGraphNode_t*, char*, size_t) { return hipErrorNotSupported; }
hipError_t hipGraphExecDestroy(hipGraphExec_t) { return hipSuccess; }
hipError_t hipGraphExecKernelNodeSetParams(hipGraphExec_t, hipGraphNode_t, const hipKernelNodeParams*) { return hipErrorNotSupported; }
hipError_t hipGraphLaunch(hipGraphExec_t, hipStream_t) { return hipErrorNotSupported; }

// ------------------------------------------------------------------------------------------------ C entry points of the emulation itself
extern "C" {
// deviation tables of the five transcendental instructions (oracle/hw_math.h); the memory stays owned by the caller
__attribute__((visibility("default"))) void nrdEmuSetHwTables(const signed char* rcp, const signed char* sqrt, const signed char* rsq, const signed char* exp2, const signed char* log2) {
    hwmath::g_RcpDelta = rcp;
    hwmath::g_SqrtDelta = sqrt;
    hwmath::g_RsqDelta = rsq;
    hwmath::g_Exp2Delta = exp2;
    hwmath::g_Log2Delta = log2;
}
__attribute__((visibility("default"))) void nrdEmuSetHwTableExp2Neg(const signed char* exp2neg) { hwmath::g_Exp2NegDelta = exp2neg; }
__attribute__((visibility("default"))) int nrdEmuSetThreads(int n) {
#ifdef _OPENMP
    if (n > 0)
        omp_set_num_threads(n);
    return omp_get_max_threads();
#else
    (void)n;
    return 1;
#endif
}
}

extern "C" __attribute__((visibility("default"))) long emu_med3_violations() { return emu::g_Med3Violations; }
