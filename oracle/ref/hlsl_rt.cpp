// ORACLE/_ref -- TEST INFRASTRUCTURE, NOT PRODUCT CODE (see hlsl_shim.h).
//
// Runtime of the reference shaders compiled as C++: the shader registry, constant-buffer unpacking (HLSL packing rules), resource binding in
// DispatchDesc order (inputs by t-register, then outputs by u-register) and the execution of a dispatch -- thread groups on OpenMP threads, the
// threads of a group one after the other, as ucontext fibers when the shader contains a group barrier (a fiber runs until it returns or reaches
// GroupMemoryBarrierWithGroupSync; the barrier opens when every thread that is still alive has arrived). Texels go through the codecs of
// oracle/tex.h, compiled here with that file's own flags, so both oracles store and load identically.
#include "hlsl_shim.h"

#include "../tex.h"

#include <sys/mman.h>
#include <ucontext.h>

#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#ifdef _OPENMP
#include <omp.h>
#endif

// oracle/hw_math.h declares these (tex.h -> hlsl.h -> hw_math.h); nothing here evaluates a transcendental through them
namespace hwmath {
const signed char* g_RcpDelta = nullptr;
const signed char* g_SqrtDelta = nullptr;
const signed char* g_RsqDelta = nullptr;
const signed char* g_Exp2Delta = nullptr;
const signed char* g_Log2Delta = nullptr;
int g_IeeeMode = 1;
void TablesMissing(const char* which) {
    fprintf(stderr, "nrdref: unexpected use of a hardware table (%s)\n", which);
    abort();
}
} // namespace hwmath

namespace hlsl {
uint f32tof16(float f) { return orc::f32tof16(f); }
float f16tof32(uint h) { return orc::f16tof32(h); }
} // namespace hlsl

namespace hlsl_rt {

static orc::Tex AsTex(const Plane& p) { return orc::Tex(orc::Plane{(uint8_t*)p.data, p.rowPitchBytes, p.format, p.width, p.height}); }

void Fetch(const Plane& p, int x, int y, float out[4]) {
    orc::float4 v = AsTex(p).Fetch(x, y);
    out[0] = v.x, out[1] = v.y, out[2] = v.z, out[3] = v.w;
}
uint32_t FetchUint(const Plane& p, int x, int y) { return AsTex(p).FetchUint(x, y); }
void Store(const Plane& p, int x, int y, const float v[4]) {
    orc::Tex t = AsTex(p);
    t.Store(x, y, orc::float4(v[0], v[1], v[2], v[3]));
}
void StoreUint(const Plane& p, int x, int y, uint32_t v) {
    orc::Tex t = AsTex(p);
    t.StoreUint(x, y, v);
}
bool IsUintFormat(uint32_t f) { return f == orc::FMT_R32_UINT || f == orc::FMT_R16_UINT || f == orc::FMT_R8_UINT; }

// ------------------------------------------------------------------------------------------------ registry
struct ShaderTable {
    std::vector<ConstantReg> constants;
    std::vector<ResourceReg> resources;
    std::string fileName;
    int groupX = 0, groupY = 0;
    bool usesBarrier = false;
    void (*thunk)(const ThreadIds&) = nullptr;
};

static std::vector<ShaderTable*>& Registry() {
    static std::vector<ShaderTable*> r;
    return r;
}

ShaderTable* NewTable() { return new ShaderTable(); }
void AddConstant(ShaderTable* t, void* ptr, CbKind kind) { t->constants.push_back({ptr, kind}); }
void AddResource(ShaderTable* t, Plane* plane, bool output, int index, const char* name) { t->resources.push_back({plane, output, index, name}); }
void RegisterShader(ShaderTable* t, const char* fileName, int gx, int gy, bool usesBarrier, void (*thunk)(const ThreadIds&)) {
    t->fileName = fileName;
    t->groupX = gx, t->groupY = gy;
    t->usesBarrier = usesBarrier;
    t->thunk = thunk;
    Registry().push_back(t);
}

// HLSL constant-buffer packing: 4-byte scalars packed into 16-byte registers, a vector never straddles a register, a matrix starts a register
// and occupies one per column (column_major, the default and what NRD.hlsli sets: `#pragma pack_matrix( column_major )`); the host writes its
// column-major float4x4 with one memcpy (reference Source/InstanceImpl.h "AddFloat4x4"), so register c = column c.
static uint32_t UnpackConstants(const ShaderTable& t, const uint8_t* blob, uint32_t size) {
    uint32_t off = 0;
    for (const ConstantReg& c : t.constants) {
        const uint32_t bytes = c.kind == CB_SCALAR ? 4 : c.kind == CB_VEC2 ? 8 : c.kind == CB_VEC3 ? 12 : c.kind == CB_VEC4 ? 16 : 64;
        if (c.kind == CB_MAT4)
            off = (off + 15u) & ~15u;
        else if ((off & 15u) + bytes > 16u)
            off = (off + 15u) & ~15u;
        if (off + bytes > size)
            return 0;
        if (c.kind == CB_MAT4) {
            const float* m = (const float*)(blob + off);
            hlsl::float4x4* M = (hlsl::float4x4*)c.ptr;
            for (int col = 0; col < 4; col++)
                for (int row = 0; row < 4; row++)
                    M->r[row].d[col] = m[col * 4 + row];
        } else {
            memcpy(c.ptr, blob + off, bytes); // vec<T, N> starts with its N components
        }
        off += bytes;
    }
    return off;
}

// ------------------------------------------------------------------------------------------------ fibers
namespace {
constexpr size_t STACK_BYTES = 1u << 20;
enum State : uint8_t { READY, AT_BARRIER, DONE };
struct Fiber {
    ucontext_t ctx;
    ThreadIds ids;
    State state;
    void* stack;
};
struct GroupRunner {
    std::vector<Fiber> fibers;
    ucontext_t scheduler;
    Fiber* running = nullptr;
    void (*thunk)(const ThreadIds&) = nullptr;
    void Ensure(size_t n) {
        while (fibers.size() < n) {
            Fiber f;
            f.stack = mmap(nullptr, STACK_BYTES, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
            if (f.stack == MAP_FAILED) {
                perror("nrdref: mmap of a fiber stack");
                abort();
            }
            fibers.push_back(f);
        }
    }
};
thread_local GroupRunner* t_runner = nullptr;
thread_local bool t_inFiber = false;

void FiberEntry() {
    GroupRunner* r = t_runner;
    r->thunk(r->running->ids);
    r->running->state = DONE;
}
} // namespace

void Barrier() {
    if (!t_inFiber) {
        fprintf(stderr, "nrdref: GroupMemoryBarrierWithGroupSync in a shader registered without barriers\n");
        abort();
    }
    GroupRunner* r = t_runner;
    Fiber* me = r->running;
    me->state = AT_BARRIER;
    swapcontext(&me->ctx, &r->scheduler);
}

static void RunGroup(const ShaderTable& t, GroupRunner& r, uint32_t gx, uint32_t gy) {
    const uint32_t n = (uint32_t)(t.groupX * t.groupY);
    ThreadIds ids;
    ids.groupId = hlsl::uint3(gx, gy, 0u);
    if (!t.usesBarrier) {
        for (uint32_t i = 0; i < n; i++) {
            const uint32_t tx = i % (uint32_t)t.groupX, ty = i / (uint32_t)t.groupX;
            ids.groupThreadId = hlsl::uint3(tx, ty, 0u);
            ids.dispatchThreadId = hlsl::uint3(gx * (uint32_t)t.groupX + tx, gy * (uint32_t)t.groupY + ty, 0u);
            ids.groupIndex = i;
            t.thunk(ids);
        }
        return;
    }
    r.Ensure(n);
    r.thunk = t.thunk;
    t_runner = &r;
    t_inFiber = true;
    for (uint32_t i = 0; i < n; i++) {
        Fiber& f = r.fibers[i];
        const uint32_t tx = i % (uint32_t)t.groupX, ty = i / (uint32_t)t.groupX;
        f.ids = ids;
        f.ids.groupThreadId = hlsl::uint3(tx, ty, 0u);
        f.ids.dispatchThreadId = hlsl::uint3(gx * (uint32_t)t.groupX + tx, gy * (uint32_t)t.groupY + ty, 0u);
        f.ids.groupIndex = i;
        f.state = READY;
        getcontext(&f.ctx);
        f.ctx.uc_stack.ss_sp = f.stack;
        f.ctx.uc_stack.ss_size = STACK_BYTES;
        f.ctx.uc_link = &r.scheduler;
        makecontext(&f.ctx, FiberEntry, 0);
    }
    for (;;) {
        bool alive = false;
        for (uint32_t i = 0; i < n; i++) {
            Fiber& f = r.fibers[i];
            if (f.state != READY)
                continue;
            r.running = &f;
            swapcontext(&r.scheduler, &f.ctx);
        }
        for (uint32_t i = 0; i < n; i++) // every live thread is at the barrier now: open it
            if (r.fibers[i].state == AT_BARRIER) {
                r.fibers[i].state = READY;
                alive = true;
            }
        if (!alive)
            break;
    }
    t_inFiber = false;
}

} // namespace hlsl_rt

using namespace hlsl_rt;

extern "C" {

__attribute__((visibility("default"))) int nrdref_count() { return (int)Registry().size(); }
__attribute__((visibility("default"))) const char* nrdref_name(int i) { return Registry()[(size_t)i]->fileName.c_str(); }
__attribute__((visibility("default"))) int nrdref_has(const char* shaderFileName) {
    for (ShaderTable* t : Registry())
        if (t->fileName == shaderFileName)
            return 1;
    return 0;
}
__attribute__((visibility("default"))) int nrdref_set_threads(int n) {
#ifdef _OPENMP
    if (n > 0)
        omp_set_num_threads(n);
    return omp_get_max_threads();
#else
    (void)n;
    return 1;
#endif
}

// Runs one dispatch of the reference shader `shaderFileName` (DispatchDesc: constants, resources in binding order, grid in thread groups).
// 0 = done, 1 = no such shader in this build, 2 = the constants / resources do not fit the shader's declarations.
__attribute__((visibility("default"))) int nrdref_dispatch(const char* shaderFileName, const void* constants, uint32_t constantsSize, const Plane* planes, uint32_t planesNum, uint32_t gridW, uint32_t gridH) {
    ShaderTable* t = nullptr;
    for (ShaderTable* c : Registry())
        if (c->fileName == shaderFileName)
            t = c;
    if (!t) {
        fprintf(stderr, "nrdref_dispatch: unknown shader '%s'\n", shaderFileName);
        return 1;
    }
    if (!t->constants.empty() && constantsSize) { // (a dispatch without constant data leaves the shader's constants untouched: Clear_*.cs never reads its dummy)
        const uint32_t used = UnpackConstants(*t, (const uint8_t*)constants, constantsSize);
        if (used == 0 || ((used + 15u) & ~15u) != ((constantsSize + 15u) & ~15u)) {
            fprintf(stderr, "nrdref_dispatch: '%s' declares %u bytes of constants, the dispatch carries %u\n", shaderFileName, used, constantsSize);
            return 2;
        }
    }
    uint32_t numInputs = 0;
    for (const ResourceReg& r : t->resources)
        if (!r.output)
            numInputs++;
    if (t->resources.size() != planesNum) {
        fprintf(stderr, "nrdref_dispatch: '%s' declares %zu resources, the dispatch binds %u\n", shaderFileName, t->resources.size(), planesNum);
        return 2;
    }
    for (const ResourceReg& r : t->resources) {
        const uint32_t slot = r.output ? numInputs + (uint32_t)r.index : (uint32_t)r.index;
        if (slot >= planesNum) {
            fprintf(stderr, "nrdref_dispatch: '%s' resource %s has no slot\n", shaderFileName, r.name);
            return 2;
        }
        *r.plane = planes[slot];
    }
    const int64_t groups = (int64_t)gridW * gridH;
#pragma omp parallel
    {
        GroupRunner runner;
#pragma omp for schedule(dynamic, 4)
        for (int64_t g = 0; g < groups; g++)
            RunGroup(*t, runner, (uint32_t)(g % gridW), (uint32_t)(g / gridW));
        for (auto& f : runner.fibers)
            munmap(f.stack, STACK_BYTES);
    }
    return 0;
}
}
