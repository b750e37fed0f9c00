// L1 / LDS gather-rate probe for gfx950: how many CU cycles one wave64 load instruction costs when every lane addresses its own texel (the access
// pattern of the Poisson taps and of the history footprints), against the same loads with adjacent lanes reading adjacent texels, and against LDS.
//   hipcc --offload-arch=gfx950 -O2 tools/gather_bench.hip -o tools/build/gather_bench && tools/build/gather_bench
// Working set: a 2 MiB window per workgroup region (L2-resident after the first touch; the pass kernels' taps are L2 / L1 hits too).
// Reported: average cycles per wave-load at 4 waves per SIMD (16 waves per CU), all CUs busy; "lines" = distinct 128-byte lines a wave touches per load.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

template <typename T, int MODE>
__global__ __launch_bounds__(256) void Gather(const T* __restrict__ src, float* out, int iters, unsigned windowTexels) {
    // MODE 0: adjacent lanes read adjacent texels (coalesced); 1: every lane its own pseudo-random texel inside a 64 x 64-texel neighbourhood (a tap cloud);
    // 2: every lane a random texel of the whole window
    unsigned lane = threadIdx.x, state = (blockIdx.x * 256u + lane) * 2654435761u + 12345u;
    const unsigned base = (blockIdx.x * 4099u) % (windowTexels - 8192u);
    float acc = 0.0f;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int k = 0; k < 8; k++) {
            state = state * 1664525u + 1013904223u;
            unsigned idx;
            if (MODE == 0)
                idx = base + (unsigned)(i * 8 + k) * 256u % 4096u + lane;
            else if (MODE == 1)
                idx = base + ((state >> 8) & 63u) + (((state >> 16) & 63u) << 6); // 4096 texels = 64 x 64 cloud
            else
                idx = (state >> 4) % windowTexels;
            T v = src[idx];
            acc += ((const float*)&v)[0];
        }
    }
    out[blockIdx.x * 256 + lane] = acc;
}

__global__ __launch_bounds__(256) void LdsGather(float* out, int iters) {
    __shared__ float4 tile[2048]; // 32 KiB
    for (int i = threadIdx.x; i < 2048; i += 256)
        tile[i] = make_float4(i, 1, 2, 3);
    __syncthreads();
    unsigned state = (blockIdx.x * 256u + threadIdx.x) * 2654435761u + 12345u;
    float acc = 0.0f;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int k = 0; k < 8; k++) {
            state = state * 1664525u + 1013904223u;
            float4 v = tile[(state >> 8) & 2047u];
            acc += v.x;
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

template <typename F>
static double TimeIt(F launch) {
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    launch(2);
    hipDeviceSynchronize();
    hipEventRecord(a, 0);
    launch(400);
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    return ms;
}

int main() {
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    const double ghz = prop.clockRate / 1e6;
    const unsigned windowBytes = 64u << 20;
    void* src = nullptr;
    float* out = nullptr;
    hipMalloc(&src, windowBytes);
    hipMemset(src, 0, windowBytes);
    hipMalloc((void**)&out, sizeof(float) * 256 * cus * 4);
    const int blocks = cus * 4; // 4 x 256 threads per CU = 4 waves per SIMD
    printf("device %s, %d CUs, %.2f GHz; cycles per wave64 load instruction per CU (16 waves per CU resident)\n", prop.name, cus, ghz);
    auto report = [&](const char* name, double ms, int iters) {
        // a CU executed 16 waves x iters x 8 loads
        const double cycles = ms * 1e-3 * ghz * 1e9 / (16.0 * iters * 8.0);
        printf("%-44s %8.1f cycles / wave-load\n", name, cycles);
    };
#define RUN(T, MODE, NAME) report(NAME, TimeIt([&](int it) { hipLaunchKernelGGL((Gather<T, MODE>), dim3(blocks), dim3(256), 0, 0, (const T*)src, out, it, (unsigned)(windowBytes / sizeof(T))); }), 400);
    RUN(float4, 0, "16 B / lane, coalesced")
    RUN(float4, 1, "16 B / lane, 64x64 tap cloud")
    RUN(float4, 2, "16 B / lane, random in 64 MiB")
    RUN(float2, 0, "8 B / lane, coalesced")
    RUN(float2, 1, "8 B / lane, 64x64 tap cloud")
    RUN(float, 0, "4 B / lane, coalesced")
    RUN(float, 1, "4 B / lane, 64x64 tap cloud")
    report("LDS 16 B / lane, random (ds_read_b128)", TimeIt([&](int it) { hipLaunchKernelGGL(LdsGather, dim3(blocks), dim3(256), 0, 0, out, it); }), 400);
    return 0;
}
