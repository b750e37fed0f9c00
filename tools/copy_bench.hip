// Which plain copy kernel gets the most out of the HBM of this GPU? (developer tooling; decides the shape of nrdHipMeasureCopyBandwidth's kernel, the
// "measured copy bandwidth" the roofline fractions of bench.py are quoted against)   hipcc --offload-arch=gfx950 -O3 tools/copy_bench.hip -o tools/build/copy_bench
#include <hip/hip_runtime.h>

#include <cstdio>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4))); // a native vector: what the nontemporal builtins accept
template <int UNROLL, bool NT>
__global__ __launch_bounds__(256) void Copy(const u32x4* __restrict__ src, u32x4* __restrict__ dst, size_t count) {
    const size_t stride = (size_t)gridDim.x * 256u;
    size_t i = (size_t)blockIdx.x * 256u + threadIdx.x;
    for (; i + (UNROLL - 1) * stride < count; i += UNROLL * stride) {
        u32x4 v[UNROLL];
#pragma unroll
        for (int k = 0; k < UNROLL; k++)
            v[k] = NT ? __builtin_nontemporal_load(src + i + k * stride) : src[i + k * stride];
#pragma unroll
        for (int k = 0; k < UNROLL; k++) {
            if (NT)
                __builtin_nontemporal_store(v[k], dst + i + k * stride);
            else
                dst[i + k * stride] = v[k];
        }
    }
    for (; i < count; i += stride)
        dst[i] = src[i];
}

template <int UNROLL, bool NT>
static void Run(const char* name, const u32x4* src, u32x4* dst, size_t bytes, int blocks) {
    hipEvent_t a, b;
    hipEventCreate(&a), hipEventCreate(&b);
    for (int i = 0; i < 3; i++)
        hipLaunchKernelGGL((Copy<UNROLL, NT>), dim3(blocks), dim3(256), 0, 0, src, dst, bytes / 16);
    hipEventRecord(a, 0);
    const int reps = 10;
    for (int i = 0; i < reps; i++)
        hipLaunchKernelGGL((Copy<UNROLL, NT>), dim3(blocks), dim3(256), 0, 0, src, dst, bytes / 16);
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    printf("%-34s blocks %7d  %7.1f GB/s (read + write)\n", name, blocks, 2.0 * bytes * reps / (ms * 1e-3) / 1e9);
}

int main() {
    for (size_t bytes : {(size_t)1 << 30, (size_t)256 << 20}) {
        u32x4 *src, *dst;
        hipMalloc(&src, bytes), hipMalloc(&dst, bytes);
        hipMemset(src, 1, bytes);
        printf("---- %zu MiB\n", bytes >> 20);
        for (int blocks : {1024, 2048, 4096, 8192, 16384, 65536, (int)(bytes / 16 / 256)}) {
            Run<1, false>("1 x 16 B per lane", src, dst, bytes, blocks);
            Run<4, false>("4 x 16 B per lane", src, dst, bytes, blocks);
            Run<4, true>("4 x 16 B per lane, nontemporal", src, dst, bytes, blocks);
            Run<8, false>("8 x 16 B per lane", src, dst, bytes, blocks);
        }
        hipEvent_t a, b;
        hipEventCreate(&a), hipEventCreate(&b);
        hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, 0);
        hipEventRecord(a, 0);
        for (int i = 0; i < 10; i++)
            hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, 0);
        hipEventRecord(b, 0);
        hipEventSynchronize(b);
        float ms;
        hipEventElapsedTime(&ms, a, b);
        printf("hipMemcpyAsync device-to-device                    %7.1f GB/s\n", 2.0 * bytes * 10 / (ms * 1e-3) / 1e9);
        hipFree(src), hipFree(dst);
    }
    return 0;
}
