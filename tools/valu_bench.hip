// VALU issue-rate probe for gfx950: how many cycles one wave64 instruction of each kind occupies a SIMD, alone and with 2 / 4 / 8 waves per SIMD.
// The pass kernels are VALU-issue-bound (profiles/), so these figures are the price list the kernel restructuring is planned against.
//   hipcc --offload-arch=gfx950 -O2 tools/valu_bench.hip -o tools/build/valu_bench && tools/build/valu_bench
// Every probe is a loop of 16 independent chains x 16 repetitions of one instruction (256 per iteration), 2000 iterations; time from s_memtime
// (constant 100 MHz) and hipEvents; reported as SIMD cycles per wave-instruction at the measured shader clock.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define REP16(x) x x x x x x x x x x x x x x x x

#define PROBE(NAME, ASM)                                                                                                      \
    __global__ __launch_bounds__(256) void NAME(float* out, int iters) {                                                       \
        float v0 = threadIdx.x * 1e-3f + 1.0f, v1 = v0 + 1.0f, v2 = v0 + 2.0f, v3 = v0 + 3.0f, v4 = v0 + 4.0f, v5 = v0 + 5.0f, v6 = v0 + 6.0f, v7 = v0 + 7.0f; \
        float v8 = v0 + 8.0f, v9 = v0 + 9.0f, v10 = v0 + 10.0f, v11 = v0 + 11.0f, v12 = v0 + 12.0f, v13 = v0 + 13.0f, v14 = v0 + 14.0f, v15 = v0 + 15.0f;      \
        float k = 0.999f;                                                                                                      \
        for (int i = 0; i < iters; i++) {                                                                                      \
            REP16(asm volatile(ASM : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7), "+v"(v8), "+v"(v9), "+v"(v10), "+v"(v11), "+v"(v12), "+v"(v13), "+v"(v14), "+v"(v15) : "v"(k) : "vcc", "s20", "s21");) \
        }                                                                                                                      \
        out[blockIdx.x * 256 + threadIdx.x] = v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7 + v8 + v9 + v10 + v11 + v12 + v13 + v14 + v15;                            \
    }

#define I16(op, fmt) \
    op " %0, " fmt(0) "\n" op " %1, " fmt(1) "\n" op " %2, " fmt(2) "\n" op " %3, " fmt(3) "\n" op " %4, " fmt(4) "\n" op " %5, " fmt(5) "\n" op " %6, " fmt(6) "\n" op " %7, " fmt(7) "\n" \
    op " %8, " fmt(8) "\n" op " %9, " fmt(9) "\n" op " %10, " fmt(10) "\n" op " %11, " fmt(11) "\n" op " %12, " fmt(12) "\n" op " %13, " fmt(13) "\n" op " %14, " fmt(14) "\n" op " %15, " fmt(15) "\n"

#define F_FMA(n) "%" #n ", %16, %" #n
#define F_MUL(n) "%" #n ", %16"
#define F_UN(n) "%" #n
#define F_CND(n) "%" #n ", %16, vcc"

PROBE(k_fma, I16("v_fma_f32", F_FMA))
PROBE(k_mul, I16("v_mul_f32", F_MUL))
PROBE(k_add, I16("v_add_f32", F_MUL))
PROBE(k_max, I16("v_max_f32", F_MUL))
PROBE(k_mov, I16("v_mov_b32", F_UN))
PROBE(k_cndmask, I16("v_cndmask_b32", F_CND))
PROBE(k_rcp, I16("v_rcp_f32", F_UN))
PROBE(k_sqrt, I16("v_sqrt_f32", F_UN))
PROBE(k_rsq, I16("v_rsq_f32", F_UN))
PROBE(k_exp, I16("v_exp_f32", F_UN))
PROBE(k_log, I16("v_log_f32", F_UN))
PROBE(k_cvt_f32_f16, I16("v_cvt_f32_f16", F_UN))
PROBE(k_cvt_f16_f32, I16("v_cvt_f16_f32", F_UN))
PROBE(k_floor, I16("v_floor_f32", F_UN))
PROBE(k_cvt_i32, I16("v_cvt_i32_f32", F_UN))
PROBE(k_mul_u24, I16("v_mul_u32_u24", F_MUL))
PROBE(k_mul_lo, I16("v_mul_lo_u32", F_MUL))
PROBE(k_min, I16("v_min_f32", F_MUL))
#define F_MED3(n) "%" #n ", %16, 0"
PROBE(k_med3, I16("v_med3_f32", F_MED3))
PROBE(k_and, I16("v_and_b32", F_MUL))
#define F_SHL(n) "1, %" #n
PROBE(k_lshl, I16("v_lshlrev_b32", F_SHL))
PROBE(k_add_u32, I16("v_add_u32", F_MUL))
#define F_MAD24(n) "%" #n ", %16, %" #n
PROBE(k_mad_u24, I16("v_mad_u32_u24", F_MAD24))
#define F_LSHLADD(n) "%" #n ", 2, %16"
PROBE(k_lshl_add, I16("v_lshl_add_u32", F_LSHLADD))
#define F_BFE(n) "%" #n ", 3, 10"
PROBE(k_bfe, I16("v_bfe_u32", F_BFE))
PROBE(k_cvt_f32_u32, I16("v_cvt_f32_u32", F_UN))
#define F_CND64(n) "%" #n ", %16, s[20:21]"
PROBE(k_cndmask64, I16("v_cndmask_b32_e64", F_CND64))
#define F_ABSSUB(n) "|%" #n "|, %16"
PROBE(k_sub_abs, I16("v_sub_f32_e64", F_ABSSUB))
#define F_MULCLAMP(n) "%" #n ", %16 clamp"
PROBE(k_mul_clamp, I16("v_mul_f32_e64", F_MULCLAMP))
// compare + select: the pair the Select() idiom costs (v_cmp writes vcc, v_cndmask reads it)
#define CMPSEL(n) "v_cmp_gt_f32 vcc, %" #n ", %16\nv_cndmask_b32 %" #n ", %" #n ", %16, vcc\n"
PROBE(k_cmp_cndmask, CMPSEL(0) CMPSEL(1) CMPSEL(2) CMPSEL(3) CMPSEL(4) CMPSEL(5) CMPSEL(6) CMPSEL(7))
#define CMP64(n) "v_cmp_gt_f32_e64 s[20:21], %" #n ", %16\n"
PROBE(k_cmp64, CMP64(0) CMP64(1) CMP64(2) CMP64(3) CMP64(4) CMP64(5) CMP64(6) CMP64(7) CMP64(8) CMP64(9) CMP64(10) CMP64(11) CMP64(12) CMP64(13) CMP64(14) CMP64(15))

// packed: register pairs
#define PROBE2(NAME, OP)                                                                                                       \
    __global__ __launch_bounds__(256) void NAME(float* out, int iters) {                                                       \
        typedef float f2 __attribute__((ext_vector_type(2)));                                                                  \
        f2 v0 = {threadIdx.x * 1e-3f + 1.0f, 2.0f}, v1 = v0 + 1.0f, v2 = v0 + 2.0f, v3 = v0 + 3.0f, v4 = v0 + 4.0f, v5 = v0 + 5.0f, v6 = v0 + 6.0f, v7 = v0 + 7.0f; \
        f2 k = {0.999f, 1.001f};                                                                                               \
        for (int i = 0; i < iters; i++) {                                                                                      \
            REP16(asm volatile(OP " %0, %0, %8, %0\n" OP " %1, %1, %8, %1\n" OP " %2, %2, %8, %2\n" OP " %3, %3, %8, %3\n" OP " %4, %4, %8, %4\n" OP " %5, %5, %8, %5\n" OP " %6, %6, %8, %6\n" OP " %7, %7, %8, %7\n" \
                               OP " %0, %0, %8, %0\n" OP " %1, %1, %8, %1\n" OP " %2, %2, %8, %2\n" OP " %3, %3, %8, %3\n" OP " %4, %4, %8, %4\n" OP " %5, %5, %8, %5\n" OP " %6, %6, %8, %6\n" OP " %7, %7, %8, %7\n" \
                               : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7) : "v"(k));)       \
        }                                                                                                                      \
        f2 s = v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7;                                                                          \
        out[blockIdx.x * 256 + threadIdx.x] = s.x + s.y;                                                                       \
    }
PROBE2(k_pk_fma, "v_pk_fma_f32")
#define PROBE2B(NAME, OP)                                                                                                      \
    __global__ __launch_bounds__(256) void NAME(float* out, int iters) {                                                       \
        typedef float f2 __attribute__((ext_vector_type(2)));                                                                  \
        f2 v0 = {threadIdx.x * 1e-3f + 1.0f, 2.0f}, v1 = v0 + 1.0f, v2 = v0 + 2.0f, v3 = v0 + 3.0f, v4 = v0 + 4.0f, v5 = v0 + 5.0f, v6 = v0 + 6.0f, v7 = v0 + 7.0f; \
        f2 k = {0.999f, 1.001f};                                                                                               \
        for (int i = 0; i < iters; i++) {                                                                                      \
            REP16(asm volatile(OP " %0, %0, %8\n" OP " %1, %1, %8\n" OP " %2, %2, %8\n" OP " %3, %3, %8\n" OP " %4, %4, %8\n" OP " %5, %5, %8\n" OP " %6, %6, %8\n" OP " %7, %7, %8\n" \
                               OP " %0, %0, %8\n" OP " %1, %1, %8\n" OP " %2, %2, %8\n" OP " %3, %3, %8\n" OP " %4, %4, %8\n" OP " %5, %5, %8\n" OP " %6, %6, %8\n" OP " %7, %7, %8\n" \
                               : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7) : "v"(k));)       \
        }                                                                                                                      \
        f2 s = v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7;                                                                          \
        out[blockIdx.x * 256 + threadIdx.x] = s.x + s.y;                                                                       \
    }
PROBE2B(k_pk_mul, "v_pk_mul_f32")
PROBE2B(k_pk_add, "v_pk_add_f32")

// fp16 operand straight into an fp32 FMA (no separate conversion): v_fma_mix_f32 dst, a (fp16 low half), b, c
#define F_MIX(n) "%16, %16, %" #n " op_sel_hi:[1,0,0]"
PROBE(k_fma_mix, I16("v_fma_mix_f32", F_MIX))

struct Probe {
    const char* name;
    void (*fn)(float*, int);
    int perIter; // wave-instructions per loop iteration
};

int main() {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, 0) != hipSuccess) {
        printf("no device\n");
        return 1;
    }
    const int cus = prop.multiProcessorCount;
    const double clockGHz = prop.clockRate / 1e6;
    printf("device %s, %d CUs, clock %.3f GHz\n", prop.name, cus, clockGHz);
    float* out = nullptr;
    hipMalloc((void**)&out, sizeof(float) * 256 * cus * 8 * 4);
    const Probe probes[] = {{"v_fma_f32", k_fma, 256}, {"v_mul_f32", k_mul, 256}, {"v_add_f32", k_add, 256}, {"v_max_f32", k_max, 256}, {"v_mov_b32", k_mov, 256}, {"v_cndmask_b32", k_cndmask, 256},
        {"v_pk_fma_f32", k_pk_fma, 256}, {"v_pk_mul_f32", k_pk_mul, 256}, {"v_pk_add_f32", k_pk_add, 256}, {"v_fma_mix_f32", k_fma_mix, 256}, {"v_rcp_f32", k_rcp, 256}, {"v_sqrt_f32", k_sqrt, 256},
        {"v_rsq_f32", k_rsq, 256}, {"v_exp_f32", k_exp, 256}, {"v_log_f32", k_log, 256}, {"v_cvt_f32_f16", k_cvt_f32_f16, 256}, {"v_cvt_f16_f32", k_cvt_f16_f32, 256}, {"v_floor_f32", k_floor, 256},
        {"v_cvt_i32_f32", k_cvt_i32, 256}, {"v_mul_u32_u24", k_mul_u24, 256}, {"v_mul_lo_u32", k_mul_lo, 256}, {"v_min_f32", k_min, 256}, {"v_med3_f32", k_med3, 256}, {"v_and_b32", k_and, 256},
        {"v_lshlrev_b32", k_lshl, 256}, {"v_add_u32", k_add_u32, 256}, {"v_mad_u32_u24", k_mad_u24, 256}, {"v_lshl_add_u32", k_lshl_add, 256}, {"v_bfe_u32", k_bfe, 256},
        {"v_cvt_f32_u32", k_cvt_f32_u32, 256}, {"v_cndmask_b32_e64(sgpr)", k_cndmask64, 256}, {"v_sub_f32 |abs|", k_sub_abs, 256}, {"v_mul_f32 clamp", k_mul_clamp, 256},
        {"v_cmp+v_cndmask pair", k_cmp_cndmask, 256}, {"v_cmp_gt_f32_e64", k_cmp64, 256}};
    const int iters = 2000;
    printf("%-16s %12s %12s %12s %12s   (SIMD cycles per wave64 instruction at 1 / 2 / 4 / 8 waves per SIMD; one 256-thread block = 1 wave per SIMD)\n", "instruction", "1 w/SIMD", "2 w/SIMD", "4 w/SIMD", "8 w/SIMD");
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    for (const Probe& p : probes) {
        printf("%-16s", p.name);
        for (int wps : {1, 2, 4, 8}) {
            const int blocks = cus * wps; // a 256-thread block puts one wave on each of the 4 SIMDs of a CU
            hipLaunchKernelGGL(p.fn, dim3(blocks), dim3(256), 0, 0, out, 10);
            hipDeviceSynchronize();
            hipEventRecord(a, 0);
            hipLaunchKernelGGL(p.fn, dim3(blocks), dim3(256), 0, 0, out, iters);
            hipEventRecord(b, 0);
            hipEventSynchronize(b);
            float ms = 0;
            hipEventElapsedTime(&ms, a, b);
            // each SIMD executed wps waves x iters x perIter instructions
            const double cyclesPerInstr = (ms * 1e-3 * clockGHz * 1e9) / ((double)wps * iters * p.perIter);
            printf(" %12.2f", cyclesPerInstr);
        }
        printf("\n");
    }
    hipFree(out);
    return 0;
}
