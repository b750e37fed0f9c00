// Measures gfx950's one-instruction transcendentals (v_rcp_f32, v_sqrt_f32, v_rsq_f32, v_exp_f32, v_log_f32) over every mantissa of the
// argument ranges the numerics contract evaluates them on, as the deviation (in ulps) from a DETERMINISTIC reference result -- the double-precision
// functions of oracle/hw_ref.h, which the CPU oracle evaluates again when it emulates the instructions (oracle/hlsl.h HwRcp / HwSqrt / HwRsq /
// HwExp2 / HwLog2). Developer tooling (run on the GPU box); the tables it writes are committed as oracle/hw_*.i8.z.
//   hipcc --offload-arch=gfx950 -O2 -I oracle tools/hw_tables.hip -o tools/build/hw_tables && tools/build/hw_tables <outdir>
// It also tests, over many binades, the range-reduction identities the contract relies on (see "hypotheses" below) and prints how often they fail.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "hw_ref.h"

enum Op { OP_RCP, OP_SQRT, OP_RSQ, OP_EXP2, OP_LOG2 };

__global__ void Eval(int op, uint32_t firstBits, uint32_t count, uint32_t* out) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count)
        return;
    float x = __uint_as_float(firstBits + i), r = 0.0f;
    switch (op) {
        case OP_RCP: r = __builtin_amdgcn_rcpf(x); break;
        case OP_SQRT: r = __builtin_amdgcn_sqrtf(x); break;
        case OP_RSQ: r = __builtin_amdgcn_rsqf(x); break;
        case OP_EXP2: r = __builtin_amdgcn_exp2f(x); break;
        case OP_LOG2: r = __builtin_amdgcn_logf(x); break;
    }
    out[i] = __float_as_uint(r);
}

static uint32_t Bits(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    return u;
}
static float FromBits(uint32_t u) {
    float f;
    memcpy(&f, &u, 4);
    return f;
}

static std::vector<uint32_t> Run(int op, uint32_t firstBits, uint32_t count) {
    uint32_t* d;
    hipMalloc(&d, (size_t)count * 4);
    hipLaunchKernelGGL(Eval, dim3((count + 255) / 256), dim3(256), 0, 0, op, firstBits, count, d);
    std::vector<uint32_t> h(count);
    hipMemcpy(h.data(), d, (size_t)count * 4, hipMemcpyDeviceToHost);
    hipFree(d);
    return h;
}

static float Ref(int op, float x) {
    switch (op) {
        case OP_RCP: return hwref::RefRcp(x);
        case OP_SQRT: return hwref::RefSqrt(x);
        case OP_RSQ: return hwref::RefRsq(x);
        case OP_EXP2: return hwref::RefExp2(x);
        default: return hwref::RefLog2(x);
    }
}

// deviation table over [firstBits, firstBits + count) as int8; returns false if a deviation does not fit
static bool Table(int op, const char* name, uint32_t firstBits, uint32_t count, const std::string& outdir, const char* file) {
    std::vector<uint32_t> hw = Run(op, firstBits, count);
    std::vector<int8_t> delta(count);
    std::map<long, long> hist;
    bool fits = true;
    for (uint32_t i = 0; i < count; i++) {
        const float x = FromBits(firstBits + i);
        const long d = (long)(int32_t)hw[i] - (long)(int32_t)Bits(Ref(op, x));
        hist[d]++;
        if (d < -127 || d > 127)
            fits = false;
        delta[i] = (int8_t)d;
    }
    printf("%s over %u inputs from %.9g (bits 0x%08x): deviation in ulps from the reference -> count:", name, count, FromBits(firstBits), firstBits);
    int shown = 0;
    for (auto& kv : hist)
        if (shown++ < 16)
            printf(" %ld:%ld", kv.first, kv.second);
    printf("%s\n", fits ? "" : "  [DOES NOT FIT int8]");
    if (file && fits) {
        FILE* fp = fopen((outdir + "/" + file).c_str(), "wb");
        fwrite(delta.data(), 1, count, fp);
        fclose(fp);
    }
    return fits;
}

int main(int argc, char** argv) {
    const std::string outdir = argc > 1 ? argv[1] : ".";
    const uint32_t ONE = 0x3f800000u, M = 1u << 23;

    // ---- the tables ------------------------------------------------------------------------------------------------------------------------
    Table(OP_RCP, "v_rcp_f32", ONE, M, outdir, "hw_rcp.i8");            // [1, 2): 1/x scales exactly with the exponent
    Table(OP_SQRT, "v_sqrt_f32", ONE, 2 * M, outdir, "hw_sqrt.i8");     // [1, 4): both exponent parities
    Table(OP_RSQ, "v_rsq_f32", ONE, 2 * M, outdir, "hw_rsq.i8");
    Table(OP_EXP2, "v_exp_f32", ONE, M + 1, outdir, "hw_exp2.i8");      // [1, 2]: the contract evaluates 2^(1 + frac(x)) and scales by 2^(floor(x) - 1)
    Table(OP_LOG2, "v_log_f32", ONE, M, outdir, "hw_log2.i8");          // [1, 2): the contract evaluates e + log2(m)
    Table(OP_LOG2, "v_log_f32 on [0.5, 1)", ONE - M, M, outdir, "hw_log2_half.i8");

    // ---- hypotheses ------------------------------------------------------------------------------------------------------------------------
    // (R) v_rcp_f32(m * 2^e) == v_rcp_f32(m) * 2^-e for normal results; (S) likewise sqrt / rsq with 4^k
    {
        std::vector<uint32_t> base = Run(OP_RCP, ONE, M);
        for (int e : {-100, -20, -1, 1, 7, 60, 120}) {
            std::vector<uint32_t> hw = Run(OP_RCP, ONE + ((uint32_t)e << 23), M);
            long bad = 0;
            for (uint32_t i = 0; i < M; i++)
                bad += hw[i] != base[i] - ((uint32_t)e << 23);
            printf("(R) v_rcp_f32 on [2^%d, 2^%d): %ld of %u differ from the scaled [1, 2) result\n", e, e + 1, bad, M);
        }
        std::vector<uint32_t> neg = Run(OP_RCP, ONE | 0x80000000u, M);
        long bad = 0;
        for (uint32_t i = 0; i < M; i++)
            bad += neg[i] != (base[i] | 0x80000000u);
        printf("(R) v_rcp_f32 on (-2, -1]: %ld differ from the negated result\n", bad);
    }
    // (E) v_exp_f32(x) for |x| >= 1 == v_exp_f32(1 + frac(x)) * 2^(floor(x) - 1) ?
    {
        std::vector<uint32_t> base = Run(OP_EXP2, ONE, M + 1); // t in [1, 2]
        for (int k : {1, 2, 3, 6}) {
            for (int sign = 0; sign < 2; sign++) {
                const uint32_t first = (ONE + ((uint32_t)k << 23)) | (sign ? 0x80000000u : 0u);
                std::vector<uint32_t> hw = Run(OP_EXP2, first, M);
                long bad = 0, badUlp1 = 0;
                for (uint32_t i = 0; i < M; i += 1) {
                    const float x = FromBits(first + i);
                    const float fl = floorf(x);
                    const float f = x - fl; // exact
                    const float t = 1.0f + f;
                    const uint32_t b = base[Bits(t) - ONE];
                    const float want = ldexpf(FromBits(b), (int)fl - 1);
                    if (Bits(want) != hw[i]) {
                        bad++;
                        if (labs((long)(int32_t)Bits(want) - (long)(int32_t)hw[i]) > 1)
                            badUlp1++;
                    }
                }
                printf("(E) v_exp_f32 on %s[2^%d, 2^%d): %ld of %u differ from the reduced form (%ld by more than 1 ulp)\n", sign ? "-" : "+", k, k + 1, bad, M, badUlp1);
            }
        }
        // small arguments: binades below 1, compared with the reduced form AND with the reference
        for (int k : {-1, -2, -4, -8, -12, -16, -20, -24}) {
            for (int sign = 0; sign < 2; sign++) {
                const uint32_t first = (ONE + ((uint32_t)k << 23)) | (sign ? 0x80000000u : 0u);
                std::vector<uint32_t> hw = Run(OP_EXP2, first, M);
                long bad = 0, badRef = 0;
                for (uint32_t i = 0; i < M; i++) {
                    const float x = FromBits(first + i);
                    const float fl = floorf(x);
                    const float f = x - fl;
                    const float t = 1.0f + f; // rounds
                    const uint32_t b = base[Bits(t) - ONE];
                    const float want = ldexpf(FromBits(b), (int)fl - 1);
                    bad += Bits(want) != hw[i];
                    badRef += Bits(hwref::RefExp2(x)) != hw[i];
                }
                printf("(E) v_exp_f32 on %s[2^%d, 2^%d): %ld of %u differ from the reduced form, %ld from the correctly rounded result\n", sign ? "-" : "+", k, k + 1, bad, M, badRef);
            }
        }
    }
    // (L) v_log_f32(m * 2^e) == float(e) + v_log_f32(m) (one fp32 addition) ?
    {
        std::vector<uint32_t> base = Run(OP_LOG2, ONE, M);
        for (int e : {-120, -10, -2, -1, 1, 2, 3, 10, 100}) {
            std::vector<uint32_t> hw = Run(OP_LOG2, ONE + ((uint32_t)e << 23), M);
            long bad = 0, bad1 = 0;
            for (uint32_t i = 0; i < M; i++) {
                const float want = (float)e + FromBits(base[i]);
                if (Bits(want) != hw[i]) {
                    bad++;
                    if (labs((long)(int32_t)Bits(want) - (long)(int32_t)hw[i]) > 1)
                        bad1++;
                }
            }
            printf("(L) v_log_f32 on [2^%d, 2^%d): %ld of %u differ from e + v_log_f32(m) (%ld by more than 1 ulp)\n", e, e + 1, bad, M, bad1);
        }
    }
    // specials
    for (int op = 0; op < 5; op++) {
        const float specials[] = {0.0f, -0.0f, INFINITY, -INFINITY, NAN, 1e-45f, 1e-39f, -1e-39f, 3.4e38f, -1.0f, 1.0f, 2.0f, 1.17549435e-38f, 128.0f, -126.0f, -127.0f, -149.0f, -150.0f, 127.99999f};
        static const char* names[] = {"v_rcp_f32", "v_sqrt_f32", "v_rsq_f32", "v_exp_f32", "v_log_f32"};
        printf("%s specials:", names[op]);
        for (float s : specials) {
            std::vector<uint32_t> r = Run(op, Bits(s), 1);
            printf("  %g -> %g (0x%08x)", s, FromBits(r[0]), r[0]);
        }
        printf("\n");
    }
    return 0;
}
