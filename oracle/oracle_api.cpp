// ORACLE -- TEST INFRASTRUCTURE, NOT PRODUCT CODE (see oracle/hlsl.h).
// C entry points used by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg (through oracle/driver.py).
#include "ml.h"
#include "passes.h"

#include <cstdio>
#include <cstdlib>
#include <vector>

#ifdef _OPENMP
#    include <omp.h>
#endif

using namespace orc;

namespace hwmath {
const signed char* g_RcpDelta = nullptr;
const signed char* g_SqrtDelta = nullptr;
const signed char* g_RsqDelta = nullptr;
const signed char* g_Exp2Delta = nullptr;
const signed char* g_Log2Delta = nullptr;
const signed char* g_Exp2NegDelta = nullptr;
int g_IeeeMode = 0;
void TablesMissing(const char* which) {
    fprintf(stderr, "oracle: hardware deviation table missing or argument out of its range: %s (oracle/hw_*.i8.z through oracle/driver.py)\n", which);
    abort();
}
} // namespace hwmath

extern "C" {

struct OraclePlane { // same layout as NrdHipPlaneDesc, but "data" is HOST memory
    void* data;
    uint32_t rowPitchBytes;
    uint32_t format;
    uint16_t width, height;
};

// Runs one pass on the CPU. Returns 0 on success, 1 if the pass is unknown to the oracle.
__attribute__((visibility("default"))) int oracle_dispatch(const char* shaderFileName, const void* constants, uint32_t constantsSize, const OraclePlane* planes, uint32_t planesNum) {
    const PassEntry* tables[4];
    uint32_t counts[4];
    tables[0] = GetCommonPasses(counts[0]);
    tables[1] = GetReblurPasses(counts[1]);
    tables[2] = GetSigmaPasses(counts[2]);
    tables[3] = GetRelaxPasses(counts[3]);
    for (int t = 0; t < 4; t++)
        for (uint32_t i = 0; i < counts[t]; i++)
            if (!strcmp(tables[t][i].shaderFileName, shaderFileName)) {
                std::vector<Tex> tex(planesNum);
                for (uint32_t p = 0; p < planesNum; p++)
                    tex[p] = Tex(Plane{(uint8_t*)planes[p].data, planes[p].rowPitchBytes, planes[p].format, planes[p].width, planes[p].height});
                PassIO io{tex.data(), planesNum, constants, constantsSize};
                tables[t][i].fn(io);
                return 0;
            }
    fprintf(stderr, "oracle_dispatch: unknown pass '%s'\n", shaderFileName);
    return 1;
}

// deviation tables of the five transcendental instructions (oracle/hw_math.h); the memory stays owned by the caller (oracle/driver.py keeps it alive)
__attribute__((visibility("default"))) void oracle_set_hw_tables(const signed char* rcp, const signed char* sqrt, const signed char* rsq, const signed char* exp2, const signed char* log2) {
    hwmath::g_RcpDelta = rcp;
    hwmath::g_SqrtDelta = sqrt;
    hwmath::g_RsqDelta = rsq;
    hwmath::g_Exp2Delta = exp2;
    hwmath::g_Log2Delta = log2;
}
__attribute__((visibility("default"))) void oracle_set_hw_table_exp2neg(const signed char* exp2neg) { hwmath::g_Exp2NegDelta = exp2neg; }
// 1 = IEEE mode: the reference results instead of the device emulation (oracle/hw_math.h). Returns the previous mode.
__attribute__((visibility("default"))) int oracle_set_ieee_mode(int on) {
    const int prev = hwmath::g_IeeeMode;
    hwmath::g_IeeeMode = on ? 1 : 0;
    return prev;
}
// 0 = v_sqrt_f32, 1 = v_rsq_f32, 2 = v_rcp_f32, 3 = the contract's exp2, 4 = the contract's log2, 5 = Exp2NonPos, 6 = SatExp2, 7 = ExpNegAbs (round 5)
__attribute__((visibility("default"))) void oracle_eval_hw(int op, const float* in, float* out, int n) {
    for (int i = 0; i < n; i++)
        out[i] = op == 0 ? orc::HwSqrt(in[i]) : op == 1 ? orc::HwRsq(in[i]) : op == 2 ? orc::Rcp(in[i]) : op == 3 ? orc::exp2(in[i]) : op == 4 ? orc::log2(in[i]) : op == 5 ? orc::Exp2NonPos(in[i]) : op == 6 ? orc::SatExp2(in[i]) : orc::ExpNegAbs(in[i]);
}
__attribute__((visibility("default"))) float oracle_pow01(float x, float y) { return orc::Math::Pow01(x, y); }

__attribute__((visibility("default"))) int oracle_set_threads(int n) {
#ifdef _OPENMP
    if (n > 0)
        omp_set_num_threads(n);
    return omp_get_max_threads();
#else
    (void)n;
    return 1;
#endif
}

// The front-end / back-end functions the oracle restates for its own passes (ml.h [nrd]), row by row: tests/test_frontend_header.py holds the
// device results of include/NRD.hip.h against them. op 0: (N.xyz, roughness, materialID) -> the R10G10B10A2 texel word (as float bits);
// 1: word -> (N.xyz, roughness, materialID); 2: (hitDist, viewZ, roughness) -> normalised hit distance with the library-default hitDistParams;
// 3: (radiance.xyz, normHitDist) -> REBLUR packed; 4: packed -> REBLUR unpacked; 5: x -> SIGMA_BackEnd_UnpackShadow
__attribute__((visibility("default"))) void oracle_frontend(int op, const float* in, int inStride, float* out, int outStride, int n) {
    for (int i = 0; i < n; i++) {
        const float* a = in + (size_t)i * inStride;
        float* o = out + (size_t)i * outStride;
        if (op == 0) {
            float4 p = NRD_FrontEnd_PackNormalAndRoughness(float3(a[0], a[1], a[2]), a[3], a[4]);
            uint32_t w = ToUnorm(p.x, 1023.0f) | (ToUnorm(p.y, 1023.0f) << 10) | (ToUnorm(p.z, 1023.0f) << 20) | (ToUnorm(p.w, 3.0f) << 30);
            memcpy(o, &w, 4);
        } else if (op == 1) {
            uint32_t w;
            memcpy(&w, a, 4);
            float4 texel(float(w & 0x3FFu) / 1023.0f, float((w >> 10) & 0x3FFu) / 1023.0f, float((w >> 20) & 0x3FFu) / 1023.0f, float(w >> 30) / 3.0f);
            float4 r = NRD_FrontEnd_UnpackNormalAndRoughness(texel, o[4]);
            o[0] = r.x, o[1] = r.y, o[2] = r.z, o[3] = r.w;
        } else if (op == 2) {
            o[0] = REBLUR_FrontEnd_GetNormHitDist(a[0], a[1], float4(3.0f, 0.1f, 20.0f, -25.0f), a[2]);
        } else if (op == 3) {
            float4 r = REBLUR_FrontEnd_PackRadianceAndNormHitDist(float3(a[0], a[1], a[2]), a[3]);
            o[0] = r.x, o[1] = r.y, o[2] = r.z, o[3] = r.w;
        } else if (op == 4) {
            float4 r = REBLUR_BackEnd_UnpackRadianceAndNormHitDist(float4(a[0], a[1], a[2], a[3]));
            o[0] = r.x, o[1] = r.y, o[2] = r.z, o[3] = r.w;
        } else {
            o[0] = a[0] * a[0];
        }
    }
}

// scalar probes so the tests can pin the numerics contract (codecs + transcendentals) value by value
__attribute__((visibility("default"))) uint32_t oracle_f32tof16(float f) { return f32tof16(f); }
__attribute__((visibility("default"))) float oracle_f16tof32(uint32_t h) { return f16tof32(h); }
__attribute__((visibility("default"))) float oracle_exp2(float x) { return orc::exp2(x); }
__attribute__((visibility("default"))) float oracle_log2(float x) { return orc::log2(x); }
__attribute__((visibility("default"))) float oracle_atan(float x) { return orc::atan(x); }
__attribute__((visibility("default"))) float oracle_pow(float x, float y) { return orc::pow(x, y); }
}
