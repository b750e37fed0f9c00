// Checks include/NRD.hip.h: every function runs on the HOST over a deterministic input set and prints / verifies known answers;
// with a GPU the same functions run on the DEVICE over the same inputs and must agree (bit-exact for the packers and codecs, which
// use only + - * / sqrt; 2e-4 relative for the resolves, which use exp / log / pow from the platform's math library).
// usage: frontend_check [--no-gpu]
//        frontend_check --dump FILE COUNT   evaluates COUNT samples ON THE DEVICE and writes inputs and results as raw float32 / uint32 rows
//                                           (tests/test_frontend_header.py holds them against models that do not include this header)
#include "NRD.hip.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

struct Sample {
    float3 N, V, radiance, direction, albedo, Rf0;
    float roughness, materialID, hitDist, viewZ;
};
struct Result {
    uint32_t normalRoughnessWord, normalRoughnessWordHi; // the IN_NORMAL_ROUGHNESS texel (the high word only with the 64-bit encodings)
    float4 unpackedNR, reblurPacked, reblurUnpacked, sh0, sh1, relaxPacked, relaxSh1, dirOcc, translucency;
    float normHitDist, penumbra, penumbraLocal, shadow, materialID;
    float3 diffFactor, specFactor, sgDiffuse, sgSpecular, shDiffuse, shSpecular, sgColor, sgDir;
    float2 rejitter;
    float4 misc; // REBLUR_GetHitDist, NRD_GetNormalizedStrandThickness, _NRD_SG_Integral, NRD_IsValidRadiance (NRD.hlsli:575-580, 1136-1162)
};

__host__ __device__ inline uint32_t Hash(uint32_t v) {
    uint32_t s = v * 747796405u + 2891336453u;
    uint32_t w = ((s >> ((s >> 28u) + 4u)) ^ s) * 277803737u;
    return (w >> 22u) ^ w;
}
__host__ __device__ inline float U(uint32_t i, uint32_t k) { return float(Hash(i * 64u + k) >> 8) * (1.0f / 16777216.0f); }
__host__ __device__ inline float3 Dir(uint32_t i, uint32_t k) {
    float3 v = make_float3(U(i, k) * 2.0f - 1.0f, U(i, k + 1) * 2.0f - 1.0f, U(i, k + 2) * 2.0f - 1.0f);
    float l = sqrtf(v.x * v.x + v.y * v.y + v.z * v.z);
    if (l < 0.05f)
        return make_float3(0.0f, 0.0f, 1.0f);
    return make_float3(v.x / l, v.y / l, v.z / l);
}

__host__ __device__ inline Sample MakeSample(uint32_t i) {
    Sample s;
    s.N = Dir(i, 0);
    s.V = Dir(i, 3);
    if (s.N.x * s.V.x + s.N.y * s.V.y + s.N.z * s.V.z < 0.0f)
        s.V = make_float3(-s.V.x, -s.V.y, -s.V.z);
    s.direction = Dir(i, 6);
    s.radiance = make_float3(U(i, 9) * 4.0f, U(i, 10) * 3.0f, U(i, 11) * 5.0f);
    s.albedo = make_float3(U(i, 12), U(i, 13), U(i, 14));
    s.Rf0 = make_float3(0.04f + 0.9f * U(i, 15), 0.04f + 0.9f * U(i, 16), 0.04f + 0.9f * U(i, 17));
    s.roughness = U(i, 18);
    s.materialID = float(Hash(i + 77u) & 3u);
    s.hitDist = U(i, 19) * 30.0f;
    s.viewZ = 0.5f + U(i, 20) * 100.0f;
    return s;
}

__host__ __device__ inline Result Evaluate(uint32_t i) {
    const Sample s = MakeSample(i);
    const float4 hitDistParams = make_float4(3.0f, 0.1f, 20.0f, -25.0f);
    Result r;
    float4 nr = NRD_FrontEnd_PackNormalAndRoughness(s.N, s.roughness, s.materialID);
    const NRD_NormalRoughnessTexel texel = NRD_StoreNormalRoughnessTexel(nr); // the texel of the library's encoding (32 or 64 bits)
    r.normalRoughnessWord = (uint32_t)texel;
    r.normalRoughnessWordHi = (uint32_t)((uint64_t)texel >> 32);
    r.unpackedNR = NRD_FrontEnd_UnpackNormalAndRoughness(NRD_LoadNormalRoughnessTexel(texel), r.materialID);
    r.normHitDist = REBLUR_FrontEnd_GetNormHitDist(s.hitDist, s.viewZ, hitDistParams, s.roughness);
    r.reblurPacked = REBLUR_FrontEnd_PackRadianceAndNormHitDist(s.radiance, r.normHitDist);
    r.reblurUnpacked = REBLUR_BackEnd_UnpackRadianceAndNormHitDist(r.reblurPacked);
    r.sh0 = REBLUR_FrontEnd_PackSh(s.radiance, r.normHitDist, s.direction, r.sh1);
    r.relaxPacked = RELAX_FrontEnd_PackSh(s.radiance, s.hitDist, s.direction, r.relaxSh1);
    r.dirOcc = REBLUR_FrontEnd_PackDirectionalOcclusion(s.direction, r.normHitDist);
    r.penumbra = SIGMA_FrontEnd_PackPenumbra(i % 5 == 0 ? NRD_FP16_MAX : s.hitDist, 0.02f);
    r.penumbraLocal = SIGMA_FrontEnd_PackPenumbra(s.hitDist, s.hitDist + 10.0f, 0.5f);
    r.translucency = SIGMA_FrontEnd_PackTranslucency(i % 5 == 0 ? NRD_FP16_MAX : s.hitDist, s.albedo);
    r.shadow = SIGMA_BackEnd_UnpackShadow(s.roughness);
    NRD_MaterialFactors(s.N, s.V, s.albedo, s.Rf0, s.roughness, r.diffFactor, r.specFactor);
    NRD_SG sg = REBLUR_BackEnd_UnpackSh(r.sh0, r.sh1);
    r.sgColor = NRD_SG_ExtractColor(sg);
    r.sgDir = NRD_SG_ExtractDirection(sg);
    r.sgDiffuse = NRD_SG_ResolveDiffuse(sg, s.N);
    r.sgSpecular = NRD_SG_ResolveSpecular(sg, s.N, s.V, s.roughness);
    r.shDiffuse = NRD_SH_ResolveDiffuse(sg, s.N);
    r.shSpecular = NRD_SH_ResolveSpecular(sg, s.N, s.V, s.roughness);
    r.rejitter = NRD_SG_ReJitter(sg, sg, s.Rf0, s.V, s.roughness, s.viewZ, s.viewZ * 1.001f, s.viewZ * 0.999f, s.viewZ, s.viewZ, s.N, s.N, Dir(i, 21), s.N, s.N);
    NRD_SG wide = sg;
    wide.sharpness = 0.5f + 4.0f * s.roughness; // (the packers leave 0: the denoiser fills it)
    const float poison = i % 5 == 0 ? 0.0f : 1.0f;
    r.misc = make_float4(REBLUR_GetHitDist(r.normHitDist, s.viewZ, hitDistParams, s.roughness), NRD_GetNormalizedStrandThickness(s.hitDist * 0.01f, s.viewZ * 0.001f), _NRD_SG_Integral(wide),
                         NRD_IsValidRadiance(make_float3(s.radiance.x / poison, s.radiance.y / poison, s.radiance.z / poison)) ? 1.0f : 0.0f);
    return r;
}

__global__ void EvaluateKernel(Result* out, uint32_t count) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count)
        out[i] = Evaluate(i);
}

#define CHECK(x)                                            \
    do {                                                    \
        if (!(x)) {                                         \
            printf("FAILED: %s (line %d)\n", #x, __LINE__); \
            return 1;                                       \
        }                                                   \
    } while (0)

static bool Close(float a, float b, float rel) { return fabsf(a - b) <= rel * fmaxf(fmaxf(fabsf(a), fabsf(b)), 1e-3f); }

// one row of 32-bit words per sample: the inputs, then every result (field order = DUMP_FIELDS in tests/test_frontend_header.py)
static void Append(std::vector<uint32_t>& row, float v) {
    uint32_t u;
    memcpy(&u, &v, 4);
    row.push_back(u);
}
static void Append(std::vector<uint32_t>& row, float2 v) { Append(row, v.x), Append(row, v.y); }
static void Append(std::vector<uint32_t>& row, float3 v) { Append(row, v.x), Append(row, v.y), Append(row, v.z); }
static void Append(std::vector<uint32_t>& row, float4 v) { Append(row, v.x), Append(row, v.y), Append(row, v.z), Append(row, v.w); }

static int Dump(const char* path, uint32_t count, bool onHost) {
    std::vector<Result> dev(count);
    if (onHost) { // (debugging the comparison without a GPU; the test dumps the DEVICE results)
        for (uint32_t i = 0; i < count; i++)
            dev[i] = Evaluate(i);
    } else {
        Result* dOut = nullptr;
        CHECK(hipMalloc((void**)&dOut, sizeof(Result) * (size_t)count) == hipSuccess);
        hipLaunchKernelGGL(EvaluateKernel, dim3((count + 255) / 256), dim3(256), 0, 0, dOut, count);
        CHECK(hipMemcpy(dev.data(), dOut, sizeof(Result) * (size_t)count, hipMemcpyDeviceToHost) == hipSuccess);
        (void)hipFree(dOut);
    }
    FILE* fp = fopen(path, "wb");
    CHECK(fp != nullptr);
    std::vector<uint32_t> row;
    for (uint32_t i = 0; i < count; i++) {
        const Sample s = MakeSample(i);
        const Result& r = dev[i];
        row.clear();
        Append(row, s.N), Append(row, s.V), Append(row, s.radiance), Append(row, s.direction), Append(row, s.albedo), Append(row, s.Rf0);
        Append(row, s.roughness), Append(row, s.materialID), Append(row, s.hitDist), Append(row, s.viewZ), Append(row, Dir(i, 21));
        row.push_back(r.normalRoughnessWord);
        Append(row, r.unpackedNR), Append(row, r.reblurPacked), Append(row, r.reblurUnpacked), Append(row, r.sh0), Append(row, r.sh1), Append(row, r.relaxPacked), Append(row, r.relaxSh1);
        Append(row, r.dirOcc), Append(row, r.translucency), Append(row, r.normHitDist), Append(row, r.penumbra), Append(row, r.penumbraLocal), Append(row, r.shadow), Append(row, r.materialID);
        Append(row, r.diffFactor), Append(row, r.specFactor), Append(row, r.sgDiffuse), Append(row, r.sgSpecular), Append(row, r.shDiffuse), Append(row, r.shSpecular), Append(row, r.sgColor);
        Append(row, r.sgDir), Append(row, r.rejitter), Append(row, r.misc);
        fwrite(row.data(), 4, row.size(), fp);
    }
    fclose(fp);
    printf("dumped %u samples x %zu words\n", count, row.size());
    return 0;
}

int main(int argc, char** argv) {
    if (argc > 3 && (!strcmp(argv[1], "--dump") || !strcmp(argv[1], "--dump-host")))
        return Dump(argv[2], (uint32_t)atoi(argv[3]), !strcmp(argv[1], "--dump-host"));
    const bool noGpu = argc > 1 && !strcmp(argv[1], "--no-gpu");
    const uint32_t count = 4096;
    std::vector<Result> host(count);
    for (uint32_t i = 0; i < count; i++)
        host[i] = Evaluate(i);

    // ---- known answers on the host
    for (uint32_t i = 0; i < count; i++) {
        const Sample s = MakeSample(i);
        const Result& r = host[i];
        // the packed normal decodes to within the quantisation of the encoding (10-bit oct by default), the roughness to half a code of its channel (in the ENCODED domain:
        // linear / squared / square root), the material id exactly where the encoding has one (R10G10B10A2) and as 0 elsewhere
        const float codes = NRD_NORMAL_ENCODING == 2 ? 1023.0f : NRD_NORMAL_ENCODING == 0 ? 255.0f : NRD_NORMAL_ENCODING == 1 ? 127.0f : NRD_NORMAL_ENCODING == 3 ? 65535.0f : 32767.0f;
        auto enc = [](float x) { return NRD_ROUGHNESS_ENCODING == 0 ? x * x : NRD_ROUGHNESS_ENCODING == 2 ? sqrtf(x) : x; };
        CHECK(r.materialID == (NRD_NORMAL_ENCODING == 2 ? s.materialID : 0.0f));
        CHECK(fabsf(enc(r.unpackedNR.w) - enc(s.roughness)) <= 0.5f / codes + 2e-6f);
        float d = r.unpackedNR.x * s.N.x + r.unpackedNR.y * s.N.y + r.unpackedNR.z * s.N.z;
        CHECK(d > (NRD_NORMAL_ENCODING == 2 ? 0.99999f : NRD_NORMAL_ENCODING < 2 ? 0.9999f : 0.999999f)); // < ~0.26 degrees for the default
        // YCoCg round trip and the SG colour
        CHECK(fabsf(r.reblurUnpacked.x - s.radiance.x) < 1e-5f && fabsf(r.reblurUnpacked.y - s.radiance.y) < 1e-5f && fabsf(r.reblurUnpacked.z - s.radiance.z) < 1e-5f);
        CHECK(fabsf(r.sgColor.x - s.radiance.x) < 1e-5f && fabsf(r.sgColor.z - s.radiance.z) < 1e-5f);
        CHECK(r.normHitDist >= 0.0f && r.normHitDist <= 1.0f && r.reblurPacked.w == r.normHitDist);
        // SH1 carries direction * luma (REBLUR: Y of YCoCg, RELAX: Rec.709 luminance)
        CHECK(Close(r.sh1.x, s.direction.x * r.sh0.x, 1e-6f) && r.sh1.w == 0.0f && r.relaxSh1.w == 0.0f && r.relaxPacked.w == s.hitDist);
        CHECK(Close(r.dirOcc.w, r.normHitDist, 1e-6f));
        CHECK((i % 5 == 0) ? (r.penumbra == NRD_FP16_MAX && r.translucency.x == 1.0f) : (Close(r.penumbra, s.hitDist * 0.02f * 0.5f, 1e-6f) && r.translucency.x == 0.0f));
        CHECK(r.shadow == s.roughness * s.roughness);
        CHECK(r.diffFactor.x >= NRD_MATERIAL_FACTOR_MIN_SCALE && r.diffFactor.x <= 1.0f && r.specFactor.y >= NRD_MATERIAL_FACTOR_MIN_SCALE && r.specFactor.y <= 1.0f);
        // resolves are finite, non-negative; the SH diffuse resolve along the light direction is 1.5x the SH0 luma (dot = c0, + 0.5 c0)
        CHECK(r.sgDiffuse.x >= 0.0f && r.sgSpecular.y >= 0.0f && r.shSpecular.z >= 0.0f && !isnan(r.sgSpecular.x) && !isinf(r.sgSpecular.x));
        CHECK(r.rejitter.x >= 1.0f / NRD_PI - 1e-6f && r.rejitter.x <= NRD_PI + 1e-6f);
    }
    {
        NRD_SG sg = _NRD_SG_Create(make_float3(1.0f, 1.0f, 1.0f), make_float3(0.0f, 0.0f, 1.0f), 0.5f);
        float3 c = NRD_SH_ResolveDiffuse(sg, make_float3(0.0f, 0.0f, 1.0f));
        CHECK(Close(c.x, 1.5f, 1e-5f) && Close(c.y, 1.5f, 1e-5f));
        float acc = NRD_FrontEnd_SpecHitDistAveraging_Begin();
        NRD_FrontEnd_SpecHitDistAveraging_Add(acc, 0.0f);
        NRD_FrontEnd_SpecHitDistAveraging_Add(acc, 3.0f);
        NRD_FrontEnd_SpecHitDistAveraging_Add(acc, 2.0f);
        NRD_FrontEnd_SpecHitDistAveraging_End(acc);
        CHECK(acc == 2.0f && NRD_FrontEnd_TrimHitDistance(0.01f, 0.1f) == 0.0f);
        // MISC (NRD.hlsli:1136-1162)
        const float4 hitDistParams = make_float4(3.0f, 0.1f, 20.0f, -25.0f);
        for (float hitDist : {0.0f, 0.37f, 2.5f})  // REBLUR_GetHitDist inverts REBLUR_FrontEnd_GetNormHitDist below the saturation point
            CHECK(Close(REBLUR_GetHitDist(REBLUR_FrontEnd_GetNormHitDist(hitDist, 12.0f, hitDistParams, 0.4f), 12.0f, hitDistParams, 0.4f), hitDist, 1e-5f));
        CHECK(REBLUR_GetHitDist(1.0f, -10.0f, hitDistParams, 0.0f) == (3.0f + 10.0f * 0.1f) * 20.0f);  // roughness 0: exp2( 0 ) = 1 -> the full z scale
        CHECK(NRD_IsValidRadiance(make_float3(1.0f, 0.0f, 65504.0f)) && !NRD_IsValidRadiance(make_float3(1.0f, NAN, 0.0f)) && !NRD_IsValidRadiance(make_float3(INFINITY, 0.0f, 0.0f)));
        CHECK(_NRD_IsInvalid(-INFINITY) && !_NRD_IsInvalid(0.0f));
        CHECK(NRD_GetNormalizedStrandThickness(0.0f, 0.01f) == 1.0f && NRD_GetNormalizedStrandThickness(0.03f, 0.01f) == 0.25f);
        NRD_SG wide = sg;
        wide.sharpness = 1.5f;  // (_NRD_SG_Create leaves 0: the denoiser fills it) a wide lobe, where the exponential term counts
        CHECK(Close(_NRD_SG_Integral(wide), _NRD_SG_IntegralApprox(wide) * (1.0f - expf(-2.0f * wide.sharpness)), 1e-6f) && _NRD_SG_Integral(wide) < _NRD_SG_IntegralApprox(wide));
    }
    uint32_t checksum = 0;
    for (uint32_t i = 0; i < count; i++)
        checksum = checksum * 31u + host[i].normalRoughnessWord, checksum = sizeof(NRD_NormalRoughnessTexel) == 8 ? checksum * 31u + host[i].normalRoughnessWordHi : checksum;
    printf("host OK: %u samples, normal/roughness word checksum %08x\n", count, checksum);
    if (noGpu)
        return 0;

    // ---- device vs host
    Result* dOut = nullptr;
    CHECK(hipMalloc((void**)&dOut, sizeof(Result) * count) == hipSuccess);
    hipLaunchKernelGGL(EvaluateKernel, dim3((count + 255) / 256), dim3(256), 0, 0, dOut, count);
    std::vector<Result> dev(count);
    CHECK(hipMemcpy(dev.data(), dOut, sizeof(Result) * count, hipMemcpyDeviceToHost) == hipSuccess);
    (void)hipFree(dOut);
    uint32_t exactMismatch = 0, looseMismatch = 0;
    for (uint32_t i = 0; i < count; i++) {
        const Result &h = host[i], &d = dev[i];
        // packers and codecs: bit-exact
        exactMismatch += h.normalRoughnessWord != d.normalRoughnessWord || h.normalRoughnessWordHi != d.normalRoughnessWordHi;
        exactMismatch += memcmp(&h.unpackedNR, &d.unpackedNR, 16) != 0;
        exactMismatch += memcmp(&h.reblurUnpacked, &d.reblurUnpacked, 12) != 0;
        exactMismatch += memcmp(&h.relaxPacked, &d.relaxPacked, 16) != 0 || memcmp(&h.relaxSh1, &d.relaxSh1, 16) != 0;
        exactMismatch += memcmp(&h.translucency, &d.translucency, 16) != 0 || h.penumbra != d.penumbra || h.penumbraLocal != d.penumbraLocal;
        // anything behind exp2 / exp / log / pow: platform math library, tolerance
        const float* hf = (const float*)&h;
        const float* df = (const float*)&d;
        for (size_t k = 1; k < sizeof(Result) / 4; k++) {
            // 2e-4 relative (1e-5 absolute below 0.05): exp / log / pow differ by a few ulp between the two math libraries, and the
            // resolves amplify that through differences of exponentials
            if (fabsf(hf[k] - df[k]) > 2e-4f * fmaxf(fmaxf(fabsf(hf[k]), fabsf(df[k])), 0.05f)) {
                if (looseMismatch < 8)
                    printf("  sample %u float #%zu: host %.9g device %.9g\n", i, k, hf[k], df[k]);
                looseMismatch++;
            }
        }
    }
    printf("device vs host: %u bit mismatches in packers/codecs, %u values beyond 2e-4\n", exactMismatch, looseMismatch);
    CHECK(exactMismatch == 0 && looseMismatch == 0);
    printf("frontend check OK\n");
    return 0;
}
