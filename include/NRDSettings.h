// Per-frame and per-denoiser settings of the NRD C-ABI, layout-compatible with the reference
// Include/NRDSettings.h (v4.14). Field order and default values are the ABI/behaviour contract and are
// identical to the reference (file:line given per struct); the commentary is ours and deliberately short.
#pragma once

#define NRD_SETTINGS_VERSION_MAJOR 4
#define NRD_SETTINGS_VERSION_MINOR 14

static_assert(NRD_VERSION_MAJOR == NRD_SETTINGS_VERSION_MAJOR && NRD_VERSION_MINOR == NRD_SETTINGS_VERSION_MINOR, "NRD.h / NRDSettings.h version mismatch");

namespace nrd {

// frames = seconds * fps
inline uint32_t GetMaxAccumulatedFrameNum(float accumulationTime, float fps) { return (uint32_t)(accumulationTime * fps); }

// Which half of the pixels carries data when tracing at half rate (pixels with ((x ^ y ^ frame) & 1) == colour).
enum class CheckerboardMode : uint8_t { OFF, BLACK, WHITE, MAX_NUM };

// CONTINUE: normal; RESTART: drop history; CLEAR_AND_RESTART: additionally zero every pool plane first.
enum class AccumulationMode : uint8_t { CONTINUE, RESTART, CLEAR_AND_RESTART, MAX_NUM };

enum class HitDistanceReconstructionMode : uint8_t { OFF, AREA_3X3, AREA_5X5, MAX_NUM };

// reference NRDSettings.h:84-198. Matrices are column-major, column vectors, non-jittered.
struct CommonSettings {
    float viewToClipMatrix[16] = {};
    float viewToClipMatrixPrev[16] = {};
    float worldToViewMatrix[16] = {};
    float worldToViewMatrixPrev[16] = {};
    // rigid motion of the world between frames, if any
    float worldPrevToWorldMatrix[16] = {1.0f, 0.0f, 0.0f, 0.0f, 0.0f, 1.0f, 0.0f, 0.0f, 0.0f, 0.0f, 1.0f, 0.0f, 0.0f, 0.0f, 0.0f, 1.0f};

    // pixelUvPrev = pixelUv + mv.xy * scale.xy (2D/2.5D);  Xprev = X + mv * scale (world-space MVs)
    float motionVectorScale[3] = {1.0f, 1.0f, 0.0f};

    float cameraJitter[2] = {};     // [-0.5; 0.5]
    float cameraJitterPrev[2] = {};

    uint16_t resourceSize[2] = {};     // texture dimensions
    uint16_t resourceSizePrev[2] = {};
    uint16_t rectSize[2] = {};         // active viewport (== resourceSize unless dynamic resolution)
    uint16_t rectSizePrev[2] = {};

    float viewZScale = 1.0f;             // viewZ = IN_VIEWZ * viewZScale
    float timeDeltaBetweenFrames = 0.0f; // ms; 0 = measure with a wall-clock timer (non-deterministic!)
    float denoisingRange = 500000.0f;    // pixels with viewZ beyond this are "sky" and untouched
    float disocclusionThreshold = 0.01f;
    float disocclusionThresholdAlternate = 0.05f;
    float cameraAttachedReflectionMaterialID = 999.0f;
    float strandMaterialID = 999.0f;
    float strandThickness = 80e-6f;
    float splitScreen = 0.0f; // [0; 1]: left part shows the noisy input

    uint16_t printfAt[2] = {9999, 9999};
    float debug = 0.0f;

    uint32_t rectOrigin[2] = {};
    uint32_t frameIndex = 0;

    AccumulationMode accumulationMode = AccumulationMode::CONTINUE;
    bool isMotionVectorInWorldSpace = false;
    bool isHistoryConfidenceAvailable = false;
    bool isDisocclusionThresholdMixAvailable = false;
    bool isBaseColorMetalnessAvailable = false;
    bool enableValidation = false;
};

// ---------------------------------------------------------------- REBLUR (reference NRDSettings.h:200-330)
const uint32_t REBLUR_MAX_HISTORY_FRAME_NUM = 63;
const float REBLUR_DEFAULT_ACCUMULATION_TIME = 0.5f;

// normHitDist = saturate(hitDist / ((A + viewZ * B) * lerp(1, C, exp2(D * roughness^2))))
struct HitDistanceParameters {
    float A = 3.0f;
    float B = 0.1f;
    float C = 20.0f;
    float D = -25.0f;
};

struct ReblurAntilagSettings {
    float luminanceSigmaScale = 4.0f;
    float luminanceSensitivity = 3.0f;
};

struct ReblurSettings {
    HitDistanceParameters hitDistanceParameters = {};
    ReblurAntilagSettings antilagSettings = {};

    uint32_t maxAccumulatedFrameNum = 30;
    uint32_t maxFastAccumulatedFrameNum = 6;
    uint32_t maxStabilizedFrameNum = REBLUR_MAX_HISTORY_FRAME_NUM; // 0 disables the temporal-stabilization pass
    uint32_t maxStabilizedFrameNumForHitDistance = REBLUR_MAX_HISTORY_FRAME_NUM;
    uint32_t historyFixFrameNum = 3;
    uint32_t historyFixBasePixelStride = 14;

    float diffusePrepassBlurRadius = 30.0f;  // pixels, 0 disables
    float specularPrepassBlurRadius = 50.0f; // pixels, 0 disables
    float minHitDistanceWeight = 0.1f;
    float minBlurRadius = 1.0f;
    float maxBlurRadius = 30.0f;
    float lobeAngleFraction = 0.15f;
    float roughnessFraction = 0.15f;
    float responsiveAccumulationRoughnessThreshold = 0.0f;
    float planeDistanceSensitivity = 0.02f;
    float specularProbabilityThresholdsForMvModification[2] = {0.5f, 0.9f};
    float fireflySuppressorMinRelativeScale = 2.0f;

    CheckerboardMode checkerboardMode = CheckerboardMode::OFF;
    HitDistanceReconstructionMode hitDistanceReconstructionMode = HitDistanceReconstructionMode::OFF;
    bool enableAntiFirefly = false;
    bool enablePerformanceMode = false;

    float minMaterialForDiffuse = 4.0f;
    float minMaterialForSpecular = 4.0f;

    bool usePrepassOnlyForSpecularMotionEstimation = false;
};

// ---------------------------------------------------------------- RELAX (reference NRDSettings.h:332-430)
const uint32_t RELAX_MAX_HISTORY_FRAME_NUM = 255;
const float RELAX_DEFAULT_ACCUMULATION_TIME = 0.5f;

struct RelaxAntilagSettings {
    float accelerationAmount = 0.3f;
    float spatialSigmaScale = 4.5f;
    float temporalSigmaScale = 0.5f;
    float resetAmount = 0.5f;
};

struct RelaxSettings {
    RelaxAntilagSettings antilagSettings = {};

    uint32_t diffuseMaxAccumulatedFrameNum = 30;
    uint32_t specularMaxAccumulatedFrameNum = 30;
    uint32_t diffuseMaxFastAccumulatedFrameNum = 6;
    uint32_t specularMaxFastAccumulatedFrameNum = 6;
    uint32_t historyFixFrameNum = 3;
    uint32_t historyFixBasePixelStride = 14;
    float historyFixEdgeStoppingNormalPower = 8.0f;
    uint32_t spatialVarianceEstimationHistoryThreshold = 3;

    float diffusePrepassBlurRadius = 30.0f;
    float specularPrepassBlurRadius = 50.0f;
    float minHitDistanceWeight = 0.1f;
    float diffusePhiLuminance = 2.0f;
    float specularPhiLuminance = 1.0f;
    float lobeAngleFraction = 0.5f;
    float roughnessFraction = 0.15f;
    float specularVarianceBoost = 0.0f;
    float specularLobeAngleSlack = 0.15f;
    float historyClampingColorBoxSigmaScale = 2.0f;

    uint32_t atrousIterationNum = 5; // [2; 8]

    float diffuseMinLuminanceWeight = 0.0f;
    float specularMinLuminanceWeight = 0.0f;
    float depthThreshold = 0.003f;

    float confidenceDrivenRelaxationMultiplier = 0.0f;
    float confidenceDrivenLuminanceEdgeStoppingRelaxation = 0.0f;
    float confidenceDrivenNormalEdgeStoppingRelaxation = 0.0f;

    float luminanceEdgeStoppingRelaxation = 0.5f;
    float normalEdgeStoppingRelaxation = 0.3f;
    float roughnessEdgeStoppingRelaxation = 1.0f;

    CheckerboardMode checkerboardMode = CheckerboardMode::OFF;
    HitDistanceReconstructionMode hitDistanceReconstructionMode = HitDistanceReconstructionMode::OFF;
    bool enableAntiFirefly = false;
    bool enableRoughnessEdgeStopping = true;

    float minMaterialForDiffuse = 4.0f;
    float minMaterialForSpecular = 4.0f;
};

// ---------------------------------------------------------------- SIGMA (reference NRDSettings.h:432-450)
const uint32_t SIGMA_MAX_HISTORY_FRAME_NUM = 7;
const float SIGMA_DEFAULT_ACCUMULATION_TIME = 0.084f;

struct SigmaSettings {
    float lightDirection[3] = {0.0f, 0.0f, 0.0f}; // direction TO the light; zero for non-directional sources
    float planeDistanceSensitivity = 0.02f;
    uint32_t maxStabilizedFrameNum = 5; // 0 disables temporal stabilization
};

// ---------------------------------------------------------------- REFERENCE (reference NRDSettings.h:452-461)
const uint32_t REFERENCE_MAX_HISTORY_FRAME_NUM = 4095;
const float REFERENCE_DEFAULT_ACCUMULATION_TIME = 17.0f;

struct ReferenceSettings {
    uint32_t maxAccumulatedFrameNum = 1020;
};

static_assert(sizeof(CommonSettings) == 428, "CommonSettings ABI");
static_assert(sizeof(ReblurSettings) == 112, "ReblurSettings ABI");
static_assert(sizeof(RelaxSettings) == 140, "RelaxSettings ABI");
static_assert(sizeof(SigmaSettings) == 20, "SigmaSettings ABI");
static_assert(sizeof(ReferenceSettings) == 4, "ReferenceSettings ABI");

} // namespace nrd
