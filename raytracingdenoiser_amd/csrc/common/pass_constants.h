// Constant-buffer layouts of the NRD passes, as plain PODs shared by the host dispatch compiler (g++) and the HIP
// kernels (hipcc). Byte-for-byte the layouts the reference host fills for its shaders, so DispatchDesc::
// constantBufferData stays meaningful to an existing caller:
//   REBLUR  832 B  reference Shaders/Include/REBLUR_Config.hlsli:113-186 (REBLUR_SHARED_CONSTANTS)
//   SIGMA   516 B  reference Shaders/Include/SIGMA_Config.hlsli:45-80
//   RELAX   704 B  reference Shaders/Include/RELAX_Config.hlsli (+ 2 per-iteration words for a-trous)
//   REFERENCE      reference Shaders/Resources/REFERENCE_{TemporalAccumulation,Copy}.resources.hlsli:11-16
// Matrices are 16 floats, column-major: element (row i, col j) = m[j * 4 + i]; M * v = sum_j col_j * v[j].
#pragma once

#include <cstdint>

namespace nrdc {

struct F2 { float x, y; };
struct F4 { float x, y, z, w; };
struct U2 { uint32_t x, y; };
struct I2 { int32_t x, y; };

struct ReblurConstants {
    float gWorldToClip[16];
    float gViewToClip[16];
    float gViewToWorld[16];
    float gWorldToViewPrev[16];
    float gWorldToClipPrev[16];
    float gWorldPrevToWorld[16];
    F4 gRotatorPre;
    F4 gRotator;
    F4 gRotatorPost;
    F4 gFrustum;
    F4 gFrustumPrev;
    F4 gCameraDelta;
    F4 gHitDistParams;
    F4 gViewVectorWorld;
    F4 gViewVectorWorldPrev;
    F4 gMvScale;
    F2 gAntilagParams;
    F2 gResourceSize;
    F2 gResourceSizeInv;
    F2 gResourceSizeInvPrev;
    F2 gRectSize;
    F2 gRectSizeInv;
    F2 gRectSizePrev;
    F2 gResolutionScale;
    F2 gResolutionScalePrev;
    F2 gRectOffset;
    F2 gSpecProbabilityThresholdsForMvModification;
    F2 gJitter;
    U2 gPrintfAt;
    U2 gRectOrigin;
    I2 gRectSizeMinusOne;
    float gDisocclusionThreshold;
    float gDisocclusionThresholdAlternate;
    float gCameraAttachedReflectionMaterialID;
    float gStrandMaterialID;
    float gStrandThickness;
    float gStabilizationStrength;
    float gHitDistStabilizationStrength;
    float gDebug;
    float gOrthoMode;
    float gUnproject;
    float gDenoisingRange;
    float gPlaneDistSensitivity;
    float gFramerateScale;
    float gMinBlurRadius;
    float gMaxBlurRadius;
    float gDiffPrepassBlurRadius;
    float gSpecPrepassBlurRadius;
    float gMaxAccumulatedFrameNum;
    float gMaxFastAccumulatedFrameNum;
    float gAntiFirefly;
    float gLobeAngleFraction;
    float gRoughnessFraction;
    float gResponsiveAccumulationRoughnessThreshold;
    float gHistoryFixFrameNum;
    float gHistoryFixBasePixelStride;
    float gMinRectDimMulUnproject;
    float gUsePrepassNotOnlyForSpecularMotionEstimation;
    float gSplitScreen;
    float gSplitScreenPrev;
    float gCheckerboardResolveAccumSpeed;
    float gViewZScale;
    float gFireflySuppressorMinRelativeScale;
    float gMinHitDistanceWeight;
    float gDiffMinMaterial;
    float gSpecMinMaterial;
    uint32_t gHasHistoryConfidence;
    uint32_t gHasDisocclusionThresholdMix;
    uint32_t gDiffCheckerboard;
    uint32_t gSpecCheckerboard;
    uint32_t gFrameIndex;
    uint32_t gIsRectChanged;
    uint32_t gResetHistory;
};
static_assert(sizeof(ReblurConstants) == 832, "REBLUR shared constants must be 832 bytes");

// The validation pass appends two words (reference REBLUR_Validation.resources.hlsli); kept for size parity only.
struct ReblurValidationConstants {
    ReblurConstants shared;
    uint32_t gHasDiffuse;
    uint32_t gHasSpecular;
};

struct ReferenceAccumulateConstants {
    U2 gRectOrigin;
    float gAccumSpeed;
    float gDebug;
    float gViewZScale; // never written by the reference host (stays 0)
};
static_assert(sizeof(ReferenceAccumulateConstants) == 20, "REFERENCE accumulate constants");

struct ReferenceCopyConstants {
    F2 gRectSizeInv;
    float gSplitScreen;
    float gDebug;
    float gViewZScale;
};
static_assert(sizeof(ReferenceCopyConstants) == 20, "REFERENCE copy constants");

struct SigmaConstants {
    float gWorldToView[16];
    float gViewToClip[16];
    float gWorldToClipPrev[16];
    float gWorldToViewPrev[16];
    F4 gRotator;
    F4 gRotatorPost;
    F4 gViewVectorWorld;
    F4 gLightDirectionView;
    F4 gFrustum;
    F4 gFrustumPrev;
    F4 gCameraDelta;
    F4 gMvScale;
    F2 gResourceSizeInv;
    F2 gResourceSizeInvPrev;
    F2 gRectSize;
    F2 gRectSizeInv;
    F2 gRectSizePrev;
    F2 gResolutionScale;
    F2 gRectOffset;
    U2 gPrintfAt;
    U2 gRectOrigin;
    I2 gRectSizeMinusOne;
    I2 gTilesSizeMinusOne;
    float gOrthoMode;
    float gUnproject;
    float gDenoisingRange;
    float gPlaneDistSensitivity;
    float gStabilizationStrength;
    float gDebug;
    float gSplitScreen;
    float gViewZScale;
    float gMinRectDimMulUnproject;
    uint32_t gFrameIndex;
    uint32_t gIsRectChanged;
};
static_assert(sizeof(SigmaConstants) == 516, "SIGMA shared constants must be 516 bytes");

// RELAX shared block: reference Shaders/Include/RELAX_Config.hlsli:21-99 (field order), filled by Source/Relax.cpp:58-177
struct RelaxConstants {
    float gWorldToClip[16];
    float gWorldToClipPrev[16];
    float gWorldToViewPrev[16];
    float gWorldPrevToWorld[16];
    F4 gRotatorPre;
    F4 gFrustumRight;
    F4 gFrustumUp;
    F4 gFrustumForward;
    F4 gPrevFrustumRight;
    F4 gPrevFrustumUp;
    F4 gPrevFrustumForward;
    F4 gCameraDelta;
    F4 gMvScale;
    F2 gJitter;
    F2 gResolutionScale;
    F2 gRectOffset;
    F2 gResourceSizeInv;
    F2 gResourceSize;
    F2 gRectSizeInv;
    F2 gRectSizePrev;
    F2 gResourceSizeInvPrev;
    U2 gPrintfAt;
    U2 gRectOrigin;
    I2 gRectSize;
    float gSpecMaxAccumulatedFrameNum;
    float gSpecMaxFastAccumulatedFrameNum;
    float gDiffMaxAccumulatedFrameNum;
    float gDiffMaxFastAccumulatedFrameNum;
    float gDisocclusionThreshold;
    float gDisocclusionThresholdAlternate;
    float gCameraAttachedReflectionMaterialID;
    float gStrandMaterialID;
    float gStrandThickness;
    float gRoughnessFraction;
    float gSpecVarianceBoost;
    float gSplitScreen;
    float gDiffBlurRadius;
    float gSpecBlurRadius;
    float gDepthThreshold;
    float gLobeAngleFraction;
    float gSpecLobeAngleSlack;
    float gHistoryFixEdgeStoppingNormalPower;
    float gRoughnessEdgeStoppingRelaxation;
    float gNormalEdgeStoppingRelaxation;
    float gColorBoxSigmaScale;
    float gHistoryAccelerationAmount;
    float gHistoryResetTemporalSigmaScale;
    float gHistoryResetSpatialSigmaScale;
    float gHistoryResetAmount;
    float gDenoisingRange;
    float gSpecPhiLuminance;
    float gDiffPhiLuminance;
    float gDiffMaxLuminanceRelativeDifference;
    float gSpecMaxLuminanceRelativeDifference;
    float gLuminanceEdgeStoppingRelaxation;
    float gConfidenceDrivenRelaxationMultiplier;
    float gConfidenceDrivenLuminanceEdgeStoppingRelaxation;
    float gConfidenceDrivenNormalEdgeStoppingRelaxation;
    float gDebug;
    float gOrthoMode;
    float gUnproject;
    float gFramerateScale;
    float gCheckerboardResolveAccumSpeed;
    float gJitterDelta;
    float gHistoryFixFrameNum;
    float gHistoryFixBasePixelStride;
    float gHistoryThreshold;
    float gViewZScale;
    float gMinHitDistanceWeight;
    float gDiffMinMaterial;
    float gSpecMinMaterial;
    uint32_t gRoughnessEdgeStoppingEnabled;
    uint32_t gFrameIndex;
    uint32_t gDiffCheckerboard;
    uint32_t gSpecCheckerboard;
    uint32_t gHasHistoryConfidence;
    uint32_t gHasDisocclusionThresholdMix;
    uint32_t gResetHistory;
};
static_assert(sizeof(RelaxConstants) == 704, "RELAX shared constants must be 704 bytes");

// A-trous passes (both flavours) append the iteration parameters: reference Shaders/Resources/RELAX_Atrous.resources.hlsli:11-15
struct RelaxAtrousConstants {
    RelaxConstants shared;
    uint32_t gStepSize;
    uint32_t gIsLastPass;
};
static_assert(sizeof(RelaxAtrousConstants) == 712, "RELAX a-trous constants");

} // namespace nrdc
