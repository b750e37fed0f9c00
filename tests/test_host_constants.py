"""The per-frame constant blocks of the host (SURVEY.md section 8a, row "Host per-frame") against an INDEPENDENT restatement.

The parity harness feeds the oracle and the HIP path the product's own constant blocks, so it cannot see a wrong constant. Here the blocks that
nrd::GetComputeDispatches emits (832 B REBLUR, 704 B RELAX, SIGMA) are parsed with the field order of the reference's *_SHARED_CONSTANTS macros
(Shaders/Include/REBLUR_Config.hlsli, RELAX_Config.hlsli, SIGMA_Config.hlsli) and every field is compared with a float64 numpy model written from the
reference host code alone -- SetCommonSettings (Source/InstanceImpl.cpp:269-473), AddSharedConstants_Reblur (Source/Reblur.cpp:297-406), _Relax
(Source/Relax.cpp:51-180), _Sigma (Source/Sigma.cpp:92-145) -- for a left-handed camera, a right-handed one (converted to LH by the host) and a
jittered one, on a restart frame and on a continuing frame. The documented quirks are part of the model: gLobeAngleFraction squared (REBLUR),
gMinHitDistanceWeight doubled, gHistoryFixFrameNum + 1 and gLuminanceEdgeStoppingRelaxation <- roughnessEdgeStoppingRelaxation (RELAX).
What cannot be restated from the reference (NVIDIA-RTX/MathLib is absent): the Weyl / Bayer sequences behind the rotators -- only their structure is
checked (unit rotators (cos, sin, -sin, cos), pre-pass angle in [0, 90) degrees) -- and DecomposeProjection, whose outputs are derived from the
requirement that ReconstructViewPosition inverts GetScreenUv.
"""
import math
import struct

import numpy as np
import pytest

import parity
from raytracingdenoiser_amd import api

F4X4, F4, F2, U2, I2, F, U = "float4x4", "float4", "float2", "uint2", "int2", "float", "uint"
SIZES = {F4X4: 64, F4: 16, F2: 8, U2: 8, I2: 8, F: 4, U: 4}


def _layout(text):
    out = []
    for item in text.split():
        kind, name = item.split(":")
        out.append((name, kind))
    return out


REBLUR_LAYOUT = _layout("""float4x4:gWorldToClip float4x4:gViewToClip float4x4:gViewToWorld float4x4:gWorldToViewPrev float4x4:gWorldToClipPrev float4x4:gWorldPrevToWorld
float4:gRotatorPre float4:gRotator float4:gRotatorPost float4:gFrustum float4:gFrustumPrev float4:gCameraDelta float4:gHitDistParams float4:gViewVectorWorld float4:gViewVectorWorldPrev
float4:gMvScale float2:gAntilagParams float2:gResourceSize float2:gResourceSizeInv float2:gResourceSizeInvPrev float2:gRectSize float2:gRectSizeInv float2:gRectSizePrev
float2:gResolutionScale float2:gResolutionScalePrev float2:gRectOffset float2:gSpecProbabilityThresholdsForMvModification float2:gJitter uint2:gPrintfAt uint2:gRectOrigin
int2:gRectSizeMinusOne float:gDisocclusionThreshold float:gDisocclusionThresholdAlternate float:gCameraAttachedReflectionMaterialID float:gStrandMaterialID float:gStrandThickness
float:gStabilizationStrength float:gHitDistStabilizationStrength float:gDebug float:gOrthoMode float:gUnproject float:gDenoisingRange float:gPlaneDistSensitivity float:gFramerateScale
float:gMinBlurRadius float:gMaxBlurRadius float:gDiffPrepassBlurRadius float:gSpecPrepassBlurRadius float:gMaxAccumulatedFrameNum float:gMaxFastAccumulatedFrameNum float:gAntiFirefly
float:gLobeAngleFraction float:gRoughnessFraction float:gResponsiveAccumulationRoughnessThreshold float:gHistoryFixFrameNum float:gHistoryFixBasePixelStride float:gMinRectDimMulUnproject
float:gUsePrepassNotOnlyForSpecularMotionEstimation float:gSplitScreen float:gSplitScreenPrev float:gCheckerboardResolveAccumSpeed float:gViewZScale float:gFireflySuppressorMinRelativeScale
float:gMinHitDistanceWeight float:gDiffMinMaterial float:gSpecMinMaterial uint:gHasHistoryConfidence uint:gHasDisocclusionThresholdMix uint:gDiffCheckerboard uint:gSpecCheckerboard
uint:gFrameIndex uint:gIsRectChanged uint:gResetHistory""")

RELAX_LAYOUT = _layout("""float4x4:gWorldToClip float4x4:gWorldToClipPrev float4x4:gWorldToViewPrev float4x4:gWorldPrevToWorld float4:gRotatorPre float4:gFrustumRight float4:gFrustumUp
float4:gFrustumForward float4:gPrevFrustumRight float4:gPrevFrustumUp float4:gPrevFrustumForward float4:gCameraDelta float4:gMvScale float2:gJitter float2:gResolutionScale float2:gRectOffset
float2:gResourceSizeInv float2:gResourceSize float2:gRectSizeInv float2:gRectSizePrev float2:gResourceSizeInvPrev uint2:gPrintfAt uint2:gRectOrigin int2:gRectSize
float:gSpecMaxAccumulatedFrameNum float:gSpecMaxFastAccumulatedFrameNum float:gDiffMaxAccumulatedFrameNum float:gDiffMaxFastAccumulatedFrameNum float:gDisocclusionThreshold
float:gDisocclusionThresholdAlternate float:gCameraAttachedReflectionMaterialID float:gStrandMaterialID float:gStrandThickness float:gRoughnessFraction float:gSpecVarianceBoost
float:gSplitScreen float:gDiffBlurRadius float:gSpecBlurRadius float:gDepthThreshold float:gLobeAngleFraction float:gSpecLobeAngleSlack float:gHistoryFixEdgeStoppingNormalPower
float:gRoughnessEdgeStoppingRelaxation float:gNormalEdgeStoppingRelaxation float:gColorBoxSigmaScale float:gHistoryAccelerationAmount float:gHistoryResetTemporalSigmaScale
float:gHistoryResetSpatialSigmaScale float:gHistoryResetAmount float:gDenoisingRange float:gSpecPhiLuminance float:gDiffPhiLuminance float:gDiffMaxLuminanceRelativeDifference
float:gSpecMaxLuminanceRelativeDifference float:gLuminanceEdgeStoppingRelaxation float:gConfidenceDrivenRelaxationMultiplier float:gConfidenceDrivenLuminanceEdgeStoppingRelaxation
float:gConfidenceDrivenNormalEdgeStoppingRelaxation float:gDebug float:gOrthoMode float:gUnproject float:gFramerateScale float:gCheckerboardResolveAccumSpeed float:gJitterDelta
float:gHistoryFixFrameNum float:gHistoryFixBasePixelStride float:gHistoryThreshold float:gViewZScale float:gMinHitDistanceWeight float:gDiffMinMaterial float:gSpecMinMaterial
uint:gRoughnessEdgeStoppingEnabled uint:gFrameIndex uint:gDiffCheckerboard uint:gSpecCheckerboard uint:gHasHistoryConfidence uint:gHasDisocclusionThresholdMix uint:gResetHistory""")

SIGMA_LAYOUT = _layout("""float4x4:gWorldToView float4x4:gViewToClip float4x4:gWorldToClipPrev float4x4:gWorldToViewPrev float4:gRotator float4:gRotatorPost float4:gViewVectorWorld
float4:gLightDirectionView float4:gFrustum float4:gFrustumPrev float4:gCameraDelta float4:gMvScale float2:gResourceSizeInv float2:gResourceSizeInvPrev float2:gRectSize float2:gRectSizeInv
float2:gRectSizePrev float2:gResolutionScale float2:gRectOffset uint2:gPrintfAt uint2:gRectOrigin int2:gRectSizeMinusOne int2:gTilesSizeMinusOne float:gOrthoMode float:gUnproject
float:gDenoisingRange float:gPlaneDistSensitivity float:gStabilizationStrength float:gDebug float:gSplitScreen float:gViewZScale float:gMinRectDimMulUnproject uint:gFrameIndex uint:gIsRectChanged""")


def parse_block(blob, layout):
    """bytes -> {name: numpy array} following HLSL constant-buffer packing as the reference structs are laid out (tightly, in declaration order: the
    macros order the members by decreasing size so no padding arises)"""
    out, off = {}, 0
    for name, kind in layout:
        n = SIZES[kind]
        chunk = blob[off:off + n]
        if kind in (U2, U):
            out[name] = np.frombuffer(chunk, dtype=np.uint32).astype(np.float64)
        elif kind == I2:
            out[name] = np.frombuffer(chunk, dtype=np.int32).astype(np.float64)
        else:
            out[name] = np.frombuffer(chunk, dtype=np.float32).astype(np.float64)
        off += n
    return out, off


# ------------------------------------------------------------------------------------------------------------------ the model
def _col_major(m16):
    """16 floats, column-major (NRDSettings.h: matrices are column-major, vectors are columns) -> numpy 4x4 with M @ v"""
    return np.array(m16, dtype=np.float64).reshape(4, 4).T


def _flat(M):
    return M.T.reshape(-1)


class HostModel:
    """float64 restatement of nrd::InstanceImpl::SetCommonSettings for perspective projections"""

    def __init__(self):
        self.first = True
        self.split_prev = 0.0
        self.world_to_view = np.eye(4)
        self.view_to_clip = np.eye(4)

    def set_common(self, cs):
        self.split_prev_out = self.split_screen if hasattr(self, "split_screen") else 0.0
        self.split_screen = float(cs.splitScreen)
        self.cs = cs
        mode = int(cs.accumulationMode)
        if self.first:
            mode = int(api.AccumulationMode.CLEAR_AND_RESTART)
            self.first = False
        self.mode = mode
        V2C, V2Cp = _col_major(list(cs.viewToClipMatrix)), _col_major(list(cs.viewToClipMatrixPrev))
        W2V, W2Vp = _col_major(list(cs.worldToViewMatrix)), _col_major(list(cs.worldToViewMatrixPrev))
        self.rect_prev, self.res_prev = tuple(cs.rectSizePrev), tuple(cs.resourceSizePrev)
        self.jitter_prev = tuple(cs.cameraJitterPrev)
        if mode != int(api.AccumulationMode.CONTINUE):
            # InstanceImpl.cpp:282-297: sizes and jitter of the "previous" frame are those of this one. (The member matrices assigned there are overwritten
            # from the settings a few lines later, :359-381: the shaders see the caller's Prev matrices also on a restart frame.)
            self.split_prev_out = 0.0
            self.rect_prev, self.res_prev = tuple(cs.rectSize), tuple(cs.resourceSize)
            self.jitter_prev = tuple(cs.cameraJitter)
        # handedness: a D3D-style LH projection has clip.w = +z, an RH one clip.w = -z
        if V2C[3, 2] < 0.0:
            V2C[:, 2] *= -1.0
            V2Cp[:, 2] *= -1.0
            W2V[2, :] *= -1.0  # Transpose, negate column 2, Transpose == negate row 2
            W2Vp[2, :] *= -1.0
        V2W, V2Wp = np.linalg.inv(W2V), np.linalg.inv(W2Vp)
        cam, cam_prev = V2W[:3, 3].copy(), V2Wp[:3, 3].copy()
        delta = cam_prev - cam
        V2W[:3, 3] = 0.0
        W2V = np.linalg.inv(V2W)
        V2Wp[:3, 3] = delta
        W2Vp = np.linalg.inv(V2Wp)
        self.V2C, self.V2Cp, self.W2V, self.W2Vp, self.V2W, self.V2Wp = V2C, V2Cp, W2V, W2Vp, V2W, V2Wp
        self.W2C, self.W2Cp = V2C @ W2V, V2Cp @ W2Vp
        self.WP2W = _col_major(list(cs.worldPrevToWorldMatrix))
        self.camera_delta = delta

        def frustum(P):  # uv -> view ray: Xv.xy = (uv * f.zw + f.xy) * viewZ inverts GetScreenUv (uv.y flipped)
            return np.array([-(1.0 + P[0, 2]) / P[0, 0], (1.0 - P[1, 2]) / P[1, 1], 2.0 / P[0, 0], -2.0 / P[1, 1]])

        self.frustum, self.frustum_prev = frustum(V2C), frustum(V2Cp)
        self.project_y = abs(V2C[1, 1])
        self.view_dir, self.view_dir_prev = -V2W[:3, 2], -V2Wp[:3, 2]
        dt = float(cs.timeDeltaBetweenFrames)
        self.time_delta = dt
        self.framerate_scale = max(33.333 / dt, 1.0)
        self.jitter_delta = max(abs(cs.cameraJitter[0] - self.jitter_prev[0]), abs(cs.cameraJitter[1] - self.jitter_prev[1]))
        fps = self.framerate_scale * 30.0
        nl = fps * 0.25 / (1.0 + fps * 0.25)
        self.cb_resolve = nl + (0.5 - nl) * self.jitter_delta

    # -------------------------------------------------------------------------------------------------------------- REBLUR
    def reblur(self, s):
        cs = self.cs
        rw, rh = cs.rectSize[0], cs.rectSize[1]
        resw, resh = cs.resourceSize[0], cs.resourceSize[1]
        rwp, rhp = self.rect_prev
        reswp, reshp = self.res_prev
        reset = self.mode != int(api.AccumulationMode.CONTINUE)
        unproject = 1.0 / (0.5 * rh * self.project_y)
        worst = min(rw / resw, rh / resh)
        bonus = (1.0 + self.jitter_delta) / rh
        cb = {0: (2, 2), 1: (0, 1), 2: (1, 0)}[int(s.checkerboardMode)]
        return {
            "gWorldToClip": _flat(self.W2C), "gViewToClip": _flat(self.V2C), "gViewToWorld": _flat(self.V2W), "gWorldToViewPrev": _flat(self.W2Vp), "gWorldToClipPrev": _flat(self.W2Cp),
            "gWorldPrevToWorld": _flat(self.WP2W), "gFrustum": self.frustum, "gFrustumPrev": self.frustum_prev, "gCameraDelta": self.camera_delta, "gHitDistParams": list(s.hitDistanceParameters),
            "gViewVectorWorld": self.view_dir, "gViewVectorWorldPrev": self.view_dir_prev,
            "gMvScale": [cs.motionVectorScale[0], cs.motionVectorScale[1], cs.motionVectorScale[2], 1.0 if cs.isMotionVectorInWorldSpace else 0.0],
            "gAntilagParams": list(s.antilagSettings), "gResourceSize": [resw, resh], "gResourceSizeInv": [1.0 / resw, 1.0 / resh], "gResourceSizeInvPrev": [1.0 / reswp, 1.0 / reshp],
            "gRectSize": [rw, rh], "gRectSizeInv": [1.0 / rw, 1.0 / rh], "gRectSizePrev": [rwp, rhp], "gResolutionScale": [rw / resw, rh / resh], "gResolutionScalePrev": [rwp / reswp, rhp / reshp],
            "gRectOffset": [cs.rectOrigin[0] / resw, cs.rectOrigin[1] / resh],
            "gSpecProbabilityThresholdsForMvModification": list(s.specularProbabilityThresholdsForMvModification) if cs.isBaseColorMetalnessAvailable else [2.0, 3.0],
            "gJitter": list(cs.cameraJitter), "gPrintfAt": list(cs.printfAt), "gRectOrigin": list(cs.rectOrigin), "gRectSizeMinusOne": [rw - 1, rh - 1],
            "gDisocclusionThreshold": cs.disocclusionThreshold + bonus, "gDisocclusionThresholdAlternate": cs.disocclusionThresholdAlternate + bonus,
            "gCameraAttachedReflectionMaterialID": cs.cameraAttachedReflectionMaterialID, "gStrandMaterialID": cs.strandMaterialID, "gStrandThickness": cs.strandThickness,
            "gStabilizationStrength": 0.0 if reset else s.maxStabilizedFrameNum / (1.0 + s.maxStabilizedFrameNum),
            "gHitDistStabilizationStrength": 0.0 if reset else s.maxStabilizedFrameNumForHitDistance / (1.0 + s.maxStabilizedFrameNumForHitDistance),
            "gDebug": cs.debug, "gOrthoMode": 0.0, "gUnproject": unproject, "gDenoisingRange": cs.denoisingRange, "gPlaneDistSensitivity": s.planeDistanceSensitivity,
            "gFramerateScale": self.framerate_scale, "gMaxBlurRadius": max(s.maxBlurRadius * worst, s.minBlurRadius), "gMinBlurRadius": s.minBlurRadius,
            "gDiffPrepassBlurRadius": s.diffusePrepassBlurRadius * worst, "gSpecPrepassBlurRadius": s.specularPrepassBlurRadius * worst,
            "gMaxAccumulatedFrameNum": 0.0 if reset else float(min(s.maxAccumulatedFrameNum, 63)), "gMaxFastAccumulatedFrameNum": 0.0 if reset else float(s.maxFastAccumulatedFrameNum),
            "gAntiFirefly": 1.0 if s.enableAntiFirefly else 0.0, "gLobeAngleFraction": s.lobeAngleFraction * s.lobeAngleFraction, "gRoughnessFraction": s.roughnessFraction,
            "gResponsiveAccumulationRoughnessThreshold": s.responsiveAccumulationRoughnessThreshold, "gHistoryFixFrameNum": float(s.historyFixFrameNum),
            "gHistoryFixBasePixelStride": float(s.historyFixBasePixelStride), "gMinRectDimMulUnproject": min(rw, rh) * unproject,
            "gUsePrepassNotOnlyForSpecularMotionEstimation": 0.0 if s.usePrepassOnlyForSpecularMotionEstimation else 1.0, "gSplitScreen": cs.splitScreen, "gSplitScreenPrev": self.split_prev_out,
            "gCheckerboardResolveAccumSpeed": self.cb_resolve, "gViewZScale": cs.viewZScale, "gFireflySuppressorMinRelativeScale": s.fireflySuppressorMinRelativeScale,
            "gMinHitDistanceWeight": s.minHitDistanceWeight, "gDiffMinMaterial": s.minMaterialForDiffuse, "gSpecMinMaterial": s.minMaterialForSpecular,
            "gHasHistoryConfidence": float(bool(cs.isHistoryConfidenceAvailable)), "gHasDisocclusionThresholdMix": float(bool(cs.isDisocclusionThresholdMixAvailable)),
            "gDiffCheckerboard": cb[0], "gSpecCheckerboard": cb[1], "gFrameIndex": cs.frameIndex, "gIsRectChanged": float((rw, rh) != (rwp, rhp)), "gResetHistory": float(reset),
        }

    # -------------------------------------------------------------------------------------------------------------- RELAX
    def relax(self, s):
        cs = self.cs
        rw, rh = cs.rectSize[0], cs.rectSize[1]
        resw, resh = cs.resourceSize[0], cs.resourceSize[1]
        rwp, rhp = self.rect_prev
        reswp, reshp = self.res_prev
        reset = self.mode != int(api.AccumulationMode.CONTINUE)
        bonus = (1.0 + self.jitter_delta) / rh

        def basis(V2C, W2V, V2W, fr):
            tan_half = 1.0 / V2C[0, 0]
            aspect = V2C[0, 0] / V2C[1, 1]
            right, up = W2V[0, :3] * tan_half, W2V[1, :3] * tan_half * aspect
            fwd_view = np.array([0.5 * fr[2] + fr[0], 0.5 * fr[3] + fr[1], 1.0, 0.0])
            return right, up, (V2W @ fwd_view)[:3]

        r, u, f = basis(self.V2C, self.W2V, self.V2W, self.frustum)
        rp, up_, fp = basis(self.V2Cp, self.W2Vp, self.V2Wp, self.frustum_prev)
        cb = {0: (2, 2), 1: (0, 1), 2: (1, 0)}[int(s.checkerboardMode)]
        sat = lambda x: min(max(x, 0.0), 1.0)
        neglog = lambda x: -math.log(x) if x > 0.0 else math.inf  # -log(0) = +inf: the default (no limit on the luminance difference)
        a = list(s.antilagSettings)  # accelerationAmount, spatialSigmaScale, temporalSigmaScale, resetAmount (NRDSettings.h RelaxAntilagSettings)
        return {
            "gWorldToClip": _flat(self.W2C), "gWorldToClipPrev": _flat(self.W2Cp), "gWorldToViewPrev": _flat(self.W2Vp), "gWorldPrevToWorld": _flat(self.WP2W),
            "gFrustumRight": list(r) + [0.0], "gFrustumUp": list(u) + [0.0], "gFrustumForward": list(f) + [0.0], "gPrevFrustumRight": list(rp) + [0.0], "gPrevFrustumUp": list(up_) + [0.0],
            "gPrevFrustumForward": list(fp) + [0.0], "gCameraDelta": list(self.camera_delta) + [0.0],
            "gMvScale": [cs.motionVectorScale[0], cs.motionVectorScale[1], cs.motionVectorScale[2], 1.0 if cs.isMotionVectorInWorldSpace else 0.0],
            "gJitter": list(cs.cameraJitter), "gResolutionScale": [rw / resw, rh / resh], "gRectOffset": [cs.rectOrigin[0] / resw, cs.rectOrigin[1] / resh],
            "gResourceSizeInv": [1.0 / resw, 1.0 / resh], "gResourceSize": [resw, resh], "gRectSizeInv": [1.0 / rw, 1.0 / rh], "gRectSizePrev": [rwp, rhp],
            "gResourceSizeInvPrev": [1.0 / reswp, 1.0 / reshp], "gPrintfAt": list(cs.printfAt), "gRectOrigin": list(cs.rectOrigin), "gRectSize": [rw, rh],
            "gSpecMaxAccumulatedFrameNum": 0.0 if reset else float(min(s.specularMaxAccumulatedFrameNum, 255)),
            "gSpecMaxFastAccumulatedFrameNum": 0.0 if reset else float(min(s.specularMaxFastAccumulatedFrameNum, 255)),
            "gDiffMaxAccumulatedFrameNum": 0.0 if reset else float(min(s.diffuseMaxAccumulatedFrameNum, 255)),
            "gDiffMaxFastAccumulatedFrameNum": 0.0 if reset else float(min(s.diffuseMaxFastAccumulatedFrameNum, 255)),
            "gDisocclusionThreshold": cs.disocclusionThreshold + bonus, "gDisocclusionThresholdAlternate": cs.disocclusionThresholdAlternate + bonus,
            "gCameraAttachedReflectionMaterialID": cs.cameraAttachedReflectionMaterialID, "gStrandMaterialID": cs.strandMaterialID, "gStrandThickness": cs.strandThickness,
            "gRoughnessFraction": s.roughnessFraction, "gSpecVarianceBoost": s.specularVarianceBoost, "gSplitScreen": cs.splitScreen, "gDiffBlurRadius": s.diffusePrepassBlurRadius,
            "gSpecBlurRadius": s.specularPrepassBlurRadius, "gDepthThreshold": s.depthThreshold, "gLobeAngleFraction": s.lobeAngleFraction,
            "gSpecLobeAngleSlack": math.radians(s.specularLobeAngleSlack), "gHistoryFixEdgeStoppingNormalPower": s.historyFixEdgeStoppingNormalPower,
            "gRoughnessEdgeStoppingRelaxation": s.roughnessEdgeStoppingRelaxation, "gNormalEdgeStoppingRelaxation": s.normalEdgeStoppingRelaxation,
            "gColorBoxSigmaScale": s.historyClampingColorBoxSigmaScale, "gHistoryAccelerationAmount": a[0], "gHistoryResetTemporalSigmaScale": a[2], "gHistoryResetSpatialSigmaScale": a[1],
            "gHistoryResetAmount": a[3], "gDenoisingRange": cs.denoisingRange, "gSpecPhiLuminance": s.specularPhiLuminance, "gDiffPhiLuminance": s.diffusePhiLuminance,
            "gDiffMaxLuminanceRelativeDifference": neglog(sat(s.diffuseMinLuminanceWeight)), "gSpecMaxLuminanceRelativeDifference": neglog(sat(s.specularMinLuminanceWeight)),
            "gLuminanceEdgeStoppingRelaxation": s.roughnessEdgeStoppingRelaxation,  # sic (Relax.cpp:156)
            "gConfidenceDrivenRelaxationMultiplier": s.confidenceDrivenRelaxationMultiplier,
            "gConfidenceDrivenLuminanceEdgeStoppingRelaxation": s.confidenceDrivenLuminanceEdgeStoppingRelaxation,
            "gConfidenceDrivenNormalEdgeStoppingRelaxation": s.confidenceDrivenNormalEdgeStoppingRelaxation, "gDebug": cs.debug, "gOrthoMode": 0.0,
            "gUnproject": 1.0 / (0.5 * rh * self.project_y), "gFramerateScale": min(max(16.66 / self.time_delta, 0.25), 4.0), "gCheckerboardResolveAccumSpeed": self.cb_resolve,
            "gJitterDelta": self.jitter_delta, "gHistoryFixFrameNum": s.historyFixFrameNum + 1.0, "gHistoryFixBasePixelStride": float(s.historyFixBasePixelStride),
            "gHistoryThreshold": float(s.spatialVarianceEstimationHistoryThreshold), "gViewZScale": cs.viewZScale, "gMinHitDistanceWeight": s.minHitDistanceWeight * 2.0,
            "gDiffMinMaterial": s.minMaterialForDiffuse, "gSpecMinMaterial": s.minMaterialForSpecular, "gRoughnessEdgeStoppingEnabled": float(bool(s.enableRoughnessEdgeStopping)),
            "gFrameIndex": cs.frameIndex, "gDiffCheckerboard": cb[0], "gSpecCheckerboard": cb[1], "gHasHistoryConfidence": float(bool(cs.isHistoryConfidenceAvailable)),
            "gHasDisocclusionThresholdMix": float(bool(cs.isDisocclusionThresholdMixAvailable)), "gResetHistory": float(reset),
        }

    # -------------------------------------------------------------------------------------------------------------- SIGMA
    def sigma(self, s):
        cs = self.cs
        rw, rh = cs.rectSize[0], cs.rectSize[1]
        resw, resh = cs.resourceSize[0], cs.resourceSize[1]
        rwp, rhp = self.rect_prev
        reswp, reshp = self.res_prev
        unproject = 1.0 / (0.5 * rh * self.project_y)
        frames = min(s.maxStabilizedFrameNum, 7)
        light_view = self.W2V[:3, :3] @ np.array(list(s.lightDirection), dtype=np.float64)
        return {
            "gWorldToView": _flat(self.W2V), "gViewToClip": _flat(self.V2C), "gWorldToClipPrev": _flat(self.W2Cp), "gWorldToViewPrev": _flat(self.W2Vp), "gViewVectorWorld": self.view_dir,
            "gLightDirectionView": list(light_view) + [0.0], "gFrustum": self.frustum, "gFrustumPrev": self.frustum_prev, "gCameraDelta": self.camera_delta,
            "gMvScale": [cs.motionVectorScale[0], cs.motionVectorScale[1], cs.motionVectorScale[2], 1.0 if cs.isMotionVectorInWorldSpace else 0.0],
            "gResourceSizeInv": [1.0 / resw, 1.0 / resh], "gResourceSizeInvPrev": [1.0 / reswp, 1.0 / reshp], "gRectSize": [rw, rh], "gRectSizeInv": [1.0 / rw, 1.0 / rh],
            "gRectSizePrev": [rwp, rhp], "gResolutionScale": [rw / resw, rh / resh], "gRectOffset": [cs.rectOrigin[0] / resw, cs.rectOrigin[1] / resh], "gPrintfAt": list(cs.printfAt),
            "gRectOrigin": list(cs.rectOrigin), "gRectSizeMinusOne": [rw - 1, rh - 1], "gTilesSizeMinusOne": [(rw + 15) // 16 - 1, (rh + 15) // 16 - 1], "gOrthoMode": 0.0,
            "gUnproject": unproject, "gDenoisingRange": cs.denoisingRange, "gPlaneDistSensitivity": s.planeDistanceSensitivity,
            "gStabilizationStrength": frames / (1.0 + frames) if self.mode == int(api.AccumulationMode.CONTINUE) else 0.0, "gDebug": cs.debug, "gSplitScreen": cs.splitScreen,
            "gViewZScale": cs.viewZScale, "gMinRectDimMulUnproject": min(rw, rh) * unproject, "gFrameIndex": cs.frameIndex, "gIsRectChanged": float((rw, rh) != (rwp, rhp)),
        }


# ------------------------------------------------------------------------------------------------------------------ cameras
def _camera(kind, frame, width, height):
    """(viewToClip, worldToView) column-major lists for frame `frame`; kind: "lh" (the synthetic sequence's camera), "rh" (the same view expressed with a
    right-handed view space: z negated), "jitter" (lh with an asymmetric, sub-pixel-jittered projection)"""
    from raytracingdenoiser_amd import synth

    cam = synth.Camera(width, height, frame)
    v2c, w2v = np.array(cam.view_to_clip, dtype=np.float64).reshape(4, 4).T, np.array(cam.world_to_view, dtype=np.float64).reshape(4, 4).T
    jitter = (0.0, 0.0)
    if kind == "rh":
        flip = np.diag([1.0, 1.0, -1.0, 1.0])
        w2v = flip @ w2v  # view z' = -z
        v2c = v2c @ flip  # clip unchanged
    if kind == "jitter":
        jitter = (0.3 * ((frame * 7) % 5 - 2) / 2.0, 0.2 * ((frame * 3) % 5 - 2) / 2.0)
        v2c = v2c.copy()
        v2c[0, 2] += 2.0 * jitter[0] / width
        v2c[1, 2] -= 2.0 * jitter[1] / height
    return list(v2c.T.reshape(-1)), list(w2v.T.reshape(-1)), jitter


def _settings_for(name):
    if name.startswith("REBLUR"):
        return api.ReblurSettings(maxBlurRadius=25.0, lobeAngleFraction=0.2, historyFixFrameNum=2, enableAntiFirefly=True, minMaterialForSpecular=2.0, checkerboardMode=1)
    if name.startswith("RELAX"):
        return api.RelaxSettings(historyFixFrameNum=4, minHitDistanceWeight=0.15, roughnessEdgeStoppingRelaxation=0.8, luminanceEdgeStoppingRelaxation=0.3, specularLobeAngleSlack=0.25,
                                 diffuseMinLuminanceWeight=0.2, checkerboardMode=2, specularMaxAccumulatedFrameNum=300)
    return api.SigmaSettings(lightDirection=(0.3, 0.8, -0.52), maxStabilizedFrameNum=9)


CASES = [("REBLUR_DIFFUSE_SPECULAR", REBLUR_LAYOUT, 832, "reblur"), ("RELAX_DIFFUSE_SPECULAR", RELAX_LAYOUT, 704, "relax"), ("SIGMA_SHADOW", SIGMA_LAYOUT, None, "sigma")]


@pytest.mark.parametrize("kind", ["lh", "rh", "jitter"])
@pytest.mark.parametrize("name,layout,size,method", CASES, ids=[c[0] for c in CASES])
def test_shared_constants_match_the_reference_host_model(name, layout, size, method, kind):
    width, height, res = 320, 176, (352, 192)  # rect < resource: the resolution scales and the blur-radius scaling are exercised too
    inst = api.Instance([(0, parity.DENOISERS[name][0])])
    model = HostModel()
    settings = _settings_for(name)
    total = sum(SIZES[k] for _, k in layout)
    if size:
        assert total == size
    for frame in range(3):
        v2c, w2v, jitter = _camera(kind, frame, width, height)
        v2cp, w2vp, jitter_prev = _camera(kind, max(frame - 1, 0), width, height)
        cs = api.CommonSettings(resourceSize=res, rectSize=(width, height), resourceSizePrev=res, rectSizePrev=(width, height), timeDeltaBetweenFrames=11.0, frameIndex=frame,
                                isMotionVectorInWorldSpace=(kind != "jitter"), motionVectorScale=(0.0, 0.0, 0.0) if kind != "jitter" else (1.0 / width, 1.0 / height, 1.0),
                                cameraJitter=jitter, cameraJitterPrev=jitter_prev, splitScreen=0.25 * frame, viewZScale=1.0 if kind != "rh" else 1.0,
                                disocclusionThreshold=0.012, strandThickness=7e-5)
        for i in range(16):
            cs.viewToClipMatrix[i], cs.viewToClipMatrixPrev[i], cs.worldToViewMatrix[i], cs.worldToViewMatrixPrev[i] = v2c[i], v2cp[i], w2v[i], w2vp[i]
        assert inst.set_denoiser_settings(0, settings) == api.Result.SUCCESS
        assert inst.set_common_settings(cs) == api.Result.SUCCESS
        r, ds = inst.get_compute_dispatches()
        assert r == api.Result.SUCCESS
        blocks = [d.constants for d in ds if len(d.constants) >= total and not d.shader.startswith("Clear")]
        assert blocks, [d.shader for d in ds]
        got, used = parse_block(blocks[-1], layout)
        assert used == total
        model.set_common(cs)
        want = getattr(model, method)(settings)
        missing = [n for n, _ in layout if n not in want and not n.startswith("gRotator")]
        assert not missing, missing
        for fname, _ in layout:
            if fname.startswith("gRotator"):
                rot = got[fname]
                assert abs(rot[0] - rot[3]) < 1e-6 and abs(rot[1] + rot[2]) < 1e-6 and abs(rot[0] ** 2 + rot[1] ** 2 - 1.0) < 1e-5, (fname, rot)
                if fname == "gRotatorPre":
                    assert rot[0] >= -1e-6 and rot[1] >= -1e-6, rot  # an angle in [0, 90] degrees
                continue
            w = np.atleast_1d(np.array(want[fname], dtype=np.float64))
            g = got[fname][: w.size]
            tol = 2e-6 * np.maximum(np.abs(w), 1.0) + 1e-7
            if fname.startswith(("gWorldToClip", "gWorldToView", "gViewToWorld", "gViewToClip")):
                tol = 4e-6 * np.maximum(np.abs(w), 1.0) + 1e-6  # products / inverses of fp32 matrices
            same = (g == w) | (np.abs(g - w) <= tol)  # == covers infinities
            assert np.all(same), "%s frame %d (%s camera): got %s, model %s" % (fname, frame, kind, g, w)
    inst.destroy()
