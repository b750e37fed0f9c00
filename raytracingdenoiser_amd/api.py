"""ctypes mirror of the C-ABI in include/NRD.h, include/NRDDescs.h, include/NRDSettings.h and include/NRDHip.h.

Plumbing only: the product is lib/libNRD_hip.so (C++ host + HIP kernels). This module loads it, mirrors the POD
structs field-for-field (sizes asserted against the values the C++ headers static_assert) and offers a small
`Instance` convenience wrapper. Nothing here computes pixels and there is NO CPU fallback: without the built
library, loading fails loudly.
"""
import ctypes as C
import enum
import os

_PKG = os.path.dirname(os.path.abspath(__file__))
# the product library (raytracingdenoiser_amd/build.py): one library, one arithmetic (DESIGN.md "Numerics")
def _encoding():
    """(NRD_NORMAL_ENCODING, NRD_ROUGHNESS_ENCODING) of the library this process binds: a build configuration, as in the reference (CMakeLists.txt:28-29); build.py builds one
    library per encoding -- lib/libNRD_hip.so for the default (2, 1), lib/enc<N><R>/libNRD_hip.so otherwise -- and the environment variables of the same names pick one"""
    n, r = int(os.environ.get("NRD_NORMAL_ENCODING", "2")), int(os.environ.get("NRD_ROUGHNESS_ENCODING", "1"))
    if not (0 <= n <= 4 and 0 <= r <= 2):
        raise ValueError("NRD_NORMAL_ENCODING must be 0..4 and NRD_ROUGHNESS_ENCODING 0..2, got %d / %d" % (n, r))
    return n, r


NORMAL_ENCODING, ROUGHNESS_ENCODING = _encoding()
ENCODING_SUFFIX = "" if (NORMAL_ENCODING, ROUGHNESS_ENCODING) == (2, 1) else "_enc%d%d" % (NORMAL_ENCODING, ROUGHNESS_ENCODING)
NORMAL_ROUGHNESS_FORMAT_NAME = ["RGBA8_UNORM", "RGBA8_SNORM", "R10_G10_B10_A2_UNORM", "RGBA16_UNORM", "RGBA16_SNORM"][NORMAL_ENCODING]  # the format IN_NORMAL_ROUGHNESS is bound in
LIB_PATH = os.path.join(_PKG, "lib", ENCODING_SUFFIX.lstrip("_"), "libNRD_hip.so") if ENCODING_SUFFIX else os.path.join(_PKG, "lib", "libNRD_hip.so")


# ----------------------------------------------------------------------------------------------- enums
class Result(enum.IntEnum):
    SUCCESS = 0
    FAILURE = 1
    INVALID_ARGUMENT = 2
    UNSUPPORTED = 3
    NON_UNIQUE_IDENTIFIER = 4


_RESOURCE_TYPES = [
    "IN_MV", "IN_NORMAL_ROUGHNESS", "IN_VIEWZ", "IN_DIFF_CONFIDENCE", "IN_SPEC_CONFIDENCE", "IN_DISOCCLUSION_THRESHOLD_MIX",
    "IN_BASECOLOR_METALNESS", "IN_DIFF_RADIANCE_HITDIST", "IN_SPEC_RADIANCE_HITDIST", "IN_DIFF_HITDIST", "IN_SPEC_HITDIST",
    "IN_DIFF_DIRECTION_HITDIST", "IN_DIFF_SH0", "IN_DIFF_SH1", "IN_SPEC_SH0", "IN_SPEC_SH1", "IN_PENUMBRA", "IN_TRANSLUCENCY",
    "IN_SIGNAL", "OUT_DIFF_RADIANCE_HITDIST", "OUT_SPEC_RADIANCE_HITDIST", "OUT_DIFF_SH0", "OUT_DIFF_SH1", "OUT_SPEC_SH0",
    "OUT_SPEC_SH1", "OUT_DIFF_HITDIST", "OUT_SPEC_HITDIST", "OUT_DIFF_DIRECTION_HITDIST", "OUT_SHADOW_TRANSLUCENCY", "OUT_SIGNAL",
    "OUT_VALIDATION", "TRANSIENT_POOL", "PERMANENT_POOL", "MAX_NUM",
]
ResourceType = enum.IntEnum("ResourceType", {n: i for i, n in enumerate(_RESOURCE_TYPES)})

_DENOISERS = [
    "REBLUR_DIFFUSE", "REBLUR_DIFFUSE_OCCLUSION", "REBLUR_DIFFUSE_SH", "REBLUR_SPECULAR", "REBLUR_SPECULAR_OCCLUSION",
    "REBLUR_SPECULAR_SH", "REBLUR_DIFFUSE_SPECULAR", "REBLUR_DIFFUSE_SPECULAR_OCCLUSION", "REBLUR_DIFFUSE_SPECULAR_SH",
    "REBLUR_DIFFUSE_DIRECTIONAL_OCCLUSION", "RELAX_DIFFUSE", "RELAX_DIFFUSE_SH", "RELAX_SPECULAR", "RELAX_SPECULAR_SH",
    "RELAX_DIFFUSE_SPECULAR", "RELAX_DIFFUSE_SPECULAR_SH", "SIGMA_SHADOW", "SIGMA_SHADOW_TRANSLUCENCY", "REFERENCE", "MAX_NUM",
]
Denoiser = enum.IntEnum("Denoiser", {n: i for i, n in enumerate(_DENOISERS)})

_FORMATS = [
    "R8_UNORM", "R8_SNORM", "R8_UINT", "R8_SINT", "RG8_UNORM", "RG8_SNORM", "RG8_UINT", "RG8_SINT", "RGBA8_UNORM", "RGBA8_SNORM",
    "RGBA8_UINT", "RGBA8_SINT", "RGBA8_SRGB", "R16_UNORM", "R16_SNORM", "R16_UINT", "R16_SINT", "R16_SFLOAT", "RG16_UNORM",
    "RG16_SNORM", "RG16_UINT", "RG16_SINT", "RG16_SFLOAT", "RGBA16_UNORM", "RGBA16_SNORM", "RGBA16_UINT", "RGBA16_SINT",
    "RGBA16_SFLOAT", "R32_UINT", "R32_SINT", "R32_SFLOAT", "RG32_UINT", "RG32_SINT", "RG32_SFLOAT", "RGB32_UINT", "RGB32_SINT",
    "RGB32_SFLOAT", "RGBA32_UINT", "RGBA32_SINT", "RGBA32_SFLOAT", "R10_G10_B10_A2_UNORM", "R10_G10_B10_A2_UINT",
    "R11_G11_B10_UFLOAT", "R9_G9_B9_E5_UFLOAT", "MAX_NUM",
]
Format = enum.IntEnum("Format", {n: i for i, n in enumerate(_FORMATS)})

FORMAT_BYTES = {
    Format.R8_UNORM: 1, Format.R8_UINT: 1, Format.RG8_UNORM: 2, Format.R16_UINT: 2, Format.R16_SFLOAT: 2, Format.R16_UNORM: 2,
    Format.RGBA8_UNORM: 4, Format.RGBA8_SNORM: 4, Format.RGBA16_UNORM: 8, Format.R32_UINT: 4, Format.R32_SFLOAT: 4, Format.R10_G10_B10_A2_UNORM: 4, Format.RG16_SFLOAT: 4,
    Format.RGBA16_SFLOAT: 8, Format.RGBA16_SNORM: 8, Format.RGBA32_SFLOAT: 16, Format.R11_G11_B10_UFLOAT: 4,
}


class DescriptorType(enum.IntEnum):
    TEXTURE = 0
    STORAGE_TEXTURE = 1


class AccumulationMode(enum.IntEnum):
    CONTINUE = 0
    RESTART = 1
    CLEAR_AND_RESTART = 2


class CheckerboardMode(enum.IntEnum):
    OFF = 0
    BLACK = 1
    WHITE = 2


# ----------------------------------------------------------------------------------------------- descs
class AllocationCallbacks(C.Structure):
    _fields_ = [("Allocate", C.c_void_p), ("Reallocate", C.c_void_p), ("Free", C.c_void_p), ("userArg", C.c_void_p)]


class DenoiserDesc(C.Structure):
    _fields_ = [("identifier", C.c_uint32), ("denoiser", C.c_uint32)]


class InstanceCreationDesc(C.Structure):
    _fields_ = [("allocationCallbacks", AllocationCallbacks), ("denoisers", C.POINTER(DenoiserDesc)), ("denoisersNum", C.c_uint32)]


class TextureDesc(C.Structure):
    _fields_ = [("format", C.c_uint32), ("downsampleFactor", C.c_uint16)]


class ResourceDesc(C.Structure):
    _fields_ = [("descriptorType", C.c_uint32), ("type", C.c_uint32), ("indexInPool", C.c_uint16)]


class ResourceRangeDesc(C.Structure):
    _fields_ = [("descriptorType", C.c_uint32), ("baseRegisterIndex", C.c_uint32), ("descriptorsNum", C.c_uint32)]


class ComputeShaderDesc(C.Structure):
    _fields_ = [("bytecode", C.c_void_p), ("size", C.c_uint64)]


class PipelineDesc(C.Structure):
    _fields_ = [
        ("computeShaderDXBC", ComputeShaderDesc), ("computeShaderDXIL", ComputeShaderDesc), ("computeShaderSPIRV", ComputeShaderDesc),
        ("shaderFileName", C.c_char_p), ("shaderEntryPointName", C.c_char_p), ("resourceRanges", C.POINTER(ResourceRangeDesc)),
        ("resourceRangesNum", C.c_uint32), ("hasConstantData", C.c_bool),
    ]


class DescriptorPoolDesc(C.Structure):
    _fields_ = [("setsMaxNum", C.c_uint32), ("constantBuffersMaxNum", C.c_uint32), ("samplersMaxNum", C.c_uint32),
                ("texturesMaxNum", C.c_uint32), ("storageTexturesMaxNum", C.c_uint32)]


class InstanceDesc(C.Structure):
    _fields_ = [
        ("constantBufferMaxDataSize", C.c_uint32), ("constantBufferSpaceIndex", C.c_uint32), ("constantBufferRegisterIndex", C.c_uint32),
        ("samplers", C.POINTER(C.c_uint32)), ("samplersNum", C.c_uint32), ("samplersSpaceIndex", C.c_uint32), ("samplersBaseRegisterIndex", C.c_uint32),
        ("pipelines", C.POINTER(PipelineDesc)), ("pipelinesNum", C.c_uint32), ("resourcesSpaceIndex", C.c_uint32),
        ("permanentPool", C.POINTER(TextureDesc)), ("permanentPoolSize", C.c_uint32),
        ("transientPool", C.POINTER(TextureDesc)), ("transientPoolSize", C.c_uint32),
        ("descriptorPoolDesc", DescriptorPoolDesc),
    ]


class DispatchDesc(C.Structure):
    _fields_ = [
        ("name", C.c_char_p), ("identifier", C.c_uint32), ("resources", C.POINTER(ResourceDesc)), ("resourcesNum", C.c_uint32),
        ("constantBufferData", C.POINTER(C.c_uint8)), ("constantBufferDataSize", C.c_uint32), ("constantBufferDataMatchesPreviousDispatch", C.c_bool),
        ("pipelineIndex", C.c_uint16), ("gridWidth", C.c_uint16), ("gridHeight", C.c_uint16),
    ]


class SPIRVBindingOffsets(C.Structure):
    _fields_ = [("samplerOffset", C.c_uint32), ("textureOffset", C.c_uint32), ("constantBufferOffset", C.c_uint32), ("storageTextureAndBufferOffset", C.c_uint32)]


class LibraryDesc(C.Structure):
    _fields_ = [
        ("spirvBindingOffsets", SPIRVBindingOffsets), ("supportedDenoisers", C.POINTER(C.c_uint32)), ("supportedDenoisersNum", C.c_uint32),
        ("versionMajor", C.c_uint8), ("versionMinor", C.c_uint8), ("versionBuild", C.c_uint8), ("normalEncoding", C.c_uint8), ("roughnessEncoding", C.c_uint8),
    ]


# ----------------------------------------------------------------------------------------------- settings
_IDENTITY = (1.0, 0.0, 0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0, 0.0, 1.0)


class _Defaulted(C.Structure):
    _defaults_ = {}

    def __init__(self, **kw):
        super().__init__()
        for k, v in self._defaults_.items():
            self._set(k, v)
        for k, v in kw.items():
            self._set(k, v)

    def _set(self, k, v):
        if isinstance(v, (tuple, list)):
            arr = getattr(self, k)
            for i, x in enumerate(v):
                arr[i] = x
        else:
            setattr(self, k, v)


class CommonSettings(_Defaulted):
    _fields_ = [
        ("viewToClipMatrix", C.c_float * 16), ("viewToClipMatrixPrev", C.c_float * 16), ("worldToViewMatrix", C.c_float * 16),
        ("worldToViewMatrixPrev", C.c_float * 16), ("worldPrevToWorldMatrix", C.c_float * 16), ("motionVectorScale", C.c_float * 3),
        ("cameraJitter", C.c_float * 2), ("cameraJitterPrev", C.c_float * 2), ("resourceSize", C.c_uint16 * 2), ("resourceSizePrev", C.c_uint16 * 2),
        ("rectSize", C.c_uint16 * 2), ("rectSizePrev", C.c_uint16 * 2), ("viewZScale", C.c_float), ("timeDeltaBetweenFrames", C.c_float),
        ("denoisingRange", C.c_float), ("disocclusionThreshold", C.c_float), ("disocclusionThresholdAlternate", C.c_float),
        ("cameraAttachedReflectionMaterialID", C.c_float), ("strandMaterialID", C.c_float), ("strandThickness", C.c_float), ("splitScreen", C.c_float),
        ("printfAt", C.c_uint16 * 2), ("debug", C.c_float), ("rectOrigin", C.c_uint32 * 2), ("frameIndex", C.c_uint32), ("accumulationMode", C.c_uint8),
        ("isMotionVectorInWorldSpace", C.c_bool), ("isHistoryConfidenceAvailable", C.c_bool), ("isDisocclusionThresholdMixAvailable", C.c_bool),
        ("isBaseColorMetalnessAvailable", C.c_bool), ("enableValidation", C.c_bool),
    ]
    _defaults_ = dict(worldPrevToWorldMatrix=_IDENTITY, motionVectorScale=(1.0, 1.0, 0.0), viewZScale=1.0, denoisingRange=500000.0,
                      disocclusionThreshold=0.01, disocclusionThresholdAlternate=0.05, cameraAttachedReflectionMaterialID=999.0,
                      strandMaterialID=999.0, strandThickness=80e-6, printfAt=(9999, 9999))


class ReblurSettings(_Defaulted):
    _fields_ = [
        ("hitDistanceParameters", C.c_float * 4), ("antilagSettings", C.c_float * 2), ("maxAccumulatedFrameNum", C.c_uint32),
        ("maxFastAccumulatedFrameNum", C.c_uint32), ("maxStabilizedFrameNum", C.c_uint32), ("maxStabilizedFrameNumForHitDistance", C.c_uint32),
        ("historyFixFrameNum", C.c_uint32), ("historyFixBasePixelStride", C.c_uint32), ("diffusePrepassBlurRadius", C.c_float),
        ("specularPrepassBlurRadius", C.c_float), ("minHitDistanceWeight", C.c_float), ("minBlurRadius", C.c_float), ("maxBlurRadius", C.c_float),
        ("lobeAngleFraction", C.c_float), ("roughnessFraction", C.c_float), ("responsiveAccumulationRoughnessThreshold", C.c_float),
        ("planeDistanceSensitivity", C.c_float), ("specularProbabilityThresholdsForMvModification", C.c_float * 2),
        ("fireflySuppressorMinRelativeScale", C.c_float), ("checkerboardMode", C.c_uint8), ("hitDistanceReconstructionMode", C.c_uint8),
        ("enableAntiFirefly", C.c_bool), ("enablePerformanceMode", C.c_bool), ("minMaterialForDiffuse", C.c_float), ("minMaterialForSpecular", C.c_float),
        ("usePrepassOnlyForSpecularMotionEstimation", C.c_bool),
    ]
    _defaults_ = dict(hitDistanceParameters=(3.0, 0.1, 20.0, -25.0), antilagSettings=(4.0, 3.0), maxAccumulatedFrameNum=30, maxFastAccumulatedFrameNum=6,
                      maxStabilizedFrameNum=63, maxStabilizedFrameNumForHitDistance=63, historyFixFrameNum=3, historyFixBasePixelStride=14,
                      diffusePrepassBlurRadius=30.0, specularPrepassBlurRadius=50.0, minHitDistanceWeight=0.1, minBlurRadius=1.0, maxBlurRadius=30.0,
                      lobeAngleFraction=0.15, roughnessFraction=0.15, planeDistanceSensitivity=0.02, specularProbabilityThresholdsForMvModification=(0.5, 0.9),
                      fireflySuppressorMinRelativeScale=2.0, minMaterialForDiffuse=4.0, minMaterialForSpecular=4.0)


class SigmaSettings(_Defaulted):
    _fields_ = [("lightDirection", C.c_float * 3), ("planeDistanceSensitivity", C.c_float), ("maxStabilizedFrameNum", C.c_uint32)]
    _defaults_ = dict(planeDistanceSensitivity=0.02, maxStabilizedFrameNum=5)


class ReferenceSettings(_Defaulted):
    _fields_ = [("maxAccumulatedFrameNum", C.c_uint32)]
    _defaults_ = dict(maxAccumulatedFrameNum=1020)


class RelaxSettings(_Defaulted):
    _fields_ = [
        ("antilagSettings", C.c_float * 4), ("diffuseMaxAccumulatedFrameNum", C.c_uint32), ("specularMaxAccumulatedFrameNum", C.c_uint32),
        ("diffuseMaxFastAccumulatedFrameNum", C.c_uint32), ("specularMaxFastAccumulatedFrameNum", C.c_uint32), ("historyFixFrameNum", C.c_uint32),
        ("historyFixBasePixelStride", C.c_uint32), ("historyFixEdgeStoppingNormalPower", C.c_float), ("spatialVarianceEstimationHistoryThreshold", C.c_uint32),
        ("diffusePrepassBlurRadius", C.c_float), ("specularPrepassBlurRadius", C.c_float), ("minHitDistanceWeight", C.c_float),
        ("diffusePhiLuminance", C.c_float), ("specularPhiLuminance", C.c_float), ("lobeAngleFraction", C.c_float), ("roughnessFraction", C.c_float),
        ("specularVarianceBoost", C.c_float), ("specularLobeAngleSlack", C.c_float), ("historyClampingColorBoxSigmaScale", C.c_float),
        ("atrousIterationNum", C.c_uint32), ("diffuseMinLuminanceWeight", C.c_float), ("specularMinLuminanceWeight", C.c_float), ("depthThreshold", C.c_float),
        ("confidenceDrivenRelaxationMultiplier", C.c_float), ("confidenceDrivenLuminanceEdgeStoppingRelaxation", C.c_float),
        ("confidenceDrivenNormalEdgeStoppingRelaxation", C.c_float), ("luminanceEdgeStoppingRelaxation", C.c_float), ("normalEdgeStoppingRelaxation", C.c_float),
        ("roughnessEdgeStoppingRelaxation", C.c_float), ("checkerboardMode", C.c_uint8), ("hitDistanceReconstructionMode", C.c_uint8),
        ("enableAntiFirefly", C.c_bool), ("enableRoughnessEdgeStopping", C.c_bool), ("minMaterialForDiffuse", C.c_float), ("minMaterialForSpecular", C.c_float),
    ]
    _defaults_ = dict(antilagSettings=(0.3, 4.5, 0.5, 0.5), diffuseMaxAccumulatedFrameNum=30, specularMaxAccumulatedFrameNum=30,
                      diffuseMaxFastAccumulatedFrameNum=6, specularMaxFastAccumulatedFrameNum=6, historyFixFrameNum=3, historyFixBasePixelStride=14,
                      historyFixEdgeStoppingNormalPower=8.0, spatialVarianceEstimationHistoryThreshold=3, diffusePrepassBlurRadius=30.0,
                      specularPrepassBlurRadius=50.0, minHitDistanceWeight=0.1, diffusePhiLuminance=2.0, specularPhiLuminance=1.0, lobeAngleFraction=0.5,
                      roughnessFraction=0.15, specularLobeAngleSlack=0.15, historyClampingColorBoxSigmaScale=2.0, atrousIterationNum=5, depthThreshold=0.003,
                      luminanceEdgeStoppingRelaxation=0.5, normalEdgeStoppingRelaxation=0.3, roughnessEdgeStoppingRelaxation=1.0,
                      enableRoughnessEdgeStopping=True, minMaterialForDiffuse=4.0, minMaterialForSpecular=4.0)


assert C.sizeof(CommonSettings) == 428 and C.sizeof(ReblurSettings) == 112 and C.sizeof(RelaxSettings) == 140
assert C.sizeof(SigmaSettings) == 20 and C.sizeof(ReferenceSettings) == 4
assert C.sizeof(DispatchDesc) == 56 and C.sizeof(InstanceDesc) == 104 and C.sizeof(PipelineDesc) == 80 and C.sizeof(LibraryDesc) == 40


class HipPlaneDesc(C.Structure):
    _fields_ = [("data", C.c_void_p), ("rowPitchBytes", C.c_uint32), ("format", C.c_uint32), ("width", C.c_uint16), ("height", C.c_uint16)]


# ----------------------------------------------------------------------------------------------- library
# every symbol the headers declare; tests assert each one resolves
NRD_SYMBOLS = ["CreateInstance", "DestroyInstance", "GetLibraryDesc", "GetInstanceDesc", "SetCommonSettings", "SetDenoiserSettings",
               "GetComputeDispatches", "GetResourceTypeString", "GetDenoiserString"]
class HipHaloItem(C.Structure):  # include/NRDHip.h NrdHipHaloItem
    _fields_ = [("resourceType", C.c_uint32), ("indexInPool", C.c_uint32), ("widthRows", C.c_uint32)]


class HipHaloStep(C.Structure):
    _fields_ = [("firstDispatch", C.c_uint32), ("dispatchCount", C.c_uint32), ("earlyCount", C.c_uint32), ("firstItem", C.c_uint32), ("itemCount", C.c_uint32)]


class HipHaloPlanInfo(C.Structure):
    _fields_ = [("fallback", C.c_uint32), ("stepsNum", C.c_uint32), ("itemsNum", C.c_uint32)]


NRD_HIP_SYMBOLS = ["nrdHipCreateExecutor", "nrdHipDestroyExecutor", "nrdHipBindResource", "nrdHipGetPoolPlane", "nrdHipExecuteDispatches",
                   "nrdHipDenoise", "nrdHipGetPoolMemoryUsage", "nrdHipGetLastError", "nrdHipEvalNumerics", "nrdHipGetArenaSize",
                   "nrdHipCreateExecutorWithArena", "nrdHipSetProfiling", "nrdHipCollectPassTimings", "nrdHipSetOwnedRows", "nrdHipGetDispatchReach",
                   "nrdHipExecuteDispatchRange", "nrdHipPlanHaloExchange", "nrdHipSetGraphMode", "nrdHipGetGraphStats", "nrdHipGetTileFallbackStats", "nrdHipGetNumericsMode", "nrdHipMeasureCopyBandwidth",
                   "nrdHipMeasureMotionRows", "nrdHipMeasureMotionRowsAsync", "nrdHipSetHistoryReachWord"]

_libs = {}


def load_library(path=None):
    """dlopen lib/libNRD_hip.so and set prototypes. Raises if the library has not been built -- there is no fallback.
    NRD_HIP_LIBRARY=<path> points at an A/B variant build of the same sources (tools/build_variant.py)."""
    path = path or os.environ.get("NRD_HIP_LIBRARY") or LIB_PATH
    if path in _libs:
        return _libs[path]
    if not os.path.exists(path):
        raise RuntimeError("NRD HIP library not built: %s (run `python -c 'import __graft_entry__ as g; g.build()'`)" % path)
    # Load order matters in a process that also uses PyTorch: torch bundles its own libamdhip64, and if this library came first the loader
    # would bind it to the system ROCm copy instead -- two HIP runtimes in one process, and the one behind the executor then reports no
    # device (nrdHipCreateExecutor -> FAILURE). Importing torch first makes both share torch's runtime; without torch nothing changes.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(path)
    P = C.POINTER
    lib.CreateInstance.argtypes, lib.CreateInstance.restype = [P(InstanceCreationDesc), P(C.c_void_p)], C.c_uint32
    lib.DestroyInstance.argtypes, lib.DestroyInstance.restype = [C.c_void_p], None
    lib.GetLibraryDesc.argtypes, lib.GetLibraryDesc.restype = [], P(LibraryDesc)
    lib.GetInstanceDesc.argtypes, lib.GetInstanceDesc.restype = [C.c_void_p], P(InstanceDesc)
    lib.SetCommonSettings.argtypes, lib.SetCommonSettings.restype = [C.c_void_p, P(CommonSettings)], C.c_uint32
    lib.SetDenoiserSettings.argtypes, lib.SetDenoiserSettings.restype = [C.c_void_p, C.c_uint32, C.c_void_p], C.c_uint32
    lib.GetComputeDispatches.argtypes = [C.c_void_p, P(C.c_uint32), C.c_uint32, P(P(DispatchDesc)), P(C.c_uint32)]
    lib.GetComputeDispatches.restype = C.c_uint32
    lib.GetResourceTypeString.argtypes, lib.GetResourceTypeString.restype = [C.c_uint32], C.c_char_p
    lib.GetDenoiserString.argtypes, lib.GetDenoiserString.restype = [C.c_uint32], C.c_char_p

    lib.nrdHipCreateExecutor.argtypes = [C.c_void_p, C.c_uint16, C.c_uint16, C.c_void_p, P(C.c_void_p)]
    lib.nrdHipCreateExecutor.restype = C.c_uint32
    lib.nrdHipDestroyExecutor.argtypes, lib.nrdHipDestroyExecutor.restype = [C.c_void_p], None
    lib.nrdHipBindResource.argtypes, lib.nrdHipBindResource.restype = [C.c_void_p, C.c_uint32, P(HipPlaneDesc)], C.c_uint32
    lib.nrdHipGetPoolPlane.argtypes, lib.nrdHipGetPoolPlane.restype = [C.c_void_p, C.c_uint32, C.c_uint32, P(HipPlaneDesc)], C.c_uint32
    lib.nrdHipExecuteDispatches.argtypes, lib.nrdHipExecuteDispatches.restype = [C.c_void_p, C.c_void_p, C.c_uint32], C.c_uint32
    lib.nrdHipDenoise.argtypes, lib.nrdHipDenoise.restype = [C.c_void_p, P(C.c_uint32), C.c_uint32], C.c_uint32
    lib.nrdHipGetPoolMemoryUsage.argtypes = [C.c_void_p, P(C.c_uint64), P(C.c_uint64)]
    lib.nrdHipGetPoolMemoryUsage.restype = C.c_uint32
    lib.nrdHipGetLastError.argtypes, lib.nrdHipGetLastError.restype = [C.c_void_p], C.c_char_p
    lib.nrdHipGetArenaSize.argtypes, lib.nrdHipGetArenaSize.restype = [C.c_void_p, C.c_uint16, C.c_uint16], C.c_uint64
    lib.nrdHipCreateExecutorWithArena.argtypes = [C.c_void_p, C.c_uint16, C.c_uint16, C.c_void_p, C.c_void_p, C.c_uint64, P(C.c_void_p)]
    lib.nrdHipCreateExecutorWithArena.restype = C.c_uint32
    lib.nrdHipSetOwnedRows.argtypes, lib.nrdHipSetOwnedRows.restype = [C.c_void_p, C.c_uint32, C.c_uint32], C.c_uint32
    lib.nrdHipGetDispatchReach.argtypes, lib.nrdHipGetDispatchReach.restype = [C.c_void_p, C.c_void_p, C.c_uint32, C.POINTER(C.c_int32)], C.c_uint32
    lib.nrdHipExecuteDispatchRange.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    lib.nrdHipExecuteDispatchRange.restype = C.c_uint32
    lib.nrdHipPlanHaloExchange.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, P(C.c_uint32), C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, P(C.c_int32), P(C.c_int32),
                                           P(HipHaloStep), C.c_uint32, P(HipHaloItem), C.c_uint32, P(HipHaloPlanInfo)]
    lib.nrdHipPlanHaloExchange.restype = C.c_uint32
    lib.nrdHipSetProfiling.argtypes, lib.nrdHipSetProfiling.restype = [C.c_void_p, C.c_uint32], C.c_uint32
    lib.nrdHipCollectPassTimings.argtypes = [C.c_void_p, P(C.c_uint32), P(C.c_double), P(C.c_uint32), C.c_uint32, P(C.c_uint32)]
    lib.nrdHipCollectPassTimings.restype = C.c_uint32
    lib.nrdHipEvalNumerics.argtypes = [C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
    lib.nrdHipEvalNumerics.restype = C.c_uint32
    lib.nrdHipSetGraphMode.argtypes, lib.nrdHipSetGraphMode.restype = [C.c_void_p, C.c_uint32], C.c_uint32
    lib.nrdHipGetGraphStats.argtypes, lib.nrdHipGetGraphStats.restype = [C.c_void_p, P(C.c_uint64), P(C.c_uint64), P(C.c_uint64)], C.c_uint32
    lib.nrdHipGetTileFallbackStats.argtypes, lib.nrdHipGetTileFallbackStats.restype = [C.c_void_p, P(C.c_uint32), P(C.c_uint32)], C.c_uint32
    lib.nrdHipMeasureMotionRows.argtypes, lib.nrdHipMeasureMotionRows.restype = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, P(C.c_float)], C.c_uint32
    lib.nrdHipSetHistoryReachWord.argtypes, lib.nrdHipSetHistoryReachWord.restype = [C.c_void_p, C.c_void_p], C.c_uint32
    if hasattr(lib, "nrdHipMeasureMotionRowsAsync"):  # (absent from the CPU emulation of the device sources: tests/emu)
        lib.nrdHipMeasureMotionRowsAsync.argtypes, lib.nrdHipMeasureMotionRowsAsync.restype = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p], C.c_uint32
    lib.nrdHipGetNumericsMode.argtypes, lib.nrdHipGetNumericsMode.restype = [], C.c_uint32
    lib.nrdHipMeasureCopyBandwidth.argtypes, lib.nrdHipMeasureCopyBandwidth.restype = [C.c_uint64, C.c_uint32, C.c_void_p, P(C.c_double)], C.c_uint32
    _libs[path] = lib
    return lib


class Dispatch:
    """Python view of one nrd::DispatchDesc (copied out, since the instance overwrites the array on the next call)."""

    def __init__(self, d, pipelines):
        self.name = d.name.decode() if d.name else ""
        self.identifier = d.identifier
        self.pipeline_index = d.pipelineIndex
        self.shader = pipelines[d.pipelineIndex]
        self.grid = (d.gridWidth, d.gridHeight)
        self.resources = [(DescriptorType(d.resources[i].descriptorType), ResourceType(d.resources[i].type), d.resources[i].indexInPool) for i in range(d.resourcesNum)]
        self.constants = bytes(C.string_at(d.constantBufferData, d.constantBufferDataSize)) if d.constantBufferDataSize and d.constantBufferData else b""
        self.constants_match_previous = bool(d.constantBufferDataMatchesPreviousDispatch)

    def __repr__(self):
        return "Dispatch(%s, %s, grid=%s)" % (self.name, self.shader, self.grid)


class Instance:
    """Thin RAII wrapper over nrd::CreateInstance / DestroyInstance."""

    def __init__(self, denoisers, lib=None):
        """denoisers: list of (identifier, Denoiser)"""
        self.lib = lib or load_library()
        self._descs = (DenoiserDesc * len(denoisers))(*[DenoiserDesc(i, int(d)) for i, d in denoisers])
        icd = InstanceCreationDesc()
        icd.denoisers = self._descs
        icd.denoisersNum = len(denoisers)
        handle = C.c_void_p()
        r = Result(self.lib.CreateInstance(C.byref(icd), C.byref(handle)))
        if r != Result.SUCCESS:
            raise RuntimeError("nrd::CreateInstance failed: %s" % r.name)
        self.handle = handle
        self.identifiers = [i for i, _ in denoisers]
        d = self.desc
        self.pipelines = [d.pipelines[i].shaderFileName.decode() for i in range(d.pipelinesNum)]
        self.permanent_pool = [(Format(d.permanentPool[i].format), d.permanentPool[i].downsampleFactor) for i in range(d.permanentPoolSize)]
        self.transient_pool = [(Format(d.transientPool[i].format), d.transientPool[i].downsampleFactor) for i in range(d.transientPoolSize)]

    @property
    def desc(self):
        return self.lib.GetInstanceDesc(self.handle).contents

    def set_common_settings(self, cs):
        self.last_common_settings = cs  # kept for hosts that derive per-frame bounds from the camera (sharding.HaloSharder)
        return Result(self.lib.SetCommonSettings(self.handle, C.byref(cs)))

    def set_denoiser_settings(self, identifier, settings):
        return Result(self.lib.SetDenoiserSettings(self.handle, identifier, C.cast(C.byref(settings), C.c_void_p)))

    def get_compute_dispatches_raw(self, identifiers=None):
        ids = identifiers if identifiers is not None else self.identifiers
        arr = (C.c_uint32 * len(ids))(*ids)
        out = C.POINTER(DispatchDesc)()
        num = C.c_uint32()
        r = Result(self.lib.GetComputeDispatches(self.handle, arr, len(ids), C.byref(out), C.byref(num)))
        return r, out, num.value

    def dispatch_reach(self, dispatch_ptr, num):
        """rows above / below a pixel each dispatch reads from planes written earlier in the frame (-1 = unknown: whole-frame pass)"""
        out = (C.c_int32 * max(num, 1))()
        r = Result(self.lib.nrdHipGetDispatchReach(self.handle, C.cast(dispatch_ptr, C.c_void_p), num, out))
        assert r == Result.SUCCESS, r
        return list(out[:num])

    def plan_halo_exchange(self, dispatch_ptr, num, strip_bounds, rank, height, max_motion_rows=32, exchange_threshold=24):
        """nrdHipPlanHaloExchange: (fallback, steps [(items [((type, index), width)], first, count, early)], row_begin, row_end)"""
        world = len(strip_bounds) - 1
        bounds = (C.c_uint32 * (world + 1))(*strip_bounds)
        rb, re = (C.c_int32 * max(num, 1))(), (C.c_int32 * max(num, 1))()
        steps, items, info = (HipHaloStep * 64)(), (HipHaloItem * 1024)(), HipHaloPlanInfo()
        r = Result(self.lib.nrdHipPlanHaloExchange(self.handle, C.cast(dispatch_ptr, C.c_void_p), num, bounds, world, rank, height, max_motion_rows, exchange_threshold, rb, re, steps, 64,
                                                   items, 1024, C.byref(info)))
        assert r == Result.SUCCESS, r
        out = [([((items[k].resourceType, items[k].indexInPool), items[k].widthRows) for k in range(st.firstItem, st.firstItem + st.itemCount)], st.firstDispatch, st.dispatchCount,
                st.earlyCount) for st in steps[: info.stepsNum]]
        return bool(info.fallback), out, list(rb[:num]), list(re[:num])

    def get_compute_dispatches(self, identifiers=None):
        r, out, num = self.get_compute_dispatches_raw(identifiers)
        if r != Result.SUCCESS:
            return r, []
        return r, [Dispatch(out[i], self.pipelines) for i in range(num)]

    def destroy(self):
        if self.handle:
            self.lib.DestroyInstance(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass
