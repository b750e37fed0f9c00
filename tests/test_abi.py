"""C-ABI surface: the library loads, exports every symbol include/*.h declares, and keeps the reference's
error behaviour (reference Source/InstanceImpl.cpp:116-124, :472, :487, :577). No compute calls -> runs without a GPU."""
import ctypes as C
import re
import os

import pytest

from raytracingdenoiser_amd import api

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    lib = api.load_library()
    hdr = open(os.path.join(ROOT, "include", "NRD.h")).read()
    declared = re.findall(r"NRD_API [^;]*?(?:NRD_CALL )?(\w+)\(", hdr)
    assert sorted(declared) == sorted(api.NRD_SYMBOLS)
    hip_hdr = open(os.path.join(ROOT, "include", "NRDHip.h")).read()
    declared_hip = re.findall(r"\b(nrdHip\w+)\(", hip_hdr)
    assert sorted(set(declared_hip)) == sorted(api.NRD_HIP_SYMBOLS)
    for name in api.NRD_SYMBOLS + api.NRD_HIP_SYMBOLS:
        assert getattr(lib, name) is not None


def test_library_desc():
    lib = api.load_library()
    d = lib.GetLibraryDesc().contents
    assert (d.versionMajor, d.versionMinor, d.versionBuild) == (4, 14, 0)
    assert d.normalEncoding == 2 and d.roughnessEncoding == 1  # R10_G10_B10_A2_UNORM, LINEAR
    supported = {api.Denoiser(d.supportedDenoisers[i]) for i in range(d.supportedDenoisersNum)}
    assert {api.Denoiser.REFERENCE, api.Denoiser.REBLUR_DIFFUSE, api.Denoiser.REBLUR_DIFFUSE_SPECULAR, api.Denoiser.SIGMA_SHADOW} <= supported
    assert lib.GetDenoiserString(int(api.Denoiser.SIGMA_SHADOW)) == b"SIGMA_SHADOW"
    assert lib.GetResourceTypeString(int(api.ResourceType.IN_DIFF_RADIANCE_HITDIST)) == b"IN_DIFF_RADIANCE_HITDIST"
    assert lib.GetResourceTypeString(999) is None


def test_create_errors():
    for d in api.Denoiser:  # every denoiser of the reference can be instantiated ...
        if d != api.Denoiser.MAX_NUM:
            api.Instance([(0, d)])
    with pytest.raises(RuntimeError, match="UNSUPPORTED|INVALID_ARGUMENT"):  # ... and nothing else
        api.Instance([(0, api.Denoiser.MAX_NUM)])
    with pytest.raises(RuntimeError, match="NON_UNIQUE_IDENTIFIER"):
        api.Instance([(3, api.Denoiser.REFERENCE), (3, api.Denoiser.SIGMA_SHADOW)])


def _settings(w, h, **kw):
    cs = api.CommonSettings(resourceSize=(w, h), rectSize=(w, h), resourceSizePrev=(w, h), rectSizePrev=(w, h), timeDeltaBetweenFrames=16.667, **kw)
    for m in (cs.viewToClipMatrix, cs.viewToClipMatrixPrev, cs.worldToViewMatrix, cs.worldToViewMatrixPrev):
        for k in (0, 5, 10, 15):
            m[k] = 1.0
    # a D3D-style LH perspective projection so that DecomposeProjection sees clip.w = z
    for m in (cs.viewToClipMatrix, cs.viewToClipMatrixPrev):
        m[11], m[15], m[14] = 1.0, 0.0, -0.1
    return cs


def test_settings_and_dispatch_errors():
    inst = api.Instance([(1, api.Denoiser.REFERENCE)])
    assert inst.set_denoiser_settings(42, api.ReferenceSettings()) == api.Result.INVALID_ARGUMENT
    assert inst.set_denoiser_settings(1, api.ReferenceSettings()) == api.Result.SUCCESS
    bad = _settings(64, 64)
    bad.denoisingRange = -1.0
    assert inst.set_common_settings(bad) == api.Result.INVALID_ARGUMENT
    assert inst.set_common_settings(_settings(64, 64)) == api.Result.SUCCESS
    r, ds = inst.get_compute_dispatches([99])
    assert r == api.Result.INVALID_ARGUMENT and ds == []
    r, out, n = inst.get_compute_dispatches_raw([])
    assert r == api.Result.SUCCESS and n == 0


def test_first_frame_clears_then_continues():
    inst = api.Instance([(1, api.Denoiser.REFERENCE)])
    inst.set_common_settings(_settings(256, 256))  # first use is forced to CLEAR_AND_RESTART
    r, ds = inst.get_compute_dispatches()
    assert [d.shader for d in ds] == ["Clear_Float.cs", "Clear_Float.cs", "REFERENCE_TemporalAccumulation.cs", "REFERENCE_Copy.cs"]
    assert ds[0].grid == (16, 16) and ds[2].grid == (16, 16)
    inst.set_common_settings(_settings(256, 256))
    r, ds = inst.get_compute_dispatches()
    assert [d.shader for d in ds] == ["REFERENCE_TemporalAccumulation.cs", "REFERENCE_Copy.cs"]


def test_reblur_pool_layout_and_pass_indices():
    """Pool formats per reference Reblur_DiffuseSpecular.hpp:39-72; 42 B/px permanent + 28 B/px transient (SURVEY 8c KAT 7);
    default-settings pass selection per reference Reblur.cpp:104-210."""
    inst = api.Instance([(0, api.Denoiser.REBLUR_DIFFUSE_SPECULAR)])
    F = api.Format
    assert [f for f, _ in inst.permanent_pool] == [F.R32_SFLOAT, F.R10_G10_B10_A2_UNORM, F.R16_UINT, F.RGBA16_SFLOAT, F.R16_SFLOAT, F.R16_SFLOAT, F.R16_SFLOAT,
                                                   F.RGBA16_SFLOAT, F.R16_SFLOAT, F.R16_SFLOAT, F.R16_SFLOAT, F.R16_SFLOAT, F.R16_SFLOAT]
    assert inst.transient_pool == [(F.RG8_UNORM, 1), (F.R32_UINT, 1), (F.R16_SFLOAT, 1), (F.RGBA16_SFLOAT, 1), (F.R16_SFLOAT, 1), (F.RGBA16_SFLOAT, 1),
                                   (F.R16_SFLOAT, 1), (F.R8_UNORM, 16)]
    assert sum(api.FORMAT_BYTES[f] for f, _ in inst.permanent_pool) == 42
    assert sum(api.FORMAT_BYTES[f] for f, d in inst.transient_pool if d == 1) == 28

    inst.set_common_settings(_settings(2560, 1440))
    r, ds = inst.get_compute_dispatches()
    ds = [d for d in ds if not d.shader.startswith("Clear")]
    assert [d.shader for d in ds] == [
        "REBLUR_ClassifyTiles.cs", "REBLUR_DiffuseSpecular_PrePass.cs", "REBLUR_DiffuseSpecular_TemporalAccumulation.cs", "REBLUR_DiffuseSpecular_HistoryFix.cs",
        "REBLUR_DiffuseSpecular_Blur.cs", "REBLUR_DiffuseSpecular_PostBlur.cs", "REBLUR_DiffuseSpecular_TemporalStabilization.cs"]
    assert ds[0].grid == (160, 90) and all(d.grid == (320, 90) for d in ds[1:])  # 16x16 and 8x16 reference groups
    assert all(len(d.constants) == 832 for d in ds)
    # TS disabled -> the NoTemporalStabilization post-blur permutation, no TS pass
    rs = api.ReblurSettings(maxStabilizedFrameNum=0)
    inst.set_denoiser_settings(0, rs)
    inst.set_common_settings(_settings(2560, 1440))
    r, ds = inst.get_compute_dispatches()
    assert [d.shader for d in ds][-1] == "REBLUR_DiffuseSpecular_PostBlur_NoTemporalStabilization.cs"
    # split screen >= 1: passthrough only
    inst.set_common_settings(_settings(2560, 1440, splitScreen=1.0))
    r, ds = inst.get_compute_dispatches()
    assert [d.shader for d in ds] == ["REBLUR_DiffuseSpecular_SplitScreen.cs"]


def test_sigma_pass_selection():
    """reference Sigma.cpp:25-90: Copy + PostBlur(stabilised) + TS only when maxStabilizedFrameNum != 0."""
    inst = api.Instance([(5, api.Denoiser.SIGMA_SHADOW)])
    inst.set_common_settings(_settings(1920, 1080))
    r, ds = inst.get_compute_dispatches()
    names = [d.shader for d in ds if not d.shader.startswith("Clear")]
    assert names == ["SIGMA_Shadow_ClassifyTiles.cs", "SIGMA_SmoothTiles.cs", "SIGMA_Copy.cs", "SIGMA_Shadow_Blur.cs", "SIGMA_Shadow_PostBlur.cs",
                     "SIGMA_Shadow_TemporalStabilization.cs"]
    smooth = [d for d in ds if d.shader == "SIGMA_SmoothTiles.cs"][0]
    assert smooth.grid == (8, 5)  # ceil(ceil(1920/16)/16), ceil(ceil(1080/16)/16)
    assert all(len(d.constants) == 528 for d in ds if not d.shader.startswith("Clear"))  # 516 bytes of fields, rounded up to 16 like the reference host struct


def test_ping_pong_alternates():
    inst = api.Instance([(0, api.Denoiser.REBLUR_DIFFUSE)])
    seen = []
    for frame in range(3):
        inst.set_common_settings(_settings(128, 128, frameIndex=frame))
        r, ds = inst.get_compute_dispatches()
        ts = [d for d in ds if d.shader.endswith("TemporalStabilization.cs")][0]
        perm = [i for _, t, i in ts.resources if t == api.ResourceType.PERMANENT_POOL]
        seen.append((perm[2], perm[-1]))  # stabilized history read / written
    assert seen[0] != seen[1] and seen[0] == seen[2]
    assert seen[0] == (seen[1][1], seen[1][0])


def test_constant_arena_overflow_fails_the_whole_call():
    """The per-frame constant arena holds 128 KiB (reference InstanceImpl.h: CONSTANT_DATA_SIZE). An instance with more denoisers than that
    covers must fail GetComputeDispatches as a whole (no dispatch with a null / shared constant block is handed out) and keep working for
    a subset of its identifiers."""
    many = [(i + 1, api.Denoiser.REBLUR_DIFFUSE_SPECULAR) for i in range(24)]  # 24 x ~9 dispatches x 832 bytes > 128 KiB
    inst = api.Instance(many)
    assert inst.set_common_settings(_settings(128, 128)) == api.Result.SUCCESS
    r, out, n = inst.get_compute_dispatches_raw()
    assert r == api.Result.FAILURE and n == 0 and not out
    r, ds = inst.get_compute_dispatches([1, 2])
    assert r == api.Result.SUCCESS and len(ds) > 10 and all(len(d.constants) in (0, 832) for d in ds)


@pytest.mark.parametrize("field,value", [
    ("viewZScale", 0.0), ("viewZScale", -1.0), ("resourceSize", (0, 64)), ("resourceSizePrev", (64, 0)), ("rectSize", (0, 64)), ("rectSizePrev", (64, 0)),
    ("cameraJitter", (0.6, 0.0)), ("cameraJitter", (0.0, -0.51)), ("cameraJitterPrev", (-0.7, 0.0)), ("denoisingRange", 0.0),
    ("disocclusionThreshold", 0.0), ("disocclusionThresholdAlternate", -0.1),
])
def test_set_common_settings_rejects_what_the_reference_asserts(field, value):
    """reference InstanceImpl.cpp:300-337: each assert is a predicate of the Result (INVALID_ARGUMENT) here; the instance keeps working afterwards"""
    inst = api.Instance([(1, api.Denoiser.REBLUR_DIFFUSE)])
    assert inst.set_common_settings(_settings(64, 64)) == api.Result.SUCCESS  # the first use overwrites the *Prev fields with the current ones (reference :283-297)
    bad = _settings(64, 64)
    if isinstance(value, tuple):
        for i, v in enumerate(value):
            getattr(bad, field)[i] = v
    else:
        setattr(bad, field, value)
    assert inst.set_common_settings(bad) == api.Result.INVALID_ARGUMENT
    assert inst.set_common_settings(_settings(64, 64)) == api.Result.SUCCESS
    r, ds = inst.get_compute_dispatches()
    assert r == api.Result.SUCCESS and len(ds) > 0


def test_screen_space_motion_vectors_need_a_scale():
    """'mvScale.xy can't be 0' unless the motion vectors are in world space (reference InstanceImpl.cpp:315-316)"""
    inst = api.Instance([(1, api.Denoiser.REBLUR_DIFFUSE)])
    cs = _settings(64, 64)
    cs.isMotionVectorInWorldSpace = False
    cs.motionVectorScale[0], cs.motionVectorScale[1] = 0.0, 1.0
    assert inst.set_common_settings(cs) == api.Result.INVALID_ARGUMENT
    cs.isMotionVectorInWorldSpace = True
    assert inst.set_common_settings(cs) == api.Result.SUCCESS
