"""REBLUR chain: oracle known-answer properties on the CPU, HIP-vs-oracle parity on the GPU.
Known answers: SURVEY.md section 8c (2) constant radiance -> unchanged, (3) all-sky -> untouched, (4) splitScreen >= 1 ->
passthrough, (6) accumulated-frame counter 1, 2, 3, ... on a static scene."""
import numpy as np
import pytest
import torch

import parity
from raytracingdenoiser_amd import api

RT = api.ResourceType
W, H = 160, 96


def _run_oracle(name, seq, overrides=None, cs_kw=None):
    ora = parity.OracleRun(name, W, H)
    for f, frame in enumerate(seq):
        cs = parity.common_settings(frame["camera"], seq[max(f - 1, 0)]["camera"], W, H, f, **(cs_kw or {}))
        parity.tag_checkerboard(frame, overrides, f)
        ora.step(frame, cs, parity.denoiser_settings(name, frame, overrides))
    return ora


def test_oracle_energy_is_preserved_and_noise_drops():
    name = "REBLUR_DIFFUSE_SPECULAR"
    seq = parity.generate_sequence(name, W, H, 8)
    ora = _run_oracle(name, seq)
    m = ~seq[-1]["is_sky"].numpy()
    for rt, key in ((RT.OUT_DIFF_RADIANCE_HITDIST, "diff"), (RT.OUT_SPEC_RADIANCE_HITDIST, "spec")):
        out = ora.output(rt)
        noisy = seq[-1][key].float().numpy()
        assert not np.isnan(out).any()
        assert abs(out[m][:, 0].mean() - noisy[m][:, 0].mean()) < 0.03 * noisy[m][:, 0].mean()  # normalised filters keep the mean
        assert out[m][:, 0].std() < 0.9 * noisy[m][:, 0].std()
    always_sky = np.all(np.stack([fr["is_sky"].numpy() for fr in seq]), axis=0)
    assert np.all(ora.output(RT.OUT_DIFF_RADIANCE_HITDIST)[always_sky] == 0)  # sky pixels are never written (cleared on frame 0)


def test_oracle_constant_signal_is_a_fixed_point():
    name = "REBLUR_DIFFUSE_SPECULAR"
    seq = parity.generate_sequence(name, W, H, 5, static_camera=True, noise=False)
    const = torch.tensor([0.5, 0.0625, -0.03125, 0.25], dtype=torch.float16)
    for fr in seq:
        fr["diff"] = const.expand(H, W, 4).contiguous()
        fr["spec"] = const.expand(H, W, 4).contiguous()
    ora = _run_oracle(name, seq)
    m = ~seq[-1]["is_sky"].numpy()
    for rt in (RT.OUT_DIFF_RADIANCE_HITDIST, RT.OUT_SPEC_RADIANCE_HITDIST):
        out = ora.output(rt)[m]
        assert np.max(np.abs(out - const.float().numpy())) < 2e-3  # weighted means of a constant, up to fp16 storage rounding


def test_oracle_accumulated_frames_count_up():
    name = "REBLUR_DIFFUSE"
    seq = parity.generate_sequence(name, W, H, 6, static_camera=True, noise=False)
    ora = parity.OracleRun(name, W, H)
    m = ~seq[0]["is_sky"].numpy()
    for f, frame in enumerate(seq):
        cs = parity.common_settings(frame["camera"], seq[max(f - 1, 0)]["camera"], W, H, f)
        ora.step(frame, cs, parity.denoiser_settings(name, frame))
        raw, fmt, w = ora.ex.pool_plane(RT.PERMANENT_POOL, 2)  # PREV_INTERNAL_DATA (R16_UINT)
        accum = (raw[:, : w * 2].copy().view(np.uint16) & 63)[m]
        assert np.median(accum) == f + 1  # frame k stores min(k, maxAccumulatedFrameNum) + 1


def test_oracle_all_sky_and_split_screen_passthrough():
    name = "REBLUR_DIFFUSE_SPECULAR"
    seq = parity.generate_sequence(name, W, H, 2)
    for fr in seq:
        fr["viewz"] = torch.full_like(fr["viewz"], 1.0e6)
    ora = _run_oracle(name, seq)
    assert np.all(ora.output(RT.OUT_DIFF_RADIANCE_HITDIST) == 0) and np.all(ora.output(RT.OUT_SPEC_RADIANCE_HITDIST) == 0)

    seq = parity.generate_sequence(name, W, H, 2)
    ora = _run_oracle(name, seq, cs_kw=dict(splitScreen=1.0))
    m = ~seq[-1]["is_sky"].numpy()
    assert np.array_equal(ora.output(RT.OUT_DIFF_RADIANCE_HITDIST)[m], seq[-1]["diff"].float().numpy()[m])
    assert [d.shader for d in ora.last_dispatches] == ["REBLUR_DiffuseSpecular_SplitScreen.cs"]


def test_oracle_neutral_confidence_inputs_change_nothing():
    # confidence = 1 and disocclusion mix = 0 are the neutral elements of the optional guide inputs
    name = "REBLUR_DIFFUSE_SPECULAR"
    seq = parity.generate_sequence(name, W, H, 4, extra_want=("confidence",))
    for fr in seq:
        fr["diff_confidence"] = torch.full_like(fr["diff_confidence"], 255)
        fr["spec_confidence"] = torch.full_like(fr["spec_confidence"], 255)
        fr["disocclusion_mix"] = torch.zeros_like(fr["disocclusion_mix"])
    with_guides = _run_oracle(name, seq, cs_kw=dict(isHistoryConfidenceAvailable=True, isDisocclusionThresholdMixAvailable=True))
    without = _run_oracle(name, seq)
    for rt in without.outs:
        assert np.array_equal(with_guides.output(rt), without.output(rt))
    # and non-neutral guides do change the result
    seq2 = parity.generate_sequence(name, W, H, 4, extra_want=("confidence",))
    changed = _run_oracle(name, seq2, cs_kw=dict(isHistoryConfidenceAvailable=True, isDisocclusionThresholdMixAvailable=True))
    assert any(not np.array_equal(changed.output(rt), without.output(rt)) for rt in without.outs)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["REBLUR_DIFFUSE_SPECULAR", "REBLUR_DIFFUSE", "REBLUR_SPECULAR"])
def test_hip_matches_oracle(name):
    worst = parity.run_parity(name, width=192, height=128, frames=6, verbose=True)
    assert worst <= parity.REL_TOL


@pytest.mark.gpu
def test_hip_matches_oracle_odd_size_no_stabilization():
    # ragged edges (not multiples of 32 / 16 / 8) and the PostBlur_NoTemporalStabilization permutation
    worst = parity.run_parity("REBLUR_DIFFUSE_SPECULAR", width=211, height=117, frames=4, verbose=True, settings_overrides=dict(maxStabilizedFrameNum=0))
    assert worst <= parity.REL_TOL


@pytest.mark.gpu
def test_hip_matches_oracle_with_confidence_and_disocclusion_mix_inputs():
    worst = parity.run_parity("REBLUR_DIFFUSE_SPECULAR", width=160, height=96, frames=5, verbose=True, extra_want=("confidence",),
                              cs_kw=dict(isHistoryConfidenceAvailable=True, isDisocclusionThresholdMixAvailable=True))
    assert worst <= parity.REL_TOL


@pytest.mark.gpu
@pytest.mark.parametrize("name,mode,prepass", [("REBLUR_DIFFUSE_SPECULAR", 1, True), ("REBLUR_DIFFUSE_SPECULAR", 2, False), ("REBLUR_DIFFUSE", 2, True)])
def test_hip_matches_oracle_hit_distance_reconstruction(name, mode, prepass):
    # half of the input hit distances are missing; mode 1 = AREA_3X3, 2 = AREA_5X5; with and without the pre-pass behind it
    overrides = dict(hitDistanceReconstructionMode=mode)
    if not prepass:
        overrides.update(diffusePrepassBlurRadius=0.0, specularPrepassBlurRadius=0.0)
    worst = parity.run_parity(name, width=176, height=104, frames=3, verbose=True, extra_want=("holes",), settings_overrides=overrides)
    assert worst <= parity.REL_TOL


@pytest.mark.gpu
def test_hip_matches_oracle_antifirefly_and_no_prepass():
    worst = parity.run_parity("REBLUR_DIFFUSE_SPECULAR", width=160, height=96, frames=4, verbose=True,
                              settings_overrides=dict(enableAntiFirefly=True, diffusePrepassBlurRadius=0.0, specularPrepassBlurRadius=0.0))
    assert worst <= parity.REL_TOL


# ---------------------------------------------------------------------------------------------- performance mode
def test_oracle_performance_mode_selects_perf_permutations_and_still_denoises():
    """ReblurSettings::enablePerformanceMode -> "REBLUR_Perf_*" pipelines (reference Source/Reblur.cpp:148-190, REBLUR_Config.hlsli:196-238):
    a cheaper filter, not a different estimator -- same mean, noise still drops, result differs from the quality mode."""
    name = "REBLUR_DIFFUSE_SPECULAR"
    seq = parity.generate_sequence(name, W, H, 8)
    quality = _run_oracle(name, seq)
    perf = _run_oracle(name, seq, overrides=dict(enablePerformanceMode=True))
    shaders = [d.shader for d in perf.last_dispatches]
    assert shaders[0] == "REBLUR_ClassifyTiles.cs" and all(s.startswith("REBLUR_Perf_DiffuseSpecular_") for s in shaders[1:]) and len(shaders) == 7
    m = ~seq[-1]["is_sky"].numpy()
    for rt, key in ((RT.OUT_DIFF_RADIANCE_HITDIST, "diff"), (RT.OUT_SPEC_RADIANCE_HITDIST, "spec")):
        q, p, noisy = quality.output(rt)[m][:, 0], perf.output(rt)[m][:, 0], seq[-1][key].float().numpy()[m][:, 0]
        assert not np.array_equal(q, p)
        assert abs(p.mean() - noisy.mean()) < 0.03 * noisy.mean()
        assert p.std() < 0.9 * noisy.std()
        assert np.abs(p - q).mean() < 0.1 * q.mean()


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["REBLUR_DIFFUSE_SPECULAR", "REBLUR_DIFFUSE", "REBLUR_SPECULAR"])
def test_hip_matches_oracle_performance_mode(name):
    worst = parity.run_parity(name, width=192, height=128, frames=6, verbose=True, settings_overrides=dict(enablePerformanceMode=True))
    assert worst <= parity.REL_TOL


@pytest.mark.gpu
def test_hip_matches_oracle_performance_mode_options():
    # Perf permutations of the optional passes: 3x3 hit-distance reconstruction, anti-firefly (radius 3), no temporal stabilisation, odd size
    worst = parity.run_parity("REBLUR_DIFFUSE_SPECULAR", width=211, height=117, frames=4, verbose=True, extra_want=("holes",),
                              settings_overrides=dict(enablePerformanceMode=True, hitDistanceReconstructionMode=1, enableAntiFirefly=True, maxStabilizedFrameNum=0))
    assert worst <= parity.REL_TOL
    worst = parity.run_parity("REBLUR_DIFFUSE_SPECULAR", width=160, height=96, frames=3, verbose=True, extra_want=("holes",),
                              settings_overrides=dict(enablePerformanceMode=True, hitDistanceReconstructionMode=2, diffusePrepassBlurRadius=0.0, specularPrepassBlurRadius=0.0))
    assert worst <= parity.REL_TOL


# ---------------------------------------------------------------------------------------------- occlusion family
OCCLUSION = ["REBLUR_DIFFUSE_SPECULAR_OCCLUSION", "REBLUR_DIFFUSE_OCCLUSION", "REBLUR_SPECULAR_OCCLUSION"]


def test_oracle_occlusion_family_tables_and_denoising():
    """REBLUR_*_OCCLUSION (reference Denoisers/Reblur_*Occlusion.hpp, Reblur.cpp:212-296): hit distance only, R16_UNORM planes, no pre-pass, no
    temporal stabilisation; the history-fix permutation follows enableAntiFirefly (a reference quirk); diffuse output is independent of the
    specular signal; the filter keeps the mean and lowers the noise."""
    seq = parity.generate_sequence("REBLUR_DIFFUSE_SPECULAR_OCCLUSION", W, H, 6)
    ds = _run_oracle("REBLUR_DIFFUSE_SPECULAR_OCCLUSION", seq)
    assert [d.shader for d in ds.last_dispatches] == [
        "REBLUR_ClassifyTiles.cs", "REBLUR_DiffuseSpecularOcclusion_TemporalAccumulation.cs", "REBLUR_Perf_DiffuseSpecularOcclusion_HistoryFix.cs",
        "REBLUR_DiffuseSpecularOcclusion_Blur.cs", "REBLUR_DiffuseSpecularOcclusion_PostBlur_NoTemporalStabilization.cs"]
    assert [f for f, _ in ds.inst.permanent_pool] == [api.Format.R32_SFLOAT, api.Format.R10_G10_B10_A2_UNORM, api.Format.R16_UINT, api.Format.R16_UNORM, api.Format.R16_UNORM,
                                                      api.Format.R16_SFLOAT, api.Format.R16_SFLOAT]
    af = _run_oracle("REBLUR_DIFFUSE_SPECULAR_OCCLUSION", seq[:2], overrides=dict(enableAntiFirefly=True, enablePerformanceMode=True, hitDistanceReconstructionMode=2))
    assert [d.shader for d in af.last_dispatches][1:4] == ["REBLUR_Perf_DiffuseSpecularOcclusion_HitDistReconstruction_5x5.cs",
                                                           "REBLUR_Perf_DiffuseSpecularOcclusion_TemporalAccumulation.cs", "REBLUR_DiffuseSpecularOcclusion_HistoryFix.cs"]
    d_only = _run_oracle("REBLUR_DIFFUSE_OCCLUSION", seq)
    assert np.array_equal(d_only.output(RT.OUT_DIFF_HITDIST), ds.output(RT.OUT_DIFF_HITDIST))
    m = ~seq[-1]["is_sky"].numpy()
    for rt, key in ((RT.OUT_DIFF_HITDIST, "diff"), (RT.OUT_SPEC_HITDIST, "spec")):
        out, noisy = ds.output(rt)[m][:, 0] / 65535.0, seq[-1][key].float().numpy()[m][:, 3]
        assert abs(out.mean() - noisy.mean()) < 0.03 and out.std() < 0.8 * noisy.std()


def test_oracle_occlusion_constant_signal_is_a_fixed_point():
    seq = parity.generate_sequence("REBLUR_DIFFUSE_OCCLUSION", W, H, 4)
    for fr in seq:
        fr["diff"][..., 3] = 0.5
    out = _run_oracle("REBLUR_DIFFUSE_OCCLUSION", seq).output(RT.OUT_DIFF_HITDIST)[..., 0]
    m = ~np.any(np.stack([fr["is_sky"].numpy() for fr in seq]), axis=0)
    d = np.abs(out[m] - 32768.0)  # 0.5 in R16_UNORM; every pass of every frame re-quantises a normalised mean of equal values
    assert d.max() <= 8.0 and np.mean(d == 0.0) > 0.99


@pytest.mark.gpu
@pytest.mark.parametrize("name", OCCLUSION)
def test_hip_matches_oracle_occlusion(name):
    worst = parity.run_parity(name, width=192, height=128, frames=6, verbose=True)
    assert worst <= parity.REL_TOL


@pytest.mark.gpu
def test_hip_matches_oracle_occlusion_options():
    # odd size (user-plane pitch != pool pitch), performance mode, 3x3 reconstruction of missing hit distances, quality history-fix permutation
    worst = parity.run_parity("REBLUR_DIFFUSE_SPECULAR_OCCLUSION", width=211, height=117, frames=4, verbose=True, extra_want=("holes",),
                              settings_overrides=dict(enablePerformanceMode=True, hitDistanceReconstructionMode=1, enableAntiFirefly=True))
    assert worst <= parity.REL_TOL
    worst = parity.run_parity("REBLUR_SPECULAR_OCCLUSION", width=160, height=96, frames=4, verbose=True, extra_want=("holes", "confidence"),
                              settings_overrides=dict(hitDistanceReconstructionMode=2), cs_kw=dict(isHistoryConfidenceAvailable=True, isDisocclusionThresholdMixAvailable=True))
    assert worst <= parity.REL_TOL


# ---------------------------------------------------------------------------------------------- SH family
SH_FAMILY = ["REBLUR_DIFFUSE_SPECULAR_SH", "REBLUR_DIFFUSE_SH", "REBLUR_SPECULAR_SH"]


def test_oracle_sh_family_adds_a_plane_without_touching_sh0():
    """REBLUR_*_SH (reference Denoisers/Reblur_*Sh.hpp): SH0 = the non-SH texel, so OUT_*_SH0 must equal the output of the non-SH denoiser bit
    for bit; SH1 (direction * luma) is filtered with the same weights -- its length follows the denoised luma, its noise drops."""
    seq = parity.generate_sequence("REBLUR_DIFFUSE_SPECULAR_SH", W, H, 6)
    sh = _run_oracle("REBLUR_DIFFUSE_SPECULAR_SH", seq)
    plain = _run_oracle("REBLUR_DIFFUSE_SPECULAR", seq)
    shaders = [d.shader for d in sh.last_dispatches]
    assert shaders == ["REBLUR_ClassifyTiles.cs"] + ["REBLUR_DiffuseSpecularSh_%s.cs" % p for p in ("PrePass", "TemporalAccumulation", "HistoryFix", "Blur", "PostBlur", "TemporalStabilization")]
    assert len(sh.inst.permanent_pool) == len(plain.inst.permanent_pool) + 2 and len(sh.inst.transient_pool) == len(plain.inst.transient_pool) + 2
    assert np.array_equal(sh.output(RT.OUT_DIFF_SH0), plain.output(RT.OUT_DIFF_RADIANCE_HITDIST))
    assert np.array_equal(sh.output(RT.OUT_SPEC_SH0), plain.output(RT.OUT_SPEC_RADIANCE_HITDIST))
    m = ~seq[-1]["is_sky"].numpy()
    for rt, key in ((RT.OUT_DIFF_SH1, "diff_sh1"), (RT.OUT_SPEC_SH1, "spec_sh1")):
        out, noisy = sh.output(rt)[m][:, :3], seq[-1][key].float().numpy()[m][:, :3]
        assert not np.isnan(out).any() and out.std() < 0.9 * noisy.std()
    # hit-distance reconstruction reuses the radiance family's shader
    hd = _run_oracle("REBLUR_DIFFUSE_SH", seq[:2], overrides=dict(hitDistanceReconstructionMode=1))
    assert [d.shader for d in hd.last_dispatches][1] == "REBLUR_Diffuse_HitDistReconstruction.cs"


@pytest.mark.gpu
@pytest.mark.parametrize("name", SH_FAMILY)
def test_hip_matches_oracle_sh(name):
    worst = parity.run_parity(name, width=192, height=128, frames=6, verbose=True)
    assert worst <= parity.REL_TOL


@pytest.mark.gpu
def test_hip_matches_oracle_sh_options():
    # odd size, no temporal stabilisation (history copies of SH0 and SH1), performance mode, 5x5 hit-distance reconstruction, anti-firefly
    worst = parity.run_parity("REBLUR_DIFFUSE_SPECULAR_SH", width=211, height=117, frames=4, verbose=True, extra_want=("holes",),
                              settings_overrides=dict(maxStabilizedFrameNum=0, enablePerformanceMode=True, hitDistanceReconstructionMode=2, enableAntiFirefly=True))
    assert worst <= parity.REL_TOL
    worst = parity.run_parity("REBLUR_SPECULAR_SH", width=160, height=96, frames=3, verbose=True,
                              settings_overrides=dict(diffusePrepassBlurRadius=0.0, specularPrepassBlurRadius=0.0), cs_kw=dict(splitScreen=0.4))
    assert worst <= parity.REL_TOL


# ---------------------------------------------------------------------------------------------- directional occlusion
def test_oracle_directional_occlusion_tables_and_denoising():
    """REBLUR_DIFFUSE_DIRECTIONAL_OCCLUSION (reference Denoisers/Reblur_DiffuseDirectionalOcclusion.hpp): the diffuse chain on RGBA16_SNORM
    (direction * hit distance, hit distance) texels with an R16_UNORM fast history; hit-distance reconstruction and split screen run on the
    radiance family's pipelines."""
    name = "REBLUR_DIFFUSE_DIRECTIONAL_OCCLUSION"
    seq = parity.generate_sequence(name, W, H, 6)
    ora = _run_oracle(name, seq)
    assert [d.shader for d in ora.last_dispatches] == ["REBLUR_ClassifyTiles.cs"] + ["REBLUR_DiffuseDirectionalOcclusion_%s.cs" % p for p in
                                                       ("PrePass", "TemporalAccumulation", "HistoryFix", "Blur", "PostBlur", "TemporalStabilization")]
    assert [f for f, _ in ora.inst.permanent_pool][3:5] == [api.Format.RGBA16_SNORM, api.Format.R16_UNORM]
    hd = _run_oracle(name, seq[:2], overrides=dict(hitDistanceReconstructionMode=1), cs_kw=dict(splitScreen=0.3))
    shaders = [d.shader for d in hd.last_dispatches]
    assert shaders[1] == "REBLUR_Diffuse_HitDistReconstruction.cs" and shaders[-1] == "REBLUR_Diffuse_SplitScreen.cs"
    m = ~seq[-1]["is_sky"].numpy()
    out = ora.output(RT.OUT_DIFF_DIRECTION_HITDIST)[m] / 32767.0
    noisy = seq[-1]["diff_direction_hitdist"].numpy().astype(np.float32)[m] / 32767.0
    assert abs(out[:, 3].mean() - noisy[:, 3].mean()) < 0.03 and out[:, 3].std() < 0.7 * noisy[:, 3].std() and out[:, :3].std() < 0.8 * noisy[:, :3].std()


@pytest.mark.gpu
def test_hip_matches_oracle_directional_occlusion():
    name = "REBLUR_DIFFUSE_DIRECTIONAL_OCCLUSION"
    worst = parity.run_parity(name, width=192, height=128, frames=6, verbose=True)
    assert worst <= parity.REL_TOL
    # odd size, reconstruction + split screen through the radiance family's pipelines, no stabilisation, performance mode
    worst = parity.run_parity(name, width=211, height=117, frames=4, verbose=True, extra_want=("holes",), cs_kw=dict(splitScreen=0.3),
                              settings_overrides=dict(hitDistanceReconstructionMode=1, maxStabilizedFrameNum=0, enablePerformanceMode=True))
    assert worst <= parity.REL_TOL


# ---------------------------------------------------------------------------------------------------- checkerboard modes
def test_oracle_checkerboard_resolves_half_rate_inputs():
    """CheckerboardMode BLACK / WHITE (reference NRDSettings.h CheckerboardMode, REBLUR_PrePass.hlsli:43-108): only every other pixel of the noisy
    inputs is traced, packed into the left half of the plane. The pre-pass stays in the chain even with zero radii (reference Reblur.cpp:127-129),
    nothing reads the unused right half (filled with a sentinel), and the result stays close to the full-rate one."""
    name = "REBLUR_DIFFUSE_SPECULAR"
    seq = parity.generate_sequence(name, W, H, 6)
    full = _run_oracle(name, seq)
    m = ~seq[-1]["is_sky"].numpy()
    for mode in (api.CheckerboardMode.BLACK, api.CheckerboardMode.WHITE):
        for radii in (None, dict(diffusePrepassBlurRadius=0.0, specularPrepassBlurRadius=0.0)):
            ora = _run_oracle(name, seq, dict(checkerboardMode=int(mode), **(radii or {})))
            assert "REBLUR_DiffuseSpecular_PrePass.cs" in [d.shader for d in ora.last_dispatches]
            for rt in (RT.OUT_DIFF_RADIANCE_HITDIST, RT.OUT_SPEC_RADIANCE_HITDIST):
                out, ref = ora.output(rt), full.output(rt)
                assert not np.isnan(out).any() and out.max() < 8.0  # the sentinel is 17
                assert np.abs(out[m] - ref[m]).mean() < 0.12 * np.abs(ref[m]).mean()
    # split screen: the passthrough reads the packed column x >> 1
    ora = _run_oracle(name, seq[:1], dict(checkerboardMode=1), cs_kw=dict(splitScreen=1.0))
    packed = parity.checkerboard_pack(seq[0]["diff"], 0, 0).float().numpy()
    out = ora.output(RT.OUT_DIFF_RADIANCE_HITDIST)
    x = np.arange(W)
    keep = (seq[0]["viewz"].numpy() < 5e5)[..., None]
    assert np.array_equal(out, packed[:, x >> 1] * keep)


@pytest.mark.gpu
@pytest.mark.parametrize("name,mode,overrides", [
    ("REBLUR_DIFFUSE_SPECULAR", 1, None),
    ("REBLUR_DIFFUSE_SPECULAR", 2, dict(diffusePrepassBlurRadius=0.0, specularPrepassBlurRadius=0.0)),  # resolve-only pre-pass
    ("REBLUR_SPECULAR", 2, dict(enablePerformanceMode=True)),
    ("REBLUR_DIFFUSE_SPECULAR_SH", 1, None),
    ("REBLUR_DIFFUSE_SPECULAR_OCCLUSION", 1, None),   # no pre-pass: resolved inside temporal accumulation
    ("REBLUR_SPECULAR_OCCLUSION", 2, None),
    ("REBLUR_DIFFUSE_DIRECTIONAL_OCCLUSION", 2, None),
])
def test_hip_matches_oracle_checkerboard(name, mode, overrides):
    worst = parity.run_parity(name, width=178, height=101, frames=5, verbose=True, settings_overrides=dict(checkerboardMode=mode, **(overrides or {})))
    assert worst <= parity.REL_TOL


@pytest.mark.gpu
def test_hip_matches_oracle_checkerboard_split_screen():
    worst = parity.run_parity("REBLUR_DIFFUSE_SPECULAR", width=160, height=96, frames=3, verbose=True, settings_overrides=dict(checkerboardMode=1), cs_kw=dict(splitScreen=0.5))
    assert worst <= parity.REL_TOL


# ---------------------------------------------------------------------------------------------------- motion vector conventions
def _mv_settings(z_scale, **kw):
    return dict(isMotionVectorInWorldSpace=False, motionVectorScale=(1.0 / W, 1.0 / H, z_scale), **kw)


@pytest.mark.parametrize("z_scale", [1.0, 0.0])
def test_oracle_screen_space_motion_vectors_agree_with_world_space_ones(z_scale):
    """The scene is static, so "world-space MVs scaled by 0" and true 2D / 2.5D screen-space MVs (CommonSettings::motionVectorScale, NRDSettings.h) describe
    the same motion: both reprojection paths of temporal accumulation / stabilization must land on (nearly: the MVs are fp16) the same result."""
    name = "REBLUR_DIFFUSE_SPECULAR"
    a = _run_oracle(name, parity.generate_sequence(name, W, H, 6))
    b = _run_oracle(name, parity.generate_sequence(name, W, H, 6, extra_want=("mv2d",)), cs_kw=_mv_settings(z_scale))
    for rt in (RT.OUT_DIFF_RADIANCE_HITDIST, RT.OUT_SPEC_RADIANCE_HITDIST):
        ref, out = a.output(rt), b.output(rt)
        assert np.mean(ref == out) > 0.9 and np.abs(ref - out).mean() < 1e-4 * np.abs(ref).mean() + 1e-5


def test_oracle_specular_mv_modification_rewrites_in_mv():
    """CommonSettings::isBaseColorMetalnessAvailable (reference REBLUR_TemporalStabilization.hlsli:250-285, Reblur.cpp:190, :359): on mostly-specular
    surfaces temporal stabilization blends IN_MV towards the motion of the reflection; the denoised outputs do not depend on it."""
    name = "REBLUR_DIFFUSE_SPECULAR"
    seq = parity.generate_sequence(name, W, H, 4, extra_want=("mv2d", "basecolor"))
    plain = _run_oracle(name, seq, cs_kw=_mv_settings(1.0))
    assert np.array_equal(plain.inputs[RT.IN_MV], seq[-1]["mv"].numpy())  # off: IN_MV untouched
    mod = _run_oracle(name, seq, cs_kw=_mv_settings(1.0, isBaseColorMetalnessAvailable=True))
    assert [d.shader for d in mod.last_dispatches][-1] == "REBLUR_DiffuseSpecular_TemporalStabilization.cs"
    src, out = seq[-1]["mv"].float().numpy(), mod.inputs[RT.IN_MV].astype(np.float32)
    changed = np.any(src != out, axis=-1)
    metal = seq[-1]["basecolor_metalness"].numpy()[..., 3] == 255
    assert not np.isnan(out).any() and changed.mean() > 0.1
    assert changed[metal & ~seq[-1]["is_sky"].numpy()].mean() > changed[~metal].mean()  # metals are all-specular
    assert np.array_equal(out[..., 3], src[..., 3])
    for rt in (RT.OUT_DIFF_RADIANCE_HITDIST, RT.OUT_SPEC_RADIANCE_HITDIST):
        assert np.array_equal(plain.output(rt), mod.output(rt))


@pytest.mark.gpu
@pytest.mark.parametrize("name,z_scale", [("REBLUR_DIFFUSE_SPECULAR", 1.0), ("REBLUR_DIFFUSE_SPECULAR", 0.0), ("REBLUR_SPECULAR_OCCLUSION", 1.0), ("REBLUR_DIFFUSE_SH", 0.0)])
def test_hip_matches_oracle_screen_space_motion_vectors(name, z_scale):
    worst = parity.run_parity(name, width=176, height=104, frames=5, verbose=True, extra_want=("mv2d",),
                              cs_kw=dict(isMotionVectorInWorldSpace=False, motionVectorScale=(1.0 / 176, 1.0 / 104, z_scale)))
    assert worst <= parity.REL_TOL


@pytest.mark.gpu
@pytest.mark.parametrize("name,world_space,z_scale", [("REBLUR_DIFFUSE_SPECULAR", False, 1.0), ("REBLUR_DIFFUSE_SPECULAR", False, 0.0), ("REBLUR_SPECULAR", True, 1.0),
                                                      ("REBLUR_DIFFUSE_SPECULAR_SH", False, 1.0)])
def test_hip_matches_oracle_specular_mv_modification(name, world_space, z_scale):
    # the in/out IN_MV plane is compared as well (parity.run_parity)
    w, h = 176, 104
    cs_kw = dict(isBaseColorMetalnessAvailable=True, isMotionVectorInWorldSpace=world_space, motionVectorScale=(1.0, 1.0, 1.0) if world_space else (1.0 / w, 1.0 / h, z_scale))
    worst = parity.run_parity(name, width=w, height=h, frames=4, verbose=True, extra_want=("basecolor",) if world_space else ("mv2d", "basecolor"), cs_kw=cs_kw,
                              settings_overrides=dict(enablePerformanceMode=True) if name == "REBLUR_SPECULAR" else None)
    assert worst <= parity.REL_TOL


def test_oracle_material_ids_separate_the_filters():
    """with material IDs in the G-buffer and minMaterialFor* below 3 the passes stop blending across material borders (reference CompareMaterials,
    REBLUR_Common.hlsli): the result must differ from the run that ignores them, and pixels deep inside one material band must not"""
    w, h, frames = 96, 64, 3
    seq = [parity.synth.render_frame(w, h, f, want=tuple(parity.DENOISERS["REBLUR_DIFFUSE_SPECULAR"][1]) + ("materials",)) for f in range(frames)]
    outs = []
    for over in (dict(minMaterialForDiffuse=0.0, minMaterialForSpecular=1.0), None):
        ora = parity.OracleRun("REBLUR_DIFFUSE_SPECULAR", w, h)
        for f, frame in enumerate(seq):
            ora.step(frame, parity.common_settings(frame["camera"], seq[max(f - 1, 0)]["camera"], w, h, f), parity.denoiser_settings("REBLUR_DIFFUSE_SPECULAR", frame, over))
        outs.append(ora.output(RT.OUT_DIFF_RADIANCE_HITDIST).copy())
    assert np.isfinite(outs[0]).all() and not np.array_equal(outs[0], outs[1])


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["REBLUR_DIFFUSE_SPECULAR", "REBLUR_DIFFUSE", "REBLUR_SPECULAR_OCCLUSION", "REBLUR_DIFFUSE_SPECULAR_SH"])
def test_hip_matches_oracle_material_ids(name):
    """material tests on (minMaterialFor* < 3) + the two special material IDs of CommonSettings: the full-rect tap variant WITH material tests (FR 1),
    CompareMaterials in temporal accumulation / history fix, the strand-material disocclusion threshold and the camera-attached reflection material"""
    worst = parity.run_parity(name, width=160, height=96, frames=4, extra_want=("materials",), settings_overrides=dict(minMaterialForDiffuse=0.0, minMaterialForSpecular=1.0),
                              cs_kw=dict(strandMaterialID=1.0, cameraAttachedReflectionMaterialID=2.0))
    assert worst == 0.0
