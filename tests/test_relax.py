"""RELAX chain: host-table checks and oracle known-answer properties on the CPU, HIP-vs-oracle parity on the GPU.
Known answers: SURVEY.md section 8c (2) constant radiance -> unchanged, (3) all-sky -> untouched, (4) splitScreen >= 1 ->
passthrough, (6) history length 1, 2, 3, ... on a static scene; a-trous iteration count and ping-pong binding order follow
reference Source/Relax.cpp:262-276."""
import numpy as np
import pytest
import torch

import parity
from raytracingdenoiser_amd import api

RT = api.ResourceType
W, H = 160, 96
ALL = ["RELAX_DIFFUSE", "RELAX_DIFFUSE_SH", "RELAX_SPECULAR", "RELAX_SPECULAR_SH", "RELAX_DIFFUSE_SPECULAR", "RELAX_DIFFUSE_SPECULAR_SH"]


def _run_oracle(name, seq, overrides=None, cs_kw=None, width=W, height=H):
    ora = parity.OracleRun(name, width, height)
    for f, frame in enumerate(seq):
        cs = parity.common_settings(frame["camera"], seq[max(f - 1, 0)]["camera"], width, height, f, **(cs_kw or {}))
        parity.tag_checkerboard(frame, overrides, f)
        ora.step(frame, cs, parity.denoiser_settings(name, frame, overrides))
    return ora


# ---------------------------------------------------------------------------------------------------- host tables
def test_pool_layout_matches_reference_tables():
    # reference Source/Denoisers/Relax_DiffuseSpecularSh.hpp:20-78 / Relax_Diffuse.hpp:18-46
    F = api.Format
    inst = api.Instance([(0, api.Denoiser.RELAX_DIFFUSE_SPECULAR_SH)])
    assert [p[0] for p in inst.permanent_pool] == [F.RGBA16_SFLOAT] * 8 + [F.R16_SFLOAT, F.R16_SFLOAT, F.R8_UNORM, F.RGBA8_UNORM, F.R8_UNORM, F.R32_SFLOAT]
    assert inst.transient_pool == [(F.RGBA16_SFLOAT, 1)] * 8 + [(F.R8_UNORM, 1), (F.R8_UNORM, 16), (F.R8_UNORM, 1)]
    inst = api.Instance([(0, api.Denoiser.RELAX_DIFFUSE)])
    assert [p[0] for p in inst.permanent_pool] == [F.RGBA16_SFLOAT] * 2 + [F.R8_UNORM, F.RGBA8_UNORM, F.R8_UNORM, F.R32_SFLOAT]
    assert inst.transient_pool == [(F.RGBA16_SFLOAT, 1)] * 2 + [(F.R8_UNORM, 16), (F.R8_UNORM, 1)]
    inst = api.Instance([(0, api.Denoiser.RELAX_SPECULAR)])
    assert len(inst.permanent_pool) == 8 and len(inst.transient_pool) == 5


@pytest.mark.parametrize("iterations", [2, 3, 5, 8])
def test_atrous_chain_ping_pongs_into_the_user_outputs(iterations):
    name = "RELAX_DIFFUSE_SPECULAR"
    seq = parity.generate_sequence(name, 64, 48, 1)
    inst = api.Instance([(0, parity.DENOISERS[name][0])])
    assert inst.set_denoiser_settings(0, api.RelaxSettings(atrousIterationNum=iterations)) == api.Result.SUCCESS
    assert inst.set_common_settings(parity.common_settings(seq[0]["camera"], seq[0]["camera"], 64, 48, 1, accumulationMode=0)) == api.Result.SUCCESS
    r, ds = inst.get_compute_dispatches()
    assert r == api.Result.SUCCESS
    atrous = [d for d in ds if "Atrous" in d.shader]
    assert [d.shader for d in atrous] == ["RELAX_DiffuseSpecular_AtrousSmem.cs"] + ["RELAX_DiffuseSpecular_Atrous.cs"] * (iterations - 1)
    # each iteration reads what the previous one wrote (resources 1, 2 = spec, diff inputs; outputs follow the 8 inputs)
    is_output = lambda r: r[0] == api.DescriptorType.STORAGE_TEXTURE
    for prev, cur in zip(atrous, atrous[1:]):
        prev_outs = [r[1:] for r in prev.resources if is_output(r)][:2]
        assert [cur.resources[1][1:], cur.resources[2][1:]] == prev_outs
    last_outs = [r[1] for r in atrous[-1].resources if is_output(r)]
    assert last_outs == [RT.OUT_SPEC_RADIANCE_HITDIST, RT.OUT_DIFF_RADIANCE_HITDIST]
    steps = [np.frombuffer(d.constants, dtype=np.uint32)[176:178].tolist() for d in atrous]
    assert steps == [[1 << i, 1 if i == iterations - 1 else 0] for i in range(iterations)]


def test_constant_block_sizes_and_frustum_basis():
    name = "RELAX_DIFFUSE_SPECULAR_SH"
    seq = parity.generate_sequence(name, 64, 48, 1)
    inst = api.Instance([(0, parity.DENOISERS[name][0])])
    cam = seq[0]["camera"]
    assert inst.set_common_settings(parity.common_settings(cam, cam, 64, 48, 0)) == api.Result.SUCCESS
    r, ds = inst.get_compute_dispatches()
    by_shader = {d.shader: d for d in ds}
    assert len(by_shader["RELAX_DiffuseSpecularSh_PrePass.cs"].constants) == 704
    assert len(by_shader["RELAX_DiffuseSpecularSh_Atrous.cs"].constants) == 720  # 712 bytes of fields, rounded up to 16 like the reference host struct
    c = np.frombuffer(by_shader["RELAX_ClassifyTiles.cs"].constants, dtype=np.float32)
    right, up, fwd = c[68:71], c[72:75], c[76:79]  # gFrustumRight / Up / Forward after 4 matrices + gRotatorPre
    # X = viewZ * (forward + right * clipX - up * clipY) must reproduce the camera rays of the synthetic scene
    for (u, v) in ((0.5, 0.5), (0.1, 0.9), (1.0, 0.0)):
        clip = (2.0 * u - 1.0, 2.0 * v - 1.0)
        x = fwd + right * clip[0] - up * clip[1]
        d = (clip[0] / cam.fx) * np.array(cam.right) + (-clip[1] / cam.fy) * np.array(cam.up) + np.array(cam.fwd)
        assert np.allclose(x, d, atol=1e-5)


# ---------------------------------------------------------------------------------------------------- oracle known answers
@pytest.mark.parametrize("name", ["RELAX_DIFFUSE_SPECULAR", "RELAX_DIFFUSE_SPECULAR_SH"])
def test_oracle_noise_drops_and_energy_is_preserved(name):
    seq = parity.generate_sequence(name, W, H, 8)
    ora = _run_oracle(name, seq)
    m = ~seq[-1]["is_sky"].numpy()
    sh = name.endswith("_SH")
    for sig in ("diff", "spec"):
        rt = getattr(RT, "OUT_%s_%s" % (sig.upper(), "SH0" if sh else "RADIANCE_HITDIST"))
        out = ora.output(rt)
        noisy = seq[-1][sig + "_relax"].float().numpy()
        assert not np.isnan(out).any()
        luma = lambda a: a[..., 0] * 0.25 + a[..., 1] * 0.5 + a[..., 2] * 0.25  # Y of YCoCg
        out_y = out[..., 0] if sh else luma(out)  # SH outputs are already YCoCg
        assert abs(out_y[m].mean() - luma(noisy)[m].mean()) < 0.06 * luma(noisy)[m].mean()
        assert out_y[m].std() < 0.9 * luma(noisy)[m].std()
    always_sky = np.all(np.stack([fr["is_sky"].numpy() for fr in seq]), axis=0)
    assert np.all(ora.output(rt)[always_sky] == 0)  # sky pixels are never written (cleared on frame 0)


def test_oracle_single_signal_variants_equal_the_combined_one():
    seq = parity.generate_sequence("RELAX_DIFFUSE_SPECULAR_SH", W, H, 4)
    both = _run_oracle("RELAX_DIFFUSE_SPECULAR_SH", seq)
    diff = _run_oracle("RELAX_DIFFUSE_SH", seq)
    spec = _run_oracle("RELAX_SPECULAR_SH", seq)
    for rt in (RT.OUT_DIFF_SH0, RT.OUT_DIFF_SH1):
        assert np.array_equal(both.output(rt), diff.output(rt))
    for rt in (RT.OUT_SPEC_SH0, RT.OUT_SPEC_SH1):
        assert np.array_equal(both.output(rt), spec.output(rt))


def test_oracle_constant_signal_is_a_fixed_point():
    name = "RELAX_DIFFUSE_SPECULAR"
    seq = parity.generate_sequence(name, W, H, 5, static_camera=True, noise=False)
    const = torch.tensor([0.5, 0.25, 0.125, 2.0], dtype=torch.float16)
    for fr in seq:
        fr["diff_relax"] = const.expand(H, W, 4).contiguous()
        fr["spec_relax"] = const.expand(H, W, 4).contiguous()
    ora = _run_oracle(name, seq)
    m = ~seq[-1]["is_sky"].numpy()
    for rt in (RT.OUT_DIFF_RADIANCE_HITDIST, RT.OUT_SPEC_RADIANCE_HITDIST):
        out = ora.output(rt)[m]
        assert np.max(np.abs(out[:, :3] - const.float().numpy()[:3])) < 2e-3  # weighted means of a constant, up to fp16 rounding
        assert np.max(out[:, 3]) < 1e-3  # .w carries the luminance variance: zero for a constant signal


def test_oracle_history_length_counts_up():
    name = "RELAX_DIFFUSE"
    seq = parity.generate_sequence(name, W, H, 6, static_camera=True, noise=False)
    ora = parity.OracleRun(name, W, H)
    m = ~seq[0]["is_sky"].numpy()
    m[:2], m[-2:], m[:, :2], m[:, -2:] = False, False, False, False
    for f, frame in enumerate(seq):
        cs = parity.common_settings(frame["camera"], seq[max(f - 1, 0)]["camera"], W, H, f)
        ora.step(frame, cs, parity.denoiser_settings(name, frame))
        raw, fmt, w = ora.ex.pool_plane(RT.PERMANENT_POOL, 2)  # HISTORY_LENGTH_PREV (R8_UNORM)
        assert np.median(raw[:, :w][m]) == f + 1


def test_oracle_all_sky_and_split_screen_passthrough():
    name = "RELAX_DIFFUSE_SPECULAR"
    seq = parity.generate_sequence(name, W, H, 2)
    for fr in seq:
        fr["viewz"] = torch.full_like(fr["viewz"], 1.0e6)
    ora = _run_oracle(name, seq)
    assert np.all(ora.output(RT.OUT_DIFF_RADIANCE_HITDIST) == 0) and np.all(ora.output(RT.OUT_SPEC_RADIANCE_HITDIST) == 0)

    seq = parity.generate_sequence(name, W, H, 2)
    ora = _run_oracle(name, seq, cs_kw=dict(splitScreen=1.0))
    m = ~seq[-1]["is_sky"].numpy()
    assert np.array_equal(ora.output(RT.OUT_DIFF_RADIANCE_HITDIST)[m], seq[-1]["diff_relax"].float().numpy()[m])
    assert [d.shader for d in ora.last_dispatches] == ["RELAX_DiffuseSpecular_SplitScreen.cs"]


def test_oracle_neutral_confidence_inputs_change_nothing():
    # confidence = 1 and disocclusion mix = 0 are the neutral elements of the optional guide inputs
    name = "RELAX_DIFFUSE_SPECULAR"
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


def test_oracle_antifirefly_removes_an_isolated_outlier():
    name = "RELAX_DIFFUSE"
    seq = parity.generate_sequence(name, W, H, 3, static_camera=True, noise=False)
    m = ~seq[0]["is_sky"].numpy()
    ys, xs = np.nonzero(m[8:-8, 8:-8])
    y, x = int(ys[len(ys) // 2]) + 8, int(xs[len(xs) // 2]) + 8
    assert m[y - 2 : y + 3, x - 2 : x + 3].all()
    const = torch.tensor([0.5, 0.5, 0.5, 2.0], dtype=torch.float16)
    for fr in seq:
        fr["diff_relax"] = const.expand(H, W, 4).clone()
        fr["diff_relax"][y, x, :3] = 200.0  # a firefly in every frame
    overrides = dict(diffusePrepassBlurRadius=0.0, atrousIterationNum=2)
    plain = _run_oracle(name, seq, overrides=overrides).output(RT.OUT_DIFF_RADIANCE_HITDIST)
    filtered = _run_oracle(name, seq, overrides=dict(enableAntiFirefly=True, **overrides)).output(RT.OUT_DIFF_RADIANCE_HITDIST)
    assert plain[y, x, 0] > 5.0  # the outlier survives temporal accumulation + 2 a-trous iterations ...
    assert abs(filtered[y, x, 0] - 0.5) < 0.05  # ... but not the rank-selection filter
    far = np.ones_like(m)
    far[max(y - 40, 0) : y + 41, max(x - 40, 0) : x + 41] = False  # history fix (stride <= 7) and the a-trous taps spread the outlier
    assert far[m].any() and np.array_equal(plain[far & m], filtered[far & m])


# ---------------------------------------------------------------------------------------------------- HIP parity
@pytest.mark.gpu
@pytest.mark.parametrize("name", ALL)
def test_hip_matches_oracle(name):
    worst = parity.run_parity(name, width=192, height=128, frames=6, verbose=True)
    assert worst <= parity.REL_TOL


@pytest.mark.gpu
def test_hip_matches_oracle_odd_size_and_more_iterations():
    # ragged edges (not multiples of 32 / 16 / 8), 7 a-trous iterations (random tap offsets at steps 8..64)
    worst = parity.run_parity("RELAX_DIFFUSE_SPECULAR_SH", width=211, height=117, frames=4, verbose=True, settings_overrides=dict(atrousIterationNum=7))
    assert worst <= parity.REL_TOL


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["RELAX_DIFFUSE_SPECULAR", "RELAX_DIFFUSE_SPECULAR_SH"])
def test_hip_matches_oracle_with_confidence_and_disocclusion_mix_inputs(name):
    # IN_DIFF/SPEC_CONFIDENCE shorten the accumulation and relax the a-trous weights, IN_DISOCCLUSION_THRESHOLD_MIX switches thresholds
    worst = parity.run_parity(name, width=160, height=96, frames=5, verbose=True, extra_want=("confidence",),
                              cs_kw=dict(isHistoryConfidenceAvailable=True, isDisocclusionThresholdMixAvailable=True),
                              settings_overrides=dict(confidenceDrivenRelaxationMultiplier=1.0, confidenceDrivenLuminanceEdgeStoppingRelaxation=0.5, confidenceDrivenNormalEdgeStoppingRelaxation=0.5))
    assert worst <= parity.REL_TOL


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["RELAX_DIFFUSE_SPECULAR", "RELAX_SPECULAR_SH"])
def test_hip_matches_oracle_antifirefly(name):
    worst = parity.run_parity(name, width=176, height=104, frames=4, verbose=True, settings_overrides=dict(enableAntiFirefly=True))
    assert worst <= parity.REL_TOL


@pytest.mark.gpu
@pytest.mark.parametrize("name,mode", [("RELAX_DIFFUSE_SPECULAR", 1), ("RELAX_DIFFUSE_SPECULAR_SH", 2), ("RELAX_SPECULAR", 2)])
def test_hip_matches_oracle_hit_distance_reconstruction(name, mode):
    # half of the input hit distances are missing; mode 1 = AREA_3X3, 2 = AREA_5X5
    worst = parity.run_parity(name, width=176, height=104, frames=3, verbose=True, extra_want=("holes",), settings_overrides=dict(hitDistanceReconstructionMode=mode))
    assert worst <= parity.REL_TOL


@pytest.mark.gpu
def test_hip_matches_oracle_no_prepass_no_roughness_edge_stopping():
    worst = parity.run_parity("RELAX_DIFFUSE_SPECULAR", width=160, height=96, frames=4, verbose=True,
                              settings_overrides=dict(diffusePrepassBlurRadius=0.0, specularPrepassBlurRadius=0.0, enableRoughnessEdgeStopping=False, historyFixFrameNum=0,
                                                      spatialVarianceEstimationHistoryThreshold=0))
    assert worst <= parity.REL_TOL


# ---------------------------------------------------------------------------------------------------- checkerboard modes
def test_oracle_checkerboard_resolves_half_rate_inputs():
    """RelaxSettings::checkerboardMode (reference RELAX_PrePass.hlsli:29-60, RELAX_TemporalAccumulation.hlsli:577-606): every other pixel of the noisy
    inputs is traced and packed into the left half of the plane; the unused right half (a sentinel here) is never read."""
    name = "RELAX_DIFFUSE_SPECULAR"
    seq = parity.generate_sequence(name, W, H, 6)
    full = _run_oracle(name, seq)
    m = ~seq[-1]["is_sky"].numpy()
    for mode in (api.CheckerboardMode.BLACK, api.CheckerboardMode.WHITE):
        ora = _run_oracle(name, seq, dict(checkerboardMode=int(mode)))
        for rt in (RT.OUT_DIFF_RADIANCE_HITDIST, RT.OUT_SPEC_RADIANCE_HITDIST):
            out, ref = ora.output(rt), full.output(rt)
            assert not np.isnan(out).any() and out[..., :3].max() < 8.0  # the sentinel is 17
            assert np.abs(out[m][:, :3] - ref[m][:, :3]).mean() < 0.15 * np.abs(ref[m][:, :3]).mean()


@pytest.mark.gpu
@pytest.mark.parametrize("name,mode,overrides", [
    ("RELAX_DIFFUSE_SPECULAR", 1, None),
    ("RELAX_DIFFUSE_SPECULAR_SH", 2, None),
    ("RELAX_SPECULAR", 1, dict(diffusePrepassBlurRadius=0.0, specularPrepassBlurRadius=0.0)),  # resolve-only pre-pass
    ("RELAX_DIFFUSE_SH", 2, None),
])
def test_hip_matches_oracle_checkerboard(name, mode, overrides):
    worst = parity.run_parity(name, width=178, height=101, frames=5, verbose=True, settings_overrides=dict(checkerboardMode=mode, **(overrides or {})))
    assert worst <= parity.REL_TOL


@pytest.mark.gpu
def test_hip_matches_oracle_checkerboard_split_screen():
    worst = parity.run_parity("RELAX_DIFFUSE_SPECULAR_SH", width=160, height=96, frames=3, verbose=True, settings_overrides=dict(checkerboardMode=2), cs_kw=dict(splitScreen=0.5))
    assert worst <= parity.REL_TOL


# ---------------------------------------------------------------------------------------------------- motion vector conventions
@pytest.mark.parametrize("z_scale", [1.0, 0.0])
def test_oracle_screen_space_motion_vectors_agree_with_world_space_ones(z_scale):
    """static scene: true 2D / 2.5D screen-space MVs and "world-space MVs scaled by 0" describe the same motion (see tests/test_reblur.py)"""
    name = "RELAX_DIFFUSE_SPECULAR"
    a = _run_oracle(name, parity.generate_sequence(name, W, H, 6))
    b = _run_oracle(name, parity.generate_sequence(name, W, H, 6, extra_want=("mv2d",)), cs_kw=dict(isMotionVectorInWorldSpace=False, motionVectorScale=(1.0 / W, 1.0 / H, z_scale)))
    for rt in (RT.OUT_DIFF_RADIANCE_HITDIST, RT.OUT_SPEC_RADIANCE_HITDIST):
        ref, out = a.output(rt), b.output(rt)
        assert np.mean(ref == out) > 0.9 and np.abs(ref - out).mean() < 1e-4 * np.abs(ref).mean() + 1e-5


@pytest.mark.gpu
@pytest.mark.parametrize("name,z_scale", [("RELAX_DIFFUSE_SPECULAR", 1.0), ("RELAX_DIFFUSE_SPECULAR_SH", 0.0), ("RELAX_SPECULAR", 0.0)])
def test_hip_matches_oracle_screen_space_motion_vectors(name, z_scale):
    worst = parity.run_parity(name, width=176, height=104, frames=5, verbose=True, extra_want=("mv2d",),
                              cs_kw=dict(isMotionVectorInWorldSpace=False, motionVectorScale=(1.0 / 176, 1.0 / 104, z_scale)))
    assert worst <= parity.REL_TOL


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["RELAX_DIFFUSE_SPECULAR", "RELAX_DIFFUSE_SPECULAR_SH", "RELAX_SPECULAR"])
def test_hip_matches_oracle_material_ids(name):
    """material tests on (minMaterialFor* < 3): CompareMaterials in the pre-pass, temporal accumulation, history fix and the a-trous chain"""
    worst = parity.run_parity(name, width=160, height=96, frames=4, extra_want=("materials",), settings_overrides=dict(minMaterialForDiffuse=0.0, minMaterialForSpecular=1.0))
    assert worst == 0.0
