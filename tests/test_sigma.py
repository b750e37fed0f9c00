"""SIGMA_SHADOW (BASELINE.json configs[1]): oracle known answers on the CPU, HIP-vs-oracle parity on the GPU.
Known answers: fully lit input -> shadow 1, fully shadowed -> 0; the denoised penumbra is smoother
than the 1-spp visibility; sky is untouched."""
import numpy as np
import pytest
import torch

import parity
from raytracingdenoiser_amd import api

RT = api.ResourceType
W, H = 160, 96


def _run(seq, frames=None):
    ora = parity.OracleRun("SIGMA_SHADOW", W, H)
    for f, frame in enumerate(seq):
        cs = parity.common_settings(frame["camera"], seq[max(f - 1, 0)]["camera"], W, H, f)
        ora.step(frame, cs, parity.denoiser_settings("SIGMA_SHADOW", frame))
    return ora


def test_oracle_lit_and_umbra_are_fixed_points():
    seq = parity.generate_sequence("SIGMA_SHADOW", W, H, 3)
    m = ~seq[-1]["is_sky"].numpy()
    for value, want in ((65504.0, 255), (0.0, 0)):
        s2 = [dict(fr) for fr in seq]
        for fr in s2:
            fr["penumbra"] = torch.full((H, W), value, dtype=torch.float16)
        out = _run(s2).output(RT.OUT_SHADOW_TRANSLUCENCY)[..., 0]
        assert np.all(out[m] == want)


def test_oracle_denoises_penumbra():
    seq = parity.generate_sequence("SIGMA_SHADOW", W, H, 6)
    ora = _run(seq)
    out = ora.output(RT.OUT_SHADOW_TRANSLUCENCY)[..., 0] / 255.0
    shadow = out * out  # SIGMA_BackEnd_UnpackShadow
    fr = seq[-1]
    m = ~fr["is_sky"].numpy()
    noisy = (fr["penumbra"].float().numpy() >= 65504.0).astype(np.float32)
    pen = (fr["penumbra"].float().numpy() > 0) & (fr["penumbra"].float().numpy() < 65504.0) & m
    assert pen.sum() > 50  # the scene has penumbra pixels
    # inside penumbrae the 1-spp visibility is binary; the denoised one has intermediate values
    assert np.mean((shadow[pen] > 0.05) & (shadow[pen] < 0.95)) > 0.15
    assert abs(shadow[m].mean() - noisy[m].mean()) < 0.05
    assert [d.shader for d in ora.last_dispatches] == ["SIGMA_Shadow_ClassifyTiles.cs", "SIGMA_SmoothTiles.cs", "SIGMA_Copy.cs", "SIGMA_Shadow_Blur.cs",
                                                       "SIGMA_Shadow_PostBlur.cs", "SIGMA_Shadow_TemporalStabilization.cs"]


@pytest.mark.gpu
@pytest.mark.parametrize("size", [(192, 128), (211, 117)])
def test_hip_matches_oracle(size):
    worst = parity.run_parity("SIGMA_SHADOW", width=size[0], height=size[1], frames=6, verbose=True)
    assert worst <= parity.REL_TOL


@pytest.mark.gpu
def test_hip_matches_oracle_without_stabilization():
    worst = parity.run_parity("SIGMA_SHADOW", width=160, height=96, frames=3, verbose=True, settings_overrides=dict(maxStabilizedFrameNum=0))
    assert worst <= parity.REL_TOL


# ---------------------------------------------------------------------------------------------- SIGMA_SHADOW_TRANSLUCENCY
def _run_translucent(seq):
    ora = parity.OracleRun("SIGMA_SHADOW_TRANSLUCENCY", W, H)
    for f, frame in enumerate(seq):
        cs = parity.common_settings(frame["camera"], seq[max(f - 1, 0)]["camera"], W, H, f)
        ora.step(frame, cs, parity.denoiser_settings("SIGMA_SHADOW_TRANSLUCENCY", frame))
    return ora


def test_translucency_host_tables():
    """reference Source/Denoisers/Sigma_ShadowTranslucency.hpp: RGBA8 shadow planes, IN_TRANSLUCENCY appended to 3 passes"""
    seq = parity.generate_sequence("SIGMA_SHADOW_TRANSLUCENCY", W, H, 2)
    ora = _run_translucent(seq)
    assert [d.shader for d in ora.last_dispatches] == [
        "SIGMA_ShadowTranslucency_ClassifyTiles.cs", "SIGMA_SmoothTiles.cs", "SIGMA_Copy.cs", "SIGMA_ShadowTranslucency_Blur.cs",
        "SIGMA_ShadowTranslucency_PostBlur.cs", "SIGMA_ShadowTranslucency_TemporalStabilization.cs"]
    fmts = [t[0] for t in ora.inst.transient_pool]
    assert fmts.count(api.Format.RGBA8_UNORM) == 4  # TEMP_1, TEMP_2, HISTORY + TILES
    assert fmts.count(api.Format.R8_UNORM) == 0
    blur = ora.last_dispatches[3]
    assert len(blur.resources) == 7 and blur.resources[4][1] == RT.IN_TRANSLUCENCY


def test_translucency_x_channel_tracks_opaque_denoiser():
    """With an all-opaque translucency input (yzw = 0 in shadow, x = lit flag) the .x channel goes through the same arithmetic as
    SIGMA_SHADOW, and yzw of a fully lit / fully shadowed frame are the fixed points 1 / 0."""
    seq = parity.generate_sequence("SIGMA_SHADOW_TRANSLUCENCY", W, H, 5)
    for fr in seq:
        lit = (fr["penumbra"].float() >= 65504.0)
        t = torch.zeros((H, W, 4), dtype=torch.uint8)
        t[..., 0] = lit.to(torch.uint8) * 255
        t[..., 1:] = (lit.to(torch.uint8) * 255).unsqueeze(-1)
        fr["translucency"] = t
    out_t = _run_translucent(seq).output(RT.OUT_SHADOW_TRANSLUCENCY)
    out_s = _run(seq).output(RT.OUT_SHADOW_TRANSLUCENCY)[..., 0]
    m = ~seq[-1]["is_sky"].numpy()
    assert np.array_equal(out_t[..., 0][m], out_s[m])
    # every channel carries the same signal here
    for ch in (1, 2, 3):
        assert np.array_equal(out_t[..., ch][m], out_t[..., 0][m])


def test_translucency_keeps_colour_under_glass():
    seq = parity.generate_sequence("SIGMA_SHADOW_TRANSLUCENCY", W, H, 6)
    out = _run_translucent(seq).output(RT.OUT_SHADOW_TRANSLUCENCY) / 255.0
    out = out * out  # SIGMA_BackEnd_UnpackShadow
    fr = seq[-1]
    t = fr["translucency"].numpy()
    glass = (t[..., 0] == 0) & (t[..., 1:].max(-1) > 0) & ~fr["is_sky"].numpy()
    assert glass.sum() > 20  # the scene has a stained-glass shadow
    # under the glass the shadow term is dark but the translucent colour survives, with the glass tint (blue > red)
    assert out[glass][:, 0].mean() < 0.5
    assert out[glass][:, 3].mean() > out[glass][:, 1].mean() + 0.05


@pytest.mark.gpu
@pytest.mark.parametrize("size", [(192, 128), (211, 117)])
def test_hip_matches_oracle_translucency(size):
    worst = parity.run_parity("SIGMA_SHADOW_TRANSLUCENCY", width=size[0], height=size[1], frames=6, verbose=True)
    assert worst <= parity.REL_TOL


@pytest.mark.gpu
def test_hip_matches_oracle_translucency_without_stabilization():
    worst = parity.run_parity("SIGMA_SHADOW_TRANSLUCENCY", width=160, height=96, frames=3, verbose=True, settings_overrides=dict(maxStabilizedFrameNum=0))
    assert worst <= parity.REL_TOL


@pytest.mark.gpu
@pytest.mark.parametrize("name,z_scale", [("SIGMA_SHADOW", 1.0), ("SIGMA_SHADOW_TRANSLUCENCY", 0.0)])
def test_hip_matches_oracle_screen_space_motion_vectors(name, z_scale):
    # true 2D / 2.5D screen-space MVs instead of "world-space MVs scaled by 0" (CommonSettings::motionVectorScale): the other reprojection branch of TS
    worst = parity.run_parity(name, width=176, height=104, frames=5, verbose=True, extra_want=("mv2d",),
                              cs_kw=dict(isMotionVectorInWorldSpace=False, motionVectorScale=(1.0 / 176, 1.0 / 104, z_scale)))
    assert worst <= parity.REL_TOL
