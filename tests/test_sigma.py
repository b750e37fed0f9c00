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
