"""CommonSettings::enableValidation: the debug overlay pass (reference Shaders/REBLUR_Validation.cs.hlsl, RELAX_Validation.cs.hlsl; appended as the last
dispatch of a frame by Reblur.cpp:785-796 / Relax.cpp:608-617). OUT_VALIDATION is an RGBA8_UNORM user texture cut into 4 x 4 viewports, each showing one
guide / internal plane resampled to a quarter of the resolution. The text labels the reference prints into the viewports are not drawn (the same in
the oracle and in the HIP pass)."""
import numpy as np
import pytest
import torch

import parity
from raytracingdenoiser_amd import api

RT = api.ResourceType


def _oracle_frames(name, w, h, frames):
    seq = parity.generate_sequence(name, w, h, frames)
    ora = parity.OracleRun(name, w, h, validation=True)
    outs, lists = [], []
    for f, frame in enumerate(seq):
        cs = parity.common_settings(frame["camera"], seq[max(f - 1, 0)]["camera"], w, h, f, enableValidation=True)
        ora.step(frame, cs, parity.denoiser_settings(name, frame))
        outs.append(ora.outs[RT.OUT_VALIDATION][0].copy())
        lists.append([d.shader for d in ora.last_dispatches])
    return seq, outs, lists


@pytest.mark.parametrize("name,shader", [("REBLUR_DIFFUSE_SPECULAR", "REBLUR_Validation.cs"), ("RELAX_DIFFUSE_SPECULAR", "RELAX_Validation.cs")])
def test_oracle_validation_overlay_shows_the_guides(name, shader):
    w, h = 192, 128
    seq, outs, lists = _oracle_frames(name, w, h, 3)
    assert all(l[-1] == shader for l in lists)
    assert not outs[0].any()  # the frame that resets the history clears the overlay
    img, frame = outs[2].astype(np.float32), seq[2]
    vw, vh = w // 4, h // 4
    viewz = frame["viewz"].numpy()
    src = viewz[2::4, 2::4][:vh, :vw]  # nearest sample of viewport pixel (x, y): source texel (4x + 2, 4y + 2)
    finite = np.abs(src) < 5e5
    # viewport 2 (first row, third column): view depth as 0.1 z / (1 + 0.1 z) in green (blue for negative z), pure red beyond the denoising range
    vp = img[:vh, 2 * vw:3 * vw]
    want = 255.0 * 0.1 * np.abs(src) / (1.0 + 0.1 * np.abs(src))
    ch = np.where(src < 0, vp[..., 2], vp[..., 1])
    # (float rounding of the viewport uv may pick the texel next to (4x + 2, 4y + 2): depth edges are excluded through a quantile)
    assert finite.any() and np.quantile(np.abs(ch - want)[finite], 0.97) <= 1.0
    assert np.mean(vp[..., 0][~finite] == 255) > 0.97 and np.mean(vp[..., 1][~finite] == 0) > 0.97
    assert np.all(vp[..., 3] == 255)
    # viewport 0: normals as N * 0.5 + 0.5 -- unit length after decoding
    n = img[:vh, :vw, :3] / 255.0 * 2.0 - 1.0
    assert np.abs(np.linalg.norm(n, axis=-1) - 1.0).max() < 0.03
    # viewport 1: roughness (grey)
    vp = img[:vh, vw:2 * vw]
    assert np.all(vp[..., 0] == vp[..., 1]) and np.all(vp[..., 1] == vp[..., 2])
    # the viewports nothing is drawn into keep what the texture held (zeros from the reset frame)
    assert not img[vh:2 * vh, vw:3 * vw].any()


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["REBLUR_DIFFUSE_SPECULAR", "REBLUR_SPECULAR", "REBLUR_DIFFUSE_OCCLUSION", "REBLUR_DIFFUSE_SPECULAR_SH", "RELAX_DIFFUSE_SPECULAR", "RELAX_DIFFUSE"])
def test_hip_validation_overlay_matches_oracle(name):
    assert parity.run_parity(name, 192, 128, frames=4, cs_kw={"enableValidation": True}) == 0.0


@pytest.mark.gpu
def test_hip_validation_overlay_dynamic_resolution_and_graph_mode():
    assert parity.run_parity("REBLUR_DIFFUSE_SPECULAR", 144, 96, frames=3, resource=(192, 128), cs_kw={"enableValidation": True}, graph=True) == 0.0
    assert parity.run_parity("RELAX_DIFFUSE_SPECULAR", 144, 96, frames=3, resource=(192, 128), cs_kw={"enableValidation": True}) == 0.0


@pytest.mark.gpu
def test_sigma_ignores_enable_validation():
    """SIGMA has no validation pass (reference Sigma.cpp adds none): the flag changes nothing"""
    assert parity.run_parity("SIGMA_SHADOW", 192, 128, frames=3, cs_kw={"enableValidation": True}) == 0.0
