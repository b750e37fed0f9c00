"""Ragged and tiny frames: sizes below one 16x16 tile, one pixel wide / high, not multiples of any tile size -- every family must stay
bit-exact against the oracle (clamped halos, partial tiles, partial workgroups)."""
import pytest

import parity

SIZES = [(1, 1), (7, 5), (17, 9), (33, 1), (1, 40), (65, 47)]
FAMILIES = ["REBLUR_DIFFUSE_SPECULAR", "REBLUR_DIFFUSE_SPECULAR_SH", "REBLUR_DIFFUSE_SPECULAR_OCCLUSION", "REBLUR_DIFFUSE_DIRECTIONAL_OCCLUSION", "RELAX_DIFFUSE_SPECULAR_SH",
            "SIGMA_SHADOW", "SIGMA_SHADOW_TRANSLUCENCY"]


@pytest.mark.gpu
@pytest.mark.parametrize("name", FAMILIES)
def test_hip_matches_oracle_on_tiny_and_ragged_frames(name):
    for w, h in SIZES:
        worst = parity.run_parity(name, width=w, height=h, frames=3)
        assert worst <= parity.REL_TOL, (name, w, h, worst)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["REBLUR_DIFFUSE_SPECULAR", "REBLUR_DIFFUSE_SPECULAR_OCCLUSION", "RELAX_DIFFUSE_SPECULAR_SH", "SIGMA_SHADOW_TRANSLUCENCY"])
def test_hip_matches_oracle_with_padded_user_planes(name):
    # application planes that live inside wider allocations: row pitch > row size, odd base offsets (3 extra texels per row, one extra row)
    worst = parity.run_parity(name, width=83, height=41, frames=3, pad=3)
    assert worst <= parity.REL_TOL
