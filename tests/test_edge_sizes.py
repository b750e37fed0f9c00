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
