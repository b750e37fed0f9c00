"""The device sources against the oracle WITHOUT a GPU: tests/emu compiles the .hip files of the product for x86-64 over a HIP shim (TEST INFRASTRUCTURE -- nothing
under raytracingdenoiser_amd/ references it, and the product still refuses to run without a GPU) and runs them through the same C-ABI. Contraction is decided by
the compiler front end and the five transcendental instructions come from the oracle's measured tables, so the emulated kernels compute what the gfx950 build
computes: a change that breaks bit-exactness shows up here, in the CPU suite, before any GPU time is spent. The -m gpu suite holds the real library to the same bar."""
import pytest

import parity


@pytest.mark.parametrize("name,width,height,frames", [
    ("REBLUR_DIFFUSE_SPECULAR", 96, 64, 4),     # window kernel + fallback launch of TemporalAccumulation, all seven passes
    ("RELAX_DIFFUSE_SPECULAR_SH", 80, 56, 3),   # LDS-tiled a-trous steps 2 / 4, global steps 8 / 16, history clamping
    ("SIGMA_SHADOW", 96, 64, 3),
    ("REBLUR_DIFFUSE_OCCLUSION", 67, 45, 3),    # odd size: clamped footprints, workgroups beyond the frame
    ("REBLUR_DIFFUSE", 544, 24, 3),             # 17 tile columns: the XCD-striped tile order and its rotated form in HistoryFix (passes.h BlockTileX / BlockTileXRotated)
    ("RELAX_DIFFUSE", 544, 24, 3),
])
def test_emulated_device_sources_match_the_oracle_bit_for_bit(name, width, height, frames):
    worst = parity.run_parity(name, width=width, height=height, frames=frames, backend="emu")
    assert worst == 0.0, (name, worst)
    # every ClampI( x, a, b ) call site ran with a <= b: v_med3_i32 (what the device executes) is then the clamp the source means
    from emu import emu_run

    lib = emu_run.load()
    lib.emu_med3_violations.restype = __import__("ctypes").c_long
    assert lib.emu_med3_violations() == 0
