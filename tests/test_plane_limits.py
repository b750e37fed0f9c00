"""Maximum sizes: the kernels address a plane with 32-bit byte offsets from a 24-bit multiply (csrc/hip/planes.h TexelOffset: row pitch < 16 MiB, plane < 4 GiB). The reference names
frame sizes with 16 bits per axis (NRDSettings.h resourceSize), so the largest frames with 16-byte texels lie beyond that -- the executor REFUSES them (Result::UNSUPPORTED) at creation
and at binding instead of wrapping around. The refusals are argument checks in front of any device call: they run on the CPU with the product library itself."""
import ctypes as C

import pytest

from raytracingdenoiser_amd import api

RT, F = api.ResourceType, api.Format


def _create(lib, inst, w, h):
    handle = C.c_void_p()
    return api.Result(lib.nrdHipCreateExecutor(inst.handle, w, h, None, C.byref(handle))), handle


def test_frames_beyond_4_gib_per_plane_are_refused_at_creation():
    inst = api.Instance([(0, api.Denoiser.REBLUR_DIFFUSE_SPECULAR)])
    for w, h in ((16384, 16384), (65535, 65535), (65535, 4097), (4097, 65535)):
        r, handle = _create(inst.lib, inst, w, h)
        assert r == api.Result.UNSUPPORTED and not handle.value, (w, h, r)
    # (the arena size of a frame that IS addressable is still reported: 16384 x 16383 RGBA32F planes end 256 KiB below 4 GiB)
    assert inst.lib.nrdHipGetArenaSize(inst.handle, 16384, 16383) > 0


def _check_bind_limits(make_executor):
    W, H = 64, 32
    inst, ex = make_executor(W, H)
    lib = inst.lib

    def bind(pitch, data=0x10000):
        desc = api.HipPlaneDesc(data, pitch, int(F.R32_SFLOAT), W, H)
        return api.Result(lib.nrdHipBindResource(ex.handle, int(RT.IN_VIEWZ), C.byref(desc)))

    assert bind(W * 4) == api.Result.SUCCESS
    assert bind((1 << 24) - 4) == api.Result.SUCCESS          # the largest pitch the 24-bit multiply takes (nothing is executed: binding records the pointer)
    assert bind(1 << 24) == api.Result.UNSUPPORTED and b"16 MiB" in lib.nrdHipGetLastError(ex.handle)
    assert bind(W * 4 - 4) == api.Result.INVALID_ARGUMENT     # (the older checks still come first: pitch below the row size)
    assert bind(W * 4) == api.Result.SUCCESS


def test_bind_refuses_a_pitch_the_offset_arithmetic_cannot_hold_emulated():
    from emu import emu_run

    def make(w, h):
        inst = api.Instance([(0, api.Denoiser.REBLUR_DIFFUSE)], lib=emu_run.load())
        return inst, emu_run.EmuExecutor(inst, w, h)

    _check_bind_limits(make)


@pytest.mark.gpu
def test_bind_refuses_a_pitch_the_offset_arithmetic_cannot_hold():
    from raytracingdenoiser_amd.executor import HipExecutor

    def make(w, h):
        inst = api.Instance([(0, api.Denoiser.REBLUR_DIFFUSE)])
        return inst, HipExecutor(inst, w, h)

    _check_bind_limits(make)


@pytest.mark.gpu
def test_bit_exact_at_7680x4320():
    """beyond BASELINE.json's sizes: 8K UHD -- x beyond 4096, 129 600 tiles of 32 x 8 pixels (more than 16 bits of tile indices), 530-MB planes. SIGMA_SHADOW here (8 s);
    REBLUR_DIFFUSE_SPECULAR and RELAX_DIFFUSE_SPECULAR_SH take 63 / 93 s and are run by tools/parity_8k.py: profiles/r06_8k_parity.log -- max relative error 0 for all three."""
    import parity

    assert parity.run_parity("SIGMA_SHADOW", 7680, 4320, 2, device="cuda") == 0.0
