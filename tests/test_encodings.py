"""The reference's other G-buffer encodings: NRD_NORMAL_ENCODING 0..4 (RGBA8_UNORM, RGBA8_SNORM, R10G10B10A2_UNORM, RGBA16_UNORM, RGBA16_SNORM) and
NRD_ROUGHNESS_ENCODING 0..2 (squared, linear, square root) -- reference CMakeLists.txt:28-29, Shaders/Include/NRD.hlsli:298-309, 600-667, Common.hlsli:76-85, Source/Reblur.cpp:52-62.

As in the reference the encoding is a BUILD configuration: raytracingdenoiser_amd/build.py builds one library per encoding (lib/enc<N><R>/libNRD_hip.so; the default 2 / 1 stays
lib/libNRD_hip.so), and so do the oracle (oracle/liboracle_enc<N><R>.so), the reference's own shader text and host (oracle/_ref/enc<N><R>/, built with the matching -D switches) and the
CPU emulation of the device sources. A process binds one of each, chosen by the environment variables NRD_NORMAL_ENCODING / NRD_ROUGHNESS_ENCODING, so every case here runs in a CHILD
process (tests/encoding_cases.py) -- what it runs is the unchanged test machinery of the default encoding.

What differs between encodings, and is therefore what these cases hold: the IN_NORMAL_ROUGHNESS texel codec and the roughness transfer function in the guide decode (and in the a-trous
gathers, SIGMA's blur, the validation overlay), REBLUR's PREV_NORMAL_ROUGHNESS pool format and the texel PostBlur forwards to it, TemporalAccumulation's reads of that plane (2x2 normals,
the stored-roughness gather, and -- for every encoding but R10G10B10A2 -- true bilinear samples without random draws instead of the stochastic tap), NRD_NORMAL_ENCODING_ERROR, the missing
material IDs (and the settings validation that follows from them), nrd::GetLibraryDesc.
"""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASES = os.path.join(ROOT, "tests", "encoding_cases.py")
HAVE_REFERENCE = os.path.isdir("/root/reference/Shaders/Source")

# All four non-default normal encodings (4- and 8-byte texels, UNORM bias and SNORM, the fp16 PREV_NORMAL_ROUGHNESS plane of encoding 4) and both non-linear roughness transfer
# functions. NRD_ENCODINGS_FULL=1 adds the default normal encoding with the two other roughness encodings (verified in round 6: all six pass).
ENCODINGS = [(0, 0), (4, 2), (1, 1), (3, 1)] + ([(2, 0), (2, 2)] if os.environ.get("NRD_ENCODINGS_FULL") else [])
IDS = ["normal%d_roughness%d" % e for e in ENCODINGS]


def _env(encoding, **extra):
    env = dict(os.environ)
    env.update({"NRD_NORMAL_ENCODING": str(encoding[0]), "NRD_ROUGHNESS_ENCODING": str(encoding[1])})
    env.pop("NRD_HIP_LIBRARY", None)  # (an A/B variant of the default encoding has no business here)
    env.update(extra)
    return env


def _child(encoding, args, timeout=1500, **extra):
    r = subprocess.run([sys.executable] + args, env=_env(encoding, **extra), cwd=ROOT, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, "encoding %d / %d: %s\n%s\n%s" % (encoding[0], encoding[1], " ".join(args), r.stdout[-4000:], r.stderr[-4000:])
    return r.stdout


def _ref_built(encoding):
    return os.path.exists(os.path.join(ROOT, "oracle", "_ref", "enc%d%d" % encoding, "libnrdref.so"))


def _build_ref(encoding):
    """oracle/_ref/enc<N><R>/ from /root/reference (the build container only; the libraries travel to the GPU box)"""
    if HAVE_REFERENCE:
        _child(encoding, ["-c", "from raytracingdenoiser_amd import build as B; assert B.build_ref() is not None"])
    if not _ref_built(encoding) and (2, 1) != encoding:
        pytest.skip("oracle/_ref/enc%d%d not built (needs /root/reference)" % encoding)


@pytest.mark.parametrize("encoding", ENCODINGS, ids=IDS)
def test_library_desc_pool_formats_and_settings_validation_follow_the_encoding(encoding):
    assert "desc OK" in _child(encoding, [CASES, "desc"])


@pytest.mark.parametrize("encoding", ENCODINGS, ids=IDS)
def test_dispatch_lists_equal_the_reference_host_built_with_the_same_encoding(encoding):
    """tests/test_ref_host.py (every DispatchDesc of all 19 denoisers, option matrix, error codes, nrd::GetLibraryDesc field by field) against the reference's Source/*.cpp compiled
    with -DNRD_NORMAL_ENCODING / -DNRD_ROUGHNESS_ENCODING of this encoding. (One test of that file drives the complete reference -- host and shader text -- for all 19 denoisers; the
    per-encoding build of the text holds one denoiser per family, so it stays with the default encoding.)"""
    _build_ref(encoding)
    out = _child(encoding, ["-m", "pytest", "tests/test_ref_host.py", "-q", "-x", "-n", "2", "-p", "no:cacheprovider", "-k", "not complete_reference_host_and_shader_text"])
    assert " passed" in out and "failed" not in out, out[-2000:]


@pytest.mark.parametrize("encoding", ENCODINGS, ids=IDS)
def test_oracle_matches_the_reference_shader_text_built_with_the_same_encoding(encoding):
    _build_ref(encoding)
    assert _child(encoding, [CASES, "ref_text"]).count("ref_text OK") == 4


@pytest.mark.parametrize("encoding", ENCODINGS, ids=IDS)
def test_device_sources_compiled_for_the_cpu_equal_the_oracle(encoding):
    """the emulation backend (tests/emu): the .hip sources of this encoding, compiled for x86, against the oracle of this encoding -- bit for bit; the GPU twin is below"""
    assert _child(encoding, [CASES, "parity_emu"], NRD_PARITY_BACKEND="emu").count("parity_emu OK") == 6


@pytest.mark.parametrize("encoding", ENCODINGS, ids=IDS)
def test_frontend_header_packs_the_texels_of_the_encoding(encoding):
    """include/NRD.hip.h compiled with the encoding's defines: NRD_FrontEnd_PackNormalAndRoughness + NRD_StoreNormalRoughnessTexel produce the texels the scene generator of the parity
    tests produces (raytracingdenoiser_amd/synth.py, written without the header), and unpack to the inputs within the quantisation of the encoding"""
    out = _child(encoding, ["-m", "pytest", "tests/test_frontend_header.py", "-q", "-x", "-n", "0", "-p", "no:cacheprovider", "-k", "known_answers"])
    assert "1 passed" in out, out[-2000:]


GPU_ENCODINGS = [(0, 0), (4, 2), (1, 1), (3, 1)]


@pytest.mark.gpu
@pytest.mark.parametrize("encoding", GPU_ENCODINGS, ids=["normal%d_roughness%d" % e for e in GPU_ENCODINGS])
def test_gpu_bit_exact_against_the_oracle(encoding):
    """lib/enc<N><R>/libNRD_hip.so on the GPU against the oracle of the same encoding: REBLUR_DIFFUSE_SPECULAR, RELAX_DIFFUSE_SPECULAR(_SH) incl. 7 a-trous iterations (the gathering
    taps decode IN_NORMAL_ROUGHNESS themselves), SIGMA_SHADOW, a ragged REBLUR_SPECULAR_SH frame: every user output and pool plane, max relative error 0"""
    assert _child(encoding, [CASES, "parity_hip"]).count("parity_hip OK") == 6


@pytest.mark.gpu
@pytest.mark.parametrize("encoding", [(0, 0), (4, 2)], ids=["normal0_roughness0", "normal4_roughness2"])
def test_gpu_bit_exact_at_1280x720(encoding):
    assert _child(encoding, [CASES, "parity_hip_large"]).count("parity_hip OK") == 2


@pytest.mark.gpu
@pytest.mark.parametrize("encoding", [(0, 0), (4, 2)], ids=["normal0_roughness0", "normal4_roughness2"])
def test_gpu_sequence_against_the_reference_shader_text_of_the_encoding(encoding):
    """tests/test_deep_parity.py's one-hop case -- the GPU against oracle/_ref over 8 frames, each side with its own history -- under this encoding"""
    if not _ref_built(encoding):
        pytest.skip("oracle/_ref/enc%d%d not built (needs /root/reference; the libraries travel)" % encoding)
    out = _child(encoding, ["-m", "pytest", "tests/test_deep_parity.py", "-q", "-x", "-n", "0", "-m", "gpu", "-p", "no:cacheprovider", "-k", "reference_shader_text"])
    assert "3 passed" in out, out[-2000:]
