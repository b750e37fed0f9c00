"""NVIDIA-RTX/MathLib is an un-vendored submodule of the reference (CMakeLists.txt:118-127, no route to it from the build container): `ml.hlsli` / `ml.h` are stood in for by
oracle/ref/ml.hlsli and oracle/ref/host/{ml.h, ml.hlsli} ("parity unpinned" for those files alone -- DESIGN.md). VERDICT r05 item 8: the stand-in must be SWAPPABLE -- a checkout of the
real MathLib drops in through a make variable, no edits -- and the one point in doubt (Sequence::Bayer4x4ui advancing by ReverseBits4( frameIndex ), the reviewer's recollection of
MathLib's ML_BAYER_REVERSEBITS default, against frameIndex) must exist as a buildable alternative in every place the function is restated."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "ref")
REFERENCE = "/root/reference"
CLANG = "/opt/rocm/lib/llvm/bin/clang++"
ENTRY = os.path.join(REFERENCE, "Shaders", "Source", "REBLUR_DiffuseSpecular_TemporalAccumulation.cs.hlsl")  # (uses Sequence::Bayer4x4: REBLUR_TemporalAccumulation.hlsli:341)
needs_reference = pytest.mark.skipif(not os.path.exists(ENTRY), reason="needs /root/reference (the build container)")


def _generate(tmp_path, *extra):
    out = os.path.join(str(tmp_path), "ta.cpp")
    subprocess.run(["python3", os.path.join(REF, "hlsl2cpp.py"), ENTRY, out, "--reference", REFERENCE] + list(extra), check=True, cwd=REF)
    return open(out).read(), out


@needs_reference
def test_a_vendored_mathlib_directory_takes_the_place_of_the_stand_in(tmp_path):
    """MATHLIB_DIR of oracle/ref/Makefile -> hlsl2cpp.py --mathlib: the directory is searched in front of oracle/ref, so ITS ml.hlsli is the one the reference's shaders include"""
    vendored = tmp_path / "MathLib"
    vendored.mkdir()
    text = open(os.path.join(REF, "ml.hlsli")).read()
    (vendored / "ml.hlsli").write_text(text + "\nfloat MarkerOfTheVendoredMathLib( float x ) { return x; }\n")
    with_dir, path = _generate(tmp_path, "--mathlib", str(vendored))
    without, _ = _generate(tmp_path)
    assert "MarkerOfTheVendoredMathLib" in with_dir and "MarkerOfTheVendoredMathLib" not in without
    # and the translation unit still compiles with the flags of oracle/ref/Makefile
    flags = "-std=c++17 -O0 -fsyntax-only -ffp-contract=off -fno-fast-math -I. -Wno-gnu-anonymous-struct -Wno-nested-anon-types -Wno-constant-logical-operand -Wno-unused-value".split()
    subprocess.run([CLANG] + flags + [path], check=True, cwd=REF)
    mk = open(os.path.join(REF, "Makefile")).read() + open(os.path.join(REF, "host", "Makefile")).read()
    assert "MATHLIB_DIR" in mk and "--mathlib $(MATHLIB_DIR)" in mk and "-I$(MATHLIB_DIR)" in mk


@needs_reference
def test_the_reversebits_reading_of_bayer4x4_builds_in_the_reference_text(tmp_path):
    alt, path = _generate(tmp_path, "-DNRD_MATHLIB_BAYER_REVERSEBITS=1")
    default, _ = _generate(tmp_path)
    assert "ReverseBits4( frameIndex )" in alt and "ReverseBits4( frameIndex )" not in default
    flags = "-std=c++17 -O0 -fsyntax-only -ffp-contract=off -fno-fast-math -I. -Wno-gnu-anonymous-struct -Wno-nested-anon-types -Wno-constant-logical-operand -Wno-unused-value".split()
    subprocess.run([CLANG] + flags + [path], check=True, cwd=REF)


def test_the_reversebits_alternative_exists_wherever_bayer4x4_is_restated(tmp_path):
    """product host + device, oracle, the reference host's stand-in: the same switch, and the two readings differ exactly by the frame offset"""
    sites = ["raytracingdenoiser_amd/csrc/hip/nrdmath.h", "raytracingdenoiser_amd/csrc/host/hostmath.h", "oracle/ml.h", "oracle/ref/host/ml.hlsli", "oracle/ref/ml.hlsli"]
    for s in sites:
        assert "NRD_MATHLIB_BAYER_REVERSEBITS" in open(os.path.join(ROOT, s)).read(), s
    src = tmp_path / "bayer.cpp"
    src.write_text('''#include <cstdint>
#include <cstdio>
#include <cmath>
#include "hostmath.h"
int main() {
    unsigned bad = 0;
    for (uint32_t f = 0; f < 64; f++)
        for (uint32_t y = 0; y < 4; y++)
            for (uint32_t x = 0; x < 4; x++) {
                const uint32_t v = f & 0xFu, rev = ((v & 1u) << 3) | ((v & 2u) << 1) | ((v & 4u) >> 1) | ((v & 8u) >> 3);
                const uint32_t base = nrdhost::Bayer4x4ui(x, y, 0);
                bad += nrdhost::Bayer4x4ui(x, y, f) != ((base + (NRD_MATHLIB_BAYER_REVERSEBITS ? rev : f)) & 0xFu);
            }
    printf("%u\\n", bad);
    return bad != 0;
}
''')
    hostmath = os.path.join(ROOT, "raytracingdenoiser_amd", "csrc", "host")
    for define in ("-DNRD_MATHLIB_BAYER_REVERSEBITS=0", "-DNRD_MATHLIB_BAYER_REVERSEBITS=1"):
        exe = str(tmp_path / ("bayer" + define[-1]))
        subprocess.run(["g++", "-std=c++17", "-O1", define, "-I" + hostmath, "-I" + os.path.join(ROOT, "include"), str(src), "-o", exe], check=True)
        assert subprocess.run([exe], capture_output=True, text=True).stdout.strip() == "0"
    shutil.rmtree(str(tmp_path), ignore_errors=True)
