"""The drop-in boundary used from C++, the reference's own language: tests/cpp/integration_reference.cpp includes the unchanged NRD
headers plus include/NRDHip.h / NRDIntegrationHip.hpp, links libNRD_hip.so and runs BASELINE.json configs[0] (REFERENCE denoiser,
256x256, bit-exact running mean). CPU: compiles, links and runs the host-only part; GPU: the whole program."""
import os
import subprocess

import pytest

from raytracingdenoiser_amd import build as native_build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cpp", "integration_reference.cpp")
OUT_DIR = os.path.join(ROOT, "tests", "cpp", "build")
EXE = os.path.join(OUT_DIR, "integration_reference")


def _build():
    lib = native_build.build_product()
    os.makedirs(OUT_DIR, exist_ok=True)
    if os.path.exists(EXE) and os.path.getmtime(EXE) > max(os.path.getmtime(SRC), os.path.getmtime(lib)):
        return
    cmd = ["g++", "-std=c++17", "-O1", "-ffp-contract=off", "-Wno-attributes", "-I" + os.path.join(ROOT, "include"), "-I/opt/rocm/include", "-D__HIP_PLATFORM_AMD__", SRC, "-o", EXE,
           "-L" + os.path.dirname(lib), "-lNRD_hip", "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath,$ORIGIN/../../../raytracingdenoiser_amd/lib", "-Wl,-rpath,/opt/rocm/lib"]
    subprocess.run(cmd, check=True, capture_output=True, text=True)


def test_cpp_application_compiles_links_and_compiles_dispatches_on_the_host():
    _build()
    r = subprocess.run([EXE, "--no-gpu"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "NRD 4.14.0" in r.stdout and "host-only OK" in r.stdout


@pytest.mark.gpu
def test_cpp_application_reference_denoiser_is_bit_exact():
    _build()
    r = subprocess.run([EXE], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "0 mismatching values" in r.stdout and "integration smoke OK" in r.stdout
