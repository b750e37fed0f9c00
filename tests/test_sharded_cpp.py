"""nrd::ShardedIntegrationHip (include/NRDShardedIntegrationHip.hpp): the multi-GPU frame from a C++ host. tests/cpp/sharded_virtual_ranks.hip connects N
virtual ranks on one GPU through a loop-back transport and holds every rank's owned rows against a single-GPU nrd::IntegrationHip run, bit for bit.
CPU: the program and the RCCL transport compile against the installed headers; GPU: 2 and 3 virtual ranks."""
import os
import subprocess

import pytest

from raytracingdenoiser_amd import build as native_build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cpp", "sharded_virtual_ranks.hip")
HEADERS = [os.path.join(ROOT, "include", h) for h in ("NRDShardedIntegrationHip.hpp", "NRDIntegrationHip.hpp", "NRDHip.h", "NRD.hip.h")]
EXE = os.path.join(ROOT, "tests", "cpp", "build", "sharded_virtual_ranks")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["-std=c++17", "-O1", "-ffp-contract=off", "--offload-arch=gfx950", "-Wno-return-type-c-linkage", "-I" + os.path.join(ROOT, "include")]


def _build():
    lib = native_build.build_product()
    os.makedirs(os.path.dirname(EXE), exist_ok=True)
    if os.path.exists(EXE) and os.path.getmtime(EXE) > max([os.path.getmtime(SRC), os.path.getmtime(lib)] + [os.path.getmtime(h) for h in HEADERS]):
        return
    cmd = [HIPCC] + FLAGS + [SRC, "-o", EXE, "-L" + os.path.dirname(lib), "-lNRD_hip", "-Wl,-rpath,$ORIGIN/../../../raytracingdenoiser_amd/lib"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]


def test_sharded_integration_and_rccl_transport_compile():
    _build()
    # the RCCL transport (ncclSend / ncclRecv groups on a stream of its own) is compiled in with NRD_SHARDED_WITH_RCCL; it needs one GPU per rank to run
    r = subprocess.run([HIPCC] + FLAGS + ["-DNRD_SHARDED_WITH_RCCL", "-fsyntax-only", SRC], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 3])
def test_virtual_ranks_reproduce_the_single_gpu_run(world):
    _build()
    r = subprocess.run([EXE, str(world)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "0 mismatching values" in r.stdout and "sharded integration OK" in r.stdout


@pytest.mark.gpu
def test_virtual_ranks_with_the_motion_bound_measured_on_the_device():
    """ShardedIntegrationHipCreationDesc::measureMotion: every rank measures its strip (PrepareFrame -> nrdHipMeasureMotionRows), the maximum goes into PlanFrame; the frame
    in which one pixel of the last strip moves 30 rows is run unsharded by all ranks, the next one sharded again; owned rows bit-identical throughout"""
    _build()
    r = subprocess.run([EXE, "3", "measure"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "0 mismatching values" in r.stdout and "sharded integration OK" in r.stdout
