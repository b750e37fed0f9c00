"""include/NRD.hip.h -- the HIP counterpart of the reference's NRD.hlsli front-end / back-end functions (what an application's own
kernels include). tests/cpp/frontend_check.hip evaluates every function on a deterministic sample set: known answers on the host
(CPU test), device == host on the GPU. Here additionally: the normal/roughness words it packs equal the ones of the generator that
feeds all parity tests (raytracingdenoiser_amd/synth.py), so the header and the denoiser agree on the encoding."""
import os
import re
import subprocess

import numpy as np
import pytest
import torch

from raytracingdenoiser_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cpp", "frontend_check.hip")
HDR = os.path.join(ROOT, "include", "NRD.hip.h")
EXE = os.path.join(ROOT, "tests", "cpp", "build", "frontend_check")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def _build():
    os.makedirs(os.path.dirname(EXE), exist_ok=True)
    if os.path.exists(EXE) and os.path.getmtime(EXE) > max(os.path.getmtime(SRC), os.path.getmtime(HDR)):
        return
    cmd = [HIPCC, "-std=c++17", "-O2", "-ffp-contract=off", "--offload-arch=gfx950", "-I" + os.path.join(ROOT, "include"), SRC, "-o", EXE]
    subprocess.run(cmd, check=True, capture_output=True, text=True)


def _pcg(v):
    v = v.astype(np.uint64)
    s = (v * 747796405 + 2891336453) & 0xFFFFFFFF
    w = (((s >> ((s >> 28) + 4)) ^ s) * 277803737) & 0xFFFFFFFF
    return ((w >> 22) ^ w).astype(np.uint32)


def _u(i, k):
    return (_pcg(i * 64 + k) >> 8).astype(np.float32) * np.float32(1.0 / 16777216.0)


def _expected_word_checksum(count):
    i = np.arange(count, dtype=np.uint64)
    v = np.stack([_u(i, 0) * np.float32(2) - np.float32(1), _u(i, 1) * np.float32(2) - np.float32(1), _u(i, 2) * np.float32(2) - np.float32(1)], -1)
    l = np.sqrt(v[:, 0] * v[:, 0] + v[:, 1] * v[:, 1] + v[:, 2] * v[:, 2], dtype=np.float32)
    n = np.where((l < np.float32(0.05))[:, None], np.array([0, 0, 1], np.float32)[None], v / np.maximum(l, np.float32(1e-20))[:, None]).astype(np.float32)
    roughness = _u(i, 18)
    material = (_pcg(i + 77) & 3).astype(np.float32)
    words = synth.pack_normal_roughness(torch.from_numpy(n), torch.from_numpy(roughness), torch.from_numpy(material)).numpy().view(np.uint32)
    c = 0
    for w in words.tolist():
        c = (c * 31 + w) & 0xFFFFFFFF
    return c


def test_frontend_header_known_answers_on_the_host():
    _build()
    r = subprocess.run([EXE, "--no-gpu"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    m = re.search(r"host OK: (\d+) samples, normal/roughness word checksum ([0-9a-f]{8})", r.stdout)
    assert m, r.stdout
    assert int(m.group(2), 16) == _expected_word_checksum(int(m.group(1)))


@pytest.mark.gpu
def test_frontend_header_device_matches_host():
    _build()
    r = subprocess.run([EXE], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "frontend check OK" in r.stdout
