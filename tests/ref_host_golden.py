"""Shared by tools/make_ref_host_golden.py (the recorder: drives the reference's compiled host, oracle/_ref/libnrdhost.so) and tests/test_ref_host_golden.py (the replay: drives the
product's host alone): the dispatch stream of a short default-settings sequence per denoiser, reduced to one sha1 per dispatch over (name, shader, grid, resources, constant bytes)."""
import hashlib
import json
import os

import parity
from raytracingdenoiser_amd import api

PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_host_dispatches.json")
FRAMES, SIZE = 3, (96, 64)
# settings on top of the defaults, chosen so that every optional pass of a family appears in some case
CASES = {name: [None] for name in parity.DENOISERS}
CASES["REBLUR_DIFFUSE_SPECULAR"] += [{"enablePerformanceMode": True, "enableAntiFirefly": True, "hitDistanceReconstructionMode": 1, "checkerboardMode": 1}, {"maxStabilizedFrameNum": 0, "hitDistanceReconstructionMode": 2}]
CASES["RELAX_DIFFUSE_SPECULAR"] += [{"enableAntiFirefly": True, "hitDistanceReconstructionMode": 1, "atrousIterationNum": 6, "checkerboardMode": 2}]
CASES["SIGMA_SHADOW"] += [{"maxStabilizedFrameNum": 0}]


def _instance_digest(inst):
    return hashlib.sha1(repr((inst.pipelines, [(int(f), d) for f, d in inst.permanent_pool], [(int(f), d) for f, d in inst.transient_pool])).encode()).hexdigest()


def stream(name, overrides, lib=None):
    """[instance digest, [per-frame [dispatch digest, ...]]] of the host behind `lib` (None: the product)"""
    inst = api.Instance([(0, parity.DENOISERS[name][0])], lib=lib)
    seq = parity.generate_sequence(name, SIZE[0], SIZE[1], FRAMES, device="cpu")
    frames = []
    for f in range(FRAMES):
        cs = parity.common_settings(seq[f]["camera"], seq[max(f - 1, 0)]["camera"], SIZE[0], SIZE[1], f, splitScreen=0.25 if f == 2 else 0.0)
        assert inst.set_common_settings(cs) == api.Result.SUCCESS
        assert inst.set_denoiser_settings(0, parity.denoiser_settings(name, seq[f], overrides)) == api.Result.SUCCESS
        r, ds = inst.get_compute_dispatches()
        assert r == api.Result.SUCCESS
        frames.append([hashlib.sha1(repr((d.name, d.shader, d.grid, [(int(a), int(b), int(c)) for a, b, c in d.resources])).encode() + d.constants).hexdigest()[:20] for d in ds])
    return [_instance_digest(inst), frames]


def key(name, overrides):
    return name + ("" if not overrides else " " + json.dumps(overrides, sort_keys=True))
