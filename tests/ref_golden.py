"""Golden fixtures of the reference's compiled shader text (tests/golden/ref_text_*.npz): recorder (tools/make_ref_golden.py, needs oracle/_ref) and replay
(tests/test_ref_golden.py, needs only the strict oracle). One fixture = one denoiser, a few frames at a small size: per dispatch the shader name, the sha1 of its
inputs (constants + every bound plane) and the raw bytes of every plane the reference text wrote."""
import json
import os

import numpy as np

import parity
import ref_parity
from oracle import driver as oracle_driver
from raytracingdenoiser_amd import api

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
# (denoiser, width, height, frames): the restart frame + frames under camera motion; sizes that keep every fixture well under 1 MB
CASES = [("REBLUR_DIFFUSE_SPECULAR", 64, 40, 3), ("RELAX_DIFFUSE_SPECULAR_SH", 48, 32, 3), ("SIGMA_SHADOW", 64, 40, 3), ("REBLUR_DIFFUSE_OCCLUSION", 64, 40, 3)]


def path_of(case):
    return os.path.join(GOLDEN_DIR, "ref_text_%s.npz" % case[0])


def _drive(name, width, height, frames, executor_factory):
    seq = parity.generate_sequence(name, width, height, frames, device="cpu")
    run = parity.OracleRun(name, width, height)
    ex = executor_factory(run)
    ex.user = run.ex.user  # the bound output planes
    run.ex = ex
    for f, frame in enumerate(seq):
        cam, cam_prev = frame["camera"], seq[max(f - 1, 0)]["camera"]
        run.step(frame, parity.common_settings(cam, cam_prev, width, height, f), parity.denoiser_settings(name, frame, None))


def record(case):
    name, width, height, frames = case
    meta, blobs = [], {}

    def on_pass(d, report):
        k = len(meta)
        slots = []
        for res, fmt, w, mine, theirs, alts in report:
            slot = d.resources.index(res)
            slots.append({"slot": slot, "format": int(fmt), "width": int(w)})
            blobs["d%d_s%d" % (k, slot)] = np.ascontiguousarray(theirs).view(np.uint8).reshape(theirs.shape[0], -1)
        meta.append({"shader": d.shader, "inputs_sha1": ex_holder[0].last_inputs_digest, "outputs": slots})

    ex_holder = []

    def factory(run):
        ex = oracle_driver.ComparingExecutor(run.inst, width, height, api.FORMAT_BYTES, on_pass=on_pass, strict=True, sensitivity=False)
        ex_holder.append(ex)
        return ex

    _drive(name, width, height, frames, factory)
    os.makedirs(GOLDEN_DIR, exist_ok=True)
    path = path_of(case)
    np.savez_compressed(path, meta=np.frombuffer(json.dumps({"case": list(case), "dispatches": meta}).encode(), dtype=np.uint8), **blobs)
    return path, len(meta), os.path.getsize(path)


def replay(case):
    """(PassStats of the strict oracle against the recorded reference-text outputs, number of dispatches, index of the first dispatch whose inputs differ or None)"""
    name, width, height, frames = case
    data = np.load(path_of(case))
    meta = json.loads(bytes(data["meta"]).decode())
    assert meta["case"] == list(case), "fixture recorded with other parameters: re-run tools/make_ref_golden.py"
    dispatches = meta["dispatches"]
    stats = ref_parity.PassStats(1e-5)
    state = {"k": 0, "first_mismatch": None}

    def on_pass(d, digest, planes):
        k = state["k"]
        state["k"] += 1
        assert k < len(dispatches) and dispatches[k]["shader"] == d.shader, "dispatch %d is %s, the fixture holds %s" % (k, d.shader, dispatches[k]["shader"] if k < len(dispatches) else "nothing")
        if digest != dispatches[k]["inputs_sha1"] and state["first_mismatch"] is None:
            state["first_mismatch"] = (k, d.shader)
        n_inputs = sum(1 for r in d.resources if r[0] == api.DescriptorType.TEXTURE)
        for out in dispatches[k]["outputs"]:
            res, fmt, w, mine = planes[out["slot"]]
            assert int(fmt) == out["format"] and int(w) == out["width"]
            theirs = data["d%d_s%d" % (k, out["slot"])]
            theirs = theirs.view(mine.dtype).reshape(mine.shape)
            slot = out["slot"]
            label = ("out%d" % (slot - n_inputs) if res[0] == api.DescriptorType.STORAGE_TEXTURE else "in%d(modified)" % slot) + ":" + res[1].name + ("[%d]" % res[2] if "POOL" in res[1].name else "")
            stats.add(d.shader, label, fmt, w, mine.copy(), theirs)

    _drive(name, width, height, frames, lambda run: oracle_driver.StrictRecordingExecutor(run.inst, width, height, api.FORMAT_BYTES, on_pass=on_pass))
    assert state["k"] == len(dispatches), "the sequence has %d dispatches, the fixture %d" % (state["k"], len(dispatches))
    return stats, state["k"], state["first_mismatch"]
