"""Frames of other kinds in MID-sequence, device against oracle bit for bit after every frame: history restarts with and without clears (CommonSettings::accumulationMode), a camera cut,
split screen switched on and off, dynamic-resolution steps, and denoiser settings that change the pass list from one frame to the next (blur radii, anti-firefly, hit-distance
reconstruction, performance mode, a-trous iterations, SIGMA without stabilisation) -- what exercises the executor's per-list state: ping-pong indices, the guide caches, the plans and,
on the GPU, the hipGraphs cached per launch topology (graph mode rebuilds or re-parametrises them as the list changes). The scenarios are those of the multi-GPU tests
(tests/test_sharding.py _scenarios). CPU: the device sources compiled by tests/emu; GPU: lib/libNRD_hip.so, eager and graph mode."""
import numpy as np
import pytest

import parity
from raytracingdenoiser_amd import api, synth
from test_sharding import DYNRES_STEPS, _scenarios

RESOURCE = (192, 128)
NAMES = ["REBLUR_DIFFUSE_SPECULAR", "RELAX_DIFFUSE_SPECULAR", "SIGMA_SHADOW", "REBLUR_DIFFUSE_SPECULAR_SH", "RELAX_DIFFUSE_SPECULAR_SH"]


def _scenarios_of(name):
    base = name if name in ("REBLUR_DIFFUSE_SPECULAR", "RELAX_DIFFUSE_SPECULAR", "SIGMA_SHADOW") else ("REBLUR_DIFFUSE_SPECULAR" if name.startswith("REBLUR") else "RELAX_DIFFUSE_SPECULAR")
    sc = dict(_scenarios(base))
    sc.pop("shifted_rect")  # (tests/test_dynamic_resolution.py holds the shifted rect against the oracle)
    sc["dynamic_resolution"] = [dict(scale=s) for s in DYNRES_STEPS]
    return sc


def _run(name, scenario, make_device, graph=False):
    RW, RH = RESOURCE
    sizes = [(int(RW * e.get("scale", (1.0, 1.0))[0]), int(RH * e.get("scale", (1.0, 1.0))[1])) for e in scenario]
    raw = [synth.render_frame(*sizes[f], scenario[f].get("camera", f), want=tuple(parity.DENOISERS[name][1]), device="cpu") for f in range(len(sizes))]
    dev, ora = make_device(name, RW, RH), parity.OracleRun(name, RW, RH)
    if graph:
        dev.ex.set_graph_mode(True)
    for f in range(len(sizes)):
        frame = parity.embed_in_resource(raw[f], RESOURCE)
        w, h = sizes[f]
        kw = dict(resourceSize=RESOURCE, resourceSizePrev=RESOURCE, rectSize=(w, h), rectSizePrev=sizes[max(f - 1, 0)])
        kw.update(scenario[f].get("cs") or {})
        settings = parity.denoiser_settings(name, frame, dict(scenario[f].get("settings") or {}))
        for run in (dev, ora):
            run.step(frame, parity.common_settings(raw[f]["camera"], raw[max(f - 1, 0)]["camera"], w, h, f, **kw), settings)
        for rt in ora.outs:
            got = dev.output(rt)
            got = got.cpu().numpy() if hasattr(got, "cpu") else got
            assert np.array_equal(np.asarray(got, dtype=np.float32), ora.output(rt), equal_nan=True), (name, "frame", f, api.ResourceType(rt).name)


@pytest.mark.parametrize("name", NAMES)
def test_emulated_device_equals_the_oracle_through_mid_sequence_events(name):
    from emu.emu_run import EmuRun

    for kind, scenario in _scenarios_of(name).items():
        _run(name, scenario, EmuRun)


@pytest.mark.gpu
@pytest.mark.parametrize("graph", [False, True], ids=["eager", "graph"])
@pytest.mark.parametrize("name", NAMES)
def test_device_equals_the_oracle_through_mid_sequence_events(name, graph):
    for kind, scenario in _scenarios_of(name).items():
        _run(name, scenario, parity.GpuRun, graph=graph)
