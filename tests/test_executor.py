"""Executor behaviour that is not arithmetic: all-or-nothing pre-flight of a dispatch list, HIP-graph execution, the per-list guide cache."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

import parity
from raytracingdenoiser_amd import api

RT = parity.RT


def _run(name, frames, graph, w=192, h=128, profile_every=None):
    seq = parity.generate_sequence(name, w, h, frames)
    hip = parity.HipRun(name, w, h)
    hip.ex.set_graph_mode(graph)
    outs = []
    for f, frame in enumerate(seq):
        cs = parity.common_settings(frame["camera"], seq[max(f - 1, 0)]["camera"], w, h, f)
        hip.step(frame, cs, parity.denoiser_settings(name, frame))
        outs.append({rt: hip.output(rt).copy() for rt in hip.outs})
    pools = [hip.ex.read_pool_plane(RT.PERMANENT_POOL, i)[0].copy() for i in range(len(hip.inst.permanent_pool))]
    return outs, pools, hip.ex.graph_stats()


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["REBLUR_DIFFUSE_SPECULAR", "RELAX_DIFFUSE_SPECULAR_SH", "SIGMA_SHADOW", "REFERENCE_LIKE_REBLUR_DIFFUSE"])
def test_graph_mode_is_bit_identical_to_eager(name):
    name = "REBLUR_DIFFUSE" if name == "REFERENCE_LIKE_REBLUR_DIFFUSE" else name
    eager, eager_pools, st0 = _run(name, 6, graph=False)
    graph, graph_pools, st1 = _run(name, 6, graph=True)
    assert st0 == (0, 0, 0)
    launches, builds, updates = st1
    assert launches == 6 and 1 <= builds <= 3 and updates > 0, st1  # frame 0 (clears) has its own topology; afterwards only node parameters change
    for a, b in zip(eager, graph):
        for rt in a:
            assert np.array_equal(a[rt], b[rt]), rt
    for a, b in zip(eager_pools, graph_pools):
        assert np.array_equal(a, b)


@pytest.mark.gpu
def test_unsupported_dispatch_launches_nothing():
    """While one pass of a frame's list cannot run (here: a launcher-level rejection in the MIDDLE of the list), nrdHipDenoise must fail before anything
    is enqueued (reference Integration::Denoise contract) -- outputs and history stay untouched"""
    name, w, h = "REBLUR_DIFFUSE_SPECULAR", 192, 128
    seq = parity.generate_sequence(name, w, h, 3)
    hip = parity.HipRun(name, w, h)
    for f in range(2):
        cs = parity.common_settings(seq[f]["camera"], seq[max(f - 1, 0)]["camera"], w, h, f)
        hip.step(seq[f], cs, parity.denoiser_settings(name, seq[f]))
    before = {rt: hip.output(rt).copy() for rt in hip.outs}
    pools = [hip.ex.read_pool_plane(RT.PERMANENT_POOL, i)[0].copy() for i in range(len(hip.inst.permanent_pool))]
    # an orthographic projection is rejected by the REBLUR launchers themselves (a launcher-level check, in the middle of the list)
    cs = parity.common_settings(seq[2]["camera"], seq[1]["camera"], w, h, 2)
    for i in range(16):
        cs.viewToClipMatrix[i] = [1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 0.01, 0, 0, 0, 0, 1][i]  # w = 1: orthographic
    with pytest.raises(RuntimeError) as err:
        hip.step(seq[2], cs, parity.denoiser_settings(name, seq[2]))
    assert "nothing was launched" in str(err.value)
    torch.cuda.synchronize()
    for rt in before:
        assert np.array_equal(before[rt], hip.output(rt))
    for i, p in enumerate(pools):
        assert np.array_equal(p, hip.ex.read_pool_plane(RT.PERMANENT_POOL, i)[0])


@pytest.mark.gpu
def test_range_without_first_range_still_decodes_guides():
    """a dispatch range with first > 0 that was not preceded by first == 0 on the same list must not read stale decoded guides"""
    name, w, h = "REBLUR_DIFFUSE", 192, 128
    seq = parity.generate_sequence(name, w, h, 2)

    def run(split):
        hip = parity.HipRun(name, w, h)
        for f, frame in enumerate(seq):
            for rt, t, fmt in parity.user_planes(name, frame):
                t = t.cuda().clone().contiguous()
                hip.inputs[rt] = t
                hip.ex.bind(rt, t, fmt)
            hip.inst.set_denoiser_settings(0, parity.denoiser_settings(name, frame))
            hip.inst.set_common_settings(parity.common_settings(frame["camera"], seq[max(f - 1, 0)]["camera"], w, h, f))
            r, ptr, num = hip.inst.get_compute_dispatches_raw()
            assert r == api.Result.SUCCESS
            if split and f == 1:
                # the tile classification (dispatch 0) reads no guides; start the list at dispatch 1 after running dispatch 0 through ANOTHER call path
                hip.ex.execute_range(ptr, num, 0, 1)
                hip.ex.bind(RT.IN_NORMAL_ROUGHNESS, hip.inputs[RT.IN_NORMAL_ROUGHNESS], parity.F.R10_G10_B10_A2_UNORM)  # rebind: invalidates the cache
                hip.ex.execute_range(ptr, num, 1, num - 1)
            else:
                hip.ex.execute_raw(ptr, num)
        return {rt: hip.output(rt).copy() for rt in hip.outs}

    a, b = run(False), run(True)
    for rt in a:
        assert np.array_equal(a[rt], b[rt])


def test_one_library_one_arithmetic():
    """round 2 shipped a "fast" and an "exact" build; there is ONE library now and it reports the pinned arithmetic (mode 0)"""
    lib = api.load_library()
    assert lib.nrdHipGetNumericsMode() == 0
    assert api.load_library() is lib and not os.path.exists(os.path.join(os.path.dirname(api.LIB_PATH), "libNRD_hip_exact.so"))


@pytest.mark.gpu
def test_mixed_reblur_and_relax_instance_matches_the_oracle():
    """ONE instance with a REBLUR and a RELAX denoiser (a legal combination in the reference: REBLUR_DIFFUSE + RELAX_SPECULAR in one
    GetComputeDispatches list). The executor writes BOTH per-frame guide planes (view position for the REBLUR taps, world position for the RELAX taps);
    outputs of both denoisers must equal the oracle's bit for bit."""
    from oracle import driver as oracle_driver

    w, h, frames = 192, 128, 4
    denoisers = [(0, api.Denoiser.REBLUR_DIFFUSE), (1, api.Denoiser.RELAX_SPECULAR)]
    emu = os.environ.get("NRD_PARITY_BACKEND") == "emu"
    if emu:
        from emu import emu_run

        lib = emu_run.load()
    else:
        lib = api.load_library()
    seq = [parity.synth.render_frame(w, h, f, want=("reblur", "relax")) for f in range(frames)]
    inst_o, inst_d = api.Instance(denoisers), api.Instance(denoisers, lib=lib)
    ora = oracle_driver.OracleExecutor(inst_o, w, h, api.FORMAT_BYTES)
    if emu:
        dev = emu_run.EmuExecutor(inst_d, w, h)
        to_dev, to_host = (lambda a: np.array(a, copy=True, order="C")), (lambda a: a)
    else:
        from raytracingdenoiser_amd.executor import HipExecutor

        dev = HipExecutor(inst_d, w, h)
        to_dev, to_host = (lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()), (lambda t: t.cpu().numpy())
    outs = {}
    for rt in (RT.OUT_DIFF_RADIANCE_HITDIST, RT.OUT_SPEC_RADIANCE_HITDIST):
        a, b = np.zeros((h, w, 4), np.float16), to_dev(np.zeros((h, w, 4), np.float16))
        ora.bind(rt, a, parity.F.RGBA16_SFLOAT)
        dev.bind(rt, b, parity.F.RGBA16_SFLOAT)
        outs[rt] = (a, b)
    for f, frame in enumerate(seq):
        planes = [(RT.IN_MV, frame["mv"], parity.F.RGBA16_SFLOAT), (RT.IN_NORMAL_ROUGHNESS, frame["normal_roughness"], parity.F.R10_G10_B10_A2_UNORM), (RT.IN_VIEWZ, frame["viewz"], parity.F.R32_SFLOAT),
                  (RT.IN_DIFF_RADIANCE_HITDIST, frame["diff"], parity.F.RGBA16_SFLOAT), (RT.IN_SPEC_RADIANCE_HITDIST, frame["spec_relax"], parity.F.RGBA16_SFLOAT)]
        keep = []
        for rt, t, fmt in planes:
            a = np.array(t.numpy(), copy=True, order="C")
            b = to_dev(a)
            keep.append((a, b))
            ora.bind(rt, a, fmt)
            dev.bind(rt, b, fmt)
        for inst in (inst_o, inst_d):
            assert inst.set_denoiser_settings(0, api.ReblurSettings()) == api.Result.SUCCESS
            assert inst.set_denoiser_settings(1, api.RelaxSettings()) == api.Result.SUCCESS
            assert inst.set_common_settings(parity.common_settings(frame["camera"], seq[max(f - 1, 0)]["camera"], w, h, f)) == api.Result.SUCCESS
        r, ds = inst_o.get_compute_dispatches()
        assert r == api.Result.SUCCESS and any(d.shader.startswith("REBLUR_") for d in ds) and any(d.shader.startswith("RELAX_") for d in ds)
        ora.execute(ds)
        dev.denoise()
        for rt, (a, b) in outs.items():
            got = to_host(b)
            assert np.array_equal(got.view(np.uint16), a.view(np.uint16)), (f, rt, int((got.view(np.uint16) != a.view(np.uint16)).sum()))


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["REBLUR_DIFFUSE_SPECULAR", "REBLUR_DIFFUSE_SPECULAR_SH", "RELAX_DIFFUSE_SPECULAR_SH", "RELAX_DIFFUSE"])
def test_temporal_accumulation_window_and_fallback_kernels_match_the_oracle(name):
    """REBLUR / RELAX TemporalAccumulation runs as a window kernel (the surface-motion footprints of a tile come from one LDS-staged rectangle of the previous frame)
    plus the plain kernel for the tiles whose rectangle does not fit. On the test scenes every tile fits, so the plain kernel would never run: the test hook
    NRD_HIP_TA_WINDOW_LIMIT shrinks the accepted rectangle (read once per process, hence the subprocess) until a good part of the tiles takes each path --
    all outputs and pool planes must still equal the oracle's bit for bit, and with NRD_HIP_TA_WINDOW=0 (plain kernel only) as well."""
    import re
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys; sys.path[:0] = [%r, %r]; import parity; "
            "print('worst', parity.run_parity(%r, width=256, height=160, frames=5, verbose=True))") % (root, os.path.join(root, "tests"), name)
    fallback = {}
    on = {"NRD_HIP_RELAX_TA_WINDOW": "1"}  # (the RELAX window kernel is opt-in: it was measured not to pay at 4K; REBLUR's is on by default)
    for tag, extra in (("default", on), ("limited", dict(on, NRD_HIP_TA_WINDOW_LIMIT="35x11")), ("off", {"NRD_HIP_TA_WINDOW": "0", "NRD_HIP_RELAX_TA_WINDOW": "0"})):
        out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **extra), capture_output=True, text=True, timeout=1200)
        assert out.returncode == 0, out.stderr[-2000:]
        assert float(re.search(r"worst ([0-9.eE+-]+)", out.stdout).group(1)) == 0.0, (tag, out.stdout[-2000:])
        fallback[tag] = [int(m) for m in re.findall(r"tiles left to a fallback kernel: (\d+) of", out.stdout)]
    assert sum(fallback["default"]) == 0 and sum(fallback["off"]) == 0, fallback  # everything fits / the flags are not touched without the window kernel
    assert all(n > 0 for n in fallback["limited"]), fallback                      # both kernels had tiles in every frame


@pytest.mark.gpu
def test_executor_runs_an_instance_created_under_the_reference_quirks_switch():
    """NRD_HIP_REFERENCE_QUIRKS=1: REBLUR_DIFFUSE_SPECULAR_SH describes the reference's 11 transient textures and its dispatches name a full-resolution RGBA16F texture as the
    tile map (Reblur_DiffuseSpecularSh.hpp:61-85); the executor binds the real tile plane there (InstanceImpl::TransientAlias). Same inputs, same outputs, bit for bit, as the
    instance without the switch -- whose outputs the parity suite holds against the oracle."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys, hashlib; sys.path[:0] = [%r, %r]; import parity; name = 'REBLUR_DIFFUSE_SPECULAR_SH'; w, h = 176, 112; "
            "seq = parity.generate_sequence(name, w, h, 4); run = parity.HipRun(name, w, h); digest = hashlib.sha1(); "
            "[(run.step(fr, parity.common_settings(fr['camera'], seq[max(f - 1, 0)]['camera'], w, h, f), parity.denoiser_settings(name, fr, None)), "
            "  [digest.update(run.output(rt).tobytes()) for rt in sorted(run.outs, key=int)]) for f, fr in enumerate(seq)]; "
            "print('planes', len(run.inst.transient_pool), 'digest', digest.hexdigest())") % (root, os.path.join(root, "tests"))
    seen = {}
    for switch in ("0", "1"):
        out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, NRD_HIP_REFERENCE_QUIRKS=switch), capture_output=True, text=True, timeout=1200)
        assert out.returncode == 0, (switch, out.stderr[-2000:])
        words = out.stdout.split()
        seen[switch] = (int(words[words.index("planes") + 1]), words[words.index("digest") + 1])
    assert seen["0"][0] == 10 and seen["1"][0] == 11, seen
    assert seen["0"][1] == seen["1"][1], seen


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["RELAX_DIFFUSE_SPECULAR_SH", "RELAX_SPECULAR"])
def test_atrous_tap_sources_match_the_oracle(name):
    """The a-trous iterations read their taps from LDS tiles (steps 2, 4), LDS bands (step 8) or global gathers with 16 x 4-pixel waves (step 16 and beyond) by default;
    NRD_HIP_ATROUS_MARCH=16 selects the marching kernel of round 6 for steps 8 and 16 (an LDS ring filled by LDS-DMA: measured slower, kept as an A/B switch), with segments of
    2 steps so that complete refills, prefetched rows and the first / last stripe's clamped columns all occur; NRD_HIP_ATROUS_LDS=0 + NRD_HIP_ATROUS_BANDS=0 gathers everything.
    One arithmetic: every variant equals the oracle bit for bit (7 iterations: steps up to 64; a size with partial stripes, steps and tiles)."""
    import re
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys; sys.path[:0] = [%r, %r]; import parity; "
            "print('worst', parity.run_parity(%r, width=211, height=117, frames=3, settings_overrides=dict(atrousIterationNum=7)))") % (root, os.path.join(root, "tests"), name)
    for tag, extra in (("default", {}), ("march", {"NRD_HIP_ATROUS_MARCH": "16", "NRD_HIP_ATROUS_MARCH_SEG": "2"}), ("bands16", {"NRD_HIP_ATROUS_BANDS": "16"}),
                       ("gathers", {"NRD_HIP_ATROUS_LDS": "0", "NRD_HIP_ATROUS_BANDS": "0"})):
        out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **extra), capture_output=True, text=True, timeout=1200)
        assert out.returncode == 0, (tag, out.stderr[-2000:])
        assert float(re.search(r"worst ([0-9.eE+-]+)", out.stdout).group(1)) == 0.0, (tag, out.stdout[-2000:])


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["REBLUR_DIFFUSE_SPECULAR", "RELAX_DIFFUSE_SPECULAR_SH"])
def test_guide_planes_from_the_classification_kernel_or_from_their_own_kernel(name):
    """The per-frame guide planes (decoded normals + view / world position) are written by the tile-classification kernel of the list (one launch, IN_VIEWZ read once:
    kernels_common.hip DecodeGuidesClassifyKernel) or, with NRD_HIP_FUSE_CLASSIFY=0 (and always under row sharding / a shifted rect), by a kernel of their own in front of
    the first pass. Both ways every output and pool plane equals the oracle's bit for bit -- on a size with partial tiles on both edges, the first frame being a
    CLEAR_AND_RESTART list (clears in front of the classification)."""
    import re
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys; sys.path[:0] = [%r, %r]; import parity; "
            "print('worst', parity.run_parity(%r, width=250, height=150, frames=3))") % (root, os.path.join(root, "tests"), name)
    for switch, fused_frames in (("1", 3), ("0", 0)):
        out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, NRD_HIP_FUSE_CLASSIFY=switch, NRD_HIP_TRACE_GUIDE_ROWS="1"), capture_output=True, text=True, timeout=1200)
        assert out.returncode == 0, out.stderr[-2000:]
        assert float(re.search(r"worst ([0-9.eE+-]+)", out.stdout).group(1)) == 0.0, (switch, out.stdout[-2000:])
        assert out.stderr.count("guide planes written by the tile classification kernel") == fused_frames, (switch, out.stderr[-2000:])
