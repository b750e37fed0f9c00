"""nrdHipMeasureMotionRows (include/NRDHip.h): the on-device motion bound of the multi-GPU contract, against a per-pixel float64 re-projection written from the
CommonSettings matrices alone (no library constants involved). CPU: the device source compiled by tests/emu; GPU: the same checks through lib/libNRD_hip.so."""
import ctypes as C

import numpy as np
import pytest
import torch

import parity
from raytracingdenoiser_amd import api, scene

RT = api.ResourceType
W, H = 192, 128


def _expected_world_space_rows(cs, viewz, row_begin, row_end, denoising_range):
    """static geometry (world-space motion vectors of zero): un-project every pixel with the current camera, re-project with the previous one"""
    def mat(m):
        return np.array(list(m), dtype=np.float64).reshape(4, 4).T

    P, Pp, V, Vp = mat(cs.viewToClipMatrix), mat(cs.viewToClipMatrixPrev), mat(cs.worldToViewMatrix), mat(cs.worldToViewMatrixPrev)
    h, w = viewz.shape
    sign = 1.0 if P[3, 2] > 0 else -1.0
    ys, xs = np.meshgrid(np.arange(h) + 0.5, np.arange(w) + 0.5, indexing="ij")
    u, v = xs / w, ys / h
    nx, ny = u * 2.0 - 1.0, -(v * 2.0 - 1.0)
    zv = sign * np.abs(viewz.astype(np.float64))
    cw = P[3, 2] * zv + P[3, 3]
    view = np.stack([(nx * cw - P[0, 2] * zv - P[0, 3]) / P[0, 0], (ny * cw - P[1, 2] * zv - P[1, 3]) / P[1, 1], zv, np.ones_like(zv)]).reshape(4, -1)
    clip = (Pp @ Vp @ np.linalg.inv(V)) @ view
    v_prev = (clip[1] / clip[3] * -0.5 + 0.5).reshape(h, w)
    rows = np.abs(v_prev - v) * float(cs.rectSizePrev[1])
    rows[np.abs(viewz) > denoising_range] = 0.0
    return float(rows[row_begin:row_end].max()) if row_end > row_begin else 0.0


class _Harness:
    """one denoiser instance + executor (emulated or real), frame inputs bound, this frame's dispatch list ready -- NOT executed"""

    def __init__(self, name, backend, cs_kw=None, frame_edit=None):
        self.seq = scene.generate_sequence(name, W, H, 2, device="cpu")
        frame = self.seq[1]
        if frame_edit:
            frame_edit(frame)
        if backend == "emu":
            from emu import emu_run

            self.inst = api.Instance([(0, scene.DENOISERS[name][0])], lib=emu_run.load())
            self.ex = emu_run.EmuExecutor(self.inst, W, H)
            self.keep = [np.array(t.numpy(), copy=True, order="C") for _, t, _ in scene.user_planes(name, frame)]
        else:
            from raytracingdenoiser_amd.executor import HipExecutor

            self.inst = api.Instance([(0, scene.DENOISERS[name][0])])
            self.ex = HipExecutor(self.inst, W, H)
            self.keep = [t.cuda().contiguous() for _, t, _ in scene.user_planes(name, frame)]
        for (rt, _, fmt), arr in zip(scene.user_planes(name, frame), self.keep):
            self.ex.bind(rt, arr, fmt)
        self.viewz = frame["viewz"].numpy().reshape(H, W)
        self.cs = scene.common_settings(frame["camera"], self.seq[0]["camera"], W, H, 1, **(cs_kw or {}))
        assert self.inst.set_denoiser_settings(0, scene.denoiser_settings(name, frame, None)) == api.Result.SUCCESS
        assert self.inst.set_common_settings(self.cs) == api.Result.SUCCESS
        r, self.ptr, self.n = self.inst.get_compute_dispatches_raw()
        assert r == api.Result.SUCCESS and self.n > 0

    def measure(self, row_begin=0, row_end=H):
        out = C.c_float(-1.0)
        r = self.inst.lib.nrdHipMeasureMotionRows(self.ex.handle, C.cast(self.ptr, C.c_void_p), self.n, row_begin, row_end, C.byref(out))
        assert api.Result(r) == api.Result.SUCCESS, self.inst.lib.nrdHipGetLastError(self.ex.handle)
        return out.value


def _check_world_space(name, backend):
    h = _Harness(name, backend)
    rng = float(h.cs.denoisingRange)
    for rows in ((0, H), (0, H // 2), (H // 2, H), (37, 38), (50, 50)):
        want = _expected_world_space_rows(h.cs, h.viewz, rows[0], rows[1], rng)
        got = h.measure(*rows)
        assert got == pytest.approx(want, rel=2e-3, abs=2e-3), (name, rows, got, want)
    assert h.measure() > 0.05  # the sequence's camera moves
    if backend == "hip":
        # nrdHipMeasureMotionRowsAsync (round 6): the same kernel, the result left in the caller's device memory in stream order -- equal to the synchronous form bit for bit
        word = torch.full((1,), -1.0, dtype=torch.float32, device="cuda")
        for rows in ((0, H), (H // 2, H), (50, 50)):
            h.ex.measure_motion_rows_async(h.ptr, h.n, rows[0], rows[1], word)
            torch.cuda.synchronize()
            assert float(word.item()) == h.measure(*rows), rows
    assert h.measure(0, 10 * H) == h.measure()  # rowEnd is clamped to the rect


def _check_screen_space(name, backend):
    """2D motion vectors (the reference's default convention): mv.xy * motionVectorScale.xy is a uv offset"""
    def edit(frame):
        mv = torch.zeros_like(frame["mv"])
        mv[..., 1] = 1.5
        frame["viewz"] = frame["viewz"].clone()
        z = frame["viewz"].view(H, W)
        z[40, 17], z[100, 3], z[5, 5] = 7.0, 3.0, 1.0e6  # geometry, geometry, sky
        mv[40, 17, 1] = -9.25  # the extreme of rows [32, 64)
        mv[100, 3, 1] = 21.0   # the extreme of the frame
        mv[100, 3, 0] = 500.0  # horizontal motion does not count
        mv[5, 5, 1] = 90.0     # sky: not denoised, never reprojected
        frame["mv"] = mv

    h = _Harness(name, backend, cs_kw=dict(isMotionVectorInWorldSpace=False, motionVectorScale=(1.0 / W, 1.0 / H, 0.0)), frame_edit=edit)
    assert h.measure() == pytest.approx(21.0, rel=1e-3)
    assert h.measure(32, 64) == pytest.approx(9.25, rel=1e-3)
    assert h.measure(0, 32) == pytest.approx(1.5 if float(np.abs(h.viewz[:32]).min()) < float(h.cs.denoisingRange) else 0.0, rel=1e-3)
    assert h.measure(100, 101) == pytest.approx(21.0, rel=1e-3)


def _check_errors(backend):
    h = _Harness("REBLUR_DIFFUSE", backend)
    out = C.c_float()
    lib = h.inst.lib
    assert api.Result(lib.nrdHipMeasureMotionRows(None, C.cast(h.ptr, C.c_void_p), h.n, 0, H, C.byref(out))) == api.Result.INVALID_ARGUMENT
    assert api.Result(lib.nrdHipMeasureMotionRows(h.ex.handle, C.cast(h.ptr, C.c_void_p), h.n, 0, H, None)) == api.Result.INVALID_ARGUMENT
    assert api.Result(lib.nrdHipMeasureMotionRows(h.ex.handle, None, 0, 0, H, C.byref(out))) == api.Result.SUCCESS and out.value == 0.0  # empty list: nothing reprojects


@pytest.mark.parametrize("name", ["REBLUR_DIFFUSE_SPECULAR", "RELAX_DIFFUSE_SPECULAR", "SIGMA_SHADOW"])
def test_emulated_motion_rows_world_space(name):
    _check_world_space(name, "emu")


@pytest.mark.parametrize("name", ["REBLUR_DIFFUSE", "RELAX_DIFFUSE"])
def test_emulated_motion_rows_screen_space(name):
    _check_screen_space(name, "emu")


def test_emulated_motion_rows_argument_errors():
    _check_errors("emu")


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["REBLUR_DIFFUSE_SPECULAR", "RELAX_DIFFUSE_SPECULAR", "SIGMA_SHADOW"])
def test_motion_rows_world_space(name):
    _check_world_space(name, "hip")


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["REBLUR_DIFFUSE", "RELAX_DIFFUSE"])
def test_motion_rows_screen_space(name):
    _check_screen_space(name, "hip")
    _check_errors("hip")


# ---- nrdHipSetHistoryReachWord (round 6): the temporal passes REPORT how far from its own row a pixel read last frame's planes -------------------------------------------
def _check_history_reach(name, backend):
    """2D motion vectors of 1.5 rows everywhere, 21 rows at one pixel of geometry, 90 rows at a sky pixel: the word holds 21 after the frame (a diffuse / shadow signal has no virtual
    motion: its reach IS the surface motion, and equals what nrdHipMeasureMotionRows measured on the inputs); a specular signal reports at least that; nothing is reported
    without a word, and the outputs do not depend on the tracking"""
    if backend == "emu":
        from emu.emu_run import EmuRun as Run
    else:
        Run = parity.GpuRun
    seq = parity.generate_sequence(name, W, H, 3, device="cpu")
    frame = dict(seq[2])
    mv = torch.zeros_like(frame["mv"])
    mv[..., 1] = 1.5
    z = frame["viewz"].clone()
    zv = z.view(H, W)
    zv[100, 3], zv[5, 5] = 3.0, 1.0e6  # geometry, sky
    mv[100, 3, 1] = 21.0
    mv[5, 5, 1] = 90.0
    frame["mv"], frame["viewz"] = mv, z
    cs_kw = dict(isMotionVectorInWorldSpace=False, motionVectorScale=(1.0 / W, 1.0 / H, 0.0))
    outs = []
    for tracked in (False, True):
        run = Run(name, W, H)
        word = (np.zeros(1, dtype=np.float32) if backend == "emu" else torch.zeros(1, dtype=torch.float32, device="cuda")) if tracked else None
        for f in range(3):
            fr = frame if f == 2 else seq[f]
            cam, cam_prev = fr["camera"], seq[max(f - 1, 0)]["camera"]
            if f == 2 and tracked:
                run.ex.set_history_reach_word(word)
            run.step(fr, parity.common_settings(cam, cam_prev, W, H, f, **(cs_kw if f == 2 else {})), parity.denoiser_settings(name, fr, None))
        outs.append({rt: run.output(rt) for rt in run.outs})
        if tracked:
            reach = float(word[0] if backend == "emu" else word.item())
            if "SPECULAR" in name:
                assert reach >= 20.99, reach
            else:
                assert abs(reach - 21.0) < 0.01, reach
            run.ex.set_history_reach_word(None)
    for rt in outs[0]:
        assert np.array_equal(outs[0][rt], outs[1][rt], equal_nan=True), rt


@pytest.mark.parametrize("name", ["REBLUR_DIFFUSE", "RELAX_DIFFUSE", "SIGMA_SHADOW", "REBLUR_DIFFUSE_SPECULAR", "RELAX_DIFFUSE_SPECULAR"])
def test_emulated_history_reach_word(name):
    _check_history_reach(name, "emu")


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["REBLUR_DIFFUSE", "RELAX_DIFFUSE", "SIGMA_SHADOW", "REBLUR_DIFFUSE_SPECULAR", "RELAX_DIFFUSE_SPECULAR_SH"])
def test_history_reach_word(name):
    """the device form of TrackHistoryReach: ballot + v_readlane over the active lanes, one atomicMax per wave"""
    _check_history_reach(name, "hip")
