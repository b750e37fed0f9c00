"""Parity harness: runs the same synthetic frame sequence through the CPU oracle and through the HIP path (both driven
by the SAME nrd::GetComputeDispatches lists from the product's host) and compares planes. Used by tests/ and smoke()."""
import os

import numpy as np
import torch

from oracle import driver as oracle_driver
from raytracingdenoiser_amd import api, synth

RT = api.ResourceType
F = api.Format

REL_TOL = 1e-3  # BASELINE.json north_star: <= 1e-3 relative per pixel (bit-exact for REFERENCE)


# the workload description (planes, settings, frame generator) lives in the package: bench.py uses it without touching tests/ or the oracle
from raytracingdenoiser_amd.scene import (DENOISERS, _relax_signals, checkerboard_pack, common_settings, denoiser_settings, embed_in_resource, generate_sequence,  # noqa: E402,F401
                                          output_planes, tag_checkerboard, user_planes)


def decode_plane(raw, fmt, width):
    """uint8 [h, pitch] -> float32 / uint32 array [h, w, c] of texel values (for error metrics)."""
    h = raw.shape[0]
    bpt = api.FORMAT_BYTES[fmt]
    body = np.ascontiguousarray(raw[:, : width * bpt])
    if fmt == F.RGBA16_SFLOAT:
        return body.view(np.float16).reshape(h, width, 4).astype(np.float32)
    if fmt == F.R16_SFLOAT:
        return body.view(np.float16).reshape(h, width, 1).astype(np.float32)
    if fmt == F.R32_SFLOAT:
        return body.view(np.float32).reshape(h, width, 1)
    if fmt == F.RGBA32_SFLOAT:
        return body.view(np.float32).reshape(h, width, 4)
    if fmt in (F.R8_UNORM, F.R8_UINT):
        return body.reshape(h, width, 1).astype(np.float32)
    if fmt == F.RG8_UNORM:
        return body.reshape(h, width, 2).astype(np.float32)
    if fmt == F.RGBA8_UNORM:
        return body.reshape(h, width, 4).astype(np.float32)
    if fmt in (F.R16_UINT, F.R16_UNORM):
        return body.view(np.uint16).reshape(h, width, 1).astype(np.float32)
    if fmt == F.RGBA16_SNORM:
        return body.view(np.int16).reshape(h, width, 4).astype(np.float32)
    if fmt == F.RGBA8_SNORM:  # (PREV_NORMAL_ROUGHNESS of NRD_NORMAL_ENCODING 1)
        return body.view(np.int8).reshape(h, width, 4).astype(np.float32)
    if fmt == F.RGBA16_UNORM:  # (NRD_NORMAL_ENCODING 3)
        return body.view(np.uint16).reshape(h, width, 4).astype(np.float32)
    if fmt in (F.R32_UINT, F.R10_G10_B10_A2_UNORM):
        return body.view(np.uint32).reshape(h, width, 1).astype(np.float64)
    raise KeyError(fmt)


def error_stats(got, want, floor=1e-3, tol=REL_TOL):
    """tolerance statistics of one plane: max / mean relative error (same definition as rel_error), the fraction of values above tol, the 99.9th
    percentile, and the position of the worst value"""
    got = got.astype(np.float64)
    want = want.astype(np.float64)
    both_nan = np.isnan(got) & np.isnan(want)
    err = np.abs(got - want) / np.maximum(np.abs(want), floor)
    err = np.where(both_nan, 0.0, err)
    err = np.where(np.isnan(err), np.inf, err)
    if not err.size:
        return {"max": 0.0, "mean": 0.0, "frac_gt_tol": 0.0, "frac_gt_tol_vec": 0.0, "p999": 0.0, "n": 0, "worst_at": None, "bit_exact_frac": 1.0}
    flat = int(np.argmax(err))
    finite = np.where(np.isfinite(err), err, 1e30)
    err_vec = err
    if got.ndim == 3 and got.shape[-1] == 4:
        # colour / direction texels (tests/ref_parity.py uses the same metric per pass): the first three channels relative to the texel's largest of them -- a YCoCg chroma or
        # SH1 direction component that cancels to ~0 inherits the absolute error of its neighbours --, the fourth (hit distance, variance, ...) relative to itself
        scale = np.maximum(np.max(np.abs(want[..., :3]), axis=-1, keepdims=True), floor)
        err_vec = np.concatenate([np.abs(got - want)[..., :3] / scale, err[..., 3:]], axis=-1)
        err_vec = np.where(np.isnan(err_vec), np.where(both_nan, 0.0, np.inf), err_vec)
    return {"max": float(err.max()), "mean": float(finite.mean()), "frac_gt_tol": float(np.mean(err > tol)), "frac_gt_tol_vec": float(np.mean(err_vec > tol)),
            "p999": float(np.quantile(finite, 0.999)), "n": int(err.size),
            "worst_at": [int(v) for v in np.unravel_index(flat, err.shape)], "bit_exact_frac": float(np.mean((got == want) | both_nan))}


class ParityStats:
    """accumulates error_stats over the frames of a run, per plane name"""

    def __init__(self):
        self.planes = {}

    def add(self, name, frame, st):
        p = self.planes.setdefault(name, {"max": 0.0, "frac_gt_tol": 0.0, "frac_gt_tol_vec": 0.0, "p999": 0.0, "mean": 0.0, "frames": 0, "worst": None, "bit_exact_frac": 1.0})
        if st["max"] >= p["max"]:
            p["worst"] = (frame, st["worst_at"], st["max"])
        p["max"] = max(p["max"], st["max"])
        p["frac_gt_tol"] = max(p["frac_gt_tol"], st["frac_gt_tol"])  # the worst frame
        p["frac_gt_tol_vec"] = max(p["frac_gt_tol_vec"], st.get("frac_gt_tol_vec", st["frac_gt_tol"]))
        p["p999"] = max(p["p999"], st["p999"])
        p["mean"] = max(p["mean"], st["mean"])
        p["bit_exact_frac"] = min(p["bit_exact_frac"], st["bit_exact_frac"])
        p["frames"] += 1

    def outputs(self):
        return {k: v for k, v in self.planes.items() if k.startswith("OUT_")}

    def summary(self, only_outputs=True):
        sel = self.outputs() if only_outputs else self.planes
        if not sel:
            return {"max_rel_err": 0.0, "frac_gt_tol": 0.0, "frac_gt_tol_vec": 0.0, "p999": 0.0, "mean": 0.0, "planes": 0}
        return {"max_rel_err": max(v["max"] for v in sel.values()), "frac_gt_tol": max(v["frac_gt_tol"] for v in sel.values()),
                "frac_gt_tol_vec": max(v["frac_gt_tol_vec"] for v in sel.values()), "p999": max(v["p999"] for v in sel.values()),
                "mean": max(v["mean"] for v in sel.values()), "bit_exact_frac": min(v["bit_exact_frac"] for v in sel.values()), "planes": len(sel)}


def rel_error(got, want, floor=1e-3):
    """max over texels of |got - want| / max(|want|, floor); NaNs count as infinite error unless both are NaN."""
    got = got.astype(np.float64)
    want = want.astype(np.float64)
    both_nan = np.isnan(got) & np.isnan(want)
    err = np.abs(got - want) / np.maximum(np.abs(want), floor)
    err = np.where(both_nan, 0.0, err)
    err = np.where(np.isnan(err), np.inf, err)
    return float(err.max()) if err.size else 0.0


class OracleRun:
    def __init__(self, name, width, height, threads=0, validation=False):
        self.name, self.width, self.height = name, width, height
        self.inst = api.Instance([(0, DENOISERS[name][0])])
        self.ex = oracle_driver.OracleExecutor(self.inst, width, height, api.FORMAT_BYTES, threads=threads)
        self.outs = {}
        for rt, dtype, ch, fmt in output_planes(name, width, height, validation):
            arr = np.zeros((height, width, ch), dtype={torch.float16: np.float16, torch.int16: np.uint16 if fmt == F.R16_UNORM else np.int16}.get(dtype, np.uint8))
            self.outs[rt] = (arr, fmt)
            self.ex.bind(rt, arr, fmt)
        self.last_dispatches = []
        self.inputs = {}

    def step(self, frame, cs, settings=None):
        for rt, t, fmt in user_planes(self.name, frame):
            arr = np.array(t.cpu().numpy(), copy=True, order="C")  # a private copy: IN_MV is an in/out plane (REBLUR specular MV modification)
            self.inputs[rt] = arr
            self.ex.bind(rt, arr, fmt)
        if settings is not None:
            assert self.inst.set_denoiser_settings(0, settings) == api.Result.SUCCESS
        assert self.inst.set_common_settings(cs) == api.Result.SUCCESS
        r, ds = self.inst.get_compute_dispatches()
        assert r == api.Result.SUCCESS
        self.last_dispatches = ds
        self.ex.execute(ds)

    def output(self, rt):
        arr, fmt = self.outs[rt]
        return arr.astype(np.float32)


def _padded(t, pad):
    """a [H, W(, C)] CUDA copy of t living inside a wider allocation (row pitch = (W + pad) texels, first texel offset by one row)"""
    shape = list(t.shape)
    big = torch.full([shape[0] + 1, shape[1] + pad] + shape[2:], 77, dtype=t.dtype, device="cuda")
    view = big[1:, : shape[1]]
    view.copy_(t)
    return view


def HipRun(name, width, height, pad=0, validation=False):
    """the device side of a parity run: the GPU through lib/libNRD_hip.so, or -- NRD_PARITY_BACKEND=emu, a developer mode for machines without a GPU -- the
    device sources compiled for the CPU (tests/emu)"""
    if os.environ.get("NRD_PARITY_BACKEND") == "emu":
        from emu.emu_run import EmuRun

        return EmuRun(name, width, height, pad=pad, validation=validation)
    return GpuRun(name, width, height, pad=pad, validation=validation)


class GpuRun:
    def __init__(self, name, width, height, pad=0, validation=False):
        from raytracingdenoiser_amd.executor import HipExecutor

        self.name, self.width, self.height, self.pad = name, width, height, pad
        self.inst = api.Instance([(0, DENOISERS[name][0])])
        self.ex = HipExecutor(self.inst, width, height)
        self.outs = {}
        for rt, dtype, ch, fmt in output_planes(name, width, height, validation):
            t = torch.zeros((height, width, ch), dtype=dtype, device="cuda")
            if pad:
                t = _padded(t, pad)
            self.outs[rt] = (t, fmt)
            self.ex.bind(rt, t, fmt)
        self.inputs = {}

    def step(self, frame, cs, settings=None):
        for rt, t, fmt in user_planes(self.name, frame):
            t = t.cuda().clone().contiguous()  # a private copy: IN_MV is an in/out plane
            t = _padded(t, self.pad) if self.pad else t
            self.inputs[rt] = t
            self.ex.bind(rt, t, fmt)
        if settings is not None:
            assert self.inst.set_denoiser_settings(0, settings) == api.Result.SUCCESS
        assert self.inst.set_common_settings(cs) == api.Result.SUCCESS
        self.ex.denoise()

    def output(self, rt):
        t, fmt = self.outs[rt]
        a = t.cpu().numpy()
        return (a.view(np.uint16) if a.dtype == np.int16 and fmt == F.R16_UNORM else a).astype(np.float32)  # int16 tensors: R16_UNORM bit patterns or SNORM16 values


def run_parity(name, width=192, height=128, frames=4, verbose=False, settings_overrides=None, static_camera=False, check_pools=True, cs_kw=None, extra_want=(), pad=0, resource=None,
               rect_sizes=None, ieee=False, stats=None, static_after=None, graph=False, device="cpu", backend=None):
    """Returns the worst relative error between the HIP path and the oracle over all frames, user outputs and pool planes.
    The library is compared with the oracle that emulates the device's rcp / sqrt / rsqrt / exp2 / log2 instructions: expected error 0, bit for bit.
    ieee = True compares it with the oracle in plain IEEE arithmetic instead (no knowledge of the device) and is judged through `stats`
    (a ParityStats that receives the per-plane tolerance statistics; the return value is then the worst error of the user OUTPUTS only).
    backend: "hip" = the GPU; "emu" = the device sources compiled for the CPU (tests/emu); default: NRD_PARITY_BACKEND or "hip".
    static_after = N: the camera stops moving after frame N (long runs: accumulation counters saturate, anti-lag fires on the stop).
    graph: the HIP side runs in graph mode (one hipGraph launch per frame).
    resource = (w, h) >= (width, height): dynamic resolution, the frame is the top-left rect of resource-sized planes;
    rect_sizes = [(w, h), ...]: the rect size of frame f is rect_sizes[f % len] (same aspect ratio as (width, height)), inside `resource`."""
    prev_ieee = oracle_driver.set_ieee_mode(ieee)
    try:
        return _run_parity(name, width, height, frames, verbose, settings_overrides, static_camera, check_pools, cs_kw, extra_want, pad, resource, rect_sizes, stats, static_after,
                           graph, device, backend or os.environ.get("NRD_PARITY_BACKEND", "hip"))
    finally:
        oracle_driver.set_ieee_mode(prev_ieee)


def _run_parity(name, width, height, frames, verbose, settings_overrides, static_camera, check_pools, cs_kw, extra_want, pad, resource, rect_sizes, stats, static_after, graph, device,
                backend="hip"):
    if rect_sizes:
        seq = [synth.render_frame(*rect_sizes[f % len(rect_sizes)], f, static_camera=static_camera, want=tuple(DENOISERS[name][1]) + tuple(extra_want)) for f in range(frames)]
    elif static_after is not None:
        # the camera of frame static_after is kept from there on; the noise keeps changing (render_frame seeds it with the frame index)
        seq = [synth.render_frame(width, height, f, device=device, want=tuple(DENOISERS[name][1]) + tuple(extra_want), camera_frame=min(f, static_after)) for f in range(frames)]
    else:
        seq = generate_sequence(name, width, height, frames, static_camera=static_camera, extra_want=extra_want, device=device)
    cs_kw = dict(cs_kw or {})
    if resource:
        seq = [embed_in_resource(fr, resource) for fr in seq]
        cs_kw.update(resourceSize=resource, resourceSizePrev=resource)
    rw, rh = resource or (width, height)
    validation = bool(cs_kw.get("enableValidation"))  # the debug overlay (OUT_VALIDATION, RGBA8) is bound and compared like any other output
    if backend == "emu":  # the device sources compiled for the CPU (tests/emu): the same comparison on a machine without a GPU
        from emu.emu_run import EmuRun as DeviceRun
    else:
        DeviceRun = GpuRun
    ora, hip = OracleRun(name, rw, rh, validation=validation), DeviceRun(name, rw, rh, pad=pad, validation=validation)
    if graph:
        hip.ex.set_graph_mode(True)
    worst = 0.0
    for f, frame in enumerate(seq):
        cam, cam_prev = frame["camera"], seq[max(f - 1, 0)]["camera"]
        if rect_sizes:
            width, height = rect_sizes[f % len(rect_sizes)]
            cs_kw.update(rectSize=(width, height), rectSizePrev=rect_sizes[max(f - 1, 0) % len(rect_sizes)])
        cs = common_settings(cam, cam_prev, width, height, f, **cs_kw)
        tag_checkerboard(frame, settings_overrides, f)
        st = denoiser_settings(name, frame, settings_overrides)
        ora.step(frame, cs, st)
        hip.step(frame, common_settings(cam, cam_prev, width, height, f, **cs_kw), denoiser_settings(name, frame, settings_overrides))
        if verbose and hasattr(hip.ex, "tile_fallback_stats"):
            print("frame %d tiles left to a fallback kernel: %d of %d" % ((f,) + tuple(hip.ex.tile_fallback_stats())))
        for rt in ora.outs:
            want, got = ora.output(rt), hip.output(rt)
            e = rel_error(got, want)
            exact = float(np.mean(got == want))
            worst = max(worst, e)
            if stats is not None:
                stats.add(rt.name, f, error_stats(got, want))
            if verbose:
                print("frame %d %-28s max rel err %.3g  bit-exact texels %.4f%%" % (f, rt.name, e, 100.0 * exact))
        if RT.IN_MV in ora.inputs:  # in/out plane: REBLUR temporal stabilization may write specular motion back into it
            want, got = ora.inputs[RT.IN_MV].astype(np.float32), hip.inputs[RT.IN_MV].cpu().numpy().astype(np.float32)
            e = rel_error(got, want)
            if stats is None:
                worst = max(worst, e)
            else:
                stats.add("IN_MV(in/out)", f, error_stats(got, want))
            if verbose and (e > 0 or cs_kw.get("isBaseColorMetalnessAvailable")):
                src = frame["mv"].cpu().numpy().astype(np.float32)
                print("frame %d IN_MV (in/out)               max rel err %.3g  texels modified by the pass %d" % (f, e, int(np.any(want != src, axis=-1).sum())))
        if check_pools:
            for pool in (RT.PERMANENT_POOL, RT.TRANSIENT_POOL):
                descs = ora.inst.permanent_pool if pool == RT.PERMANENT_POOL else ora.inst.transient_pool
                for i in range(len(descs)):
                    o_raw, fmt, w = ora.ex.pool_plane(pool, i)
                    h_raw, hfmt, hw = hip.ex.read_pool_plane(pool, i)
                    assert fmt == hfmt and w == hw
                    row_bytes = w * api.FORMAT_BYTES[fmt]
                    if stats is None and not verbose and np.array_equal(o_raw[:, :row_bytes], h_raw[:, :row_bytes]):
                        continue  # identical bytes: relative error 0 without decoding 30 MB planes to float64 (the common case of the bit-exact runs)
                    want, got = decode_plane(o_raw, fmt, w), decode_plane(h_raw, fmt, w)
                    e = rel_error(got, want)
                    if stats is None:
                        worst = max(worst, e)
                    else:  # quantised internal planes (UNORM8 counters, packed bits) are reported, the verdict is taken on the user outputs
                        stats.add("%s[%d] %s" % (pool.name, i, fmt.name), f, error_stats(got, want))
                    if verbose and e > 0:
                        bad = np.argwhere(np.any(got != want, axis=-1))
                        print("frame %d %s[%d] %-22s max rel err %.3g  differing texels %d  first %s" % (f, pool.name, i, fmt.name, e, len(bad), bad[:3].tolist()))
    return worst


def run_parity_from_gpu_state(name, width, height, warm, frames, verbose=False, device="cuda"):
    """Bit-exactness deep in a sequence without paying the oracle for the way there: the GPU runs frames 0 .. warm - 1 alone, then its whole state -- the user outputs (they
    double as history), every permanent and transient pool plane -- is copied into the oracle's planes, the oracle's host instance is stepped through the same frames without
    executing them (frame counters, ping-pong state), and frames warm .. warm + frames - 1 run on both: returns the worst relative error over all outputs and pool planes.
    The copied state is itself the product of `warm` frames that the shorter runs hold bit for bit from frame 0; what this adds is the REGIME (saturated history, narrow blur
    radii, the accumulation counters at their caps) at the full size."""
    seq = generate_sequence(name, width, height, warm + frames, device=device)
    ora, hip = OracleRun(name, width, height), HipRun(name, width, height)
    cams = [fr["camera"] for fr in seq]
    for f in range(warm):
        frame = seq[f]
        cam, cam_prev = cams[f], cams[max(f - 1, 0)]
        hip.step(frame, common_settings(cam, cam_prev, width, height, f), denoiser_settings(name, frame, None))
        assert ora.inst.set_denoiser_settings(0, denoiser_settings(name, frame, None)) == api.Result.SUCCESS
        assert ora.inst.set_common_settings(common_settings(cam, cam_prev, width, height, f)) == api.Result.SUCCESS
        assert ora.inst.get_compute_dispatches()[0] == api.Result.SUCCESS  # (advances the instance exactly as executing the frame would; nothing is run)
        seq[f] = None  # (118 MB of inputs per 1440p frame)
    for rt, (arr, fmt) in ora.outs.items():
        got = hip.outs[rt][0]
        got = got.cpu().numpy() if hasattr(got, "cpu") else np.asarray(got)  # (GpuRun: a CUDA tensor; EmuRun: a host array)
        arr[...] = np.ascontiguousarray(got).view(arr.dtype).reshape(arr.shape)
    for pool in (RT.PERMANENT_POOL, RT.TRANSIENT_POOL):
        descs = ora.inst.permanent_pool if pool == RT.PERMANENT_POOL else ora.inst.transient_pool
        for i in range(len(descs)):
            o_raw, fmt, w = ora.ex.pool_plane(pool, i)
            h_raw, hfmt, hw = hip.ex.read_pool_plane(pool, i)
            assert fmt == hfmt and w == hw and o_raw.shape[0] == h_raw.shape[0]
            row_bytes = w * api.FORMAT_BYTES[fmt]
            o_raw[:, :row_bytes] = h_raw[:, :row_bytes]
    worst = 0.0
    for f in range(warm, warm + frames):
        frame = seq[f]
        cam, cam_prev = cams[f], cams[max(f - 1, 0)]
        ora.step(frame, common_settings(cam, cam_prev, width, height, f), denoiser_settings(name, frame, None))
        hip.step(frame, common_settings(cam, cam_prev, width, height, f), denoiser_settings(name, frame, None))
        for rt in ora.outs:
            e = rel_error(hip.output(rt), ora.output(rt))
            worst = max(worst, e)
            if verbose:
                print("frame %d %-28s max rel err %.3g" % (f, rt.name, e))
        for pool in (RT.PERMANENT_POOL, RT.TRANSIENT_POOL):
            descs = ora.inst.permanent_pool if pool == RT.PERMANENT_POOL else ora.inst.transient_pool
            for i in range(len(descs)):
                o_raw, fmt, w = ora.ex.pool_plane(pool, i)
                h_raw, hfmt, hw = hip.ex.read_pool_plane(pool, i)
                row_bytes = w * api.FORMAT_BYTES[fmt]
                if not np.array_equal(o_raw[:, :row_bytes], h_raw[:, :row_bytes]):
                    e = rel_error(decode_plane(h_raw, fmt, w), decode_plane(o_raw, fmt, w))
                    worst = max(worst, e if e > 0 else 1e-30)
                    if verbose:
                        print("frame %d %s[%d] %s differs: max rel err %.3g" % (f, pool.name, i, fmt.name, e))
    return worst
