"""Parity harness: runs the same synthetic frame sequence through the CPU oracle and through the HIP path (both driven
by the SAME nrd::GetComputeDispatches lists from the product's host) and compares planes. Used by tests/ and smoke()."""
import numpy as np
import torch

from oracle import driver as oracle_driver
from raytracingdenoiser_amd import api, synth

RT = api.ResourceType
F = api.Format

REL_TOL = 1e-3  # BASELINE.json north_star: <= 1e-3 relative per pixel (bit-exact for REFERENCE)


def common_settings(cam, cam_prev, width, height, frame_index, **kw):
    args = dict(resourceSize=(width, height), rectSize=(width, height), resourceSizePrev=(width, height), rectSizePrev=(width, height),
                timeDeltaBetweenFrames=16.667, frameIndex=frame_index, isMotionVectorInWorldSpace=True, motionVectorScale=(0.0, 0.0, 0.0))
    args.update(kw)
    cs = api.CommonSettings(**args)
    for i in range(16):
        cs.viewToClipMatrix[i] = cam.view_to_clip[i]
        cs.viewToClipMatrixPrev[i] = cam_prev.view_to_clip[i]
        cs.worldToViewMatrix[i] = cam.world_to_view[i]
        cs.worldToViewMatrixPrev[i] = cam_prev.world_to_view[i]
    return cs


DENOISERS = {
    "REBLUR_DIFFUSE": (api.Denoiser.REBLUR_DIFFUSE, ("reblur",)),
    "REBLUR_SPECULAR": (api.Denoiser.REBLUR_SPECULAR, ("reblur",)),
    "REBLUR_DIFFUSE_SPECULAR": (api.Denoiser.REBLUR_DIFFUSE_SPECULAR, ("reblur",)),
    "REBLUR_DIFFUSE_SH": (api.Denoiser.REBLUR_DIFFUSE_SH, ("reblur",)),
    "REBLUR_SPECULAR_SH": (api.Denoiser.REBLUR_SPECULAR_SH, ("reblur",)),
    "REBLUR_DIFFUSE_SPECULAR_SH": (api.Denoiser.REBLUR_DIFFUSE_SPECULAR_SH, ("reblur",)),
    "REBLUR_DIFFUSE_DIRECTIONAL_OCCLUSION": (api.Denoiser.REBLUR_DIFFUSE_DIRECTIONAL_OCCLUSION, ("reblur",)),
    "REBLUR_DIFFUSE_OCCLUSION": (api.Denoiser.REBLUR_DIFFUSE_OCCLUSION, ("reblur",)),
    "REBLUR_SPECULAR_OCCLUSION": (api.Denoiser.REBLUR_SPECULAR_OCCLUSION, ("reblur",)),
    "REBLUR_DIFFUSE_SPECULAR_OCCLUSION": (api.Denoiser.REBLUR_DIFFUSE_SPECULAR_OCCLUSION, ("reblur",)),
    "SIGMA_SHADOW": (api.Denoiser.SIGMA_SHADOW, ("sigma",)),
    "SIGMA_SHADOW_TRANSLUCENCY": (api.Denoiser.SIGMA_SHADOW_TRANSLUCENCY, ("sigma",)),
    "RELAX_DIFFUSE": (api.Denoiser.RELAX_DIFFUSE, ("relax",)),
    "RELAX_DIFFUSE_SH": (api.Denoiser.RELAX_DIFFUSE_SH, ("relax",)),
    "RELAX_SPECULAR": (api.Denoiser.RELAX_SPECULAR, ("relax",)),
    "RELAX_SPECULAR_SH": (api.Denoiser.RELAX_SPECULAR_SH, ("relax",)),
    "RELAX_DIFFUSE_SPECULAR": (api.Denoiser.RELAX_DIFFUSE_SPECULAR, ("relax",)),
    "RELAX_DIFFUSE_SPECULAR_SH": (api.Denoiser.RELAX_DIFFUSE_SPECULAR_SH, ("relax",)),
}


def _relax_signals(name):
    """(has diffuse, has specular, SH) of a RELAX variant name"""
    body = name[len("RELAX_"):]
    sh = body.endswith("_SH")
    body = body[:-3] if sh else body
    return "DIFFUSE" in body, "SPECULAR" in body, sh


def user_planes(name, frame):
    """(ResourceType, tensor, Format) inputs of a denoiser for one generated frame."""
    extra = []
    if "diff_confidence" in frame:  # generated with want=(..., "confidence"): optional guides, consumed when CommonSettings enables them
        extra = [(RT.IN_DIFF_CONFIDENCE, frame["diff_confidence"], F.R8_UNORM), (RT.IN_SPEC_CONFIDENCE, frame["spec_confidence"], F.R8_UNORM),
                 (RT.IN_DISOCCLUSION_THRESHOLD_MIX, frame["disocclusion_mix"], F.R8_UNORM)]
    planes = _user_planes(name, frame)
    if frame.get("_checkerboard"):  # (CheckerboardMode, frame index): noisy signals traced for every other pixel and packed into the left half
        mode, frame_index = frame["_checkerboard"]
        diff_mode, spec_mode = (0, 1) if mode == api.CheckerboardMode.BLACK else (1, 0)  # reference Reblur.cpp / Relax.cpp: BLACK -> diffuse 0, specular 1
        planes = [(rt, checkerboard_pack(t, diff_mode if rt.name.startswith("IN_DIFF") else spec_mode, frame_index) if rt.name.startswith(("IN_DIFF", "IN_SPEC")) else t, fmt)
                  for rt, t, fmt in planes]
    if "basecolor_metalness" in frame:  # consumed when CommonSettings::isBaseColorMetalnessAvailable (REBLUR: specular motion written back into IN_MV)
        extra.append((RT.IN_BASECOLOR_METALNESS, frame["basecolor_metalness"], F.RGBA8_UNORM))
    return planes + extra


def tag_checkerboard(frame, overrides, frame_index):
    """marks a generated frame so that user_planes() hands out checkerboarded noisy inputs when the settings ask for them"""
    mode = (overrides or {}).get("checkerboardMode")
    frame["_checkerboard"] = (api.CheckerboardMode(mode), frame_index) if mode else None


def checkerboard_pack(plane, mode, frame_index):
    """Checkerboarded noisy input (reference README "checkerboard": the pixels with ((x ^ y) ^ frameIndex) & 1 == mode carry data and are packed into the
    left half of the plane, column x >> 1). The right half is filled with a sentinel: nothing may read it."""
    h, w = plane.shape[0], plane.shape[1]
    y = torch.arange(h, device=plane.device)
    b = (mode ^ (y & 1) ^ (frame_index & 1)).view(h, 1)  # per row: which pixel of each horizontal pair has data
    k = torch.arange((w + 1) // 2, device=plane.device).view(1, -1)
    src = (2 * k + b).clamp(max=w - 1)
    idx = src.view(h, -1, *([1] * (plane.dim() - 2))).expand(h, src.shape[1], *plane.shape[2:])
    out = torch.full_like(plane, 17)
    out[:, : src.shape[1]] = torch.gather(plane, 1, idx)
    return out.contiguous()


def _hitdist_unorm16(signal):
    """normalised hit distance (.w of a packed REBLUR signal) as R16_UNORM texels (int16 tensor holding the uint16 bit patterns)"""
    q = torch.floor(signal[..., 3].float().clamp(0.0, 1.0) * 65535.0 + 0.5).to(torch.int32)
    return torch.where(q >= 32768, q - 65536, q).to(torch.int16).contiguous()


def _user_planes(name, frame):
    planes = [(RT.IN_MV, frame["mv"], F.RGBA16_SFLOAT), (RT.IN_NORMAL_ROUGHNESS, frame["normal_roughness"], F.R10_G10_B10_A2_UNORM), (RT.IN_VIEWZ, frame["viewz"], F.R32_SFLOAT)]
    if name in ("REBLUR_DIFFUSE", "REBLUR_DIFFUSE_SPECULAR"):
        planes.append((RT.IN_DIFF_RADIANCE_HITDIST, frame["diff"], F.RGBA16_SFLOAT))
    if name in ("REBLUR_SPECULAR", "REBLUR_DIFFUSE_SPECULAR"):
        planes.append((RT.IN_SPEC_RADIANCE_HITDIST, frame["spec"], F.RGBA16_SFLOAT))
    if name in ("REBLUR_DIFFUSE_SH", "REBLUR_DIFFUSE_SPECULAR_SH"):
        planes += [(RT.IN_DIFF_SH0, frame["diff"], F.RGBA16_SFLOAT), (RT.IN_DIFF_SH1, frame["diff_sh1"], F.RGBA16_SFLOAT)]
    if name in ("REBLUR_SPECULAR_SH", "REBLUR_DIFFUSE_SPECULAR_SH"):
        planes += [(RT.IN_SPEC_SH0, frame["spec"], F.RGBA16_SFLOAT), (RT.IN_SPEC_SH1, frame["spec_sh1"], F.RGBA16_SFLOAT)]
    if name == "REBLUR_DIFFUSE_DIRECTIONAL_OCCLUSION":
        planes.append((RT.IN_DIFF_DIRECTION_HITDIST, frame["diff_direction_hitdist"], F.RGBA16_SNORM))
    if name in ("REBLUR_DIFFUSE_OCCLUSION", "REBLUR_DIFFUSE_SPECULAR_OCCLUSION"):
        planes.append((RT.IN_DIFF_HITDIST, _hitdist_unorm16(frame["diff"]), F.R16_UNORM))
    if name in ("REBLUR_SPECULAR_OCCLUSION", "REBLUR_DIFFUSE_SPECULAR_OCCLUSION"):
        planes.append((RT.IN_SPEC_HITDIST, _hitdist_unorm16(frame["spec"]), F.R16_UNORM))
    if name.startswith("SIGMA_SHADOW"):
        planes.append((RT.IN_PENUMBRA, frame["penumbra"], F.R16_SFLOAT))
    if name == "SIGMA_SHADOW_TRANSLUCENCY":
        planes.append((RT.IN_TRANSLUCENCY, frame["translucency"], F.RGBA8_UNORM))
    if name.startswith("RELAX"):
        has_diff, has_spec, sh = _relax_signals(name)
        if has_diff:
            planes.append((RT.IN_DIFF_SH0 if sh else RT.IN_DIFF_RADIANCE_HITDIST, frame["diff_relax"], F.RGBA16_SFLOAT))
            if sh:
                planes.append((RT.IN_DIFF_SH1, frame["diff_relax_sh1"], F.RGBA16_SFLOAT))
        if has_spec:
            planes.append((RT.IN_SPEC_SH0 if sh else RT.IN_SPEC_RADIANCE_HITDIST, frame["spec_relax"], F.RGBA16_SFLOAT))
            if sh:
                planes.append((RT.IN_SPEC_SH1, frame["spec_relax_sh1"], F.RGBA16_SFLOAT))
    return planes


def output_planes(name, width, height, validation=False):
    """(ResourceType, dtype, channels, Format)"""
    outs = [(RT.OUT_VALIDATION, torch.uint8, 4, F.RGBA8_UNORM)] if validation else []
    if name in ("REBLUR_DIFFUSE", "REBLUR_DIFFUSE_SPECULAR"):
        outs.append((RT.OUT_DIFF_RADIANCE_HITDIST, torch.float16, 4, F.RGBA16_SFLOAT))
    if name in ("REBLUR_SPECULAR", "REBLUR_DIFFUSE_SPECULAR"):
        outs.append((RT.OUT_SPEC_RADIANCE_HITDIST, torch.float16, 4, F.RGBA16_SFLOAT))
    if name in ("REBLUR_DIFFUSE_SH", "REBLUR_DIFFUSE_SPECULAR_SH"):
        outs += [(RT.OUT_DIFF_SH0, torch.float16, 4, F.RGBA16_SFLOAT), (RT.OUT_DIFF_SH1, torch.float16, 4, F.RGBA16_SFLOAT)]
    if name in ("REBLUR_SPECULAR_SH", "REBLUR_DIFFUSE_SPECULAR_SH"):
        outs += [(RT.OUT_SPEC_SH0, torch.float16, 4, F.RGBA16_SFLOAT), (RT.OUT_SPEC_SH1, torch.float16, 4, F.RGBA16_SFLOAT)]
    if name == "REBLUR_DIFFUSE_DIRECTIONAL_OCCLUSION":
        outs.append((RT.OUT_DIFF_DIRECTION_HITDIST, torch.int16, 4, F.RGBA16_SNORM))
    if name in ("REBLUR_DIFFUSE_OCCLUSION", "REBLUR_DIFFUSE_SPECULAR_OCCLUSION"):
        outs.append((RT.OUT_DIFF_HITDIST, torch.int16, 1, F.R16_UNORM))
    if name in ("REBLUR_SPECULAR_OCCLUSION", "REBLUR_DIFFUSE_SPECULAR_OCCLUSION"):
        outs.append((RT.OUT_SPEC_HITDIST, torch.int16, 1, F.R16_UNORM))
    if name == "SIGMA_SHADOW":
        outs.append((RT.OUT_SHADOW_TRANSLUCENCY, torch.uint8, 1, F.R8_UNORM))
    if name == "SIGMA_SHADOW_TRANSLUCENCY":
        outs.append((RT.OUT_SHADOW_TRANSLUCENCY, torch.uint8, 4, F.RGBA8_UNORM))
    if name.startswith("RELAX"):
        has_diff, has_spec, sh = _relax_signals(name)
        if has_diff:
            outs.append((RT.OUT_DIFF_SH0 if sh else RT.OUT_DIFF_RADIANCE_HITDIST, torch.float16, 4, F.RGBA16_SFLOAT))
            if sh:
                outs.append((RT.OUT_DIFF_SH1, torch.float16, 4, F.RGBA16_SFLOAT))
        if has_spec:
            outs.append((RT.OUT_SPEC_SH0 if sh else RT.OUT_SPEC_RADIANCE_HITDIST, torch.float16, 4, F.RGBA16_SFLOAT))
            if sh:
                outs.append((RT.OUT_SPEC_SH1, torch.float16, 4, F.RGBA16_SFLOAT))
    return outs


def denoiser_settings(name, frame, overrides=None):
    if name.startswith("REBLUR"):
        s = api.ReblurSettings(**(overrides or {}))
    elif name.startswith("SIGMA_SHADOW"):
        s = api.SigmaSettings(lightDirection=frame["light_dir"], **(overrides or {}))
    elif name.startswith("RELAX"):
        s = api.RelaxSettings(**(overrides or {}))
    else:
        raise KeyError(name)
    return s


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
        return {"max": 0.0, "mean": 0.0, "frac_gt_tol": 0.0, "p999": 0.0, "n": 0, "worst_at": None, "bit_exact_frac": 1.0}
    flat = int(np.argmax(err))
    finite = np.where(np.isfinite(err), err, 1e30)
    return {"max": float(err.max()), "mean": float(finite.mean()), "frac_gt_tol": float(np.mean(err > tol)), "p999": float(np.quantile(finite, 0.999)), "n": int(err.size),
            "worst_at": [int(v) for v in np.unravel_index(flat, err.shape)], "bit_exact_frac": float(np.mean((got == want) | both_nan))}


class ParityStats:
    """accumulates error_stats over the frames of a run, per plane name"""

    def __init__(self):
        self.planes = {}

    def add(self, name, frame, st):
        p = self.planes.setdefault(name, {"max": 0.0, "frac_gt_tol": 0.0, "p999": 0.0, "mean": 0.0, "frames": 0, "worst": None, "bit_exact_frac": 1.0})
        if st["max"] >= p["max"]:
            p["worst"] = (frame, st["worst_at"], st["max"])
        p["max"] = max(p["max"], st["max"])
        p["frac_gt_tol"] = max(p["frac_gt_tol"], st["frac_gt_tol"])  # the worst frame
        p["p999"] = max(p["p999"], st["p999"])
        p["mean"] = max(p["mean"], st["mean"])
        p["bit_exact_frac"] = min(p["bit_exact_frac"], st["bit_exact_frac"])
        p["frames"] += 1

    def outputs(self):
        return {k: v for k, v in self.planes.items() if k.startswith("OUT_")}

    def summary(self, only_outputs=True):
        sel = self.outputs() if only_outputs else self.planes
        if not sel:
            return {"max_rel_err": 0.0, "frac_gt_tol": 0.0, "p999": 0.0, "mean": 0.0, "planes": 0}
        return {"max_rel_err": max(v["max"] for v in sel.values()), "frac_gt_tol": max(v["frac_gt_tol"] for v in sel.values()), "p999": max(v["p999"] for v in sel.values()),
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


class HipRun:
    def __init__(self, name, width, height, pad=0, numerics="exact", validation=False):
        """numerics: which build of the library runs -- "exact" (libNRD_hip_exact.so, bit-identical to the oracle) or "fast" (libNRD_hip.so, the product)"""
        from raytracingdenoiser_amd.executor import HipExecutor

        self.name, self.width, self.height, self.pad = name, width, height, pad
        self.inst = api.Instance([(0, DENOISERS[name][0])], numerics=numerics)
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


def generate_sequence(name, width, height, frames, static_camera=False, noise=True, device="cpu", extra_want=()):
    """frames 0 .. frames-1 of the synthetic sequence (analytic scene, moving camera, 1-rpp noise) with the planes the denoiser consumes"""
    return [synth.render_frame(width, height, f, device=device, static_camera=static_camera, noise=noise, want=tuple(DENOISERS[name][1]) + tuple(extra_want)) for f in range(frames)]


def embed_in_resource(frame, resource):
    """dynamic resolution: every plane of a generated (rect-sized) frame placed at the top-left of a resource-sized plane; the rest is a sentinel"""
    rw, rh = resource
    out = {}
    for k, v in frame.items():
        if torch.is_tensor(v) and v.dim() >= 2 and v.dtype != torch.bool:
            big = torch.full([rh, rw] + list(v.shape[2:]), 33.0 if v.dtype.is_floating_point else 9, dtype=v.dtype, device=v.device)
            big[: v.shape[0], : v.shape[1]] = v
            v = big
        out[k] = v
    return out


def run_parity(name, width=192, height=128, frames=4, verbose=False, settings_overrides=None, static_camera=False, check_pools=True, cs_kw=None, extra_want=(), pad=0, resource=None,
               rect_sizes=None, numerics="exact", ieee=False, stats=None, static_after=None, graph=False, device="cpu", backend="hip"):
    """Returns the worst relative error between the HIP path and the oracle over all frames, user outputs and pool planes.
    numerics = "exact": the bit-exact regression build against the oracle that emulates the device's sqrt / rsqrt (expected error: 0);
    numerics = "fast" (the product build) is compared with ieee = True, the oracle in plain IEEE arithmetic, and judged through `stats`
    (a ParityStats that receives the per-plane tolerance statistics; the return value is then the worst error of the user OUTPUTS only).
    static_after = N: the camera stops moving after frame N (long runs: accumulation counters saturate, anti-lag fires on the stop).
    graph: the HIP side runs in graph mode (one hipGraph launch per frame).
    resource = (w, h) >= (width, height): dynamic resolution, the frame is the top-left rect of resource-sized planes;
    rect_sizes = [(w, h), ...]: the rect size of frame f is rect_sizes[f % len] (same aspect ratio as (width, height)), inside `resource`."""
    prev_ieee = oracle_driver.set_ieee_mode(ieee)
    try:
        return _run_parity(name, width, height, frames, verbose, settings_overrides, static_camera, check_pools, cs_kw, extra_want, pad, resource, rect_sizes, numerics, stats, static_after,
                           graph, device, backend)
    finally:
        oracle_driver.set_ieee_mode(prev_ieee)


def _run_parity(name, width, height, frames, verbose, settings_overrides, static_camera, check_pools, cs_kw, extra_want, pad, resource, rect_sizes, numerics, stats, static_after, graph, device,
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
        DeviceRun = HipRun
    ora, hip = OracleRun(name, rw, rh, validation=validation), DeviceRun(name, rw, rh, pad=pad, numerics=numerics, validation=validation)
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
