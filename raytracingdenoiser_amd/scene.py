"""The synthetic workload of SURVEY.md section 8d as the benchmark and the tests use it: which user planes a denoiser consumes and produces, the
CommonSettings of a frame of the moving-camera sequence, the denoiser settings, the frame generator (raytracingdenoiser_amd/synth.py renders the
analytic scene). Pure plumbing over the public API -- nothing here knows about the oracle."""
import torch

from . import api, synth

RT = api.ResourceType
F = api.Format


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
    planes = [(RT.IN_MV, frame["mv"], F.RGBA16_SFLOAT), (RT.IN_NORMAL_ROUGHNESS, frame["normal_roughness"], F[api.NORMAL_ROUGHNESS_FORMAT_NAME]), (RT.IN_VIEWZ, frame["viewz"], F.R32_SFLOAT)]
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
