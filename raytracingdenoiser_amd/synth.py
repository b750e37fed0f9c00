"""Synthetic G-buffer + 1-rpp noisy signal generator: the input side of the path.

An analytic scene (ground plane y = 0, three spheres, sky) seen by a slowly dollying / yawing left-handed pinhole
camera, "path traced" with one diffuse and one specular bounce ray per pixel against the same analytic primitives.
Outputs are packed exactly as an application would hand them to NRD:
  IN_VIEWZ R32F | IN_NORMAL_ROUGHNESS in the library's encoding, R10G10B10A2 by default (NRD_FrontEnd_PackNormalAndRoughness, reference NRD.hlsli:640-667)
  IN_MV RGBA16F | IN_DIFF/SPEC_RADIANCE_HITDIST RGBA16F (REBLUR_FrontEnd_PackRadianceAndNormHitDist, NRD.hlsli:732-743,
  hit distances normalised with REBLUR_FrontEnd_GetNormHitDist, NRD.hlsli:722-727) | IN_PENUMBRA R16F (+ IN_TRANSLUCENCY RGBA8) for SIGMA.
Pure torch, device-agnostic: tests generate on the CPU (and feed the same tensors to the oracle and to the GPU), the
benchmark generates directly in HBM. Not part of the denoiser.
"""
import math

import torch

BACKDROP = False        # set by bench.py --no-sky: primary rays that miss the scene hit a dome of BACKDROP_RADIUS instead of the sky
BACKDROP_RADIUS = 150.0
SKY_VIEWZ = 1.0e6  # > CommonSettings::denoisingRange (5e5): exercises the tile early-outs
HIT_DIST_PARAMS = (3.0, 0.1, 20.0, -25.0)
FP16_MAX = 65504.0

SPHERES = [  # centre xyz, radius, albedo rgb, roughness
    ((-1.6, 0.9, 4.5), 0.9, (0.9, 0.25, 0.2), 0.15),
    ((0.3, 0.6, 3.2), 0.6, (0.2, 0.7, 0.9), 0.45),
    ((1.9, 1.2, 5.5), 1.2, (0.85, 0.8, 0.3), 0.05),
]
LIGHT_DIR = (0.35, 0.8, -0.48)  # direction TO the sun (normalised below)


CAMERA_RISE = 0.0  # scene units per frame of VERTICAL camera translation (tests: vertical parallax of many rows per frame; 0 = the bench / parity camera path)


class Camera:
    """LH camera: +x right, +y up, +z forward; clip = viewToClip * view (D3D style, depth = z / w)."""

    def __init__(self, width, height, frame, fov_y_deg=60.0, z_near=0.1, z_far=1000.0, static=False):
        self.width, self.height = width, height
        t = 0.0 if static else float(frame)
        yaw = math.radians(0.1 * t) + math.radians(8.0)
        pitch = math.radians(-9.0)
        self.pos = (0.2 + 0.004 * t, 1.5 + CAMERA_RISE * t, -1.0 + 0.01 * t)
        cy, sy, cp, sp = math.cos(yaw), math.sin(yaw), math.cos(pitch), math.sin(pitch)
        fwd = (sy * cp, sp, cy * cp)
        right = (cy, 0.0, -sy)
        up = (fwd[1] * right[2] - fwd[2] * right[1], fwd[2] * right[0] - fwd[0] * right[2], fwd[0] * right[1] - fwd[1] * right[0])
        self.right, self.up, self.fwd = right, up, fwd
        f = 1.0 / math.tan(math.radians(fov_y_deg) * 0.5)
        a = width / height
        self.fx, self.fy = f / a, f
        # column-major 4x4
        self.view_to_clip = [self.fx, 0, 0, 0, 0, self.fy, 0, 0, 0, 0, z_far / (z_far - z_near), 1.0, 0, 0, -z_near * z_far / (z_far - z_near), 0]
        rows = (right, up, fwd)
        tr = [-(r[0] * self.pos[0] + r[1] * self.pos[1] + r[2] * self.pos[2]) for r in rows]
        self.world_to_view = [right[0], up[0], fwd[0], 0, right[1], up[1], fwd[1], 0, right[2], up[2], fwd[2], 0, tr[0], tr[1], tr[2], 1.0]


def _hash_uniform(x, y, frame, seed):
    """Integer hash of (x, y, frame, seed) -> float32 in (0, 1); int64 arithmetic so CPU and GPU agree bit-for-bit."""
    m = 0xFFFFFFFF
    h = (x * 0x9E3779B1 + y * 0x85EBCA77 + frame * 0xC2B2AE3D + seed * 0x27D4EB2F + 0x165667B1) & m
    h = ((h ^ (h >> 15)) * 0x2C1B3C6D) & m
    h = ((h ^ (h >> 12)) * 0x297A2D39) & m
    h = h ^ (h >> 15)
    return ((h >> 8).to(torch.float32) + 0.5) * (1.0 / 16777216.0)


def _dot(a, b):
    return (a * b).sum(-1)


def _normalize(v):
    return v / torch.sqrt(_dot(v, v)).clamp_min(1e-20).unsqueeze(-1)


def _trace(o, d, dev):
    """Nearest hit of rays o + t d (d need not be normalised) with the scene. Returns t (inf on miss), normal, albedo, roughness."""
    shape = o.shape[:-1]
    t_best = torch.full(shape, float("inf"), device=dev)
    n_best = torch.zeros(shape + (3,), device=dev)
    alb = torch.zeros(shape + (3,), device=dev)
    rough = torch.zeros(shape, device=dev)

    # ground plane y = 0 with a roughness / albedo checker
    dy = d[..., 1]
    tp = torch.where(dy < -1e-6, -o[..., 1] / dy.clamp(max=-1e-6), torch.full_like(dy, float("inf")))
    tp = torch.where(tp > 1e-3, tp, torch.full_like(tp, float("inf")))
    hit = tp < t_best
    p = o + d * torch.where(hit, tp, torch.zeros_like(tp)).unsqueeze(-1)
    checker = ((torch.floor(p[..., 0] * 0.5) + torch.floor(p[..., 2] * 0.5)) % 2.0) != 0
    t_best = torch.where(hit, tp, t_best)
    n_best = torch.where(hit.unsqueeze(-1), torch.tensor([0.0, 1.0, 0.0], device=dev).expand_as(n_best), n_best)
    plane_alb = torch.where(checker.unsqueeze(-1), torch.tensor([0.75, 0.75, 0.75], device=dev), torch.tensor([0.3, 0.32, 0.35], device=dev))
    alb = torch.where(hit.unsqueeze(-1), plane_alb, alb)
    rough = torch.where(hit, torch.where(checker, torch.full_like(rough, 0.9), torch.full_like(rough, 0.25)), rough)

    for (c, r, a, ro) in SPHERES:
        cc = torch.tensor(c, device=dev)
        oc = o - cc
        A = _dot(d, d)
        B = 2.0 * _dot(d, oc)
        C = _dot(oc, oc) - r * r
        disc = B * B - 4.0 * A * C
        sq = torch.sqrt(disc.clamp_min(0.0))
        t0 = (-B - sq) / (2.0 * A)
        ts = torch.where((disc > 0) & (t0 > 1e-3), t0, torch.full_like(t0, float("inf")))
        hit = ts < t_best
        ph = o + d * torch.where(hit, ts, torch.zeros_like(ts)).unsqueeze(-1)
        n = (ph - cc) / r
        t_best = torch.where(hit, ts, t_best)
        n_best = torch.where(hit.unsqueeze(-1), n, n_best)
        alb = torch.where(hit.unsqueeze(-1), torch.tensor(a, device=dev).expand_as(alb), alb)
        rough = torch.where(hit, torch.full_like(rough, ro), rough)
    return t_best, n_best, alb, rough


def _sky(d, dev):
    up = d[..., 1].clamp(0.0, 1.0).unsqueeze(-1)
    return torch.tensor([0.35, 0.5, 0.8], device=dev) * (0.4 + 0.6 * up)


def _shade(p, n, alb, dev):
    """Direct sun (with hard shadow) + ambient: the 'radiance' a bounce ray brings back."""
    L = _normalize(torch.tensor(LIGHT_DIR, device=dev)).expand_as(n)
    ts, _, _, _ = _trace(p + n * 1e-3, L, dev)
    lit = torch.isinf(ts).to(torch.float32)
    ndl = _dot(n, L).clamp_min(0.0) * lit
    return alb * (2.5 * ndl.unsqueeze(-1) + 0.15)


def _basis(n):
    s = torch.where(n[..., 2] >= 0, torch.ones_like(n[..., 2]), -torch.ones_like(n[..., 2]))
    a = -1.0 / (s + n[..., 2])
    b = n[..., 0] * n[..., 1] * a
    t = torch.stack([1.0 + s * n[..., 0] * n[..., 0] * a, s * b, -s * n[..., 0]], -1)
    bt = torch.stack([b, s + n[..., 1] * n[..., 1] * a, -n[..., 1]], -1)
    return t, bt


def _oct_encode(n):
    n = n / n.abs().sum(-1, keepdim=True)
    wrap = (1.0 - n[..., [1, 0]].abs()) * torch.where(n[..., :2] >= 0, torch.ones_like(n[..., :2]), -torch.ones_like(n[..., :2]))
    xy = torch.where((n[..., 2] >= 0).unsqueeze(-1), n[..., :2], wrap)
    return xy * 0.5 + 0.5


def pack_normal_roughness(n, roughness, material_id, encoding=None):
    """NRD_FrontEnd_PackNormalAndRoughness (reference NRD.hlsli:640-667) in the library's G-buffer encoding (api.NORMAL_ENCODING / ROUGHNESS_ENCODING, or `encoding` = (n, r)):
    R10G10B10A2_UNORM words (int32 [H, W]) for normal encoding 2; RGBA8_UNORM / RGBA8_SNORM words (int32 [H, W], byte 0 = x) for 0 / 1; RGBA16_UNORM / RGBA16_SNORM texels
    (int16 [H, W, 4]) for 3 / 4. Encodings other than 2 carry no material ID."""
    from . import api

    ne, re = encoding if encoding is not None else (api.NORMAL_ENCODING, api.ROUGHNESS_ENCODING)
    if re == 2:
        roughness = roughness.clamp(0.0, 1.0).sqrt()
    elif re == 0:
        roughness = roughness * roughness
    q = lambda v, m: torch.floor(v.clamp(0.0, 1.0) * m + 0.5).to(torch.int64)
    if ne == 2:
        e = _oct_encode(n)
        word = q(e[..., 0], 1023.0) | (q(e[..., 1], 1023.0) << 10) | (q(roughness, 1023.0) << 20) | (q(material_id / 3.0, 3.0) << 30)
        word = torch.where(word >= 2 ** 31, word - 2 ** 32, word)
        return word.to(torch.int32)
    n = n / n.abs().max(-1, keepdim=True).values  # "best fit (optional)"
    p = torch.cat([n, roughness.unsqueeze(-1)], -1)
    if ne in (0, 3):
        p = torch.cat([n * 0.5 + 0.5, roughness.unsqueeze(-1)], -1)
        c = q(p, 255.0 if ne == 0 else 65535.0)
    else:  # SNORM: clamp, scale, round half away from zero
        t = p.clamp(-1.0, 1.0) * (127.0 if ne == 1 else 32767.0)
        c = torch.where(t >= 0, torch.floor(t + 0.5), -torch.floor(-t + 0.5)).to(torch.int64)
    if ne <= 1:
        c = c & 0xFF
        word = c[..., 0] | (c[..., 1] << 8) | (c[..., 2] << 16) | (c[..., 3] << 24)
        word = torch.where(word >= 2 ** 31, word - 2 ** 32, word)
        return word.to(torch.int32)
    c = c & 0xFFFF
    c = torch.where(c >= 2 ** 15, c - 2 ** 16, c)
    return c.to(torch.int16)


def _ycocg(c):
    y = c[..., 0] * 0.25 + c[..., 1] * 0.5 + c[..., 2] * 0.25
    co = c[..., 0] * 0.5 - c[..., 2] * 0.5
    cg = -c[..., 0] * 0.25 + c[..., 1] * 0.5 - c[..., 2] * 0.25
    return torch.stack([y, co, cg], -1)


def _norm_hit_dist(hit_dist, view_z, roughness):
    A, B, C, D = HIT_DIST_PARAMS
    f = (A + view_z.abs() * B) * (1.0 + (C - 1.0) * torch.exp2(D * roughness * roughness).clamp(0.0, 1.0))
    return (hit_dist / f).clamp(0.0, 1.0)


def _reblur_sh1(radiance, direction):
    """REBLUR_FrontEnd_PackSh (reference NRD.hlsli:745-762): SH0 = (Y, Co, Cg, normHitDist) -- the same texel as the non-SH packing --
    and SH1 = (direction * Y, sharpness = 0), RGBA16F."""
    y = _ycocg(radiance.clamp(0.0, FP16_MAX))[..., 0]
    sh1 = torch.cat([direction.clamp(-1.0, 1.0) * y.unsqueeze(-1), torch.zeros_like(y).unsqueeze(-1)], -1)
    return sh1.clamp(-FP16_MAX, FP16_MAX).to(torch.float16).contiguous()


def _pack_relax(out, name, radiance, hit_dist, direction):
    """RELAX_FrontEnd_PackRadianceAndHitDist / RELAX_FrontEnd_PackSh (reference NRD.hlsli:789-818), both RGBA16F:
    <name>_relax = SH0 = (radiance, hitDist in world units), <name>_relax_sh1 = (direction * luminance, 0)."""
    radiance = radiance.clamp(0.0, FP16_MAX)
    hit_dist = hit_dist.clamp(0.0, FP16_MAX)
    out[name + "_relax"] = torch.cat([radiance, hit_dist.unsqueeze(-1)], -1).to(torch.float16).contiguous()
    luma = radiance[..., 0] * 0.2126 + radiance[..., 1] * 0.7152 + radiance[..., 2] * 0.0722
    sh1 = torch.cat([direction.clamp(-1.0, 1.0) * luma.unsqueeze(-1), torch.zeros_like(luma).unsqueeze(-1)], -1)
    out[name + "_relax_sh1"] = sh1.clamp(-FP16_MAX, FP16_MAX).to(torch.float16).contiguous()


def render_frame(width, height, frame, device="cpu", static_camera=False, noise=True, seed=7, want=("reblur",), camera_frame=None):
    """Returns a dict with the packed planes and the camera (for CommonSettings). camera_frame: the frame index the camera pose is taken from
    (default: frame) -- a camera that stops moving while the noise keeps changing."""
    dev = torch.device(device)
    cam = Camera(width, height, frame if camera_frame is None else camera_frame, static=static_camera)
    ys, xs = torch.meshgrid(torch.arange(height, device=dev), torch.arange(width, device=dev), indexing="ij")
    u = (xs.to(torch.float32) + 0.5) / width
    v = (ys.to(torch.float32) + 0.5) / height
    dvx = (2.0 * u - 1.0) / cam.fx
    dvy = (1.0 - 2.0 * v) / cam.fy
    R = torch.tensor([cam.right, cam.up, cam.fwd], device=dev, dtype=torch.float32)  # rows: view axes in world space
    d = dvx.unsqueeze(-1) * R[0] + dvy.unsqueeze(-1) * R[1] + R[2]  # view z = 1 => ray parameter == viewZ
    o = torch.tensor(cam.pos, device=dev, dtype=torch.float32).expand_as(d)

    t, n, alb, rough = _trace(o, d, dev)
    if BACKDROP:  # a large dome behind the scene: no sky pixels at all (every pixel of the frame is denoised; bench.py --no-sky)
        A, B, C = _dot(d, d), 2.0 * _dot(d, o), _dot(o, o) - BACKDROP_RADIUS * BACKDROP_RADIUS
        td = (-B + torch.sqrt((B * B - 4.0 * A * C).clamp_min(0.0))) / (2.0 * A)
        miss = torch.isinf(t)
        pd = o + d * td.unsqueeze(-1)
        t = torch.where(miss, td, t)
        n = torch.where(miss.unsqueeze(-1), -pd / BACKDROP_RADIUS, n)
        alb = torch.where(miss.unsqueeze(-1), torch.tensor([0.5, 0.5, 0.55], device=dev).expand_as(alb), alb)
        rough = torch.where(miss, torch.full_like(rough, 0.7), rough)
    is_sky = torch.isinf(t)
    view_z = torch.where(is_sky, torch.full_like(t, SKY_VIEWZ), t)
    n = torch.where(is_sky.unsqueeze(-1), torch.tensor([0.0, 0.0, -1.0], device=dev).expand_as(n), _normalize(n))
    p = o + d * torch.where(is_sky, torch.zeros_like(t), t).unsqueeze(-1)

    out = {"camera": cam, "viewz": view_z.contiguous(), "is_sky": is_sky}
    material_id = torch.zeros_like(rough)
    if "materials" in want:  # material IDs 0..3 in bands of the world position (consumed when minMaterialForDiffuse / ForSpecular < 3 or a special material ID matches)
        material_id = torch.where(is_sky, torch.zeros_like(rough), torch.remainder(torch.floor(p[..., 0] * 0.9) + torch.floor(p[..., 2] * 0.9), 4.0))
    out["normal_roughness"] = pack_normal_roughness(n, rough, material_id).contiguous()
    out["mv"] = torch.zeros((height, width, 4), dtype=torch.float16, device=dev)  # static scene, world-space MVs scaled by 0
    if "basecolor" in want:
        # IN_BASECOLOR_METALNESS (RGBA8_UNORM): the surface albedo as base colour, metalness in bands {0, 0.5, 1} of the world position
        metal = torch.remainder(torch.floor(p[..., 0] * 1.3) + torch.floor(p[..., 2] * 1.3), 3.0) * 0.5
        bcm = torch.cat([alb.clamp(0.0, 1.0), metal.unsqueeze(-1)], -1)
        bcm = torch.where(is_sky.unsqueeze(-1), torch.zeros_like(bcm), bcm)
        out["basecolor_metalness"] = torch.floor(bcm * 255.0 + 0.5).to(torch.uint8).contiguous()
    if "mv2d" in want:
        # screen-space ("2.5D") motion vectors of the static scene under the moving camera: .xy in pixels towards the previous frame,
        # .z = viewZprev - viewZ; for CommonSettings motionVectorScale = (1 / w, 1 / h, 1) [or (.., 0) for 2D], isMotionVectorInWorldSpace = false
        camp = Camera(width, height, max((frame if camera_frame is None else camera_frame) - 1, 0), static=static_camera)
        Rp = torch.tensor([camp.right, camp.up, camp.fwd], device=dev, dtype=torch.float32)
        pv = (p - torch.tensor(camp.pos, device=dev, dtype=torch.float32)) @ Rp.T
        zp = pv[..., 2].clamp_min(1e-3)
        up_, vp_ = 0.5 + 0.5 * pv[..., 0] / zp * camp.fx, 0.5 - 0.5 * pv[..., 1] / zp * camp.fy
        mv = torch.stack([(up_ - u) * width, (vp_ - v) * height, pv[..., 2] - view_z, torch.zeros_like(u)], -1)
        out["mv"] = torch.where(is_sky.unsqueeze(-1), torch.zeros_like(mv), mv).clamp(-FP16_MAX, FP16_MAX).to(torch.float16).contiguous()

    xi, yi = xs.to(torch.int64), ys.to(torch.int64)
    fr = frame if noise else 0
    if "reblur" in want or "relax" in want:
        vdir = _normalize(-d)
        # diffuse bounce: cosine-weighted direction around n
        u1, u2 = _hash_uniform(xi, yi, fr, seed), _hash_uniform(xi, yi, fr, seed + 1)
        tb, bb = _basis(n)
        r, phi = torch.sqrt(u1), 2.0 * math.pi * u2
        ld = torch.stack([r * torch.cos(phi), r * torch.sin(phi), torch.sqrt((1.0 - u1).clamp_min(0.0))], -1)
        wd = _normalize(tb * ld[..., 0:1] + bb * ld[..., 1:2] + n * ld[..., 2:3])
        th, nh, ah, _ = _trace(p + n * 1e-3, wd, dev)
        miss = torch.isinf(th)
        ph = p + wd * torch.where(miss, torch.zeros_like(th), th).unsqueeze(-1)
        rad = torch.where(miss.unsqueeze(-1), _sky(wd, dev), _shade(ph, _normalize(nh + 1e-9), ah, dev))
        hit_d = torch.where(miss, torch.full_like(th, 1000.0), th)
        rad = torch.where(is_sky.unsqueeze(-1), torch.zeros_like(rad), rad).clamp(0.0, 250.0)
        nhd = torch.where(is_sky, torch.zeros_like(hit_d), _norm_hit_dist(hit_d, view_z, torch.ones_like(rough)))
        out["diff"] = torch.cat([_ycocg(rad), nhd.unsqueeze(-1)], -1).clamp(-FP16_MAX, FP16_MAX).to(torch.float16).contiguous()
        out["diff_sh1"] = _reblur_sh1(rad, wd)
        # REBLUR_FrontEnd_PackDirectionalOcclusion (NRD.hlsli:776-787): (direction * normHitDist, normHitDist) as RGBA16_SNORM texels
        do = torch.cat([wd.clamp(-1.0, 1.0) * nhd.unsqueeze(-1), nhd.unsqueeze(-1)], -1).clamp(-1.0, 1.0) * 32767.0
        out["diff_direction_hitdist"] = torch.where(do >= 0, torch.floor(do + 0.5), -torch.floor(-do + 0.5)).to(torch.int16).contiguous()
        if "relax" in want:
            _pack_relax(out, "diff", rad, torch.where(is_sky, torch.zeros_like(hit_d), hit_d), wd)

        # specular bounce: mirror direction jittered inside a roughness-sized lobe
        u3, u4 = _hash_uniform(xi, yi, fr, seed + 2), _hash_uniform(xi, yi, fr, seed + 3)
        refl = _normalize(-vdir + n * (2.0 * _dot(n, vdir)).unsqueeze(-1))
        tr_, br_ = _basis(refl)
        rr = (rough * rough).unsqueeze(-1) * torch.sqrt(u3).unsqueeze(-1)
        ws = _normalize(refl + (tr_ * torch.cos(2.0 * math.pi * u4).unsqueeze(-1) + br_ * torch.sin(2.0 * math.pi * u4).unsqueeze(-1)) * rr)
        below = _dot(ws, n) <= 0.0
        th, nh, ah, _ = _trace(p + n * 1e-3, ws, dev)
        miss = torch.isinf(th)
        ph = p + ws * torch.where(miss, torch.zeros_like(th), th).unsqueeze(-1)
        rad = torch.where(miss.unsqueeze(-1), _sky(ws, dev), _shade(ph, _normalize(nh + 1e-9), ah, dev))
        hit_d = torch.where(miss, torch.full_like(th, 1000.0), th)
        hit_d = torch.where(below, torch.zeros_like(hit_d), hit_d)  # rays into the surface: hitDist = 0 (handled by NRD)
        rad = torch.where((is_sky | below).unsqueeze(-1), torch.zeros_like(rad), rad).clamp(0.0, 250.0)
        nhd = torch.where(is_sky, torch.zeros_like(hit_d), _norm_hit_dist(hit_d, view_z, rough))
        out["spec"] = torch.cat([_ycocg(rad), nhd.unsqueeze(-1)], -1).clamp(-FP16_MAX, FP16_MAX).to(torch.float16).contiguous()
        out["spec_sh1"] = _reblur_sh1(rad, ws)
        if "relax" in want:
            _pack_relax(out, "spec", rad, torch.where(is_sky, torch.zeros_like(hit_d), hit_d), ws)

    if "holes" in want:
        # probabilistic sampling: about half of the pixels carry no hit distance (0 = "invalid", to be reconstructed by NRD)
        hole = _hash_uniform(xi, yi, fr, seed + 20) < 0.5
        for key in ("diff", "spec", "diff_relax", "spec_relax"):
            if key in out:
                out[key][..., 3] = torch.where(hole, torch.zeros_like(out[key][..., 3]), out[key][..., 3])

    if "confidence" in want:
        # optional guides (R8_UNORM): history confidence per signal and the disocclusion-threshold mix, smooth fields + per-frame hash noise
        for k, key in enumerate(("diff_confidence", "spec_confidence", "disocclusion_mix")):
            field = 0.5 + 0.5 * torch.sin(u * (7.0 + 3.0 * k) + 0.37 * frame) * torch.cos(v * (5.0 + 2.0 * k) - 0.21 * frame)
            field = (0.75 * field + 0.25 * _hash_uniform(xi, yi, fr, seed + 10 + k)).clamp(0.0, 1.0)
            out[key] = torch.floor(field * 255.0 + 0.5).to(torch.uint8).contiguous()

    if "sigma" in want:
        # sun with a 0.25 deg angular radius... widened to 1.5 deg so penumbrae span pixels at test resolutions
        L = _normalize(torch.tensor(LIGHT_DIR, device=dev))
        tan_r = math.tan(math.radians(1.5))
        u5, u6 = _hash_uniform(xi, yi, fr, seed + 4), _hash_uniform(xi, yi, fr, seed + 5)
        tl, bl = _basis(L.expand_as(n))
        rr = tan_r * torch.sqrt(u5).unsqueeze(-1)
        wl = _normalize(L + (tl * torch.cos(2.0 * math.pi * u6).unsqueeze(-1) + bl * torch.sin(2.0 * math.pi * u6).unsqueeze(-1)) * rr)
        ndl = _dot(n, L.expand_as(n))
        ts, _, occluder_albedo, occluder_roughness = _trace(p + n * 1e-3, wl, dev)
        # SIGMA_FrontEnd_PackPenumbra (NRD.hlsli:828-834): distanceToOccluder * tanOfLightAngularRadius, NoL<=0 -> 0, miss -> FP16_MAX
        pen = torch.where(torch.isinf(ts), torch.full_like(ts, FP16_MAX), (ts * tan_r).clamp(max=FP16_MAX))
        pen = torch.where(ndl <= 0.0, torch.zeros_like(pen), pen)
        pen = torch.where(is_sky, torch.full_like(pen, FP16_MAX), pen)
        out["penumbra"] = pen.to(torch.float16).contiguous()
        # IN_TRANSLUCENCY for SIGMA_SHADOW_TRANSLUCENCY = SIGMA_FrontEnd_PackTranslucency (NRD.hlsli:848-855): x = "distance to occluder
        # >= FP16_MAX" (lit), yzw = saturate(translucency). The cyan sphere (roughness 0.45) is stained glass, everything else is opaque.
        lit = pen >= FP16_MAX
        glass = (occluder_roughness == 0.45) & ~lit
        tr = torch.where(lit.unsqueeze(-1), torch.ones_like(occluder_albedo), torch.where(glass.unsqueeze(-1), occluder_albedo, torch.zeros_like(occluder_albedo)))
        tr4 = torch.cat([lit.to(torch.float32).unsqueeze(-1), tr.clamp(0.0, 1.0)], -1)
        out["translucency"] = torch.floor(tr4 * 255.0 + 0.5).to(torch.uint8).contiguous()
        out["light_dir"] = tuple(float(x) for x in L)
    return out
