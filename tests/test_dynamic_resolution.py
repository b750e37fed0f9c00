"""Dynamic resolution (CommonSettings::rectSize < resourceSize, rectSizePrev != rectSize; reference NRDSettings.h "resourceSize / rectSize",
Common.hlsli ClampUvToViewport, gResolutionScale[Prev]): the denoised rect is the top-left part of resource-sized planes and may change
from frame to frame; a shifted rect (rectOrigin != 0: the guide inputs live at an offset inside their planes) is served through rect-at-origin copies."""
import numpy as np
import pytest
import torch

import parity
from raytracingdenoiser_amd import api, synth

RT = api.ResourceType
RECT, RESOURCE = (128, 80), (160, 96)
SIZES = [(144, 88), (108, 66), (126, 77), (144, 88), (90, 55), (144, 88)]  # one aspect ratio, inside RESOURCE


def _oracle_sub_rect(name, seq):
    ora = parity.OracleRun(name, *RESOURCE)
    for f, frame in enumerate(seq):
        cs = parity.common_settings(frame["camera"], seq[max(f - 1, 0)]["camera"], *RECT, f, resourceSize=RESOURCE, resourceSizePrev=RESOURCE)
        ora.step(parity.embed_in_resource(frame, RESOURCE), cs, parity.denoiser_settings(name, frame))
    return ora


def _oracle_full(name, seq):
    ora = parity.OracleRun(name, *RECT)
    for f, frame in enumerate(seq):
        ora.step(frame, parity.common_settings(frame["camera"], seq[max(f - 1, 0)]["camera"], *RECT, f), parity.denoiser_settings(name, frame))
    return ora


def test_oracle_sigma_sub_rect_equals_full_resolution_run():
    # SIGMA has no resolution-dependent constant: denoising a rect inside larger planes must give exactly the rect-sized result
    name = "SIGMA_SHADOW"
    seq = parity.generate_sequence(name, *RECT, 5)
    a, b = _oracle_full(name, seq), _oracle_sub_rect(name, seq)
    out = b.output(RT.OUT_SHADOW_TRANSLUCENCY)
    assert np.array_equal(a.output(RT.OUT_SHADOW_TRANSLUCENCY), out[: RECT[1], : RECT[0]])
    assert not out[RECT[1]:].any() and not out[:, RECT[0]:].any()  # nothing outside the rect is written


def test_oracle_relax_sub_rect_tracks_full_resolution_run():
    # RELAX: identical on the first frame, then equal up to the rounding of uv * resolutionScale in the history fetches
    name = "RELAX_DIFFUSE_SPECULAR"
    seq = parity.generate_sequence(name, *RECT, 4)
    for n in (1, 4):
        a, b = _oracle_full(name, seq[:n]), _oracle_sub_rect(name, seq[:n])
        for rt in (RT.OUT_DIFF_RADIANCE_HITDIST, RT.OUT_SPEC_RADIANCE_HITDIST):
            x, y = a.output(rt), b.output(rt)[: RECT[1], : RECT[0]]
            if n == 1:
                assert np.array_equal(x, y)
            else:
                assert np.mean(x == y) > 0.75 and np.abs(x - y).mean() < 2e-3 * np.abs(x).mean()


def test_reblur_radii_scale_with_the_resolution():
    # reference Reblur.cpp: blur radii are multiplied by min(rect / resource) -- so a REBLUR sub-rect run legitimately differs from a full-resolution one
    def radii(resource):
        inst = api.Instance([(0, api.Denoiser.REBLUR_DIFFUSE_SPECULAR)])
        cam = synth.Camera(*RECT, 0)
        assert inst.set_common_settings(parity.common_settings(cam, cam, *RECT, 0, resourceSize=resource, resourceSizePrev=resource)) == api.Result.SUCCESS
        _, ds = inst.get_compute_dispatches()
        c = np.frombuffer(bytes([d for d in ds if "PrePass" in d.shader][0].constants), dtype=np.float32)
        return c[180:183]  # gMaxBlurRadius, gDiffPrepassBlurRadius, gSpecPrepassBlurRadius
    assert np.allclose(radii(RECT), (30.0, 30.0, 50.0)) and np.allclose(radii(RESOURCE), np.array((30.0, 30.0, 50.0)) * 0.8)


@pytest.mark.parametrize("name", ["REBLUR_DIFFUSE", "RELAX_DIFFUSE"])
def test_oracle_history_survives_resolution_changes(name):
    """static camera, rect size changing every frame: the accumulated-frame counters keep counting (the previous frame is found through
    gResolutionScalePrev / gRectSizePrev), they do not restart at the resolution steps"""
    raw = [synth.render_frame(*SIZES[f], f, static_camera=True, noise=False, want=tuple(parity.DENOISERS[name][1])) for f in range(len(SIZES))]
    ora = parity.OracleRun(name, *RESOURCE)
    for f, frame in enumerate(raw):
        w, h = SIZES[f]
        cs = parity.common_settings(frame["camera"], raw[max(f - 1, 0)]["camera"], w, h, f, resourceSize=RESOURCE, resourceSizePrev=RESOURCE, rectSize=(w, h), rectSizePrev=SIZES[max(f - 1, 0)])
        ora.step(parity.embed_in_resource(frame, RESOURCE), cs, parity.denoiser_settings(name, frame))
        m = ~frame["is_sky"].numpy()
        if name.startswith("REBLUR"):
            plane, fmt, pw = ora.ex.pool_plane(RT.PERMANENT_POOL, 2)  # PREV_INTERNAL_DATA (R16_UINT): 6-bit accumulated frames
            count = (plane[:, : pw * 2].copy().view(np.uint16) & 63)[:h, :w][m]
        else:
            plane, fmt, pw = ora.ex.pool_plane(RT.PERMANENT_POOL, 2)  # history length (R8_UNORM * 255)
            assert fmt == api.Format.R8_UNORM
            count = plane[:h, :w][m]
        assert np.median(count) == f + 1 and np.mean(count == f + 1) > 0.85


def test_classify_tiles_leaves_tiles_outside_the_rect_alone():
    name = "RELAX_DIFFUSE"
    frame = synth.render_frame(*RECT, 0, want=tuple(parity.DENOISERS[name][1]))
    ora = parity.OracleRun(name, *RESOURCE)
    tiles_index = [i for i, (fmt, ds) in enumerate(ora.inst.transient_pool) if ds == 16][0]
    ora.ex.pool_plane(RT.TRANSIENT_POOL, tiles_index)[0][:] = 7
    cs = parity.common_settings(frame["camera"], frame["camera"], *RECT, 0, resourceSize=RESOURCE, resourceSizePrev=RESOURCE)
    ora.step(parity.embed_in_resource(frame, RESOURCE), cs, parity.denoiser_settings(name, frame))
    tiles, _, tw = ora.ex.pool_plane(RT.TRANSIENT_POOL, tiles_index)
    assert set(np.unique(tiles[: RECT[1] // 16, : RECT[0] // 16])) <= {0, 255}
    assert not tiles[:, RECT[0] // 16: tw].any() and not tiles[RECT[1] // 16:, :tw].any()  # cleared by the restart frame, never classified (a sentinel-filled tile would not be sky)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["REBLUR_DIFFUSE_SPECULAR", "REBLUR_DIFFUSE_SPECULAR_OCCLUSION", "REBLUR_SPECULAR_SH", "RELAX_DIFFUSE_SPECULAR", "RELAX_DIFFUSE_SPECULAR_SH", "SIGMA_SHADOW",
                                  "SIGMA_SHADOW_TRANSLUCENCY"])
def test_hip_matches_oracle_sub_rect(name):
    worst = parity.run_parity(name, width=144, height=88, frames=4, verbose=True, resource=(192, 112))
    assert worst <= parity.REL_TOL


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["REBLUR_DIFFUSE_SPECULAR", "REBLUR_DIFFUSE_OCCLUSION", "REBLUR_DIFFUSE_SH", "REBLUR_DIFFUSE_DIRECTIONAL_OCCLUSION", "RELAX_DIFFUSE_SPECULAR", "RELAX_SPECULAR_SH",
                                  "SIGMA_SHADOW", "SIGMA_SHADOW_TRANSLUCENCY"])
def test_hip_matches_oracle_rect_size_changing_every_frame(name):
    worst = parity.run_parity(name, frames=6, verbose=True, resource=RESOURCE, rect_sizes=SIZES)
    assert worst <= parity.REL_TOL


@pytest.mark.gpu
def test_hip_matches_oracle_sub_rect_with_options():
    # checkerboard + 2.5D motion vectors + performance mode inside a sub-rect
    w, h = 144, 88
    worst = parity.run_parity("REBLUR_DIFFUSE_SPECULAR", width=w, height=h, frames=4, verbose=True, resource=(192, 112), extra_want=("mv2d",),
                              settings_overrides=dict(checkerboardMode=1, enablePerformanceMode=True), cs_kw=dict(isMotionVectorInWorldSpace=False, motionVectorScale=(1.0 / w, 1.0 / h, 1.0)))
    assert worst <= parity.REL_TOL


def _embed_guides_at(frame, resource, origin):
    """a rect-sized generated frame inside resource-sized planes: guide inputs at `origin`, everything else (noisy inputs) at (0, 0) -- the layout the
    reference addresses with CommonSettings::rectOrigin (WithRectOrigin: guides only)"""
    rw, rh = resource
    ox, oy = origin
    guides = ("mv", "normal_roughness", "viewz", "diff_confidence", "spec_confidence", "disocclusion_mix", "basecolor_metalness")
    out = {}
    for k, v in frame.items():
        if torch.is_tensor(v) and v.dim() >= 2 and v.dtype != torch.bool:
            big = torch.full([rh, rw] + list(v.shape[2:]), 33.0 if v.dtype.is_floating_point else 9, dtype=v.dtype, device=v.device)
            x0, y0 = (ox, oy) if k in guides else (0, 0)
            big[y0 : y0 + v.shape[0], x0 : x0 + v.shape[1]] = v
            v = big
        out[k] = v
    return out


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["REBLUR_DIFFUSE_SPECULAR", "RELAX_DIFFUSE_SPECULAR", "SIGMA_SHADOW"])
def test_shifted_rect_equals_the_rect_at_the_origin(name):
    """CommonSettings::rectOrigin != 0 (reference NRD_USE_VIEWPORT_OFFSET, Common.hlsli:64, :200-206): the guide inputs live at rectOrigin inside their
    resource-sized planes. The outputs must be those of the same frames with the guides at (0, 0) -- bit for bit -- and must
    still agree with the oracle."""
    from raytracingdenoiser_amd import synth

    frames, origin = 4, (16, 8)
    seq = [synth.render_frame(*RECT, f, want=tuple(parity.DENOISERS[name][1])) for f in range(frames)]
    results = []
    for org in ((0, 0), origin):
        run = parity.HipRun(name, *RESOURCE)
        outs = []
        for f, fr in enumerate(seq):
            frame = _embed_guides_at(fr, RESOURCE, org)
            cs = parity.common_settings(fr["camera"], seq[max(f - 1, 0)]["camera"], *RECT, f, resourceSize=RESOURCE, resourceSizePrev=RESOURCE, rectOrigin=org)
            run.step(frame, cs, parity.denoiser_settings(name, frame))
            outs.append({rt: run.output(rt)[: RECT[1], : RECT[0]].copy() for rt in run.outs})
        results.append(outs)
    for a, b in zip(*results):
        for rt in a:
            assert np.array_equal(a[rt], b[rt]), rt
    # and against the oracle with the shifted rect
    ora, hip = parity.OracleRun(name, *RESOURCE), parity.HipRun(name, *RESOURCE)
    for f, fr in enumerate(seq):
        frame = _embed_guides_at(fr, RESOURCE, origin)
        mk = lambda: parity.common_settings(fr["camera"], seq[max(f - 1, 0)]["camera"], *RECT, f, resourceSize=RESOURCE, resourceSizePrev=RESOURCE, rectOrigin=origin)
        ora.step(frame, mk(), parity.denoiser_settings(name, frame))
        hip.step(frame, mk(), parity.denoiser_settings(name, frame))
        for rt in ora.outs:
            assert parity.rel_error(hip.output(rt)[: RECT[1], : RECT[0]], ora.output(rt)[: RECT[1], : RECT[0]]) == 0.0, (f, rt)
