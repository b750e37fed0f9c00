"""The HIP path at BASELINE.json's full sizes (1440p REBLUR, 4K RELAX, 1080p SIGMA): the oracle would need minutes there, so the results are held to
size-independent properties of the chain (SURVEY.md section 8c known answers): constants are fixed points of the normalised filters, an
all-sky frame and splitScreen >= 1 leave / pass the data untouched, the accumulated-frame counters count 1, 2, 3, ... on a static scene, the
denoised mean equals the noisy mean while the variance drops, a frame cut into row strips (virtual ranks with halo exchange) equals the
uncut frame bit for bit, and REFERENCE is the running mean evaluated in fp32."""
import numpy as np
import pytest
import torch

import parity
from raytracingdenoiser_amd import api, sharding

RT = api.ResourceType
pytestmark = pytest.mark.gpu


def _run(name, seq, w, h, overrides=None, cs_kw=None, every_frame=None):
    run = parity.HipRun(name, w, h)
    for f, frame in enumerate(seq):
        cs = parity.common_settings(frame["camera"], seq[max(f - 1, 0)]["camera"], w, h, f, **(cs_kw or {}))
        run.step(frame, cs, parity.denoiser_settings(name, frame, overrides))
        if every_frame:
            every_frame(f, run)
    torch.cuda.synchronize()
    return run


@pytest.mark.parametrize("name,size,keys", [("REBLUR_DIFFUSE_SPECULAR", (2560, 1440), ("diff", "spec")), ("RELAX_DIFFUSE_SPECULAR", (3840, 2160), ("diff_relax", "spec_relax"))])
def test_constant_signal_is_a_fixed_point_at_full_size(name, size, keys):
    w, h = size
    seq = parity.generate_sequence(name, w, h, 4, static_camera=True, noise=False, device="cuda")
    const = torch.tensor([0.5, 0.0625, -0.03125, 0.25] if name.startswith("REBLUR") else [0.5, 0.25, 0.125, 2.0], dtype=torch.float16, device="cuda")
    for fr in seq:
        for k in keys:
            fr[k] = const.expand(h, w, 4).contiguous()
    run = _run(name, seq, w, h)
    m = ~seq[-1]["is_sky"].cpu().numpy()
    for rt in run.outs:
        out = run.output(rt)[m]
        assert np.max(np.abs(out[:, :3] - const.float().cpu().numpy()[:3])) < 4e-3, rt  # weighted means of a constant, up to fp16 storage rounding


@pytest.mark.parametrize("name,size", [("REBLUR_DIFFUSE_SPECULAR", (2560, 1440)), ("RELAX_DIFFUSE_SPECULAR_SH", (3840, 2160)), ("SIGMA_SHADOW", (1920, 1080))])
def test_all_sky_and_split_screen_at_full_size(name, size):
    w, h = size
    seq = parity.generate_sequence(name, w, h, 2, device="cuda")
    sky = [dict(fr, viewz=torch.full_like(fr["viewz"], 1.0e6)) for fr in seq]
    run = _run(name, sky, w, h)
    for rt in run.outs:
        assert not run.output(rt).any() or name.startswith("SIGMA")  # sky pixels are never written (cleared on the restart frame)
    run = _run(name, seq, w, h, cs_kw=dict(splitScreen=1.0))
    m = ~seq[-1]["is_sky"].cpu().numpy()
    if name.startswith("REBLUR"):
        assert np.array_equal(run.output(RT.OUT_DIFF_RADIANCE_HITDIST)[m], seq[-1]["diff"].float().cpu().numpy()[m])
        assert np.array_equal(run.output(RT.OUT_SPEC_RADIANCE_HITDIST)[m], seq[-1]["spec"].float().cpu().numpy()[m])
    elif name.startswith("RELAX"):
        assert np.array_equal(run.output(RT.OUT_DIFF_SH1)[m], seq[-1]["diff_relax_sh1"].float().cpu().numpy()[m])  # SH1 passes through unchanged (SH0 goes to YCoCg)


def test_accumulated_frames_count_up_at_full_size():
    name, (w, h) = "REBLUR_DIFFUSE_SPECULAR", (2560, 1440)
    seq = parity.generate_sequence(name, w, h, 6, static_camera=True, noise=False, device="cuda")
    m = ~seq[0]["is_sky"].cpu().numpy()

    def check(f, run):
        raw, fmt, pw = run.ex.read_pool_plane(RT.PERMANENT_POOL, 2)  # PREV_INTERNAL_DATA (R16_UINT): 6 + 6 bits of accumulated frames
        packed = raw[:, : pw * 2].copy().view(np.uint16)
        assert np.median((packed & 63)[m]) == f + 1 and np.median(((packed >> 6) & 63)[m]) == f + 1

    _run(name, seq, w, h, every_frame=check)


@pytest.mark.parametrize("name,size,pairs", [
    ("REBLUR_DIFFUSE_SPECULAR", (2560, 1440), ((RT.OUT_DIFF_RADIANCE_HITDIST, "diff"), (RT.OUT_SPEC_RADIANCE_HITDIST, "spec"))),
    ("RELAX_DIFFUSE_SPECULAR", (3840, 2160), ((RT.OUT_DIFF_RADIANCE_HITDIST, "diff_relax"), (RT.OUT_SPEC_RADIANCE_HITDIST, "spec_relax"))),
])
def test_mean_is_kept_and_noise_drops_at_full_size(name, size, pairs):
    w, h = size
    seq = parity.generate_sequence(name, w, h, 8, device="cuda")
    run = _run(name, seq, w, h)
    m = ~seq[-1]["is_sky"].cpu().numpy()
    for rt, key in pairs:
        out, noisy = run.output(rt)[m][:, 0], seq[-1][key].float().cpu().numpy()[m][:, 0]
        assert not np.isnan(out).any()
        assert abs(out.mean() - noisy.mean()) < 0.03 * noisy.mean() and out.std() < 0.9 * noisy.std()


def _virtual_rank_run(name, w, h, world, frames, scheme):
    """`world` virtual ranks on the one GPU (each its own instance, executor, arena and output planes; transfers emulated by copies between them: what RCCL would move) against
    an uncut frame. scheme "halo": HaloSharder -- halo exchange between pass segments, strips re-cut from the tile map, then the output all-gather (sharding.output_gather_ops);
    "allgather": FrameSharder -- redundant halo compute + one all-gather of every permanent plane and output. Returns per frame (sharded?, every rank's outputs == the uncut frame's)."""
    from test_sharding import _local_completion, _local_exchange

    from raytracingdenoiser_amd.executor import HipExecutor

    seq = parity.generate_sequence(name, w, h, frames, device="cuda")

    def make():
        inst = api.Instance([(0, parity.DENOISERS[name][0])])
        ex = HipExecutor(inst, w, h)
        outs = []
        for rt, dtype, ch, fmt in parity.output_planes(name, w, h):
            outs.append(torch.zeros((h, w, ch), dtype=dtype, device="cuda"))
            ex.bind(rt, outs[-1], fmt)
        return inst, ex, outs

    def prepare(inst, ex, f, frame):
        for rt, t, fmt in parity.user_planes(name, frame):
            ex.bind(rt, t, fmt)
        inst.set_denoiser_settings(0, parity.denoiser_settings(name, frame))
        assert inst.set_common_settings(parity.common_settings(frame["camera"], seq[max(f - 1, 0)]["camera"], w, h, f)) == api.Result.SUCCESS

    ref = make()
    runs = [make() for _ in range(world)]
    if scheme == "halo":
        ranks = [sharding.HaloSharder(ex, inst, w, h, r, world) for r, (inst, ex, outs) in enumerate(runs)]
    else:
        ranks = [sharding.FrameSharder(ex, inst, w, h, r, world, outs) for r, (inst, ex, outs) in enumerate(runs)]
        assert all(sh.rows is not None for sh in ranks)
    result = []
    for f, frame in enumerate(seq):
        prepare(ref[0], ref[1], f, frame)
        ref[1].denoise()
        if scheme == "halo":
            begun = []
            for (inst, ex, outs), sh in zip(runs, ranks):
                prepare(inst, ex, f, frame)
                begun.append(sh.begin_frame())
            plans = [b[0] for b in begun]
            assert len({p.fallback for p in plans}) == 1 and all(sh.bounds == ranks[0].bounds for sh in ranks)
            if plans[0].fallback:
                torch.cuda.synchronize()
                _local_completion(ranks, plans)
                for sh, (plan, ptr, n) in zip(ranks, begun):
                    sh.ex.execute_range(ptr, n, 0, n)
            else:
                for step in range(len(plans[0].steps)):
                    torch.cuda.synchronize()
                    _local_exchange(ranks, plans, step)
                    for sh, (plan, ptr, n) in zip(ranks, begun):
                        sh.run_step(plan, ptr, n, step)
                torch.cuda.synchronize()
                # owned rows first (before the reassembly can paper over anything), then the output all-gather, replayed with copies
                for (inst, ex, outs), sh in zip(runs, ranks):
                    rb, re = sh.rows
                    assert all(torch.equal(o[rb:re], ro[rb:re]) for o, ro in zip(outs, ref[2])), (f, sh.rank)
                assert plans[0].output_keys and len(plans[0].output_keys) == len(ref[2])
            # the reassembly: every rank stages its rows in its complete planes (HaloSharder.stage_outputs), then the all-gather (sharding.output_gather_ops), replayed with copies
            for sh, plan in zip(ranks, plans):
                sh.stage_outputs(plan)
            if not plans[0].fallback:
                for key, src, r0, r1 in sharding.output_gather_ops(ranks[0].bounds, plans[0].output_keys):
                    for dst, sh in enumerate(ranks):
                        if dst != src:
                            sh._complete_plane(key)[r0:r1].copy_(ranks[src]._complete_plane(key)[r0:r1])
            for sh, plan in zip(ranks, plans):
                sh.finish_frame(plan)
            sharded = not plans[0].fallback
        else:
            for (inst, ex, outs), sh in zip(runs, ranks):
                prepare(inst, ex, f, frame)
                sh.ex.denoise()
            torch.cuda.synchronize()
            for src, sh in enumerate(ranks):  # FrameSharder.exchange: the in-place all-gather of every plane's owned rows, replayed with copies
                rb, re = sh.rows
                for dst, other in enumerate(ranks):
                    if dst != src:
                        for mine, theirs in zip(sh.planes, other.planes):
                            theirs[rb:re].copy_(mine[rb:re])
            sharded = True
        torch.cuda.synchronize()
        rts = [rt for rt, dtype, ch, fmt in parity.output_planes(name, w, h)]
        if scheme == "halo":  # the complete planes are the sharder's (the bound OUT_* planes are working planes of the pass chain and hold the rank's rows)
            result.append((sharded, all(torch.equal(sh.complete_output(rt), ro) for sh in ranks for rt, ro in zip(rts, ref[2]))))
        else:
            result.append((sharded, all(torch.equal(o, ro) for (inst, ex, outs) in runs for o, ro in zip(outs, ref[2]))))
    return result, ranks


def test_row_strips_equal_the_whole_frame_at_full_size():
    """4 virtual ranks (halo exchange emulated by copies, strips re-cut from the tile map) at 1440p == the uncut frame, every output, every frame"""
    result, ranks = _virtual_rank_run("REBLUR_DIFFUSE_SPECULAR", 2560, 1440, 4, 4, "halo")
    assert [s for s, _ in result] == [False, True, True, True] and all(ok for _, ok in result)
    assert ranks[0].rebalanced == 1 and ranks[0].bounds[1] > 1440 // 4  # the sky strip at the top grew


# BASELINE.json configs[3] and configs[4] at N = 8: the screen cut into 8 row strips, both multi-GPU schemes; afterwards EVERY rank holds the complete,
# bit-identical output planes of every frame -- the north-star's "single all-gather to reassemble the output plane"
@pytest.mark.parametrize("name,size,scheme", [
    ("REBLUR_DIFFUSE_SPECULAR", (2560, 1440), "halo"), ("REBLUR_DIFFUSE_SPECULAR", (2560, 1440), "allgather"),
    ("RELAX_DIFFUSE_SPECULAR_SH", (3840, 2160), "halo"), ("RELAX_DIFFUSE_SPECULAR_SH", (3840, 2160), "allgather"),
])
def test_eight_virtual_ranks_hold_the_complete_output(name, size, scheme):
    result, ranks = _virtual_rank_run(name, size[0], size[1], 8, 3, scheme)
    assert all(ok for _, ok in result), result
    assert [s for s, _ in result][1:] == [True, True]  # the frames after the restart frame are really cut into strips


def test_reference_is_the_fp32_running_mean_at_full_size():
    from raytracingdenoiser_amd.executor import HipExecutor

    w, h, frames = 2560, 1440, 5
    inst = api.Instance([(0, api.Denoiser.REFERENCE)])
    ex = HipExecutor(inst, w, h)
    out = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda")
    ex.bind(RT.OUT_SIGNAL, out, api.Format.RGBA32_SFLOAT)
    g = torch.Generator(device="cuda").manual_seed(1)
    mean = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda")
    cam = __import__("raytracingdenoiser_amd.synth", fromlist=["Camera"]).Camera(w, h, 0, static=True)
    for f in range(frames):
        x = torch.rand((h, w, 4), generator=g, dtype=torch.float32, device="cuda")
        ex.bind(RT.IN_SIGNAL, x, api.Format.RGBA32_SFLOAT)
        assert inst.set_denoiser_settings(0, api.ReferenceSettings()) == api.Result.SUCCESS
        assert inst.set_common_settings(parity.common_settings(cam, cam, w, h, f)) == api.Result.SUCCESS
        ex.denoise()
        a = torch.tensor(1.0 / (1.0 + f), dtype=torch.float32, device="cuda")
        mean = mean + (x - mean) * a  # lerp(history, input, 1 / (1 + N)) with the reference's operation order (reference Reference.hpp:73, REFERENCE_TemporalAccumulation.cs.hlsl:24)
        torch.cuda.synchronize()
        assert torch.equal(out, mean), f
