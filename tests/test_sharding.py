"""Row-strip sharding. CPU: the strip all-gather with gloo, world_size 2. GPU: N virtual ranks on one
MI355X (one executor per rank, all-gather emulated by copies) must reproduce the single-GPU planes bit for bit."""
import os
import sys

import numpy as np
import pytest
import torch

from raytracingdenoiser_amd import api, sharding


def _gloo_worker(rank, world, port, h, pitch, q):
    import torch.distributed as dist

    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rb, re = sharding.strip_rows(h, rank, world)
    planes = [torch.full((h, pitch), 255, dtype=torch.uint8), torch.full((h, pitch * 2), 255, dtype=torch.uint8)]
    for k, p in enumerate(planes):
        p[rb:re] = (torch.arange(rb, re, dtype=torch.int32).unsqueeze(1) * (k + 1) % 251).to(torch.uint8)
    sharding.exchange_strips(planes, rb, re)
    ok = all(torch.equal(p, (torch.arange(h, dtype=torch.int32).unsqueeze(1) * (k + 1) % 251).to(torch.uint8).expand_as(p)) for k, p in enumerate(planes))
    q.put((rank, ok))
    dist.destroy_process_group()


def test_strip_all_gather_gloo_world2():
    import torch.multiprocessing as mp

    assert sharding.strip_rows(1440, 3, 8) == (540, 720) and sharding.strip_rows(1081, 1, 8) is None and sharding.strip_rows(100, 0, 1) is None
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 500)
    procs = [ctx.Process(target=_gloo_worker, args=(r, 2, port, 64, 256, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    assert results == [(0, True), (1, True)]


def _nccl_world1_worker(q):
    import torch.distributed as dist

    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(29600 + (os.getpid() % 300))
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1)
    planes = [torch.arange(64 * 256, dtype=torch.int32, device="cuda").to(torch.uint8).view(64, 256).contiguous(), torch.full((64, 512), 7, dtype=torch.uint8, device="cuda")]
    want = [p.clone() for p in planes]
    sharding.exchange_strips(planes, 0, 64)  # world size 1: the grouped RCCL all-gather must be the identity
    torch.cuda.synchronize()
    q.put(all(torch.equal(a, b) for a, b in zip(planes, want)))
    dist.destroy_process_group()


@pytest.mark.gpu
def test_grouped_rccl_all_gather_world1():
    # exercises the RCCL (backend "nccl") code path of exchange_strips, including the grouped-collective fast path, on one GPU
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_nccl_world1_worker, args=(q,))
    p.start()
    assert q.get(timeout=180) is True
    p.join(timeout=60)


@pytest.mark.gpu
@pytest.mark.parametrize("name,world,height,overrides", [
    ("REBLUR_DIFFUSE_SPECULAR", 3, 288, dict(maxBlurRadius=4.0, diffusePrepassBlurRadius=6.0, specularPrepassBlurRadius=6.0)),  # margins < strip: real partial compute
    ("REBLUR_DIFFUSE_SPECULAR", 2, 720, None),  # default radii: margin ~200 rows on a 360-row strip
    ("RELAX_DIFFUSE_SPECULAR_SH", 3, 288, dict(atrousIterationNum=4)),  # a-trous reach 2+2+5+10 rows plus history fix / clamping
    ("RELAX_DIFFUSE_SPECULAR", 2, 360, None),
])
def test_virtual_ranks_reproduce_single_gpu(name, world, height, overrides):
    import parity
    from raytracingdenoiser_amd.executor import HipExecutor

    W, H, frames = 256, height, 5
    RT, F = api.ResourceType, api.Format
    seq = parity.generate_sequence(name, W, H, frames)

    def make_run():
        inst = api.Instance([(0, parity.DENOISERS[name][0])])
        ex = HipExecutor(inst, W, H)
        outs = []
        for rt, dtype, ch, fmt in parity.output_planes(name, W, H):
            outs.append(torch.zeros((H, W, ch), dtype=dtype, device="cuda"))
            ex.bind(rt, outs[-1], fmt)
        return inst, ex, outs

    ref_inst, ref_ex, ref_outs = make_run()
    ranks = []
    for r in range(world):
        inst, ex, outs = make_run()
        ranks.append((inst, ex, outs, sharding.FrameSharder(ex, inst, W, H, r, world, outs)))
    assert all(s.rows is not None for *_, s in ranks)

    for f, frame in enumerate(seq):
        def step(inst, ex):
            for rt, t, fmt in parity.user_planes(name, frame):
                ex.bind(rt, t.cuda().contiguous(), fmt)
            inst.set_denoiser_settings(0, parity.denoiser_settings(name, frame, overrides))
            assert inst.set_common_settings(parity.common_settings(frame["camera"], seq[max(f - 1, 0)]["camera"], W, H, f)) == api.Result.SUCCESS
            ex.denoise()

        step(ref_inst, ref_ex)
        for inst, ex, outs, s in ranks:
            step(inst, ex)
        # emulate the all-gather: every rank's owned strip goes to all the others
        for _, _, _, src in ranks:
            rb, re = src.rows
            for _, _, _, dst in ranks:
                if dst is not src:
                    for ps, pd in zip(src.planes, dst.planes):
                        pd[rb:re].copy_(ps[rb:re])
        torch.cuda.synchronize()
        # the same list FrameSharder holds: full-resolution permanent planes, the outputs, full-resolution transient planes (round 6: their unwritten -- sky -- texels are read by later frames)
        ref_planes = ([ref_ex.pool_plane_tensor(RT.PERMANENT_POOL, i) for i, (_, ds) in enumerate(ref_inst.permanent_pool) if ds == 1]
                      + [o.view(-1).view(dtype=torch.uint8).view(H, -1) for o in ref_outs]
                      + [ref_ex.pool_plane_tensor(RT.TRANSIENT_POOL, i) for i, (_, ds) in enumerate(ref_inst.transient_pool) if ds == 1])
        assert len(ref_planes) == len(ranks[0][3].planes)
        for r, (_, _, _, s) in enumerate(ranks):
            for k, (a, b) in enumerate(zip(s.planes, ref_planes)):
                assert torch.equal(a, b), "frame %d rank %d plane %d differs from the single-GPU run" % (f, r, k)


@pytest.mark.parametrize("name,world,height,overrides", [
    ("RELAX_DIFFUSE_SPECULAR_SH", 2, 192, dict(atrousIterationNum=3)),
    ("REBLUR_DIFFUSE_SPECULAR", 3, 240, dict(maxBlurRadius=4.0, diffusePrepassBlurRadius=6.0, specularPrepassBlurRadius=6.0)),
])
def test_allgather_sharding_on_emulated_kernels(name, world, height, overrides):
    """FrameSharder on the CPU emulation of the device sources (no GPU): virtual ranks, the all-gather replayed with copies, a horizon that moves DOWN through the first strip
    boundary frame by frame -- texels that were geometry turn into sky and keep what their last writer left. Every plane FrameSharder holds (permanent, outputs, and since round 6
    the transient ones) equals the uncut frame's after every frame."""
    import parity
    from emu import emu_run

    W, H, frames = 128, height, 5
    RT = api.ResourceType
    lib = emu_run.load()
    seq = parity.generate_sequence(name, W, H, frames, device="cpu")
    for f, fr in enumerate(seq):
        z = fr["viewz"].clone()
        z.view(H, W)[: H // world - 26 + 5 * f] = 1.0e6
        fr["viewz"] = z

    def make_run():
        inst = api.Instance([(0, parity.DENOISERS[name][0])], lib=lib)
        ex = emu_run.EmuTorchExecutor(inst, W, H)
        outs = []
        for rt, dtype, ch, fmt in parity.output_planes(name, W, H):
            outs.append(torch.zeros((H, W, ch), dtype=dtype))
            ex.bind(rt, outs[-1], fmt)
        return inst, ex, outs

    def planes_of(inst, ex, outs):
        return ([ex.pool_plane_tensor(RT.PERMANENT_POOL, i) for i, (_, ds) in enumerate(inst.permanent_pool) if ds == 1]
                + [o.view(-1).view(dtype=torch.uint8).view(H, -1) for o in outs]
                + [ex.pool_plane_tensor(RT.TRANSIENT_POOL, i) for i, (_, ds) in enumerate(inst.transient_pool) if ds == 1])

    ref = make_run()
    ranks = []
    for r in range(world):
        run = make_run()
        ranks.append(run + (sharding.FrameSharder(run[1], run[0], W, H, r, world, run[2]),))
    assert all(s.rows is not None and s.rows[1] - s.rows[0] < H for *_, s in ranks)
    keep = []
    for f, frame in enumerate(seq):
        for inst, ex in [ref[:2]] + [rk[:2] for rk in ranks]:
            for rt, t, fmt in parity.user_planes(name, frame):
                keep.append(t.contiguous())
                ex.bind(rt, keep[-1], fmt)
            inst.set_denoiser_settings(0, parity.denoiser_settings(name, frame, overrides))
            assert inst.set_common_settings(parity.common_settings(frame["camera"], seq[max(f - 1, 0)]["camera"], W, H, f)) == api.Result.SUCCESS
            ex.denoise()
        for *_, src in ranks:
            rb, re = src.rows
            for *_, dst in ranks:
                if dst is not src:
                    for ps, pd in zip(src.planes, dst.planes):
                        pd[rb:re].copy_(ps[rb:re])
        ref_planes = planes_of(*ref)
        for r, (*_, s) in enumerate(ranks):
            assert len(s.planes) == len(ref_planes)
            for k, (a, b) in enumerate(zip(s.planes, ref_planes)):
                assert torch.equal(a, b), "frame %d rank %d plane %d differs from the uncut frame" % (f, r, k)


# ---------------------------------------------------------------------------------------------- halo-exchange sharding
def _halo_gloo_worker(rank, world, port, q):
    import torch.distributed as dist

    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    h, rows = 48, sharding.strip_rows(48, rank, world)
    truth = [(torch.arange(h, dtype=torch.int32).unsqueeze(1) * (k + 3) % 251).to(torch.uint8).expand(h, 64 * (k + 1)).contiguous() for k in range(2)]
    planes = [torch.full_like(t, 255) for t in truth]
    for p, t in zip(planes, truth):
        p[rows[0]:rows[1]] = t[rows[0]:rows[1]]
    sharding.exchange_halos(planes, rows, rank, world, [(0, 5), (1, 16)])
    ok = True
    for p, t, w in zip(planes, truth, (5, 16)):
        lo, hi = max(rows[0] - w, 0), min(rows[1] + w, h)
        ok &= torch.equal(p[lo:hi], t[lo:hi])                                        # own strip + halos are now correct
        ok &= bool((p[:lo] == 255).all()) and bool((p[hi:] == 255).all())            # nothing else was touched
    # the sharder's own transfer path: the batch is built once per plan step and reissued from the cache on later frames (the RCCL hot path)
    sh = sharding.HaloSharder(None, None, 64, h, rank, world, balance=False)
    sh.plane_tensor = lambda key: planes[key[1]]
    plan = sharding.HaloPlan()
    plan.steps = [([((0, 0), 5), ((0, 1), 16)], 0, 1)]
    for frame in range(3):
        for p, t in zip(planes, truth):
            p.fill_(255)
            p[rows[0]:rows[1]] = t[rows[0]:rows[1]] + frame  # new data every frame (uint8 wrap-around is fine)
        sharding.finish_halo_exchange(sh.start_exchange(plan, 0))
        for p, t, w in zip(planes, truth, (5, 16)):
            lo, hi = max(rows[0] - w, 0), min(rows[1] + w, h)
            ok &= torch.equal(p[lo:hi], t[lo:hi] + frame)
        ok &= (0 in plan._ops) == (world > 1)
    q.put((rank, ok))
    dist.destroy_process_group()


def test_halo_exchange_gloo_world3():
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29800 + (os.getpid() % 150)
    procs = [ctx.Process(target=_halo_gloo_worker, args=(r, 3, port, q)) for r in range(3)]
    for p in procs:
        p.start()
    results = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    assert results == [(0, True), (1, True), (2, True)]


def test_halo_plan_for_reblur_and_relax():
    """The plan is pure host logic: segments in front of the wide passes, margins that shrink to 0 at every segment end, carried-over planes
    exchanged at the frame start (widened by the motion bound), tile maps never exchanged, unknown reach -> unsharded frame."""
    import parity

    for name, (w, h), world, expect_segments in (("REBLUR_DIFFUSE_SPECULAR", (2560, 1440), 8, 3), ("RELAX_DIFFUSE_SPECULAR_SH", (3840, 2160), 8, 1)):
        inst = api.Instance([(0, parity.DENOISERS[name][0])])
        seq = parity.generate_sequence(name, 32, 18, 2)
        plans = []
        for f in range(2):
            inst.set_denoiser_settings(0, parity.denoiser_settings(name, seq[f]))
            assert inst.set_common_settings(parity.common_settings(seq[f]["camera"], seq[max(f - 1, 0)]["camera"], w, h, f)) == api.Result.SUCCESS
            r, ptr, n = inst.get_compute_dispatches_raw()
            ds = [api.Dispatch(ptr[i], inst.pipelines) for i in range(n)]
            small = {(int(api.ResourceType.TRANSIENT_POOL), i) for i, (fmt, d) in enumerate(inst.transient_pool) if d != 1}
            plans.append((sharding.plan_halo_exchange(ds, inst.dispatch_reach(ptr, n), sharding.strip_rows(h, 3, world), h, small_planes=small), ds))
        # the restart frame is sharded too since round 6: its clears are texel-local (reach 0), and a plane that is cleared before it is read is not carried over -- nothing
        # of last frame's is exchanged for it
        restart, rds = plans[0]
        clears = [i for i, d in enumerate(rds) if d.shader.startswith("Clear_")]
        cleared = {(int(t), idx) for i in clears for dt, t, idx in rds[i].resources if dt == api.DescriptorType.STORAGE_TEXTURE}
        assert clears and not restart.fallback and all(restart.reach[i] == 0 for i in clears)
        assert all(key not in cleared for items, _, _ in restart.steps for key, _ in items if items is restart.steps[0][0])
        plan, ds = plans[1]
        assert not plan.fallback and len(plan.steps) == expect_segments
        rb, re = sharding.strip_rows(h, 3, world)
        for items, first, count in plan.steps:
            assert plan.margins[first + count - 1] == 0  # the last pass of a segment produces exactly the owned rows
            assert all(key not in small for key, _ in items) and all(0 < width <= re - rb for _, width in items)
        # frame start: the history (planes read before they are written: + the motion bound) and, since round 6, the rows of the planes a later pass of the first segment reads
        # with a neighbourhood -- texels their writer skips (sky) hold last frame's content, which only the owner of a row has (plan_halo_exchange)
        carried = set(sharding.carried_over_planes(ds, small))
        assert plan.steps[0][0] and carried <= {key for key, _ in plan.steps[0][0]}
        assert all(width >= 32 for key, width in plan.steps[0][0] if key in carried) and any(key not in carried for key, _ in plan.steps[0][0])
        tiles = [i for i, d in enumerate(ds) if "ClassifyTiles" in d.shader]
        assert all(plan.row_begin[i] == -1 for i in tiles)
        others = [i for i in range(len(ds)) if i not in tiles]
        assert all(plan.row_begin[i] == rb - plan.margins[i] and plan.row_end[i] == re + plan.margins[i] for i in others)


@pytest.mark.parametrize("name,size,world,overrides,cs_kw", [
    ("REBLUR_DIFFUSE_SPECULAR", (2560, 1440), 8, None, None),
    ("REBLUR_DIFFUSE_SPECULAR", (2560, 1440), 4, dict(enablePerformanceMode=True, maxBlurRadius=12.0), None),
    ("REBLUR_DIFFUSE_SPECULAR", (640, 360), 8, None, None),                                   # 45-row strips: the PostBlur halo does not fit -> unsharded
    ("REBLUR_DIFFUSE_SPECULAR_OCCLUSION", (1920, 1080), 4, None, None),                        # the user's OUT planes are the history
    ("REBLUR_DIFFUSE_SH", (1920, 1080), 3, dict(hitDistanceReconstructionMode=1), None),       # the pre-pass gathers from the reconstruction's output: reach = its blur radius
    ("RELAX_DIFFUSE_SPECULAR", (1920, 1080), 3, dict(hitDistanceReconstructionMode=2), None),
    ("RELAX_DIFFUSE_SPECULAR_SH", (3840, 2160), 8, None, None),
    ("RELAX_SPECULAR", (1920, 1080), 2, dict(atrousIterationNum=8), None),
    ("RELAX_DIFFUSE", (1920, 1080), 2, dict(enableAntiFirefly=True), None),                    # Copy (texel to texel) + AntiFirefly (3x3)
    ("SIGMA_SHADOW", (1920, 1080), 4, None, None),
])
def test_c_planner_equals_python_planner(name, size, world, overrides, cs_kw):
    """nrdHipPlanHaloExchange (the plan a C++ host uses to drive RCCL itself) against sharding.plan_halo_exchange, for every rank, uneven strips,
    the restart frame and two steady-state frames (ping-pong)"""
    import parity

    w, h = size
    inst = api.Instance([(0, parity.DENOISERS[name][0])])
    seq = parity.generate_sequence(name, 32, 18, 3)
    small = {(int(pool), i) for pool, descs in ((api.ResourceType.PERMANENT_POOL, inst.permanent_pool), (api.ResourceType.TRANSIENT_POOL, inst.transient_pool))
             for i, (fmt, d) in enumerate(descs) if d != 1}
    bounds = [0] + [(r * h // world) + (7 if r % 2 else -5) for r in range(1, world)] + [h]  # uneven on purpose
    sharded = 0
    for f in range(3):
        inst.set_denoiser_settings(0, parity.denoiser_settings(name, seq[f], overrides))
        assert inst.set_common_settings(parity.common_settings(seq[f]["camera"], seq[max(f - 1, 0)]["camera"], w, h, f, **(cs_kw or {}))) == api.Result.SUCCESS
        r, ptr, n = inst.get_compute_dispatches_raw()
        ds = [api.Dispatch(ptr[i], inst.pipelines) for i in range(n)]
        reach = inst.dispatch_reach(ptr, n)
        for rank in range(world):
            want = sharding.plan_halo_exchange(ds, reach, (bounds[rank], bounds[rank + 1]), h, 32, 24, small, min(b - a for a, b in zip(bounds, bounds[1:])))
            fallback, steps, row_begin, row_end = inst.plan_halo_exchange(ptr, n, bounds, rank, h, 32, 24)
            assert fallback == want.fallback, (f, rank)
            if fallback:
                assert steps == [] and row_begin == [-1] * n and row_end == [h] * n
                continue
            sharded += 1
            assert [(items, first, count) for items, first, count, early in steps] == [([(tuple(k), wd) for k, wd in items], first, count) for items, first, count in want.steps]
            assert [early for *_, early in steps] == want.early
            assert row_begin == want.row_begin and row_end == want.row_end
    # round 6: every frame of every family is sharded -- restart frames (texel-local clears), hit-distance reconstruction (the pre-pass behind it reaches its blur radius), RELAX's
    # anti-firefly pair, SIGMA (Blur / PostBlur 35 rows, TemporalStabilization 2 + the motion bound on the history Copy forwards) -- unless a halo does not fit the strips
    expect_sharded = size != (640, 360)
    assert sharded == (3 * world if expect_sharded else 0)


def _local_exchange(ranks, plans, step):
    """what exchange_halos does over RCCL, emulated with copies between the executors of one process"""
    for r, (sh, plan) in enumerate(zip(ranks, plans)):
        items = plan.steps[step][0]
        for k, kind, peer, r0, r1 in sharding.halo_transfers(sh.rows, r, len(ranks), [(i, w) for i, (_, w) in enumerate(items)]):
            if kind == "recv":
                key = items[k][0]
                sh.plane_tensor(key)[r0:r1].copy_(ranks[peer].plane_tensor(key)[r0:r1])


def _poison_beyond_halo(ranks, plans, step):
    """every row of an exchanged plane that lies OUTSIDE the strip + the halo the plan declared is overwritten with 0xFF bytes (NaN in the fp16 / fp32 planes, the
    largest code in the UNORM / UINT ones) before the segment runs. A pass that reads further than nrdHipGetDispatchReach says then produces NaNs (or visibly different values) in the
    rows the rank owns, instead of quietly reading whatever an earlier frame left there."""
    for sh, plan in zip(ranks, plans):
        rb, re = sh.rows
        for key, width in plan.steps[step][0]:
            t = sh.plane_tensor(key)
            rows = t.shape[0]
            scale = rows / float(sh.height)  # (no down-sampled plane is ever exchanged: small_planes)
            assert scale == 1.0
            if rb - width > 0:
                t[: rb - width].fill_(0xFF)
            if re + width < rows:
                t[re + width:].fill_(0xFF)


def _local_completion(ranks, plans):
    """what HaloSharder.complete_planes does with broadcasts, emulated with copies: every rank receives the other ranks' strips"""
    for r, (sh, plan) in enumerate(zip(ranks, plans)):
        for key in plan.complete_keys:
            for src, other in enumerate(ranks):
                if src != r:
                    b0, b1 = other.bounds[src], other.bounds[src + 1]
                    sh.plane_tensor(key)[b0:b1].copy_(other.plane_tensor(key)[b0:b1])


@pytest.mark.gpu
@pytest.mark.parametrize("name,world,height,overrides,balance,fallback_frame", [
    ("REBLUR_DIFFUSE_SPECULAR", 2, 720, None, False, None),                     # default radii: segments in front of Blur and PostBlur, 123-row halos
    ("REBLUR_DIFFUSE_SPECULAR", 2, 720, None, True, 3),                         # strips re-cut from the tile map; an unsharded frame in mid-sequence
    ("REBLUR_DIFFUSE_SPECULAR", 3, 288, dict(maxBlurRadius=10.0), True, 2),      # three ranks: a middle strip with two neighbours
    ("REBLUR_DIFFUSE_SPECULAR_OCCLUSION", 2, 360, dict(maxBlurRadius=15.0), True, 3),  # history = the user's OUT planes
    ("RELAX_DIFFUSE_SPECULAR_SH", 2, 360, None, True, 2),
    ("REBLUR_DIFFUSE_SPECULAR", 2, 480, dict(maxBlurRadius=15.0), "recut", None),  # a deliberate unsharded frame after every 2 sharded ones
    ("REBLUR_DIFFUSE_SPECULAR", 2, 720, dict(maxBlurRadius=60.0), False, None),     # huge radii: Blur / PostBlur reach 80 / 181 rows (round 6: the derived bound; 2 x the radius said 122 / 242)
    ("SIGMA_SHADOW", 2, 360, None, False, None),                                    # round 6: SIGMA's passes declare their reach and take row bands
    ("REBLUR_DIFFUSE_SPECULAR", 2, 480, dict(maxBlurRadius=0.0, minBlurRadius=5.0), False, None),  # the minimum radius wins: specular rings reach 4 x, which 2 x the radius did not cover
])
def test_halo_sharding_virtual_ranks_reproduce_single_gpu(name, world, height, overrides, balance, fallback_frame):
    import parity
    from raytracingdenoiser_amd.executor import HipExecutor

    W, H, frames = 256, height, 6
    RT = api.ResourceType
    seq = parity.generate_sequence(name, W, H, frames)

    def make_run():
        inst = api.Instance([(0, parity.DENOISERS[name][0])])
        ex = HipExecutor(inst, W, H)
        outs = []
        for rt, dtype, ch, fmt in parity.output_planes(name, W, H):
            outs.append(torch.zeros((H, W, ch), dtype=dtype, device="cuda"))
            ex.bind(rt, outs[-1], fmt)
        return inst, ex, outs

    def prepare(inst, ex, f, frame):
        for rt, t, fmt in parity.user_planes(name, frame):
            ex.bind(rt, t.cuda().contiguous(), fmt)
        ov = dict(overrides or {})
        inst.set_denoiser_settings(0, parity.denoiser_settings(name, frame, ov))
        assert inst.set_common_settings(parity.common_settings(frame["camera"], seq[max(f - 1, 0)]["camera"], W, H, f)) == api.Result.SUCCESS

    ref_inst, ref_ex, ref_outs = make_run()
    runs = [make_run() for _ in range(world)]
    recut = balance == "recut"
    ranks = [sharding.HaloSharder(ex, inst, W, H, r, world, max_motion_rows=16, balance=bool(balance), recut_every=2 if recut else 0) for r, (inst, ex, outs) in enumerate(runs)]
    uniform = list(ranks[0].bounds)
    sharded_frames = 0
    for f, frame in enumerate(seq):
        prepare(ref_inst, ref_ex, f, frame)
        ref_ex.denoise()
        begun = []
        for (inst, ex, outs), sh in zip(runs, ranks):
            prepare(inst, ex, f, frame)
            # the fallback frame: the application reports object motion beyond the history halo (begin_frame's motion_rows) -- this frame runs unsharded on every rank
            begun.append(sh.begin_frame(motion_rows=1000.0 if f == fallback_frame else None))
        plans = [b[0] for b in begun]
        assert len({p.fallback for p in plans}) == 1 and all(sh.bounds == ranks[0].bounds for sh in ranks)  # every rank takes the same decisions
        if plans[0].fallback:
            assert all(bool(p.complete_keys) == (f > 0) for p in plans)  # nothing to complete on the very first frame
            torch.cuda.synchronize()
            _local_completion(ranks, plans)
            for sh, (plan, ptr, n) in zip(ranks, begun):
                sh.ex.execute_range(ptr, n, 0, n)
        else:
            sharded_frames += 1
            for step in range(len(plans[0].steps)):
                torch.cuda.synchronize()
                _local_exchange(ranks, plans, step)
                _poison_beyond_halo(ranks, plans, step)
                for sh, (plan, ptr, n) in zip(ranks, begun):
                    sh.run_step(plan, ptr, n, step)
        for sh, plan in zip(ranks, plans):
            sh.finish_frame(plan)
        torch.cuda.synchronize()
        # every rank's owned rows of every output equal the single-GPU result, every frame (history errors would surface one frame later)
        for r, ((inst, ex, outs), sh) in enumerate(zip(runs, ranks)):
            rb, re = sh.rows
            for o, ro in zip(outs, ref_outs):
                assert torch.equal(o[rb:re], ro[rb:re]), (name, f, r)
    # the restart frame: run whole where the strips are balanced (its tile map is what they are cut from), sharded otherwise (round 6); recut: frames 0 and 3 run unsharded (0 | 1 2 | 3 | 4 5)
    assert sharded_frames == (4 if recut else frames - (1 if balance else 0) - (fallback_frame is not None))
    if balance:
        assert ranks[0].rebalanced >= 1  # (a re-cut that lands on the same strips does not count) and ranks[0].bounds != uniform and ranks[0].bounds[0] == 0 and ranks[0].bounds[-1] == H  # sky at the top: the top strip grows
        assert ranks[0].bounds[1] > uniform[1]
    else:
        assert ranks[0].bounds == uniform


# ---- dynamic resolution under sharding (round 6): a sub-rect of the resource is cut into strips like a full frame; a resolution STEP has no bounded halo (nrdHipGetDispatchReach: -1) and
# runs unsharded after the planes have been completed -- and "completed" has to include the planes whose sky texels a neighbourhood pass reads after an earlier pass of the SAME frame
# wrote them (sharding.carried_over_planes): at a step the silhouettes move by whole pixels, and a rank's copy of such a plane is stale outside its strip. Found with 3 uniform strips.
DYNRES_STEPS = [(1.0, 1.0), (1.0, 1.0), (0.8125, 0.8125), (0.8125, 0.8125), (0.8125, 0.8125), (1.0, 1.0), (1.0, 1.0), (0.5, 0.5), (0.5, 0.5)]


def _dynamic_resolution_case(name, world, resource, balance, backend, overrides=None, steps=DYNRES_STEPS):
    how, sizes = _scenario_case(name, world, resource, balance, backend, [dict(scale=s, settings=overrides) for s in steps])
    # the frame after every change of the rect size runs whole, a constant sub-rect is cut into strips
    changed = [f > 0 and sizes[f] != sizes[f - 1] for f in range(len(sizes))]
    assert all(how[f] == "whole" for f in range(len(sizes)) if changed[f]) and sum(h == "strips" for h in how) >= len(sizes) - sum(changed) - (1 if balance else 0), how


def _scenario_case(name, world, resource, balance, backend, scenario, graph=False):
    """`world` virtual ranks (HaloSharder, transfers replayed with copies, rows beyond the declared halos poisoned) against an uncut run over a SCENARIO: one dict per frame with
    scale = (sx, sy) of the rect inside the resource (default 1, 1), origin = CommonSettings::rectOrigin, camera = index of the generated camera path (default: the frame index),
    cs = CommonSettings fields, settings = denoiser-settings overrides. Every rank's owned rows of every output are compared after every frame. Returns (["whole" | "strips" per frame], rect sizes)."""
    import parity
    from raytracingdenoiser_amd import synth

    RW, RH = resource
    sizes = [(int(RW * e.get("scale", (1.0, 1.0))[0]), int(RH * e.get("scale", (1.0, 1.0))[1])) for e in scenario]
    device = "cuda" if backend == "hip" else "cpu"
    raw = [synth.render_frame(*sizes[f], scenario[f].get("camera", f), want=tuple(parity.DENOISERS[name][1]), device=device) for f in range(len(sizes))]
    if backend == "hip":
        from raytracingdenoiser_amd.executor import HipExecutor as Executor

        lib = None
    else:
        from emu import emu_run

        Executor, lib = emu_run.EmuTorchExecutor, emu_run.load()

    def make_run():
        inst = api.Instance([(0, parity.DENOISERS[name][0])], **({"lib": lib} if lib is not None else {}))
        ex = Executor(inst, RW, RH)
        outs = []
        for rt, dtype, ch, fmt in parity.output_planes(name, RW, RH):
            outs.append(torch.zeros((RH, RW, ch), dtype=dtype, device=device))
            ex.bind(rt, outs[-1], fmt)
        return inst, ex, outs

    keep = []

    def prepare(inst, ex, f):
        origin = scenario[f].get("origin")  # CommonSettings::rectOrigin: the guide inputs live at an offset inside their planes (served through rect-at-origin copies)
        if origin:
            from test_dynamic_resolution import _embed_guides_at

            frame = _embed_guides_at(raw[f], resource, origin)
        else:
            frame = parity.embed_in_resource(raw[f], resource)
        for rt, t, fmt in parity.user_planes(name, frame):
            keep.append(t.contiguous())
            ex.bind(rt, keep[-1], fmt)
        assert inst.set_denoiser_settings(0, parity.denoiser_settings(name, frame, dict(scenario[f].get("settings") or {}))) == api.Result.SUCCESS
        w, h = sizes[f]
        kw = dict(resourceSize=resource, resourceSizePrev=resource, rectSize=(w, h), rectSizePrev=sizes[max(f - 1, 0)])
        kw.update(dict(rectOrigin=origin) if origin else {})
        kw.update(scenario[f].get("cs") or {})
        assert inst.set_common_settings(parity.common_settings(raw[f]["camera"], raw[max(f - 1, 0)]["camera"], w, h, f, **kw)) == api.Result.SUCCESS

    def sync():
        if backend == "hip":
            torch.cuda.synchronize()

    ref = make_run()
    runs = [make_run() for _ in range(world)]
    if graph:  # the ranks launch every pass segment as a hipGraph (nrdHipSetGraphMode), the uncut run stays eager
        for inst, ex, outs in runs:
            ex.set_graph_mode(True)
    ranks = [sharding.HaloSharder(ex, inst, RW, RH, r, world, balance=balance) for r, (inst, ex, outs) in enumerate(runs)]
    how = []
    for f in range(len(sizes)):
        prepare(ref[0], ref[1], f)
        ref[1].denoise()
        begun = []
        for (inst, ex, outs), sh in zip(runs, ranks):
            prepare(inst, ex, f)
            begun.append(sh.begin_frame())
        plans = [b[0] for b in begun]
        assert len({p.fallback for p in plans}) == 1
        how.append("whole" if plans[0].fallback else "strips")
        if plans[0].fallback:
            sync()
            _local_completion(ranks, plans)
            for sh, (plan, ptr, n) in zip(ranks, begun):
                sh.ex.execute_range(ptr, n, 0, n)
        else:
            for step in range(len(plans[0].steps)):
                sync()
                _local_exchange(ranks, plans, step)
                _poison_beyond_halo(ranks, plans, step)
                for sh, (plan, ptr, n) in zip(ranks, begun):
                    sh.run_step(plan, ptr, n, step)
        sync()
        for r, ((inst, ex, outs), sh) in enumerate(zip(runs, ranks)):  # owned rows first: before the reassembly can paper over anything
            rb, re = sh.rows
            for o, ro in zip(outs, ref[2]):
                assert torch.equal(o[rb:re], ro[rb:re]), (name, world, "frame", f, how[-1], sizes[f], "rank", r)
        # the reassembly (HaloSharder.stage_outputs + the output all-gather, replayed with copies): afterwards EVERY rank holds the complete OUT_* planes, texels no frame wrote included
        for sh, plan in zip(ranks, plans):
            sh.stage_outputs(plan)
        if not plans[0].fallback:
            for key, src, r0, r1 in sharding.output_gather_ops(ranks[0].bounds, plans[0].output_keys):
                for dst, sh in enumerate(ranks):
                    if dst != src:
                        sh._complete_plane(key)[r0:r1].copy_(ranks[src]._complete_plane(key)[r0:r1])
        for sh, plan in zip(ranks, plans):
            sh.finish_frame(plan)
        sync()
        for r, sh in enumerate(ranks):
            for (rt, dtype, ch, fmt), ro in zip(parity.output_planes(name, RW, RH), ref[2]):
                assert torch.equal(sh.complete_output(rt), ro), (name, world, "frame", f, how[-1], sizes[f], "rank", r, "complete output", api.ResourceType(rt).name)
    return how, sizes


@pytest.mark.gpu
@pytest.mark.parametrize("name,world,balance", [("REBLUR_DIFFUSE_SPECULAR", 3, False), ("REBLUR_DIFFUSE_SPECULAR", 2, True), ("RELAX_DIFFUSE_SPECULAR", 3, False), ("RELAX_DIFFUSE_SPECULAR_SH", 2, True),
                                                ("SIGMA_SHADOW", 3, False)])
def test_halo_sharding_under_dynamic_resolution(name, world, balance):
    _dynamic_resolution_case(name, world, (256, 480), balance, "hip")


@pytest.mark.parametrize("name,world,resource,overrides", [
    ("REBLUR_DIFFUSE_SPECULAR", 3, (256, 480), None),
    ("RELAX_DIFFUSE_SPECULAR", 3, (128, 240), dict(atrousIterationNum=3, diffusePrepassBlurRadius=6.0, specularPrepassBlurRadius=6.0)),
])
def test_halo_sharding_under_dynamic_resolution_on_emulated_kernels(name, world, resource, overrides):
    """the same on the CPU emulation of the device sources (no GPU)"""
    _dynamic_resolution_case(name, world, resource, False, "emu", overrides)


def test_an_unsharded_frame_needs_the_same_frame_planes_completed_too(monkeypatch):
    """negative control: with the completion list of rounds 2-5 (planes read before they are written, nothing else) the resolution step of the REBLUR case above is NOT
    bit-identical on the middle rank"""
    orig = sharding.carried_over_planes
    monkeypatch.setattr(sharding, "carried_over_planes", lambda dispatches, small_planes=(), reach=None: orig(dispatches, small_planes, None))
    with pytest.raises(AssertionError, match="whole"):
        _dynamic_resolution_case("REBLUR_DIFFUSE_SPECULAR", 3, (256, 480), False, "emu")


# ---- other kinds of frames in mid-sequence (round 6): history restarts with and without clears, a camera cut (the motion exceeds every halo: that frame runs whole), split screen,
# settings that change the pass list and every reach from one frame to the next (blur radii, anti-firefly, hit-distance reconstruction, performance mode, a-trous iterations)
def _scenarios(name):
    AM = api.AccumulationMode
    settings = {
        "REBLUR_DIFFUSE_SPECULAR": [{}, {}, {}, dict(maxBlurRadius=12.0), dict(maxBlurRadius=12.0), dict(maxBlurRadius=12.0, enableAntiFirefly=True),
                                    dict(enableAntiFirefly=True, hitDistanceReconstructionMode=1), dict(hitDistanceReconstructionMode=1), {}, dict(enablePerformanceMode=True), {}],
        "RELAX_DIFFUSE_SPECULAR": [{}, {}, {}, dict(atrousIterationNum=3), dict(atrousIterationNum=3), dict(atrousIterationNum=3, enableAntiFirefly=True),
                                   dict(enableAntiFirefly=True, hitDistanceReconstructionMode=1), dict(hitDistanceReconstructionMode=1), {}, dict(atrousIterationNum=6), {}],
        "SIGMA_SHADOW": [{}, {}, {}, dict(maxStabilizedFrameNum=0), {}, {}],
    }[name]
    return {
        "restarts": [{}, {}, {}, dict(cs=dict(accumulationMode=int(AM.RESTART))), {}, {}, dict(cs=dict(accumulationMode=int(AM.CLEAR_AND_RESTART))), {}, {}],
        "camera_cut": [dict(camera=c) for c in (0, 1, 2, 3, 40, 41, 42, 5, 6)],
        "split_screen": [{}, {}, {}, dict(cs=dict(splitScreen=0.5)), dict(cs=dict(splitScreen=0.5)), {}, {}],
        "settings_change": [dict(settings=e) for e in settings],
        "shifted_rect": [dict(scale=(0.8125, 0.8125), origin=o) for o in ((0, 0), (0, 0), (16, 8), (16, 8), (16, 8), (32, 40), (32, 40), (0, 0), (0, 0))],  # CommonSettings::rectOrigin
    }


def _check_scenario(name, kind, world, balance, backend, graph=False):
    how, _ = _scenario_case(name, world, (256, 480), balance, backend, _scenarios(name)[kind], graph=graph)
    if kind == "camera_cut":
        assert how[4] == "whole" and how[7] == "whole" and how[5] == "strips", how  # the cuts run unsharded, the frames between them in strips
    else:
        assert how.count("whole") == (1 if balance else 0), how


@pytest.mark.parametrize("name,kind,world,balance", [("REBLUR_DIFFUSE_SPECULAR", "settings_change", 3, False), ("RELAX_DIFFUSE_SPECULAR", "camera_cut", 3, True),
                                                     ("SIGMA_SHADOW", "restarts", 2, False)])
def test_halo_sharding_scenarios_on_emulated_kernels(name, kind, world, balance):
    _check_scenario(name, kind, world, balance, "emu")


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["REBLUR_DIFFUSE_SPECULAR", "RELAX_DIFFUSE_SPECULAR", "SIGMA_SHADOW"])
@pytest.mark.parametrize("kind", ["restarts", "camera_cut", "split_screen", "settings_change", "shifted_rect"])
def test_halo_sharding_scenarios(name, kind):
    for world, balance in ((2, False), (3, False), (3, True)):
        _check_scenario(name, kind, world, balance, "hip")
    _check_scenario(name, kind, 3, False, "hip", graph=True)  # pass segments with row ranges as hipGraphs: graphs are re-parametrised or rebuilt as the lists and ranges change


@pytest.mark.gpu
def test_the_poisoned_halo_check_fires_when_the_reach_is_under_declared():
    """negative control of the test above: the same default-radius case in a child process whose NRD_HIP_SPECULAR_REACH_SLACK halves the declared Blur / PostBlur reach must FAIL"""
    import subprocess
    import sys

    case = "test_halo_sharding_virtual_ranks_reproduce_single_gpu and REBLUR_DIFFUSE_SPECULAR-2-720-None-False-None"
    cmd = [sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-x", "-n", "0", "-m", "gpu", "-p", "no:cacheprovider", "-k", case]
    ok = subprocess.run(cmd, env=dict(os.environ), capture_output=True, text=True, timeout=900)
    assert ok.returncode == 0 and "1 passed" in ok.stdout, ok.stdout[-2000:]
    bad = subprocess.run(cmd, env=dict(os.environ, NRD_HIP_SPECULAR_REACH_SLACK="0.5"), capture_output=True, text=True, timeout=900)
    assert bad.returncode != 0 and "1 failed" in bad.stdout, bad.stdout[-2000:]


def test_balanced_bounds():
    # 30 cheap tile rows (sky) above 60 expensive ones, 8 ranks, strips of at least 123 rows: the geometry is spread over 128-row strips
    b = sharding.balanced_bounds([0.5] * 30 + [10.0] * 60, 1440, 8, 123)
    assert b == [0, 544, 672, 800, 928, 1056, 1184, 1312, 1440]
    assert sharding.balanced_bounds([1.0] * 10, 160, 4, 16) == [0, 32, 64, 112, 160]       # uniform cost: near-uniform strips, tile aligned
    assert sharding.balanced_bounds([1.0] * 10, 150, 4, 64) is None                         # 4 strips of >= 64 rows do not fit into 150 rows
    b = sharding.balanced_bounds([1.0] * 7, 101, 2, 16)                                     # ragged height: the last strip ends at the frame height
    assert b[0] == 0 and b[-1] == 101 and b[1] % 16 == 0


def _halo_nccl_world1_worker(q):
    import torch.distributed as dist

    import parity
    from raytracingdenoiser_amd.executor import HipExecutor

    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(29650 + (os.getpid() % 300))
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1)
    name, W, H = "REBLUR_DIFFUSE_SPECULAR", 128, 64
    seq = parity.generate_sequence(name, W, H, 3)
    results = []
    for sharded in (False, True):
        inst = api.Instance([(0, parity.DENOISERS[name][0])])
        ex = HipExecutor(inst, W, H)
        outs = []
        for rt, dtype, ch, fmt in parity.output_planes(name, W, H):
            outs.append(torch.zeros((H, W, ch), dtype=dtype, device="cuda"))
            ex.bind(rt, outs[-1], fmt)
        sh = sharding.HaloSharder(ex, inst, W, H, 0, 1) if sharded else None
        for f, frame in enumerate(seq):
            for rt, t, fmt in parity.user_planes(name, frame):
                ex.bind(rt, t.cuda().contiguous(), fmt)
            inst.set_denoiser_settings(0, parity.denoiser_settings(name, frame))
            inst.set_common_settings(parity.common_settings(frame["camera"], seq[max(f - 1, 0)]["camera"], W, H, f))
            sh.denoise() if sharded else ex.denoise()
        if sharded:
            sh.wait_outputs()  # the output all-gather of the last frame: with one rank it is the whole plane gathered onto itself -- the RCCL code path (grouped, asynchronous) this box can run
        torch.cuda.synchronize()
        results.append([o.clone() for o in outs])
        if sharded:
            assert sh.gather_frames >= 1
            # the device-resident motion reduction (round 6): kernel -> device word -> RCCL all-reduce (MAX; one rank here) -> one read-back == the synchronous measurement
            r, ptr, n = inst.get_compute_dispatches_raw()
            assert sh._measure_motion_over_ranks((ptr, n), (0, H)) == ex.measure_motion_rows(ptr, n, 0, H) > 0.0
            results.append([sh.complete_output(rt).clone() for rt, dtype, ch, fmt in parity.output_planes(name, W, H)])
    q.put(all(torch.equal(a, b) for a, b in zip(results[0], results[1])) and all(torch.equal(a, b) for a, b in zip(results[0], results[2])))
    dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("list_form", [False, True])
def test_halo_sharder_under_rccl_world1(list_form):
    # world size 1: no neighbours, so no transfers -- but the whole HaloSharder.denoise() path (planning, segment execution) runs under the
    # RCCL process group exactly as bench.py drives it, and must reproduce executor.denoise()
    import torch.multiprocessing as mp

    # (round 5: HaloSharder.denoise() ends with the output all-gather; a group of one rank gathers the plane onto itself -- the grouped asynchronous
    # all_gather_into_tensor and, list_form, the list form RCCL runs for unequal strips: the only RCCL collectives this one-GPU box can execute)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    if list_form:
        os.environ["NRD_HIP_GATHER_LIST_FORM"] = "1"
    try:
        p = ctx.Process(target=_halo_nccl_world1_worker, args=(q,))
        p.start()
        assert q.get(timeout=240) is True
        p.join(timeout=60)
    finally:
        os.environ.pop("NRD_HIP_GATHER_LIST_FORM", None)


def _halo_two_process_worker(rank, world, port, name, W, H, frames, q):
    import torch.distributed as dist

    import parity
    from raytracingdenoiser_amd.executor import HipExecutor

    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)  # both ranks share cuda:0, which RCCL refuses; gloo stages the bands through the host
    seq = parity.generate_sequence(name, W, H, frames)

    def run(sharded):
        inst = api.Instance([(0, parity.DENOISERS[name][0])])
        ex = HipExecutor(inst, W, H)
        outs = []
        for rt, dtype, ch, fmt in parity.output_planes(name, W, H):
            outs.append(torch.zeros((H, W, ch), dtype=dtype, device="cuda"))
            ex.bind(rt, outs[-1], fmt)
        sh = sharding.HaloSharder(ex, inst, W, H, rank, world, max_motion_rows=16) if sharded else None
        per_frame = []
        for f, frame in enumerate(seq):
            for rt, t, fmt in parity.user_planes(name, frame):
                ex.bind(rt, t.cuda().contiguous(), fmt)
            # frame 2 has a pass without a bounded reach (blur rings as large as their distance: ReblurBlurReachRows): it runs unsharded, after every rank has received the
            # other strips of the history planes
            inst.set_denoiser_settings(0, parity.denoiser_settings(name, frame, dict(maxBlurRadius=400.0) if f == 2 else None))
            inst.set_common_settings(parity.common_settings(frame["camera"], seq[max(f - 1, 0)]["camera"], W, H, f))
            sh.denoise() if sharded else ex.denoise()
            if sharded:
                sh.wait_outputs()
            torch.cuda.synchronize()
            rows = sh.rows if sharded else (0, H)
            per_frame.append(([sh.complete_output(rt).clone() for rt, dtype, ch, fmt in parity.output_planes(name, W, H)] if sharded else [o.clone() for o in outs], rows))
        return per_frame, (sh.rebalanced if sharded else 0), (sh.exchanged_bytes if sharded else 0)

    ref, _, _ = run(False)
    got, rebalanced, exchanged = run(True)
    # the owned strip can change from frame to frame (re-cut from the tile map after every unsharded frame); since round 5 the frame ends with the output all-gather
    # (host-staged broadcasts over gloo), so the complete planes are compared on both ranks
    ok = all(torch.equal(a, b) for (fa, _), (fb, rows) in zip(ref, got) for a, b in zip(fa, fb))
    q.put((rank, ok, exchanged > 0 and rebalanced >= 1))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
def test_halo_sharding_two_processes_one_gpu():
    # the real thing end to end -- two processes, each planning and running its strip, exchanging halo bands by message passing (odd height:
    # uneven strips) -- only with gloo + host staging instead of RCCL, because the box has a single GPU
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29400 + (os.getpid() % 100)
    procs = [ctx.Process(target=_halo_two_process_worker, args=(r, 2, port, "REBLUR_DIFFUSE_SPECULAR", 192, 601, 5, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = sorted(q.get(timeout=400) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    assert results == [(0, True, True), (1, True, True)]


def _halo_two_process_emulated_worker(rank, world, port, name, W, H, frames, overrides, measure, q):
    import numpy as np
    import torch.distributed as dist

    import parity
    from emu import emu_run

    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lib = emu_run.load()
    overrides = dict(overrides or {})
    rise, halo, near = overrides.pop("_camera_rise", 0.0), overrides.pop("_history_halo", 16), overrides.pop("_near_depth", 1.0)
    balance = overrides.pop("_balance", True)  # False: uniform strips for good -- then the restart frame is sharded too (round 6)
    parity.synth.CAMERA_RISE = rise  # (a fresh process: nothing to restore)
    seq = parity.generate_sequence(name, W, H, frames, device="cpu")
    steps = [(frame, {}) for frame in seq]
    if measure:
        # one more frame whose motion vectors (2D convention) leave the 16-row history halo at ONE pixel of the LAST rank's strip: only that rank can see it on
        # its own rows, every rank has to run the frame unsharded (HaloSharder(measure_motion=True): device reduction per strip + MAX over ranks)
        fast = dict(seq[-1])
        mv = torch.zeros_like(fast["mv"])
        z = fast["viewz"].clone()
        z.view(H, W)[H - 7, 11] = 5.0
        mv[H - 7, 11, 1] = -40.0
        fast["mv"], fast["viewz"] = mv, z
        steps.append((fast, dict(isMotionVectorInWorldSpace=False, motionVectorScale=(1.0 / W, 1.0 / H, 0.0))))
        steps.append((seq[-1], {}))  # the frame after: still unsharded -- the temporal kernels REPORTED a 40-row history reach for the fast frame, and a frame is decided with
        steps.append((seq[-1], {}))  # 1.25 x the reach of the one before it (round 6: HaloSharder.history_reach_rows) -- and then back to a sharded frame

    sh_stats = {"violations": 0, "reach_after_fast": 0.0}

    def run(sharded):
        inst = api.Instance([(0, parity.DENOISERS[name][0])], lib=lib)
        ex = emu_run.EmuTorchExecutor(inst, W, H)
        outs = []
        for rt, dtype, ch, fmt in parity.output_planes(name, W, H):
            outs.append(torch.zeros((H, W, ch), dtype=dtype))
            ex.bind(rt, outs[-1], fmt)
        sh = sharding.HaloSharder(ex, inst, W, H, rank, world, max_motion_rows=halo, measure_motion=measure, near_depth=near, balance=balance) if sharded else None
        per_frame, sharded_frames, measured = [], 0, []
        for f, (frame, cs_kw) in enumerate(steps):
            for rt, t, fmt in parity.user_planes(name, frame):
                ex.bind(rt, t.cpu().contiguous(), fmt)
            inst.set_denoiser_settings(0, parity.denoiser_settings(name, frame, overrides))
            inst.set_common_settings(parity.common_settings(frame["camera"], steps[max(f - 1, 0)][0]["camera"], W, H, f, **cs_kw))
            if sharded:
                sharded_frames += 0 if sh.denoise().fallback else 1
                measured.append(sh.measured_motion_rows)
                if measure and f == frames + 1:
                    sh_stats["reach_after_fast"] = sh.history_reach_rows
                sh.wait_outputs()
                per_frame.append(([sh.complete_output(rt).clone() for rt, dtype, ch, fmt in parity.output_planes(name, W, H)], sh.rows))  # the reassembled planes
            else:
                ex.denoise()
                per_frame.append(([o.clone() for o in outs], (0, H)))
        if sharded:
            sh_stats["violations"] = sh.history_halo_violations
        return per_frame, sharded_frames, (sh.exchanged_bytes if sharded else 0), (sh.motion_fallbacks if sharded else 0), measured

    ref = run(False)[0]
    got, sharded_frames, exchanged, motion_fallbacks, measured = run(True)
    # round 5: HaloSharder.denoise() ends with the output all-gather (synchronous over gloo), so EVERY rank holds the COMPLETE output planes of every frame -- not only its rows
    ok = all(torch.equal(a, b) for (fa, _), (fb, rows) in zip(ref, got) for a, b in zip(fa, fb))
    if os.environ.get("NRD_TEST_DEBUG"):
        print("rank", rank, [[bool(torch.equal(a, b)) for a, b in zip(fa, fb)] for (fa, _), (fb, rows) in zip(ref, got)], [r for _, r in got], motion_fallbacks, measured, file=sys.stderr, flush=True)
        for f, ((fa, _), (fb, rows)) in enumerate(zip(ref, got)):
            for a, b in zip(fa, fb):
                if not torch.equal(a, b):
                    bad = (a != b).reshape(a.shape[0], -1).any(dim=1).nonzero().flatten().tolist()
                    print("rank", rank, "frame", f, "rows differing", bad[:5], "...", bad[-5:], len(bad), file=sys.stderr, flush=True)
    if measure:
        # every rank saw the same (reduced) value on every frame; the fast frame measured its 40 rows and was the only motion fallback
        ok = ok and motion_fallbacks == 2 and abs(measured[frames] - 40.0) < 0.05 and all(m is not None and m < 7.0 for i, m in enumerate(measured) if i != frames)
        ok = ok and sh_stats["violations"] == 0 and sh_stats["reach_after_fast"] >= 39.0  # no SHARDED frame read beyond its halo; the fast frame's reach was seen one frame later
    q.put((rank, ok, sharded_frames, exchanged > 0))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("name,world,W,H,overrides,measure", [
    ("REBLUR_DIFFUSE_SPECULAR", 2, 96, 240, dict(maxBlurRadius=4.0, diffusePrepassBlurRadius=6.0, specularPrepassBlurRadius=6.0), False),  # small radii: the halos fit 120-row strips
    ("REBLUR_DIFFUSE_SPECULAR", 3, 64, 300, dict(maxBlurRadius=4.0, diffusePrepassBlurRadius=6.0, specularPrepassBlurRadius=6.0), True),  # a middle strip with two neighbours
    ("RELAX_DIFFUSE_SPECULAR", 2, 96, 240, dict(atrousIterationNum=3, diffusePrepassBlurRadius=6.0, specularPrepassBlurRadius=6.0), True),
    # no pre-pass (its guide reach used to cover the tap by accident) and a camera that RISES ~10 rows of parallax per frame: the curvature estimate's
    # high-parallax tap of TemporalAccumulation reads the decoded normals many rows away along the motion direction -- the guide planes must be decoded there (poisoned otherwise)
    ("REBLUR_DIFFUSE_SPECULAR", 2, 64, 480, dict(maxBlurRadius=4.0, diffusePrepassBlurRadius=0.0, specularPrepassBlurRadius=0.0, _camera_rise=0.25, _history_halo=110, _near_depth=2.5), False),
    ("RELAX_DIFFUSE_SPECULAR", 2, 64, 480, dict(atrousIterationNum=3, diffusePrepassBlurRadius=0.0, specularPrepassBlurRadius=0.0, _camera_rise=0.25, _history_halo=110, _near_depth=2.5), False),
    # round 6: the passes that used to run unsharded on every rank
    ("SIGMA_SHADOW", 2, 96, 240, None, False),                                                   # BASELINE config 2's denoiser: Blur / PostBlur 35 rows, the history Copy + motion
    ("SIGMA_SHADOW_TRANSLUCENCY", 3, 64, 420, dict(_balance=False), False),                      # uniform strips: the restart frame (its clears) is sharded as well
    ("REBLUR_DIFFUSE_SPECULAR", 2, 96, 240, dict(maxBlurRadius=4.0, diffusePrepassBlurRadius=6.0, specularPrepassBlurRadius=6.0, hitDistanceReconstructionMode=2), False),  # 5x5 reconstruction
    ("RELAX_DIFFUSE_SPECULAR", 2, 96, 240, dict(atrousIterationNum=3, diffusePrepassBlurRadius=6.0, specularPrepassBlurRadius=6.0, hitDistanceReconstructionMode=1, enableAntiFirefly=True, _balance=False), False),
])
def test_halo_sharding_processes_over_gloo_on_emulated_kernels(name, world, W, H, overrides, measure):
    """The N > 1 path end to end WITHOUT a GPU: `world` processes over gloo, each planning its strip from the dispatch list, exchanging halo bands by message passing
    and running its pass segments -- on the CPU emulation of the device sources (tests/emu: the .hip files compiled for x86, test infrastructure). Every rank's
    owned rows of every output, every frame, must equal a full-frame run of the same emulated kernels bit for bit; the frames after the restart frame must really
    be sharded (an unsharded fallback would pass trivially). measure: the motion side of the contract comes from the device (nrdHipMeasureMotionRows on the
    rank's own rows + an all-reduce), and a frame whose motion leaves the halo on one rank's rows only is run unsharded by all of them."""
    import torch.multiprocessing as mp

    from emu import emu_run

    emu_run.load()  # build the emulation library once, before the workers race for it
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 100) + world
    frames = 4
    # the workers (fresh processes) decode their guide planes only on the rows their passes declared (executor.hip GuideReachRows) and, with this hook, write NaNs
    # everywhere else: a pass that reads a guide row outside its declared reach cannot stay bit-identical
    os.environ["NRD_HIP_POISON_GUIDES"] = "1"
    try:
        procs = [ctx.Process(target=_halo_two_process_emulated_worker, args=(r, world, port, name, W, H, frames, overrides, measure, q)) for r in range(world)]
        for p in procs:
            p.start()
        results = sorted(q.get(timeout=900) for _ in procs)
        for p in procs:
            p.join(timeout=60)
    finally:
        del os.environ["NRD_HIP_POISON_GUIDES"]
    # sharded frames: all but the restart frame -- which is run whole where the strips are balanced (they are cut from its tile map) and sharded like any other frame since
    # round 6 where they are not (+ with measure: the frame after the fast one; the fast one itself falls back)
    restart_sharded = (overrides or {}).get("_balance", True) is False
    assert results == [(r, True, frames - 1 + (1 if restart_sharded else 0) + (1 if measure else 0), True) for r in range(world)], results


@pytest.mark.gpu
def test_bench_two_ranks_rehearsal():
    """bench.py exactly as the driver launches it for N = 2 (torch.distributed.run, one rank per process, barrier + max over ranks, rank 0 prints
    the JSON line) -- with gloo and both ranks on the one GPU of the box, because RCCL needs one GPU per rank."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, NRD_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    port = 29500 + (os.getpid() % 100)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "3", "--width", "512", "--height", "600"]
    out = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["steps"] == 6 and r["warmup"] == 3 and r["value"] > 0 and r["scaling"] == "strong"
    assert "halo exchange" in r["config"]["parallelism"] and r["roofline"]["kernel"].startswith("REBLUR_")


def test_camera_motion_estimate_and_fallback_decision():
    """nothing checked that a frame's reprojection stays inside the history halo. camera_motion_rows bounds the vertical motion of static
    geometry from the CommonSettings matrices; HaloSharder.motion_exceeds_halo turns it into the per-frame decision (no GPU needed)."""
    import math

    import parity
    from raytracingdenoiser_amd import sharding, synth

    w, h = 2560, 1440
    a, b = synth.Camera(w, h, 10), synth.Camera(w, h, 11)
    cs = parity.common_settings(b, a, w, h, 11)
    slow = sharding.camera_motion_rows(cs)
    assert 0.0 <= slow < 12.0, slow  # the bench sequence: 0.1 degree of yaw and a centimetre of dolly per frame, nearest geometry at >= 1 unit

    # a camera pitching by 1.5 degrees per frame: ~37 rows at 1440p and 60 degrees of vertical field of view (the advisor's example)
    def pitched(cam, deg):
        c = synth.Camera(w, h, 10)
        p = math.radians(deg)
        m = [c.world_to_view[i] for i in range(16)]
        R = [[1, 0, 0], [0, math.cos(p), -math.sin(p)], [0, math.sin(p), math.cos(p)]]
        cols = [[m[4 * k + r] for r in range(4)] for k in range(4)]
        out = [[sum(R[r][t] * cols[k][t] for t in range(3)) for r in range(3)] + [cols[k][3]] for k in range(4)]
        c.world_to_view = [v for col in out for v in col]
        return c

    fast = sharding.camera_motion_rows(parity.common_settings(pitched(a, 1.5), a, w, h, 11))
    assert 30.0 < fast < 45.0, fast

    class FakeInstance:
        pass

    inst = FakeInstance()
    sh = sharding.HaloSharder.__new__(sharding.HaloSharder)
    sh.inst, sh.max_motion_rows = inst, 32
    inst.last_common_settings = cs
    assert not sh.motion_exceeds_halo()
    assert sh.motion_exceeds_halo(motion_rows=29.0)  # the application's own bound for moving objects counts on top
    inst.last_common_settings = parity.common_settings(pitched(a, 1.5), a, w, h, 11)
    assert sh.motion_exceeds_halo()


def test_dry_plan_runs_for_every_bench_workload():
    """bench.py --gpus N --dry-plan: host only (no GPU, no process group) for all workloads, including the SIGMA one whose settings need the scene's light direction"""
    import bench
    from raytracingdenoiser_amd import sharding

    for workload, (name, size, _, overrides) in bench.WORKLOADS.items():
        plan = sharding.dry_plan(name, size[0], size[1], 4, overrides)
        assert plan["denoiser"] == name and plan["ranks"] == 4 and len(plan["per_rank"]) == 4, workload
        # (round 6: SIGMA is sharded like the others -- until round 5 its passes declared no reach and the chain ran as replicas)
        assert not plan["per_rank"][1]["fallback_unsharded"] and plan["per_rank"][1]["received_bytes_per_frame"] > 0, workload
        if name.startswith("SIGMA"):
            assert plan["reach_rows"] == [0, 0, 0, 35, 35, 2]
