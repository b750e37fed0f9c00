"""Row-strip sharding. CPU: the strip all-gather with gloo, world_size 2. GPU: N virtual ranks on one
MI355X (one executor per rank, all-gather emulated by copies) must reproduce the single-GPU planes bit for bit."""
import os

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
        ref_planes = [ref_ex.pool_plane_tensor(RT.PERMANENT_POOL, i) for i in range(len(ref_inst.permanent_pool))] + [o.view(-1).view(dtype=torch.uint8).view(H, -1) for o in ref_outs]
        for r, (_, _, _, s) in enumerate(ranks):
            for k, (a, b) in enumerate(zip(s.planes, ref_planes)):
                assert torch.equal(a, b), "frame %d rank %d plane %d differs from the single-GPU run" % (f, r, k)
