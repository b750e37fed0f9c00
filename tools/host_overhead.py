"""Host-side cost of one sharded frame (planning + launches, no transfers): a virtual rank of an 8-way split runs the bench sequence without any
synchronisation inside the loop; reported are the host time per frame spent in HaloSharder (begin_frame + segment launches) and the GPU time per
frame -- the host must stay well below the GPU for the launches to run ahead. usage: python tools/host_overhead.py [--world 8]"""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import bench
    import parity
    from raytracingdenoiser_amd import api, sharding
    from raytracingdenoiser_amd.executor import HipExecutor

    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="reblur_ds")
    ap.add_argument("--world", type=int, default=8)
    ap.add_argument("--frames", type=int, default=64)
    args = ap.parse_args()
    name, (W, H), _, _ = bench.WORKLOADS[args.workload]
    seq = parity.generate_sequence(name, W, H, 8, device="cuda")
    inst = api.Instance([(0, parity.DENOISERS[name][0])])
    ex = HipExecutor(inst, W, H)
    for rt, dtype, ch, fmt in parity.output_planes(name, W, H):
        ex.bind(rt, torch.zeros((H, W, ch), dtype=dtype, device="cuda"), fmt)
    inst.set_denoiser_settings(0, parity.denoiser_settings(name, seq[0]))
    sh = sharding.HaloSharder(ex, inst, W, H, args.world // 2, args.world, balance=False)
    cs = [parity.common_settings(seq[f % 8]["camera"], seq[max(f - 1, 0) % 8]["camera"], W, H, f) for f in range(args.frames + 8)]

    def frame(f):
        for rt, t, fmt in parity.user_planes(name, seq[f % 8]):
            ex.bind(rt, t, fmt)
        inst.set_common_settings(cs[f])
        t0 = time.perf_counter()
        plan, ptr, n = sh.begin_frame()
        if plan.fallback:
            ex.execute_range(ptr, n, 0, n)
        else:
            for step in range(len(plan.steps)):
                sh.run_step(plan, ptr, n, step)
        sh.finish_frame(plan)
        return time.perf_counter() - t0

    for f in range(8):
        frame(f)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    host = sum(frame(8 + f) for f in range(args.frames))
    issue = time.perf_counter() - t0
    torch.cuda.synchronize()
    total = time.perf_counter() - t0
    print("%s, rank %d of %d: host in HaloSharder %.3f ms/frame, host loop %.3f ms/frame (incl. binds + SetCommonSettings), GPU %.3f ms/frame" %
          (name, args.world // 2, args.world, 1e3 * host / args.frames, 1e3 * issue / args.frames, 1e3 * total / args.frames))


if __name__ == "__main__":
    main()
