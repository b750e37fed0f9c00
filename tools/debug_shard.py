"""(development) one virtual rank of the halo scheme in lock-step with a full-frame run: after every pass segment, the first plane whose rows this rank needs differ from the full-frame
run's -- which pass reads further than it declared?   usage: python tools/debug_shard.py WORKLOAD WORLD RANK [FRAMES] [BALANCE]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import bench
    import parity
    from raytracingdenoiser_amd import api, sharding
    from raytracingdenoiser_amd.executor import HipExecutor

    workload, world, rank = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    frames = int(sys.argv[4]) if len(sys.argv) > 4 else 24
    balance = bool(int(sys.argv[5])) if len(sys.argv) > 5 else True
    motion_rows = int(sys.argv[6]) if len(sys.argv) > 6 else None  # the history halo (rows); default: HaloSharder.default_motion_rows
    name, (W, H), _, overrides = bench.WORKLOADS[workload]
    seq = parity.generate_sequence(name, W, H, frames, device="cuda")  # (as tools/model_scaling.py)

    def make():
        inst = api.Instance([(0, parity.DENOISERS[name][0])])
        ex = HipExecutor(inst, W, H)
        outs = []
        for rt, dtype, ch, fmt in parity.output_planes(name, W, H):
            outs.append(torch.zeros((H, W, ch), dtype=dtype, device="cuda"))
            ex.bind(rt, outs[-1], fmt)
        return inst, ex, outs

    ref, run = make(), make()
    sh = sharding.HaloSharder(run[1], run[0], W, H, rank, world, balance=balance, max_motion_rows=motion_rows)
    reach_word = torch.zeros(1, dtype=torch.float32, device="cuda")  # the full-frame run's temporal kernels report the history reach of ALL rows (nrdHipSetHistoryReachWord)
    ref[1].set_history_reach_word(reach_word)
    RT = api.ResourceType

    def plane(ex_tuple, key):
        inst, ex, outs = ex_tuple
        if key[0] in (int(RT.PERMANENT_POOL), int(RT.TRANSIENT_POOL)):
            return ex.pool_plane_tensor(RT(key[0]), key[1])
        return ex._bound[key[0]].view(-1).view(dtype=torch.uint8).view(H, -1)

    for f in range(frames):
        frame = seq[f]
        cs = parity.common_settings(frame["camera"], seq[max(f - 1, 0)]["camera"], W, H, f)
        for inst, ex, _ in (ref, run):
            for rt, t, fmt in parity.user_planes(name, frame):
                ex.bind(rt, t.cuda() if not t.is_cuda else t, fmt)
            inst.set_denoiser_settings(0, parity.denoiser_settings(name, frame, overrides))
            assert inst.set_common_settings(cs) == api.Result.SUCCESS
        reach_prev = float(reach_word.item())
        reach_word.zero_()
        plan, ptr, n = sh.begin_frame(history_reach=reach_prev)
        measured = run[1].measure_motion_rows(ptr, n, 0, H)
        estimate = sharding.camera_motion_rows(run[0].last_common_settings, (sh.near_depth, 1.0e4)) if getattr(run[0], "last_common_settings", None) is not None else None
        print("frame", f, "surface motion measured over the frame: %.2f rows; camera estimate %s; history reach reported by the last frame's temporal kernels %.1f rows; history halo %d rows" % (measured, estimate, reach_prev, sh.max_motion_rows))
        r2, rptr, rn = ref[0].get_compute_dispatches_raw()
        ds = [api.Dispatch(ptr[i], run[0].pipelines) for i in range(n)]
        if plan.fallback:
            for key in plan.complete_keys:
                plane(run, key).copy_(plane(ref, key))
            run[1].execute_range(ptr, n, 0, n)
            ref[1].execute_range(rptr, rn, 0, rn)
            sh.finish_frame(plan)
            print("frame", f, "unsharded", "bounds", sh.bounds)
            continue
        rb, re = sh.rows
        for step, (items, first, count) in enumerate(plan.steps):
            for key, w in items:
                lo, hi = max(rb - w, 0), min(re + w, H)
                plane(run, key)[lo:rb].copy_(plane(ref, key)[lo:rb])
                plane(run, key)[re:hi].copy_(plane(ref, key)[re:hi])
            rbs, res = plan.c_rows()
            for i in range(first, first + count):
                whole = plan.row_begin[i] < 0
                m = plan.margins[i]
                lo, hi = (0, H) if whole else (max(rb - m, 0), min(re + m, H))
                # (a) the inputs of pass i, right before it runs, on the rows its declared reach covers (planes carried over from last frame: + nothing here -- their motion-dependent
                #     reads are what the history reach reports; a stale row inside [lo - reach, hi + reach) is a planning error)
                r = max(plan.reach[i], 0)
                for dt, t, idx in ds[i].resources:
                    key = (int(t), idx)
                    if dt != api.DescriptorType.TEXTURE or int(t) < int(RT.OUT_DIFF_RADIANCE_HITDIST):
                        continue
                    a, b = plane(run, key), plane(ref, key)
                    if a.shape[0] != H:
                        continue
                    ilo, ihi = max(lo - r, 0), min(hi + r, H)
                    if not torch.equal(a[ilo:ihi], b[ilo:ihi]):
                        bad = (a[ilo:ihi] != b[ilo:ihi]).any(dim=1).nonzero().flatten() + ilo
                        print("frame %d pass %d %s: INPUT plane %s is stale in rows %d..%d (%d rows) of [%d, %d) = produced rows [%d, %d) +- reach %d" % (
                            f, i, ds[i].shader, key, int(bad.min()), int(bad.max()), len(bad), ilo, ihi, lo, hi, r))
                # (b) the pass, on both
                before = {(int(t), idx): plane(run, (int(t), idx)).clone() for dt, t, idx in ds[i].resources if dt == api.DescriptorType.STORAGE_TEXTURE and plane(run, (int(t), idx)).shape[0] == H}
                before_ref = {k: plane(ref, k).clone() for k in before}
                run[1].execute_range(ptr, n, i, 1, rbs, res)
                ref[1].execute_range(rptr, rn, i, 1)
                torch.cuda.synchronize()
                # (c) its outputs on the rows it had to produce
                for dt, t, idx in ds[i].resources:
                    if dt != api.DescriptorType.STORAGE_TEXTURE:
                        continue
                    key = (int(t), idx)
                    a, b = plane(run, key), plane(ref, key)
                    if whole:  # tile maps: complete on every rank
                        if not torch.equal(a, b):
                            bad = (a != b).nonzero()
                            print("frame %d pass %d %s (whole frame): OUTPUT plane %s differs at %d places, first (row, byte) %s: run %d ref %d" % (f, i, ds[i].shader, key, len(bad), bad[0].tolist(), int(a[tuple(bad[0].tolist())]), int(b[tuple(bad[0].tolist())])))
                            return 1
                        continue
                    if a.shape[0] != H:
                        continue
                    # only what the pass WROTE on either side counts: a texel neither wrote (sky) keeps whatever an earlier frame left there, which differs between a rank and a
                    # full-frame run outside the rank's strip and is read by nobody
                    wrote = (a[lo:hi] != before[key][lo:hi]) | (b[lo:hi] != before_ref[key][lo:hi])
                    if bool(((a[lo:hi] != b[lo:hi]) & wrote).any()):
                        dmask = (a[lo:hi] != b[lo:hi]) & wrote
                        bad = dmask.any(dim=1).nonzero().flatten() + lo
                        cols = dmask.any(dim=0).nonzero().flatten()
                        print("frame %d step %d pass %d %s (reach %d margin %d): OUTPUT plane %s differs in rows %d..%d (%d rows), byte columns %d..%d, of the produced [%d, %d); strip [%d, %d)" % (
                            f, step, i, ds[i].shader, plan.reach[i], m, key, int(bad.min()), int(bad.max()), len(bad), int(cols.min()), int(cols.max()), lo, hi, rb, re))
                        d = dmask
                        unwritten = int(((a[lo:hi] == before[key][lo:hi]) & d).sum()), int(d.sum())
                        yy, xx = d.nonzero()[0].tolist()
                        print("   %d of the %d differing bytes still hold what the plane held BEFORE the pass (= never written by this rank); first at row %d byte %d: run %d ref %d before %d" % (
                            unwritten[0], unwritten[1], yy + lo, xx, int(a[yy + lo, xx]), int(b[yy + lo, xx]), int(before[key][yy + lo, xx])))
                        rowsum = d.any(dim=1).nonzero().flatten() + lo
                        print("   differing rows:", rowsum.tolist()[:40])
                        return 1
        sh.finish_frame(plan)
        print("frame", f, "ok", "strip", sh.rows)
    return 0


if __name__ == "__main__":
    sys.exit(main())
