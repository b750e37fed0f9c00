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
    sh = sharding.HaloSharder(run[1], run[0], W, H, rank, world, balance=balance)
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
            sh.run_step(plan, ptr, n, step)
            ref[1].execute_range(rptr, rn, first, count)
            torch.cuda.synchronize()
            last_writer = {}
            for i in range(first, first + count):
                for dt, t, idx in ds[i].resources:
                    if dt == api.DescriptorType.STORAGE_TEXTURE:
                        last_writer[(int(t), idx)] = i
            for i in range(first, first + count):
                if plan.row_begin[i] < 0:
                    continue
                m = plan.margins[i]
                lo, hi = max(rb - m, 0), min(re + m, H)
                for dt, t, idx in ds[i].resources:
                    if dt != api.DescriptorType.STORAGE_TEXTURE:
                        continue
                    key = (int(t), idx)
                    if last_writer[key] != i:
                        continue  # overwritten later in this segment (ping-pong / scratch use of the OUT planes): only its last version can be compared after the segment
                    a, b = plane(run, key), plane(ref, key)
                    if a.shape[0] != H:
                        continue
                    if not torch.equal(a[lo:hi], b[lo:hi]):
                        bad = (a[lo:hi] != b[lo:hi]).any(dim=1).nonzero().flatten() + lo
                        print("frame %d step %d pass %d %s (reach %d margin %d): plane %s differs in rows %d..%d (%d rows) of the needed [%d, %d); strip [%d, %d)" % (
                            f, step, i, ds[i].shader, plan.reach[i], m, key, int(bad.min()), int(bad.max()), len(bad), lo, hi, rb, re))
                        return 1
        sh.finish_frame(plan)
        print("frame", f, "ok", "strip", sh.rows)
    return 0


if __name__ == "__main__":
    sys.exit(main())
