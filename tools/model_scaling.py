"""Per-rank compute time of the row-strip sharding, measured on ONE GPU (the box has no second GPU to run the real thing).

For N in (1, 2, 4, 8) every rank in turn (or only the middle strip, --all-ranks 0) runs the bench sequence as one virtual rank; a
full-frame executor runs next to it in lock-step and the rows the other ranks would deliver (halo bands before each pass segment, or
the all-gathered planes after the frame) are copied out of its planes, untimed. Reported per world size: the slowest rank's ms/frame
(cuda events around its launches) = the compute bound of the frame time, the redundant-compute factor against the ideal 1/N, the
strips (the halo scheme re-cuts them from the tile map, --balance 0 keeps them uniform) and the bytes received per frame. The
transfers themselves are NOT measured here.
usage: python tools/model_scaling.py [--workload reblur_ds] [--frames 24] [--warmup 16] > gpurun_out/scaling_model.json"""
import argparse
import json
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

    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="reblur_ds")
    ap.add_argument("--frames", type=int, default=24)
    ap.add_argument("--warmup", type=int, default=16)
    ap.add_argument("--worlds", default="1,2,4,8")
    ap.add_argument("--balance", type=int, default=1, help="halo scheme: strips re-cut from the tile map (1) or uniform (0)")
    ap.add_argument("--all-ranks", type=int, default=1, help="measure every rank of each world size (the frame time is the slowest one) instead of the middle strip only")
    ap.add_argument("--scheme", choices=["allgather", "halo"], default="halo", help="allgather = FrameSharder (redundant halo compute), halo = HaloSharder (halo exchange between segments)")
    ap.add_argument("--max-motion-rows", type=int, default=None, help="history-halo width in rows (default: HaloSharder.default_motion_rows)")
    ap.add_argument("--no-sky", action="store_true", help="the bench scene with a backdrop dome: every pixel is denoised (bench.py --no-sky)")
    ap.add_argument("--link-GBps", type=float, default=50.0, help="what one xGMI link delivers to one neighbour (modelled transfers)")
    ap.add_argument("--denoiser", default=None, help="any denoiser of the library instead of a bench workload (verification sweeps), with --size and --settings")
    ap.add_argument("--size", default="1280x720")
    ap.add_argument("--settings", default=None, help="JSON object of denoiser-settings overrides, e.g. '{\"hitDistanceReconstructionMode\": 1}'")
    args = ap.parse_args()
    if args.no_sky:
        from raytracingdenoiser_amd import synth

        synth.BACKDROP = True
    name, (W, H), _, overrides = bench.WORKLOADS[args.workload]
    if args.denoiser:
        name, (W, H), overrides = args.denoiser, tuple(int(v) for v in args.size.split("x")), (json.loads(args.settings) if args.settings else None)
    total = args.warmup + args.frames
    seq = parity.generate_sequence(name, W, H, total, device="cuda")

    def make():
        inst = api.Instance([(0, parity.DENOISERS[name][0])])
        ex = HipExecutor(inst, W, H)
        outs = []
        for rt, dtype, ch, fmt in parity.output_planes(name, W, H):
            outs.append(torch.zeros((H, W, ch), dtype=dtype, device="cuda"))
            ex.bind(rt, outs[-1], fmt)
        assert inst.set_denoiser_settings(0, parity.denoiser_settings(name, seq[0], overrides)) == api.Result.SUCCESS
        return inst, ex, outs

    def planes_of(inst, ex, outs):
        # what FrameSharder reassembles every frame: full-resolution permanent planes, the outputs and -- round 6 -- the full-resolution transient planes
        ps = [ex.pool_plane_tensor(api.ResourceType.PERMANENT_POOL, i) for i, (fmt, ds) in enumerate(inst.permanent_pool) if ds == 1]
        ps += [o.view(-1).view(dtype=torch.uint8).view(H, -1) for o in outs]
        return ps + [ex.pool_plane_tensor(api.ResourceType.TRANSIENT_POOL, i) for i, (fmt, ds) in enumerate(inst.transient_pool) if ds == 1]

    out_row_bytes = sum(W * ch * torch.empty(0, dtype=dtype).element_size() for rt, dtype, ch, fmt in parity.output_planes(name, W, H))  # bytes of one row of all OUT_* planes
    results = []
    for world, rank in [(int(w), r) for w in args.worlds.split(",") for r in (range(int(w)) if args.all_ranks else [int(w) // 2])]:
        ref = make()
        ref_planes = planes_of(*ref)
        run = make()
        halo = args.scheme == "halo" and world > 1
        shard = None
        if world > 1:
            shard = sharding.HaloSharder(run[1], run[0], W, H, rank, world, max_motion_rows=args.max_motion_rows, balance=bool(args.balance)) if halo else sharding.FrameSharder(run[1], run[0], W, H, rank, world, run[2])
        run_planes = planes_of(*run)
        reach_word = torch.zeros(1, dtype=torch.float32, device="cuda")
        ref[1].set_history_reach_word(reach_word)
        ms = []
        exchanged = []
        exact = True
        unsharded_frames = 0
        for f in range(total):
            frame = seq[f]
            cs = parity.common_settings(frame["camera"], seq[max(f - 1, 0)]["camera"], W, H, f)
            for inst, ex, _ in (ref, run):
                for rt, t, fmt in parity.user_planes(name, frame):
                    ex.bind(rt, t, fmt)
                assert inst.set_common_settings(cs) == api.Result.SUCCESS
            if halo:
                # lock-step with the full-frame executor: it runs the same segments, and the bands the two neighbours would send are
                # copied out of its planes before each segment (the copies stand in for the RCCL transfers and are not timed)
                # what a real group's per-frame all-reduce carries (HaloSharder._measure_motion_over_ranks): the history reach the temporal kernels reported LAST frame, MAX over
                # the ranks -- here read from the full-frame executor, whose kernels see every rank's pixels
                reach_prev = float(reach_word.item())
                reach_word.zero_()
                plan, ptr, n = shard.begin_frame(history_reach=reach_prev)
                r2, rptr, rn = ref[0].get_compute_dispatches_raw()
                assert rn == n
                t, got = 0.0, 0
                steps = [(None, 0, n)] if plan.fallback else plan.steps
                if plan.fallback:
                    unsharded_frames += 1 if f >= args.warmup else 0
                    for key in plan.complete_keys:  # what HaloSharder.complete_planes broadcasts before an unsharded frame that follows sharded ones (untimed, like the halo copies)
                        src = ref[1].pool_plane_tensor(api.ResourceType(key[0]), key[1]) if key[0] in (int(api.ResourceType.PERMANENT_POOL), int(api.ResourceType.TRANSIENT_POOL)) else \
                            ref[1]._bound[key[0]].view(-1).view(dtype=torch.uint8).view(H, -1)
                        shard.plane_tensor(key).copy_(src)
                for step, (items, first, count) in enumerate(steps):
                    if not plan.fallback:
                        rb, re = shard.rows
                        for key, w in items:
                            dst = shard.plane_tensor(key)
                            src = ref[1].pool_plane_tensor(api.ResourceType(key[0]), key[1]) if key[0] in (int(api.ResourceType.PERMANENT_POOL), int(api.ResourceType.TRANSIENT_POOL)) else \
                                ref[1]._bound[key[0]].view(-1).view(dtype=torch.uint8).view(H, -1)
                            lo, hi = max(rb - w, 0), min(re + w, H)
                            dst[lo:rb].copy_(src[lo:rb])
                            dst[re:hi].copy_(src[re:hi])
                            got += dst.shape[1] * ((rb - lo) + (hi - re))
                    torch.cuda.synchronize()
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    if plan.fallback:
                        run[1].execute_range(ptr, n, 0, n)
                    else:
                        shard.run_step(plan, ptr, n, step)
                    e1.record()
                    ref[1].execute_range(rptr, rn, first, count)
                    torch.cuda.synchronize()
                    t += e0.elapsed_time(e1)
                shard.finish_frame(plan)
                rb, re = shard.rows
                exact = exact and all(torch.equal(a[rb:re], b[rb:re]) for a, b in zip(run[2], ref[2]))  # owned rows of every output == the full-frame run
                if f >= args.warmup:
                    ms.append(t)
                    exchanged.append(got)
                continue
            ref[1].denoise()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            run[1].denoise()
            e1.record()
            torch.cuda.synchronize()
            if f >= args.warmup:
                ms.append(e0.elapsed_time(e1))
            if shard is not None and not halo and shard.rows is not None:  # what the all-gather delivers: every row this rank does not own
                rb, re = shard.rows
                assert len(run_planes) == len(shard.planes)
                exact = exact and all(torch.equal(a[rb:re], b[rb:re]) for a, b in zip(run[2], ref[2]))  # owned rows of every output == the full-frame run, before the reassembly
                for src, dst in zip(ref_planes, run_planes):
                    dst[:rb].copy_(src[:rb])
                    dst[re:].copy_(src[re:])
        torch.cuda.synchronize()
        gather_bytes = sum(p.shape[0] * p.shape[1] for p in run_planes) if world > 1 and not halo else 0
        results.append({"world": world, "rank": rank, "rows": list(shard.rows) if shard and shard.rows else [0, H], "ms_per_frame": round(sum(ms) / len(ms), 4),
                        "all_gather_bytes_per_frame": gather_bytes, "halo_bytes_received_per_frame": int(sum(exchanged) / len(exchanged)) if exchanged else 0,
                        "owned_rows_bit_identical_to_full_frame_run": exact if halo else None, "timed_frames_run_unsharded": unsharded_frames})
        for inst, ex, _ in (ref, run):
            ex.destroy()
    base = results[0]["ms_per_frame"]
    summary = []
    for world in sorted({r["world"] for r in results}):
        rs = [r for r in results if r["world"] == world]
        slowest = max(r["ms_per_frame"] for r in rs)
        summary.append({"world": world, "slowest_rank_ms": slowest, "mean_rank_ms": round(sum(r["ms_per_frame"] for r in rs) / len(rs), 4), "compute_speedup_bound": round(base / slowest, 3),
                        "redundant_compute_factor": round(sum(r["ms_per_frame"] for r in rs) / base, 3),
                        "max_halo_bytes_received_per_frame": max(r["halo_bytes_received_per_frame"] for r in rs), "strips": [r["rows"] for r in rs],
                        "owned_rows_bit_identical_to_full_frame_run": all(r["owned_rows_bit_identical_to_full_frame_run"] is not False for r in rs)})
    # MODELLED transfers (nothing here ran on more than one GPU): an inner rank receives its bands from two neighbours over two different xGMI links, so the
    # slower direction carries about half of what the rank receives; three bounds per world size -- no overlap at all (compute + transfer), perfect overlap
    # (max of the two), and the speed-up against the measured one-GPU frame each would give
    for srow in summary:
        per_link = srow["max_halo_bytes_received_per_frame"] / (2.0 if srow["world"] > 2 else 1.0)
        xfer_ms = per_link / (args.link_GBps * 1e9) * 1e3
        srow["modelled_transfer_ms_per_frame_at_%g_GBps_per_link" % args.link_GBps] = round(xfer_ms, 4)
        srow["modelled_frame_ms_no_overlap"] = round(srow["slowest_rank_ms"] + xfer_ms, 4)
        srow["modelled_frame_ms_full_overlap"] = round(max(srow["slowest_rank_ms"], xfer_ms), 4)
        srow["modelled_speedup_no_overlap"] = round(base / (srow["slowest_rank_ms"] + xfer_ms), 3)
        srow["modelled_speedup_full_overlap"] = round(base / max(srow["slowest_rank_ms"], xfer_ms), 3)
        # round 5: the output all-gather (HaloSharder.start_output_gather): every rank sends its rows of the OUT_* planes to each of the other ranks over that rank's own link of
        # the xGMI mesh, next to the WHOLE next frame (the complete planes are separate tensors) -- a bound on the frame RATE, max(compute, gather), not an addend
        if srow["world"] > 1 and args.scheme == "halo":
            tallest = max(r[1] - r[0] for r in srow["strips"])
            gather_ms = tallest * out_row_bytes / (args.link_GBps * 1e9) * 1e3
            srow["output_all_gather_bytes_per_link"] = tallest * out_row_bytes
            srow["modelled_output_gather_ms_per_link"] = round(gather_ms, 4)
            srow["modelled_frame_ms_full_overlap_with_gather"] = round(max(srow["slowest_rank_ms"], xfer_ms, gather_ms), 4)
    print(json.dumps({"workload": "%s %dx%d%s" % (name, W, H, " (no sky)" if args.no_sky else ""), "scheme": args.scheme, "balance": bool(args.balance),
                      "note": "MODELLED, not measured on several GPUs: per-rank compute measured with virtual ranks on one MI355X (%s); transfers = halo bytes / %g GB/s per link, "
                              "neighbours only" % ("every rank measured" if args.all_ranks else "middle strip", args.link_GBps), "summary": summary, "ranks": results}))


if __name__ == "__main__":
    main()
