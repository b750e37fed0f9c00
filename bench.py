#!/usr/bin/env python3
"""Benchmark of the NRD hot path on MI355X: REBLUR_DIFFUSE_SPECULAR @ 2560x1440 (BASELINE.json metric) by default;
--workload relax_ds_sh runs BASELINE.json config 5 (RELAX_DIFFUSE_SPECULAR_SH, 3840x2160, 5 a-trous iterations).

A "step" is one denoised frame = one full pass list (ClassifyTiles, PrePass, TemporalAccumulation, HistoryFix, Blur,
PostBlur, TemporalStabilization) over one synthetic frame whose inputs are ALREADY resident in HBM (generated on the
GPU before the timed region). Prints ONE JSON line (see DESIGN.md "Measurement"):

  value         Mpixels/s of the whole job = steps * 2560 * 1440 / wall time of the timed region (max over ranks)
  roofline      for the dominant kernel: algorithmic bytes per launch / its mean duration,
                measured live with HIP events recorded on the executor's stream around every dispatch
  cpu_baseline  the CPU oracle (oracle/, kind "port": the reference has no CPU implementation) on the host cores,
                on a bounded sample of the same workload (N = 1, rank 0 only)
  parity        the planes the cpu_baseline leg computes anyway (first frames of the same sequence, full size) compared with the GPU's: the library that is
                timed is the library that is checked, and the expected maximum relative error is 0 (bit-identical user outputs)

--gpus N with N > 1 and no torch.distributed environment re-launches itself under torch.distributed.run (one rank per GPU, RCCL).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload reblur_ds|reblur_diffuse|relax_ds_sh|relax_ds|sigma_shadow] [--width 2560 --height 1440]
                  [--graph] [--no-sky] [--no-cpu-baseline] [--no-parity]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch

from raytracingdenoiser_amd import api, scene, sharding, synth
from raytracingdenoiser_amd import build as native_build
from raytracingdenoiser_amd.executor import HipExecutor

# the per-frame guide decode: IN_NORMAL_ROUGHNESS + IN_VIEWZ read (8 B); written: REBLUR float4 (normal, viewZ) + the 4-byte roughness word, RELAX two float4 planes
GUIDE_BYTES_PER_PIXEL = {"REBLUR": 28, "RELAX": 40}
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak (MI355X_MICROARCH.md); ~6300 GB/s is what a float4 copy achieves (measured live below)
VALU_SIMDS, VALU_CLOCK_GHZ = 1024, 2.4  # 256 CUs x 4 SIMD-32; a wave64 v_fma_f32 issues over 2 cycles (MI355X_MICROARCH.md "Per-instruction cycle constants")

# Published reference numbers for the exact metric (reference README.md:18, RTX 4080, 1440p native)
PUBLISHED_MPIX_S = {("REBLUR_DIFFUSE_SPECULAR", 2560, 1440): 1603.0, ("RELAX_DIFFUSE_SPECULAR", 2560, 1440): 1229.0, ("RELAX_DIFFUSE_SPECULAR_SH", 2560, 1440): 760.0}

# HBM traffic per launch from hardware counters: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, tools/pmc_run.sh), stored
# by tools/pmc_to_json.py. FETCH_SIZE is doubled (MI355X_MICROARCH.md "HBM": gfx950 tallies 128-B requests at 64 B); both are KiB.
ISSUE_FLOOR_FILE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "issue_floor.json")
PMC_TRAFFIC_FILE = os.path.join(ROOT, "profiles", "pmc_traffic.json")

# Algorithmic (compulsory) bytes per pixel and per pass for REBLUR_DIFFUSE_SPECULAR with the reference pool formats
# (each plane read / written once per pass).
REBLUR_DS_BYTES_PER_PIXEL = {
    "REBLUR_ClassifyTiles.cs": 4,
    "REBLUR_DiffuseSpecular_PrePass.cs": 42,
    "REBLUR_DiffuseSpecular_TemporalAccumulation.cs": 94,
    "REBLUR_DiffuseSpecular_HistoryFix.cs": 50,
    "REBLUR_DiffuseSpecular_Blur.cs": 46,
    "REBLUR_DiffuseSpecular_PostBlur.cs": 46,
    "REBLUR_DiffuseSpecular_TemporalStabilization.cs": 66,
}
# REBLUR_DIFFUSE (BASELINE.json configs[2]; SURVEY.md section 8a: 4 + 24 + 56 + 29 + 29 + 29 + 40 = 211 B/px)
REBLUR_D_BYTES_PER_PIXEL = {
    "REBLUR_ClassifyTiles.cs": 4,
    "REBLUR_Diffuse_PrePass.cs": 24,
    "REBLUR_Diffuse_TemporalAccumulation.cs": 56,
    "REBLUR_Diffuse_HistoryFix.cs": 29,
    "REBLUR_Diffuse_Blur.cs": 29,
    "REBLUR_Diffuse_PostBlur.cs": 29,
    "REBLUR_Diffuse_TemporalStabilization.cs": 40,
}
# RELAX: SH variant = BASELINE.json config 5; the a-trous entry is per iteration (42 B/px of it are reads)
RELAX_DS_SH_BYTES_PER_PIXEL = {
    "RELAX_ClassifyTiles.cs": 4,
    "RELAX_DiffuseSpecularSh_PrePass.cs": 72,
    "RELAX_DiffuseSpecularSh_TemporalAccumulation.cs": 192,
    "RELAX_DiffuseSpecularSh_HistoryFix.cs": 5,
    "RELAX_DiffuseSpecularSh_HistoryClamping.cs": 150,
    "RELAX_DiffuseSpecularSh_AtrousSmem.cs": 83,
    "RELAX_DiffuseSpecularSh_Atrous.cs": 74,
}
RELAX_DS_BYTES_PER_PIXEL = {
    "RELAX_ClassifyTiles.cs": 4,
    "RELAX_DiffuseSpecular_PrePass.cs": 40,
    "RELAX_DiffuseSpecular_TemporalAccumulation.cs": 112,
    "RELAX_DiffuseSpecular_HistoryFix.cs": 5,
    "RELAX_DiffuseSpecular_HistoryClamping.cs": 86,
    "RELAX_DiffuseSpecular_AtrousSmem.cs": 51,
    "RELAX_DiffuseSpecular_Atrous.cs": 42,
}

# performance mode (ReblurSettings::enablePerformanceMode): same planes per pass, "REBLUR_Perf_*" pipelines
REBLUR_DS_PERF_BYTES_PER_PIXEL = {k.replace("REBLUR_DiffuseSpecular_", "REBLUR_Perf_DiffuseSpecular_"): v for k, v in REBLUR_DS_BYTES_PER_PIXEL.items()}

# SIGMA_SHADOW (BASELINE.json configs[1], 1080p): SURVEY.md section 8a, 68 B/px in total; tiles passes are per 16x16 tile (negligible)
SIGMA_SHADOW_BYTES_PER_PIXEL = {
    "SIGMA_Shadow_ClassifyTiles.cs": 6,
    "SIGMA_Copy.cs": 10,
    "SIGMA_Shadow_Blur.cs": 13,
    "SIGMA_Shadow_PostBlur.cs": 14,
    "SIGMA_Shadow_TemporalStabilization.cs": 25,
}
SIGMA_TRANSLUCENCY_BYTES_PER_PIXEL = {
    "SIGMA_ShadowTranslucency_ClassifyTiles.cs": 10,
    "SIGMA_Copy.cs": 16,
    "SIGMA_ShadowTranslucency_Blur.cs": 20,
    "SIGMA_ShadowTranslucency_PostBlur.cs": 20,
    "SIGMA_ShadowTranslucency_TemporalStabilization.cs": 34,
}

# workload -> (denoiser, default size, bytes/px per pass, denoiser-settings overrides (None = library defaults))
WORKLOADS = {
    "reblur_ds": ("REBLUR_DIFFUSE_SPECULAR", (2560, 1440), REBLUR_DS_BYTES_PER_PIXEL, None),
    "reblur_ds_perf": ("REBLUR_DIFFUSE_SPECULAR", (2560, 1440), REBLUR_DS_PERF_BYTES_PER_PIXEL, {"enablePerformanceMode": True}),
    "reblur_diffuse": ("REBLUR_DIFFUSE", (2560, 1440), REBLUR_D_BYTES_PER_PIXEL, None),
    "relax_ds_sh": ("RELAX_DIFFUSE_SPECULAR_SH", (3840, 2160), RELAX_DS_SH_BYTES_PER_PIXEL, None),
    "sigma_shadow": ("SIGMA_SHADOW", (1920, 1080), SIGMA_SHADOW_BYTES_PER_PIXEL, None),
    "sigma_translucency": ("SIGMA_SHADOW_TRANSLUCENCY", (1920, 1080), SIGMA_TRANSLUCENCY_BYTES_PER_PIXEL, None),
    "relax_ds": ("RELAX_DIFFUSE_SPECULAR", (3840, 2160), RELAX_DS_BYTES_PER_PIXEL, None),
}


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=64)
    ap.add_argument("--warmup", type=int, default=32)
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="reblur_ds", help="reblur_ds = the BASELINE.json metric (default); reblur_ds_perf = same in NRD's performance mode; relax_ds_sh = config 5 (4K)")
    ap.add_argument("--width", type=int, default=0)
    ap.add_argument("--height", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--sharding", choices=["halo", "allgather"], default=os.environ.get("NRD_SHARDING", "halo"),
                    help="multi-GPU scheme: halo = halo exchange between pass segments (point-to-point to the two neighbouring ranks), allgather = redundant halo compute + one all-gather per frame")
    ap.add_argument("--motion-bound", choices=("measure", "estimate"), default="measure", help="halo scheme: where the per-frame motion bound comes from -- measure: a device reduction over "
                    "IN_VIEWZ / IN_MV of every strip + MAX over ranks (nrdHipMeasureMotionRows); estimate: the 5 x 5 camera heuristic of round 3 (no synchronisation)")
    ap.add_argument("--max-motion-rows", type=int, default=None, help="halo scheme: largest vertical motion (rows per frame) the history halos cover (default: 32 up to 1440p, scaled with the height above)")
    ap.add_argument("--cpu-frames", type=int, default=8)
    ap.add_argument("--distinct-frames", type=int, default=0, help="number of distinct generated frames to cycle through (0 = warmup + steps)")
    ap.add_argument("--graph", action="store_true", help="launch each frame as ONE hipGraph (nrdHipSetGraphMode) instead of one launch per pass: bit-identical, less host work per frame, "
                    "but 4-6 us per frame slower than back-to-back launches on this runtime (profiles/r04_t_*: 0.7734 against 0.7689 ms) -- the default is therefore eager since round 4")
    ap.add_argument("--no-graph", action="store_true", help="(the default since round 4; kept for the scripts that pass it)")
    ap.add_argument("--no-sky", action="store_true", help="a dome behind the scene: no sky pixels (34 %% of the default frame are sky and leave at the tile test)")
    ap.add_argument("--uniform", action="store_true", help="tuning runs: every input plane constant (the value of one ground pixel), static camera -- the scene on which an L1-resident A/B build "
                    "(NRD_EXPERIMENT_L1_RESIDENT, csrc/hip/planes.h) computes the same values as the product and so measures each kernel's issue floor")
    ap.add_argument("--dry-plan", action="store_true", help="no GPU: print the multi-GPU plan of --gpus N ranks for this workload (strips, halo bands, bytes, RCCL operations) and exit")
    ap.add_argument("--no-parity", action="store_true", help="skip the GPU-vs-oracle comparison of the cpu_baseline frames")
    return ap.parse_args()


def measure_copy_bandwidth(lib, nbytes=1 << 30, reps=10):
    """GB/s (read + write) of the library's own 16-bytes-per-lane copy kernel over nbytes (include/NRDHip.h nrdHipMeasureCopyBandwidth): the achievable
    HBM rate the roofline is also quoted against ("measured copy bandwidth on the same device")"""
    import ctypes as C

    out = C.c_double()
    r = lib.nrdHipMeasureCopyBandwidth(nbytes, reps, C.c_void_p(torch.cuda.current_stream().cuda_stream), C.byref(out))
    assert r == 0, "nrdHipMeasureCopyBandwidth failed"
    return out.value


def _usable_cores():
    """host cores this process may actually use: the affinity mask, further limited by a cgroup CPU quota if there is one"""
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            fields = open(path).read().split()
            if path.endswith("cpu.max"):
                quota, period = fields[0], float(fields[1])
            else:
                quota, period = fields[0], float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota not in ("max", "-1"):
                cores = max(1, min(cores, int(float(quota) / period + 0.5)))
            break
        except (OSError, ValueError, IndexError):
            continue
    return max(cores, 1)


def cpu_baseline(name, width, height, frames, seq, overrides=None, check_parity=True, ieee_frames=4):
    """Times the CPU oracle on the first `frames` frames of the same sequence (all host cores, OpenMP over rows). The oracle's outputs are then
    compared with a fresh GPU run over the same frames (outside the timed region of either): returns (cpu_baseline, parity).
    The only place of this file that touches tests/ and oracle/ (the checker, never the thing measured)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import parity
    from oracle import driver as oracle_driver

    cores = _usable_cores()
    threads = oracle_driver.load().oracle_set_threads(cores)
    prev_mode = oracle_driver.set_ieee_mode(False)  # the oracle emulates the device's five transcendental instructions: a bit-for-bit comparison
    try:
        ora = parity.OracleRun(name, width, height, threads=threads)
        host_seq = [{k: (v.cpu() if torch.is_tensor(v) else v) for k, v in fr.items()} for fr in seq[:frames]]
        oracle_outputs = []
        dt = 0.0
        for f, frame in enumerate(host_seq):
            cs = scene.common_settings(frame["camera"], host_seq[max(f - 1, 0)]["camera"], width, height, f)
            t0 = time.perf_counter()
            ora.step(frame, cs, scene.denoiser_settings(name, frame, overrides))
            dt += time.perf_counter() - t0
            if check_parity:
                oracle_outputs.append({rt: ora.output(rt).copy() for rt in ora.outs})
    finally:
        oracle_driver.set_ieee_mode(prev_mode)
    baseline = {
        "value": round(frames * width * height / dt / 1e6, 4),
        "unit": "Mpixels/s",
        "cores": threads,
        "kind": "port",
        "sample": "first %d frames of the same %dx%d %s sequence (incl. the CLEAR_AND_RESTART frame), %.1f s" % (frames, width, height, name, dt),
    }
    par = None
    if check_parity:
        stats = parity.ParityStats()
        hip = parity.GpuRun(name, width, height)
        for f, frame in enumerate(seq[:frames]):
            cs = scene.common_settings(frame["camera"], seq[max(f - 1, 0)]["camera"], width, height, f)
            hip.step(frame, cs, scene.denoiser_settings(name, frame, overrides))
            for rt in hip.outs:
                stats.add(rt.name, f, parity.error_stats(hip.output(rt), oracle_outputs[f][rt]))
        sm = stats.summary(True)
        par = {"vs": "CPU oracle (oracle/), the device's v_rcp / v_sqrt / v_rsq / v_exp / v_log emulated from measured tables", "library": "lib/libNRD_hip.so (the library timed above)",
               "oracle_pinned_to": "the reference's own shader text and host code compiled as C++ (oracle/_ref; tests/test_ref_parity.py, tests/test_ref_host.py; DESIGN.md 4.1)",
               "frames": frames, "planes": sorted(stats.outputs()),
               "max_rel_err": sm["max_rel_err"], "p999_rel_err": sm["p999"], "frac_gt_1e-3": sm["frac_gt_tol"], "mean_rel_err": sm["mean"], "bit_exact_frac": sm["bit_exact_frac"],
               "definition": "|gpu - cpu| / max(|cpu|, 1e-3) per value of the user outputs, worst frame"}
        if ieee_frames:
            # the second statement: the same GPU frames against the oracle in plain IEEE arithmetic -- an oracle that knows nothing about the device.
            # A distribution by nature (DESIGN.md "Numerics": tap snaps + recurrence amplify the device's 1-ulp transcendentals), printed beside the 0.
            prev_mode = oracle_driver.set_ieee_mode(True)
            try:
                ora2, hip2, st2 = parity.OracleRun(name, width, height, threads=threads), parity.GpuRun(name, width, height), parity.ParityStats()
                for f, frame in enumerate(seq[:ieee_frames]):
                    host = {k: (v.cpu() if torch.is_tensor(v) else v) for k, v in frame.items()}
                    cs = scene.common_settings(frame["camera"], seq[max(f - 1, 0)]["camera"], width, height, f)
                    ora2.step(host, cs, scene.denoiser_settings(name, host, overrides))
                    hip2.step(frame, scene.common_settings(frame["camera"], seq[max(f - 1, 0)]["camera"], width, height, f), scene.denoiser_settings(name, frame, overrides))
                    for rt in hip2.outs:
                        st2.add(rt.name, f, parity.error_stats(hip2.output(rt), ora2.output(rt)))
            finally:
                oracle_driver.set_ieee_mode(prev_mode)
            s2 = st2.summary(True)
            par["vs_ieee"] = {"vs": "the same oracle in plain IEEE arithmetic (no knowledge of the device)", "frames": ieee_frames, "max_rel_err": s2["max_rel_err"], "p999_rel_err": s2["p999"],
                              "frac_gt_1e-3": s2["frac_gt_tol"], "mean_rel_err": s2["mean"], "bit_exact_frac": s2["bit_exact_frac"]}
        ref_summary = os.path.join(ROOT, "profiles", "r05_ref_parity_summary.txt")
        if os.path.exists(ref_summary):
            # the recorded per-pass comparison with the reference's compiled shader text (tools/ref_report.py; CPU-only, not re-measured here): both metrics of the DEVICE
            # arithmetic -- what this library computes -- for this denoiser, and the plane furthest from the north-star's 1e-3
            row = next((line for line in open(ref_summary) if line.startswith("device") and line.split()[1] == name), None)
            detail = None
            if row:
                import re
                m = re.search(r"bit-exact ([\d.]+)\s+min ok ([\d.]+)\s+min within 1e-3 ([\d.]+)\s+min within 1e-3 \(vector\) ([\d.]+).*worst plane: (.*?) \(bit-exact ([\d.]+), per component ([\d.]+), vector ([\d.]+), max ([\d.eE+-]+)\)", row)
                if m:
                    detail = {"bit_exact_frac_all_planes": float(m.group(1)), "min_within_1e-5_or_1ulp_frac": float(m.group(2)), "min_within_1e-3_frac_per_component": float(m.group(3)),
                              "min_within_1e-3_frac_vector_metric": float(m.group(4)), "worst_plane": m.group(5), "worst_plane_bit_exact_frac": float(m.group(6)),
                              "worst_plane_within_1e-3_per_component": float(m.group(7)), "worst_plane_within_1e-3_vector": float(m.group(8)), "worst_plane_max_rel_err": float(m.group(9))}
            par["vs_reference_text"] = {"what": "the oracle in the DEVICE arithmetic, pass by pass on identical inputs, against the reference's own HLSL shaders compiled as C++ (oracle/_ref, tests/test_ref_parity.py): "
                                                "one pass, no recurrence; 3 frames at 192x128", "device_arithmetic": detail,
                                        "recorded_in": "profiles/r05_ref_parity_summary.txt (CPU-only statistic, not measured in this run)"}
    return baseline, par


def _relaunch_under_torchrun(n):
    """`python bench.py --gpus N` without a torch.distributed environment: become N ranks (one per GPU) under torch.distributed.run"""
    import socket
    import subprocess

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    args = parse_args()
    if args.dry_plan:
        name, default_size, _, overrides = WORKLOADS[args.workload]
        print(json.dumps(sharding.dry_plan(name, args.width or default_size[0], args.height or default_size[1], max(args.gpus, 1), overrides, args.max_motion_rows), indent=1))
        return
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(_relaunch_under_torchrun(args.gpus))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    distributed = world > 1
    assert args.gpus == world, "bench.py: --gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run --nproc-per-node %d, or without any launcher)" % (args.gpus, world, args.gpus)
    assert torch.cuda.is_available(), "bench.py needs a GPU: the HIP path has no CPU fallback"
    # one process per GPU; NRD_DIST_BACKEND=gloo + fewer GPUs than ranks is the single-GPU rehearsal of the N>1 path (tests/test_sharding.py test_bench_two_ranks_rehearsal)
    torch.cuda.set_device(local_rank % torch.cuda.device_count())
    backend = os.environ.get("NRD_DIST_BACKEND", "nccl")  # "nccl" is RCCL on ROCm
    if distributed:
        import torch.distributed as dist

        import datetime

        # a collective that never completes (a peer died, a link is down) ends the run after 5 minutes instead of hanging it for the default 10 / 30
        dist.init_process_group(backend, timeout=datetime.timedelta(seconds=int(os.environ.get("NRD_DIST_TIMEOUT_S", "300"))))
        assert dist.get_world_size() == args.gpus

    if rank == 0:
        native_build.build_product()
    if distributed:
        dist.barrier()
    copy_gbs = measure_copy_bandwidth(api.load_library())

    name, default_size, bytes_per_pixel, overrides = WORKLOADS[args.workload]
    W, H = args.width or default_size[0], args.height or default_size[1]
    total = args.warmup + args.steps
    distinct = args.distinct_frames or total

    if args.no_sky:
        synth.BACKDROP = True
    # ---- synthetic inputs, generated straight into HBM (118 MB per 1440p frame; 96 frames = 11 GB of 288 GB)
    seq = scene.generate_sequence(name, W, H, distinct, device="cuda")
    if args.uniform:
        py, px = int(H * 0.8), W // 2
        seq = seq[:1]
        distinct = 1
        for k, v in list(seq[0].items()):
            if torch.is_tensor(v) and v.dim() >= 2 and v.shape[0] == H and v.shape[1] == W:
                seq[0][k] = v[py:py + 1, px:px + 1].expand_as(v).contiguous()
    torch.cuda.synchronize()
    denoised_fraction = 1.0 - float(torch.stack([fr["is_sky"].float().mean() for fr in seq[:: max(1, len(seq) // 8)]]).mean())  # sky pixels leave at the tile test

    inst = api.Instance([(0, scene.DENOISERS[name][0])])
    ex = HipExecutor(inst, W, H)
    use_graph = args.graph and not args.no_graph
    ex.set_graph_mode(use_graph)
    outputs = []
    for rt, dtype, ch, fmt in scene.output_planes(name, W, H):
        t = torch.zeros((H, W, ch), dtype=dtype, device="cuda")
        ex.bind(rt, t, fmt)
        outputs.append(t)
    shard = None
    if distributed:
        shard = sharding.HaloSharder(ex, inst, W, H, rank, world, max_motion_rows=args.max_motion_rows, measure_motion=args.motion_bound == "measure") if args.sharding == "halo" else sharding.FrameSharder(ex, inst, W, H, rank, world, outputs)

    settings = scene.denoiser_settings(name, seq[0], overrides)
    assert inst.set_denoiser_settings(0, settings) == api.Result.SUCCESS
    def frame_of(f):
        # distinct < total: walk the generated frames back and forth so consecutive frames always have neighbouring cameras
        if distinct >= total or distinct == 1:
            return seq[min(f, distinct - 1)]
        period = 2 * (distinct - 1)
        k = f % period
        return seq[k if k < distinct else period - k]

    frames_cs = []
    for f in range(total):
        cur, prev = frame_of(f), frame_of(max(f - 1, 0))
        frames_cs.append(scene.common_settings(cur["camera"], prev["camera"], W, H, f))

    def step(f):
        frame = frame_of(f)
        for rt, t, fmt in scene.user_planes(name, frame):
            ex.bind(rt, t, fmt)
        assert inst.set_common_settings(frames_cs[f]) == api.Result.SUCCESS
        if shard is not None:
            shard.denoise()
        else:
            ex.denoise()

    def fence():
        if shard is not None and hasattr(shard, "wait_outputs"):
            shard.wait_outputs()  # the last frame's output all-gather belongs to the timed region: every rank holds the complete OUT_* planes when the clock stops
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    # The interpreter's cyclic garbage collector stays out of the measured loops: the frame sequence keeps ~10^5 live Python objects, a generation-2 pass over them takes
    # 5-10 ms -- as long as the whole 20-frame timed region (seen on one box: 1.22 ms per step in the wall clock with 0.71 ms per frame on the GPU, profiles/r05_q_*).
    # No work is skipped: reference counting still frees everything the loops allocate; the collector runs again behind the measurements.
    import gc

    gc.collect()
    gc.freeze()
    gc.disable()

    for f in range(args.warmup):
        step(f)
    fence()

    # ---- the timed region: exactly args.steps frames, barrier + synchronize on both sides
    t0 = time.perf_counter()
    for f in range(args.warmup, total):
        step(f)
    fence()
    elapsed = time.perf_counter() - t0
    graph_stats = ex.graph_stats()
    tile_fallback = ex.tile_fallback_stats()  # REBLUR TemporalAccumulation: tiles of the last frame whose surface-motion window did not fit the LDS window

    # ---- per-pass durations for the roofline: the SAME frames once more with HIP events around every dispatch on the executor's stream (eager launches:
    # events cannot bracket the nodes of a graph). Outside the timed region; the history carried over differs, the work per pass does not.
    ex.set_profiling(True)
    for f in range(args.warmup, total):
        step(f)
    fence()
    timings = ex.collect_pass_timings()
    ex.set_profiling(False)

    # ---- per-frame GPU time (a real-time denoiser is judged on its worst frame): the timed frames once more with an event between consecutive
    # frames (torch events on the executor's stream = torch's current stream), then a CAMERA CUT -- one CLEAR_AND_RESTART frame in mid-sequence (the clears + a frame whose
    # every pixel is reconstructed by HistoryFix) -- and the frames that follow it while the history regrows
    def timed_frames(frames):
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(len(frames) + 1)]
        evs[0].record()
        for i, run in enumerate(frames):
            run()
            evs[i + 1].record()
        fence()
        return [evs[i].elapsed_time(evs[i + 1]) for i in range(len(frames))]

    per_frame_ms = timed_frames([(lambda f=f: step(f)) for f in range(args.warmup, total)])
    last = total - 1

    def cut_frame(k):
        def run():
            frame = frame_of(last)
            for rt, t, fmt in scene.user_planes(name, frame):
                ex.bind(rt, t, fmt)
            cs = scene.common_settings(frame["camera"], frame["camera"], W, H, last + 1 + k, **({"accumulationMode": int(api.AccumulationMode.CLEAR_AND_RESTART)} if k == 0 else {}))
            assert inst.set_common_settings(cs) == api.Result.SUCCESS
            shard.denoise() if shard is not None else ex.denoise()
        return run

    cut_ms = timed_frames([cut_frame(k) for k in range(5)])
    srt = sorted(per_frame_ms)
    frame_ms = {"mean": round(sum(srt) / len(srt), 4), "p50": round(srt[len(srt) // 2], 4), "p99": round(srt[min(len(srt) - 1, int(0.99 * len(srt)))], 4), "max": round(srt[-1], 4), "min": round(srt[0], 4),
                "frames": len(srt), "camera_cut": {"restart_frame_ms": round(cut_ms[0], 4), "following_frames_ms": [round(v, 4) for v in cut_ms[1:]],
                                                  "what": "one CLEAR_AND_RESTART frame after the timed sequence (pool clears + every pixel through the history-fix reconstruction), then 4 frames of regrowing history"},
                "note": "GPU time between consecutive frame boundaries (events on the executor's stream) of a replay of the timed frames, rank 0"}
    gc.enable()  # (the measured loops are over)
    gc.unfreeze()

    if distributed:
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    if rank != 0:
        if distributed:
            dist.barrier()
            dist.destroy_process_group()
        return

    ms_per_step = elapsed * 1e3 / args.steps
    mpix_s = args.steps * W * H / elapsed / 1e6

    # ---- roofline of the dominant kernel (live HIP-event timings of this very run)
    rows = W * H if shard is None else shard.pixels_per_rank()
    passes = {}
    # one GPU: the tile-classification kernel writes the guide planes too (kernels_common.hip DecodeGuidesClassifyKernel) and there is no separate guide row
    guides_in_classify = timings.get(HipExecutor.GUIDE_PREPARATION, (0.0, 0))[1] == 0
    for shader, (ms, n) in timings.items():
        bpp = bytes_per_pixel.get(shader)
        if shader == HipExecutor.GUIDE_PREPARATION or (guides_in_classify and shader in ("REBLUR_ClassifyTiles.cs", "RELAX_ClassifyTiles.cs")):
            bpp = GUIDE_BYTES_PER_PIXEL.get(name.split("_")[0])  # not part of the reference's compulsory traffic: a cost of this design (None: SIGMA / REFERENCE decode nothing)
        if bpp is None or n == 0:
            continue
        avg_ms = ms / n
        gbps = bpp * rows / (avg_ms * 1e-3) / 1e9
        # (several launches of one shader -- the a-trous iterations -- are averaged; frac_* = this pass's algorithmic bytes against the 8 TB/s peak and against the copy rate measured above)
        passes[shader] = {"avg_ms": round(avg_ms, 4), "launches": n, "bytes_per_launch": bpp * rows, "GBps": round(gbps, 1), "frac_of_peak": round(gbps / HBM_PEAK_GBS, 4),
                          "frac_of_measured_copy": round(gbps / copy_gbs, 4)}
    dominant = max(passes, key=lambda k: passes[k]["avg_ms"]) if passes else None
    roofline = None
    if dominant:
        achieved = passes[dominant]["GBps"]
        traffic, traffic_source, valu = None, None, None
        # the recorded evidence (hardware counters, issue floors) carries the digest of the library it was taken from; `stale` = the library timed here is another one
        digest_file = os.path.join(native_build.LIB_DIR, native_build.LIB_NAME + ".digest")
        lib_digest = open(digest_file).read().strip() if os.path.exists(digest_file) and not os.environ.get("NRD_HIP_LIBRARY") else None
        stale = {}
        if world == 1 and os.path.exists(PMC_TRAFFIC_FILE):
            entry = json.load(open(PMC_TRAFFIC_FILE)).get("%s_%dx%d%s" % (name, W, H, "_nosky" if args.no_sky else ""), {})
            k = entry.get("kernels", {}).get(dominant)
            if k:
                stale["counters"] = entry.get("library_digest") is None or entry.get("library_digest") != lib_digest
                traffic = int((2.0 * k["FETCH_SIZE_KiB"] + k["WRITE_SIZE_KiB"]) * 1024)
                traffic_source = entry.get("source")
                if k.get("SQ_INSTS_VALU"):
                    # the OTHER roofline of a kernel that computes: executed VALU instructions against the issue rate of the chip -- a wave64 VALU instruction occupies its
                    # SIMD-32 for 2 cycles (MI355X_MICROARCH.md "Per-instruction cycle constants": v_fma_f32), 256 CUs x 4 SIMDs, 2.4 GHz. The instruction count is the recorded
                    # counter (SQ_INSTS_VALU / launch), the time is this run's: frac = the share of the kernel's time the VALU pipes would need if every instruction were an fma.
                    issue_ms = k["SQ_INSTS_VALU"] * 2.0 / (VALU_SIMDS * VALU_CLOCK_GHZ * 1e9) * 1e3
                    valu = {"executed_valu_per_wave": k["valu_per_wave"], "waves_per_launch": int(k["SQ_WAVES"]), "source": entry.get("source"),
                            "fma_issue_bound_ms": round(issue_ms, 4), "frac": round(issue_ms / passes[dominant]["avg_ms"], 4),
                            "what": "SQ_INSTS_VALU x 2 cycles / (1024 SIMDs x 2.4 GHz) against this run's kernel time: the pure-FMA issue bound (selects, conversions, integer "
                                    "instructions issue at 4, transcendentals at 8 cycles: the measured issue floor below is the tight bound)"}
        # What the dominant kernel is bound by, measured directly (round 4; the counter-derived "VALU busy" of round 3 read up to 109 % and is gone): its time in
        # a build whose loads all hit the L1, on a scene where that build computes the same values (bench.py --uniform, csrc/hip/planes.h NRD_EXPERIMENT_L1_RESIDENT).
        issue_floor = None
        if world == 1 and os.path.exists(ISSUE_FLOOR_FILE):
            fl = json.load(open(ISSUE_FLOOR_FILE))
            frag = {"TemporalAccumulation": "TemporalAccumulationKernel", "TemporalStabilization": "TemporalStabilizationKernel", "HistoryFix": "HistoryFixKernel", "HistoryClamping": "HistoryClampingKernel",
                    "AtrousSmem": "AtrousSmemKernel", "PrePass": "PrePassKernel" if name.startswith("RELAX") else "ReblurSpatialKernel", "Blur": "ReblurSpatialKernel", "PostBlur": "ReblurSpatialKernel",
                    "Atrous": "RelaxAtrousKernel"}.get(dominant.rsplit("_", 1)[-1].replace(".cs", ""))
            rows_fl = next((v for k2, v in fl.get("workloads", {}).items() if k2.startswith(name + " ")), {})
            hit = next((v for k2, v in rows_fl.items() if frag and frag in k2), None)  # (the first match: the window kernel precedes its fallback)
            if hit:
                issue_floor = dict(hit, source=fl.get("source"), what="uniform scene, every pixel denoised: this kernel's time / its time with every load an L1 hit (not measured in this run)")
                stale["issue_floor"] = fl.get("library_digest") is None or fl.get("library_digest") != lib_digest
        # what the evidence says bounds the kernel: at most 25 % above its L1-resident time -> instruction issue; beyond that the memory system behind the L1 (the request path of
        # scattered taps / dependent round trips); HBM itself is a quarter busy (frac). `bound` names that; achieved / peak / frac stay the HBM roofline BASELINE.json asks for.
        bound = "hbm"
        if issue_floor and issue_floor.get("measured_over_floor"):
            bound = "valu-issue" if issue_floor["measured_over_floor"] <= 1.25 else "l1-request"
        roofline = {"bound": bound, "roofline_class": "hbm", "kernel": dominant, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                    "measured_copy_GBps": round(copy_gbs, 1), "frac_of_measured_copy": round(achieved / copy_gbs, 4),
                    "denoised_pixel_fraction": round(denoised_fraction, 4), "frac_denoised_pixels": round(achieved / HBM_PEAK_GBS * denoised_fraction, 4),
                    "traffic": traffic, "traffic_source": traffic_source, "valu": valu, "issue_floor": issue_floor, "avg_kernel_ms": passes[dominant]["avg_ms"],
                    "library_digest": lib_digest, "stale": (any(stale.values()) if stale else None), "stale_fields": sorted(k2 for k2, v in stale.items() if v),
                    "algorithmic_bytes_per_launch": passes[dominant]["bytes_per_launch"],
                    "note": "bound = what the recorded issue-floor evidence says limits the kernel (valu-issue: within 25 % of its L1-resident time; l1-request: the memory system behind the L1); "
                            "achieved / peak / frac = its algorithmic bytes against the HBM peak (roofline_class), valu.frac = its executed VALU instructions against the chip's fma issue rate; "
                            "traffic / valu / issue_floor are recorded measurements (profiles/), stamped with the digest of the library they were taken from: stale = not the library timed here; "
                            "dominant = the pass with the longest average launch; per-pass durations from HIP events on the executor's stream (eager replay of the timed frames); "
                            "frac counts the algorithmic bytes of EVERY pixel of the frame as the metric does, frac_denoised_pixels only those of the pixels that are not sky; "
                            "measured_copy_GBps = the library's own 16-B/lane copy kernel on this GPU"}
    # per frame: every pass once, except the dilated a-trous pass which runs (launches / steps) times
    per_frame = {k: p["launches"] / args.steps for k, p in passes.items()}
    gpu_ms = sum(p["avg_ms"] * per_frame[k] for k, p in passes.items())  # (includes the guide-decode kernel; the TemporalAccumulation row includes its fallback launch)
    total_bpp = sum(bytes_per_pixel[k] * per_frame[k] for k in passes if k in bytes_per_pixel)
    whole_chain = {"algorithmic_bytes_per_pixel": round(total_bpp, 1), "algorithmic_bytes_per_frame": int(total_bpp * W * H), "sum_kernel_ms": round(gpu_ms, 4),
                   "GBps": round(total_bpp * rows / (gpu_ms * 1e-3) / 1e9, 1) if gpu_ms else None,
                   "frac_of_peak": round(total_bpp * rows / (gpu_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if gpu_ms else None,
                   "frac_of_measured_copy": round(total_bpp * rows / (gpu_ms * 1e-3) / 1e9 / copy_gbs, 4) if gpu_ms else None}

    result = {
        "metric": "Mpixels/s %s @%s%s" % (name, {(2560, 1440): "1440p", (3840, 2160): "4K", (1920, 1080): "1080p"}.get((W, H), "%dx%d" % (W, H)),
                                         "" if overrides is None else " (%s)" % ", ".join("%s=%s" % kv for kv in sorted(overrides.items()))),
        "value": round(mpix_s, 2),
        "unit": "Mpixels/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4),
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": round(mpix_s / PUBLISHED_MPIX_S[(name, W, H)], 3) if world == 1 and overrides is None and (name, W, H) in PUBLISHED_MPIX_S else None,
        "dtype": "f32",
        "data": "synthetic",
        "numerics": "exact",  # one library, one arithmetic: what is timed here is bit-identical to the CPU oracle (see "parity")
        "tile_fallback": {"tiles": tile_fallback[0], "of": tile_fallback[1], "what": "32x8-pixel tiles of the last frame's rect left to the plain TemporalAccumulation kernel (LDS window too small)"},
        "launch": "eager (one launch per pass, back to back)" if not use_graph else "hipGraph (%d launches, %d builds, %d node updates in the run)" % graph_stats,
        "rccl_ranks": world if distributed and backend == "nccl" else (0 if not distributed else None),
        "config": {"workload": "%s %dx%d, %s, analytic scene%s + 1rpp noise, moving camera" % (name, W, H, "default settings" if overrides is None else "settings %s" % overrides,
                                                                                                 " with a backdrop dome (no sky pixels)" if args.no_sky else ""),
                   "parallelism": "1 GPU" if world == 1 else ("row strips x%d, halo exchange between pass segments (RCCL send/recv to the 2 neighbours, %.1f MB received per rank per frame)"
                                                               % (world, shard.exchanged_bytes / max(total, 1) / 1e6) if args.sharding == "halo" else "row strips x%d + RCCL all-gather" % world),
                   "strips": (list(shard.bounds) if shard is not None and getattr(shard, "bounds", None) else None),  # halo scheme: rows owned by each rank (re-cut from the tile map)
                   # the north-star's reassembly (BASELINE.json configs[3]): every rank ends the frame with the complete OUT_* planes
                   "reassembly": (None if shard is None else
                                  "all-gather of the owned rows of every OUT_* plane after the last pass into separate complete planes (RCCL, asynchronous: it overlaps the whole next frame and is awaited by wait_outputs()), "
                                  "%.2f MB received per rank per frame, inside the timed region" % (shard.gathered_bytes / max(shard.gather_frames, 1) / 1e6) if args.sharding == "halo" else
                                  "in-place all-gather of every permanent and transient full-resolution plane and every output (FrameSharder)"),
                   "halo_bytes_received_per_rank_per_frame": (int(shard.exchanged_bytes / max(total + args.steps + 5, 1)) if shard is not None and args.sharding == "halo" else None),
                   "gather_bytes_received_per_rank_per_frame": (int(shard.gathered_bytes / max(shard.gather_frames, 1)) if shard is not None and args.sharding == "halo" else None),
                   "motion_bound": (None if shard is None or args.sharding != "halo" else
                                    {"source": "device reduction per strip + MAX over ranks (nrdHipMeasureMotionRows)" if args.motion_bound == "measure" else "camera estimate (5 x 5 samples)",
                                     "last_measured_rows": shard.measured_motion_rows, "halo_rows": shard.max_motion_rows, "frames_run_unsharded_for_motion": shard.motion_fallbacks,
                                     # (round 6) what the measured surface motion cannot bound -- virtual motion and look-back taps of the specular signal -- as the temporal kernels
                                     # reported it for the previous frame (nrdHipSetHistoryReachWord), and the sharded frames that turned out to have read beyond their halo
                                     "last_history_reach_rows": shard.history_reach_rows, "history_halo_violations": shard.history_halo_violations}),
                   "storage": "reference pool formats (fp16 history, R10G10B10A2 normals), %.0f B/px/frame compulsory traffic" % total_bpp},
        "roofline": roofline,
        "whole_chain": whole_chain,
        "passes": passes,
        "frame_ms": frame_ms,
        # which regime of the temporal chain the timed frames are in: SURVEY section 8d specifies 32 warm-up frames + the mean of 64 (this file's
        # default); a shorter warm-up times frames whose history is still growing (maxAccumulatedFrameNum 30), with wider blur radii: slightly more work per frame
        "protocol": {"timed_frames": "%d..%d after the CLEAR_AND_RESTART frame 0" % (args.warmup, total - 1),
                     "regime": "steady state (history saturated: SURVEY 8d, 32 warm-up + 64 timed)" if args.warmup >= 30 else "accumulating (history below maxAccumulatedFrameNum = 30 during part of the timed frames)"},
    }
    if not args.no_cpu_baseline and world == 1:
        result["cpu_baseline"], result["parity"] = cpu_baseline(name, W, H, min(args.cpu_frames, distinct), seq, overrides, check_parity=not args.no_parity)
    else:
        result["cpu_baseline"], result["parity"] = None, None
    print(json.dumps(result))
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
