"""HIP path vs CPU oracle at BASELINE.json's own sizes and over long sequences (VERDICT r01 "next round" items 1 and 2).

Three kinds of runs, all through the C-ABI:

* exact build (libNRD_hip_exact.so) vs the oracle in device-emulation mode: bit-exact expected (max relative error 0) on every user output and every
  pool plane -- at 2560x1440 (REBLUR_DIFFUSE_SPECULAR, REBLUR_DIFFUSE), 1920x1080 (SIGMA_SHADOW), 3840x2160 (RELAX_DIFFUSE_SPECULAR_SH, 5 a-trous
  iterations), and over 48 frames at 192x128 (40 frames of camera motion, then 8 frames standing still: accumulation counters saturate, anti-lag
  reacts to the stop, RELAX's a-trous takes its long-history branch);
* fast build (libNRD_hip.so, the product) vs the oracle in IEEE mode (correctly rounded sqrt / rsqrt, no knowledge of the device): the north-star's
  "<= 1e-3 relative per pixel" read as a distribution, because the chain is recurrent and full of thresholds, so 1-ulp differences flip branches for
  a few pixels. The tests assert on the user outputs: the 99.9th percentile of the per-value relative error <= 1e-3, the fraction above 1e-3 below a
  small bound, a small mean; they print the maximum and where it is.
* exact build vs the oracle in IEEE mode: what the hardware sqrt / rsqrt alone cost (reported, loosely bounded).

The relative error is |got - want| / max(|want|, 1e-3), as everywhere in tests/parity.py.
"""
import json
import os

import numpy as np
import pytest
import torch

import parity

pytestmark = pytest.mark.gpu

REPORT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "parity_report.jsonl")


def _report(tag, name, size, frames, stats):
    """one JSON line per run (gpurun_out/parity_report.jsonl -> profiles/): the numbers DESIGN.md quotes"""
    row = {"run": tag, "denoiser": name, "size": list(size), "frames": frames, "outputs": stats.summary(True), "all_planes": stats.summary(False),
           "per_plane": {k: {kk: vv for kk, vv in v.items() if kk != "worst"} | {"worst": v["worst"]} for k, v in stats.planes.items()}}
    try:
        os.makedirs(os.path.dirname(REPORT), exist_ok=True)
        with open(REPORT, "a") as fp:
            fp.write(json.dumps(row) + "\n")
    except OSError:
        pass
    print("%s %s %dx%d x%d: outputs max %.3g  p99.9 %.3g  frac>1e-3 %.3g  mean %.3g | all planes max %.3g frac>1e-3 %.3g" % (
        tag, name, size[0], size[1], frames, row["outputs"]["max_rel_err"], row["outputs"]["p999"], row["outputs"]["frac_gt_tol"], row["outputs"]["mean"],
        row["all_planes"]["max_rel_err"], row["all_planes"]["frac_gt_tol"]))
    return row


FULL_SIZE = [
    ("REBLUR_DIFFUSE_SPECULAR", 2560, 1440, 3, None),   # BASELINE.json configs[3] / the metric's configuration
    ("REBLUR_DIFFUSE", 2560, 1440, 3, None),            # configs[2]
    ("SIGMA_SHADOW", 1920, 1080, 4, None),              # configs[1]
    ("RELAX_DIFFUSE_SPECULAR_SH", 3840, 2160, 2, None), # configs[4] (atrousIterationNum = 5 is the library default)
]


@pytest.mark.parametrize("name,width,height,frames,overrides", FULL_SIZE, ids=[c[0] + "_%dx%d" % (c[1], c[2]) for c in FULL_SIZE])
def test_exact_build_bit_exact_at_baseline_size(name, width, height, frames, overrides):
    worst = parity.run_parity(name, width, height, frames, settings_overrides=overrides, numerics="exact", device="cuda")
    assert worst == 0.0, "exact build differs from the oracle at %dx%d: max rel err %g" % (width, height, worst)


@pytest.mark.parametrize("name,width,height,frames,overrides", FULL_SIZE, ids=[c[0] + "_%dx%d" % (c[1], c[2]) for c in FULL_SIZE])
def test_fast_build_within_tolerance_at_baseline_size(name, width, height, frames, overrides):
    stats = parity.ParityStats()
    parity.run_parity(name, width, height, frames, settings_overrides=overrides, numerics="fast", ieee=True, stats=stats, device="cuda")
    row = _report("fast_vs_ieee_oracle", name, (width, height), frames, stats)
    out = row["outputs"]
    assert out["p999"] <= parity.REL_TOL, out
    assert out["frac_gt_tol"] <= 2e-3, out
    assert out["mean"] <= 1e-4, out


LONG = ["REBLUR_DIFFUSE_SPECULAR", "RELAX_DIFFUSE_SPECULAR_SH", "RELAX_DIFFUSE_SPECULAR", "SIGMA_SHADOW", "REBLUR_DIFFUSE_SPECULAR_OCCLUSION"]


@pytest.mark.parametrize("name", LONG)
def test_exact_build_bit_exact_over_48_frames(name):
    worst = parity.run_parity(name, 192, 128, 48, numerics="exact", static_after=39)
    assert worst == 0.0, "exact build differs from the oracle within 48 frames: max rel err %g" % worst


@pytest.mark.parametrize("name", LONG)
def test_fast_build_within_tolerance_over_48_frames(name):
    stats = parity.ParityStats()
    parity.run_parity(name, 192, 128, 48, numerics="fast", ieee=True, stats=stats, static_after=39)
    row = _report("fast_vs_ieee_oracle_48f", name, (192, 128), 48, stats)
    out = row["outputs"]
    assert out["p999"] <= 2 * parity.REL_TOL, out  # 24.6k texels: the 99.9th percentile is the 25th-worst value
    assert out["frac_gt_tol"] <= 5e-3, out
    assert out["mean"] <= 1e-4, out


@pytest.mark.parametrize("name", ["REBLUR_DIFFUSE_SPECULAR", "RELAX_DIFFUSE_SPECULAR_SH"])
def test_exact_build_vs_ieee_oracle_32_frames(name):
    """what the device's v_sqrt_f32 / v_rsq_f32 (within 1 ulp of the correctly rounded result) cost against an oracle that knows nothing about them"""
    stats = parity.ParityStats()
    parity.run_parity(name, 192, 128, 32, numerics="exact", ieee=True, stats=stats)
    row = _report("exact_vs_ieee_oracle_32f", name, (192, 128), 32, stats)
    out = row["outputs"]
    assert out["p999"] <= 2 * parity.REL_TOL, out
    assert out["frac_gt_tol"] <= 5e-3, out


def test_history_threshold_branch_of_atrous_is_reached():
    """RELAX AtrousSmem takes its 3x3 filtered-variance branch once historyLength >= gHistoryThreshold (reference RELAX_AtrousSmem.hlsli:251-336):
    the history-length plane of the long run must get there (the 4..6-frame runs of test_relax.py never do)"""
    name = "RELAX_DIFFUSE_SPECULAR_SH"
    seq = parity.generate_sequence(name, 192, 128, 12)
    hip = parity.HipRun(name, 192, 128, numerics="exact")
    RT = parity.RT
    for f, frame in enumerate(seq):
        cs = parity.common_settings(frame["camera"], seq[max(f - 1, 0)]["camera"], 192, 128, f)
        hip.step(frame, cs, parity.denoiser_settings(name, frame))
    longest = 0
    for i, (fmt, _) in enumerate(hip.inst.permanent_pool):
        if fmt == parity.F.R8_UNORM:
            raw, pf, w = hip.ex.read_pool_plane(RT.PERMANENT_POOL, i)
            longest = max(longest, int(parity.decode_plane(raw, pf, w).max()))
    assert longest >= 8, longest  # default historyFixFrameNum = 3 -> gHistoryThreshold = 4
