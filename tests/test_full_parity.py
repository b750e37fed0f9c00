"""HIP path vs CPU oracle at BASELINE.json's own sizes and over long sequences.

Two kinds of runs, all through the C-ABI, all on THE library (lib/libNRD_hip.so -- the one the benchmark times):

* against the oracle in device-emulation mode: bit-exact expected (max relative error 0) on every user output and every pool plane -- at 2560x1440
  (REBLUR_DIFFUSE_SPECULAR, REBLUR_DIFFUSE), 1920x1080 (SIGMA_SHADOW), 3840x2160 (RELAX_DIFFUSE_SPECULAR_SH, 5 a-trous iterations), and over 48 frames
  at 192x128 (40 frames of camera motion, then 8 frames standing still: accumulation counters saturate, anti-lag reacts to the stop, RELAX's a-trous
  takes its long-history branch). This is the north-star's "<= 1e-3 max relative error vs the CPU reference", met with error 0.
* against the oracle in IEEE mode (reference results for rcp / sqrt / rsqrt / exp2 / log2, no knowledge of the device): the ONLY difference between the
  two sides is that the device's five transcendental instructions are off by one ulp for 4-23 % of their inputs (profiles/r03_a_hw_tables_report.txt).
  After 32 frames a few per cent of the output values differ by more than 1e-3: the chain amplifies ulp-level differences -- every pass snaps its 16
  Poisson taps per pixel to pixel centres (floor(uv * rectSize)) at radii of up to 60 px, so a relative perturbation of 1e-7 in a blur radius moves a tap
  to the neighbouring 1-rpp texel with probability ~1e-5 per pixel and pass; that pixel then differs by ~noise / 8, the pixels that tap it by ~1 %, and
  the temporal history carries it on. "<= 1e-3 per pixel against a CPU reference" is therefore only attainable bit-exactly -- which is why the oracle
  emulates the instructions -- and against anything else the honest statement is a distribution plus "denoises equally well" (last test).

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
    ("RELAX_DIFFUSE_SPECULAR_SH", 3840, 2160, 4, None), # configs[4] (atrousIterationNum = 5 is the library default); 4 frames since round 6
]


@pytest.mark.parametrize("name,width,height,frames,overrides", FULL_SIZE, ids=[c[0] + "_%dx%d" % (c[1], c[2]) for c in FULL_SIZE])
def test_bit_exact_at_baseline_size(name, width, height, frames, overrides):
    worst = parity.run_parity(name, width, height, frames, settings_overrides=overrides, device="cuda")
    assert worst == 0.0, "the library differs from the oracle at %dx%d: max rel err %g" % (width, height, worst)


LONG = ["REBLUR_DIFFUSE_SPECULAR", "RELAX_DIFFUSE_SPECULAR_SH", "RELAX_DIFFUSE_SPECULAR", "SIGMA_SHADOW", "REBLUR_DIFFUSE_SPECULAR_OCCLUSION"]


@pytest.mark.parametrize("name", LONG)
def test_bit_exact_over_48_frames(name):
    worst = parity.run_parity(name, 192, 128, 48, static_after=39)
    assert worst == 0.0, "the library differs from the oracle within 48 frames: max rel err %g" % worst


# measured (profiles/r03 parity_report, profiles/r04_ieee_parity.jsonl): REBLUR_DS 3.3 % beyond 1e-3 / 85 % bit-exact, RELAX_DS_SH 5.5 % / 86 % -- the bounds sit just above
@pytest.mark.parametrize("name,max_frac,min_exact", [("REBLUR_DIFFUSE_SPECULAR", 0.05, 0.84), ("RELAX_DIFFUSE_SPECULAR_SH", 0.07, 0.84)])
def test_vs_ieee_oracle_32_frames(name, max_frac, min_exact):
    """what the device's v_rcp / v_sqrt / v_rsq / v_exp / v_log (each within 1 ulp of the reference result) cost against an oracle that knows nothing about them"""
    stats = parity.ParityStats()
    parity.run_parity(name, 192, 128, 32, ieee=True, stats=stats)
    row = _report("vs_ieee_oracle_32f", name, (192, 128), 32, stats)
    out = row["outputs"]
    assert 0.0 < out["frac_gt_tol"] <= max_frac, out  # the amplification the module docstring describes (round 2, sqrt / rsqrt alone: 2.1 % REBLUR_DS, 5.1 % RELAX SH)
    assert out["bit_exact_frac"] >= min_exact, out


# measured on the emulation backend (round 4): REBLUR_DS 4.5 % beyond 1e-3 / 82.4 % bit-exact, RELAX_DS_SH 6.5 % / 84.5 % -- camera stop at frame 39 included
@pytest.mark.parametrize("name,max_frac,min_exact", [("REBLUR_DIFFUSE_SPECULAR", 0.06, 0.80), ("RELAX_DIFFUSE_SPECULAR_SH", 0.08, 0.82)])
def test_vs_ieee_oracle_48_frames(name, max_frac, min_exact):
    """the same statistic once the history is saturated and the camera has stopped (anti-lag): the distribution does not keep growing with the frame count"""
    stats = parity.ParityStats()
    parity.run_parity(name, 192, 128, 48, ieee=True, stats=stats, static_after=39)
    out = _report("vs_ieee_oracle_48f", name, (192, 128), 48, stats)["outputs"]
    assert 0.0 < out["frac_gt_tol"] <= max_frac and out["bit_exact_frac"] >= min_exact, out


def test_vs_ieee_oracle_at_baseline_size():
    """the headline configuration (REBLUR_DIFFUSE_SPECULAR 2560x1440) against the IEEE oracle after a few frames (keep the device-agnostic statistic at
    the BASELINE size): mean <= 5e-4, <= 3 % of the output values beyond 1e-3"""
    stats = parity.ParityStats()
    parity.run_parity("REBLUR_DIFFUSE_SPECULAR", 2560, 1440, 4, ieee=True, stats=stats, device="cuda", check_pools=False)
    row = _report("vs_ieee_oracle_1440p", "REBLUR_DIFFUSE_SPECULAR", (2560, 1440), 4, stats)
    out = row["outputs"]
    assert out["mean"] <= 5e-4 and out["frac_gt_tol"] <= 0.03 and out["bit_exact_frac"] >= 0.9, out


@pytest.mark.parametrize("name,plane", [("REBLUR_DIFFUSE_SPECULAR", "OUT_DIFF_RADIANCE_HITDIST"), ("RELAX_DIFFUSE_SPECULAR", "OUT_DIFF_RADIANCE_HITDIST")])
def test_denoises_as_well_as_the_ieee_oracle(name, plane):
    """functional parity against an oracle that knows nothing about the device: static camera, 24 frames; the error of the denoised diffuse luminance
    against a converged image (the mean of 128 independent 1-rpp inputs) must be the same for the HIP path and for the IEEE oracle to within 2 %"""
    from oracle import driver as oracle_driver

    w, h, frames = 192, 128, 24
    key = "diff_relax" if name.startswith("RELAX") else "diff"
    seq = [parity.synth.render_frame(w, h, f, want=parity.DENOISERS[name][1], camera_frame=0) for f in range(128)]
    lum = lambda t: (t[..., 0] if name.startswith("REBLUR") else t[..., 0] * 0.2126 + t[..., 1] * 0.7152 + t[..., 2] * 0.0722)  # REBLUR signals are YCoCg, RELAX RGB
    converged = np.mean([lum(fr[key].float().numpy()) for fr in seq], axis=0)
    geometry = ~seq[0]["is_sky"].numpy()
    prev = oracle_driver.set_ieee_mode(True)
    try:
        ora, hip = parity.OracleRun(name, w, h), parity.HipRun(name, w, h)
        for f in range(frames):
            cs = parity.common_settings(seq[f]["camera"], seq[max(f - 1, 0)]["camera"], w, h, f)
            ora.step(seq[f], cs, parity.denoiser_settings(name, seq[f]))
            hip.step(seq[f], parity.common_settings(seq[f]["camera"], seq[max(f - 1, 0)]["camera"], w, h, f), parity.denoiser_settings(name, seq[f]))
    finally:
        oracle_driver.set_ieee_mode(prev)
    rt = getattr(parity.RT, plane)
    rmse = lambda out: float(np.sqrt(np.mean((lum(out)[geometry] - converged[geometry]) ** 2)))
    e_ora, e_hip, e_in = rmse(ora.output(rt)), rmse(hip.output(rt)), rmse(seq[frames - 1][key].float().numpy())
    print("%s: RMSE vs converged -- 1-rpp input %.4g, oracle %.4g, HIP %.4g" % (name, e_in, e_ora, e_hip))
    assert e_ora < 0.5 * e_in  # the denoiser denoises
    assert abs(e_hip - e_ora) <= 0.02 * e_ora, (e_hip, e_ora)


def test_history_threshold_branch_of_atrous_is_reached():
    """RELAX AtrousSmem takes its 3x3 filtered-variance branch once historyLength >= gHistoryThreshold (reference RELAX_AtrousSmem.hlsli:251-336):
    the history-length plane of the long run must get there (the 4..6-frame runs of test_relax.py never do)"""
    name = "RELAX_DIFFUSE_SPECULAR_SH"
    seq = parity.generate_sequence(name, 192, 128, 12)
    hip = parity.HipRun(name, 192, 128)
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
