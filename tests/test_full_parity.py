"""HIP path vs CPU oracle at BASELINE.json's own sizes and over long sequences (VERDICT r01 "next round" items 1 and 2).

Three kinds of runs, all through the C-ABI:

* exact build (libNRD_hip_exact.so) vs the oracle in device-emulation mode: bit-exact expected (max relative error 0) on every user output and every
  pool plane -- at 2560x1440 (REBLUR_DIFFUSE_SPECULAR, REBLUR_DIFFUSE), 1920x1080 (SIGMA_SHADOW), 3840x2160 (RELAX_DIFFUSE_SPECULAR_SH, 5 a-trous
  iterations), and over 48 frames at 192x128 (40 frames of camera motion, then 8 frames standing still: accumulation counters saturate, anti-lag
  reacts to the stop, RELAX's a-trous takes its long-history branch);
* exact build vs the oracle in IEEE mode (correctly rounded sqrt / rsqrt, no knowledge of the device): the ONLY difference between the two sides is that
  the device's v_sqrt_f32 / v_rsq_f32 are off by one ulp for ~15 % of their inputs. Measured (r02_a, profiles/r02_parity_report.jsonl): after 32 frames
  2 % (REBLUR) to 5 % (RELAX SH) of the output values differ by more than 1e-3. The chain amplifies ulp-level differences: every pass snaps its 16
  Poisson taps per pixel to pixel centres (floor(uv * rectSize)) at radii of up to 60 px, so a relative perturbation of 1e-7 in a blur radius moves a tap
  to the neighbouring 1-rpp texel with probability ~1e-5 per pixel and pass; that pixel then differs by ~noise / 8, the pixels that tap it by ~1 %, and
  the temporal history carries it on. "<= 1e-3 per pixel against a CPU reference" is therefore only attainable bit-exactly, which is what the exact
  build delivers against the device-emulating oracle; against anything else the honest statement is a distribution.
* fast build (libNRD_hip.so, the product: hardware rcp / exp2 / log2, FMA contraction, reassociation, tap positions generated in pixel units) vs the
  oracle in IEEE mode: the same mechanism with more perturbed operations -- after 3 frames at 1440p 2.3 % of the output values are beyond 1e-3 (mean
  relative error 4e-4) with -DNRD_FAST_TAP_POSITIONS=1; with the default (tap positions in the reference's operation order) 0.74 % / 6e-5; after 48 frames 4-11 %.
  The tests bound the mean error and the fraction beyond 1e-3, hold them against the exact-vs-IEEE figures of the same sequence (same mechanism, same
  order of magnitude) and check that the fast build DENOISES as well as the oracle (error against a converged image within 2 %).

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
    sh = name.endswith("_SH")  # the SH1 planes are signed and small: relative errors against a 1e-3 floor are inflated there
    # measured r02_final (product flags: hardware transcendentals, contraction, reassociation; tap positions in the reference's operation order): 0.74 % of the
    # REBLUR_DS output values beyond 1e-3, mean 6e-5, p99.9 0.005; r02_m with -DNRD_FAST_TAP_POSITIONS=1: 2.3 % / 3.9e-4 / 0.12 (a tap position that differs by a
    # few ulp crosses a pixel boundary ~1e-4 of the time, 16 taps x 3 passes per pixel); REBLUR_D 0.26 %, SIGMA 7e-6, RELAX SH 10 % (2 frames at 4K)
    assert out["frac_gt_tol"] <= (0.2 if sh else 0.03), out
    assert out["mean"] <= (5e-3 if sh else 5e-4), out


LONG = ["REBLUR_DIFFUSE_SPECULAR", "RELAX_DIFFUSE_SPECULAR_SH", "RELAX_DIFFUSE_SPECULAR", "SIGMA_SHADOW", "REBLUR_DIFFUSE_SPECULAR_OCCLUSION"]


@pytest.mark.parametrize("name", LONG)
def test_exact_build_bit_exact_over_48_frames(name):
    worst = parity.run_parity(name, 192, 128, 48, numerics="exact", static_after=39)
    assert worst == 0.0, "exact build differs from the oracle within 48 frames: max rel err %g" % worst


@pytest.mark.parametrize("name", ["REBLUR_DIFFUSE_SPECULAR", "RELAX_DIFFUSE_SPECULAR_SH", "SIGMA_SHADOW"])  # one per family (three oracle runs of 48 frames each)
def test_fast_build_within_tolerance_over_48_frames(name):
    stats = parity.ParityStats()
    parity.run_parity(name, 192, 128, 48, numerics="fast", ieee=True, stats=stats, static_after=39)
    row = _report("fast_vs_ieee_oracle_48f", name, (192, 128), 48, stats)
    out = row["outputs"]
    # the same sequence, exact build vs the IEEE oracle: differences that stem from 1-ulp sqrt / rsqrt deviations alone
    base = parity.ParityStats()
    parity.run_parity(name, 192, 128, 48, numerics="exact", ieee=True, stats=base, static_after=39)
    ref = _report("exact_vs_ieee_oracle_48f", name, (192, 128), 48, base)["outputs"]
    sh = name.endswith("_SH")
    assert out["frac_gt_tol"] <= (0.3 if sh else 0.15), out   # measured r02_a: 7 % REBLUR_DS, 11 % RELAX SH, 4 % RELAX, 1e-4 SIGMA, 0.6 % occlusion
    assert out["mean"] <= (0.06 if sh else 5e-3), out
    assert out["frac_gt_tol"] <= 8.0 * max(ref["frac_gt_tol"], 1e-3), (out, ref)  # same mechanism, same order of magnitude as +-1 ulp in sqrt alone


@pytest.mark.parametrize("name", ["REBLUR_DIFFUSE_SPECULAR", "RELAX_DIFFUSE_SPECULAR_SH"])
def test_exact_build_vs_ieee_oracle_32_frames(name):
    """what the device's v_sqrt_f32 / v_rsq_f32 (within 1 ulp of the correctly rounded result) cost against an oracle that knows nothing about them"""
    stats = parity.ParityStats()
    parity.run_parity(name, 192, 128, 32, numerics="exact", ieee=True, stats=stats)
    row = _report("exact_vs_ieee_oracle_32f", name, (192, 128), 32, stats)
    out = row["outputs"]
    assert 0.0 < out["frac_gt_tol"] <= 0.15, out  # measured r02_a: 2.1 % REBLUR_DS, 5.1 % RELAX SH -- the amplification the module docstring describes
    assert out["bit_exact_frac"] >= 0.7, out


@pytest.mark.parametrize("name,plane", [("REBLUR_DIFFUSE_SPECULAR", "OUT_DIFF_RADIANCE_HITDIST"), ("RELAX_DIFFUSE_SPECULAR", "OUT_DIFF_RADIANCE_HITDIST")])
def test_fast_build_denoises_as_well_as_the_oracle(name, plane):
    """functional parity of the product build: static camera, 24 frames; the error of the denoised diffuse luminance against a converged image (the mean
    of 128 independent 1-rpp inputs) must be the same for the HIP fast build and for the IEEE oracle to within 2 %"""
    from oracle import driver as oracle_driver

    w, h, frames = 192, 128, 24
    key = "diff_relax" if name.startswith("RELAX") else "diff"
    seq = [parity.synth.render_frame(w, h, f, want=parity.DENOISERS[name][1], camera_frame=0) for f in range(128)]
    lum = lambda t: (t[..., 0] if name.startswith("REBLUR") else t[..., 0] * 0.2126 + t[..., 1] * 0.7152 + t[..., 2] * 0.0722)  # REBLUR signals are YCoCg, RELAX RGB
    converged = np.mean([lum(fr[key].float().numpy()) for fr in seq], axis=0)
    geometry = ~seq[0]["is_sky"].numpy()
    prev = oracle_driver.set_ieee_mode(True)
    try:
        ora, hip = parity.OracleRun(name, w, h), parity.HipRun(name, w, h, numerics="fast")
        for f in range(frames):
            cs = parity.common_settings(seq[f]["camera"], seq[max(f - 1, 0)]["camera"], w, h, f)
            ora.step(seq[f], cs, parity.denoiser_settings(name, seq[f]))
            hip.step(seq[f], parity.common_settings(seq[f]["camera"], seq[max(f - 1, 0)]["camera"], w, h, f), parity.denoiser_settings(name, seq[f]))
    finally:
        oracle_driver.set_ieee_mode(prev)
    rt = getattr(parity.RT, plane)
    rmse = lambda out: float(np.sqrt(np.mean((lum(out)[geometry] - converged[geometry]) ** 2)))
    e_ora, e_hip, e_in = rmse(ora.output(rt)), rmse(hip.output(rt)), rmse(seq[frames - 1][key].float().numpy())
    print("%s: RMSE vs converged -- 1-rpp input %.4g, oracle %.4g, HIP fast build %.4g" % (name, e_in, e_ora, e_hip))
    assert e_ora < 0.5 * e_in  # the denoiser denoises
    assert abs(e_hip - e_ora) <= 0.02 * e_ora, (e_hip, e_ora)


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
