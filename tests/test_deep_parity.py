"""The parity chain in depth (round 6): the GPU against the reference's own shader text over a sequence, and bit-exactness against the oracle in the
saturated-history regime at BASELINE.json's sizes. Kept apart from tests/test_full_parity.py so that pytest-xdist (--dist loadfile) runs the two files side by side."""
import pytest

import parity
from test_full_parity import _report

pytestmark = pytest.mark.gpu


# ---- the parity chain in ONE hop: the GPU against the reference's own shader text
# GPU == device-mode oracle is bit-exact (above); oracle vs reference text is held pass by pass on the CPU (tests/test_ref_parity.py). Here nothing of ours stands in between:
# the library on the GPU and oracle/_ref/libnrdref.so -- /root/reference/Shaders/Source/*.cs.hlsl compiled as C++ in plain IEEE arithmetic; the .so travels to the GPU box -- run
# the same 8-frame sequence, each with its own history. Two implementations that differ in roundings (the device's v_rcp / v_sqrt / v_rsq / v_exp / v_log and fused multiply-adds
# on one side) are a distribution under this recurrent, threshold-laden chain (module docstring): the floors are those of the IEEE-oracle tests above. Colour / direction texels
# are measured against the texel's largest channel, as in tests/ref_parity.py (an SH1 component that cancels to ~0 inherits its neighbours' absolute error); the per-component
# figure is reported beside it.
# measured, worst plane of the worst frame (emulation backend, round 6): REBLUR_DS 0.05 % of the values beyond 1e-3 (per component 0.38 %) / 95.5 % bit-exact;
# RELAX_DS_SH 1.1 % (per component 11 %: OUT_DIFF_SH1) / 80 %; SIGMA_SHADOW 0.012 % / 99.99 %. Under the other G-buffer encodings (tests/test_encodings.py runs this test with
# NRD_NORMAL_ENCODING / NRD_ROUGHNESS_ENCODING set): RELAX_DS_SH 3.1 % / 79.6 % with RGBA16_SNORM normals + square-root roughness, which is what the RELAX floor leaves room for
REF_TEXT_SEQUENCES = [("REBLUR_DIFFUSE_SPECULAR", 0.01, 0.90), ("RELAX_DIFFUSE_SPECULAR_SH", 0.05, 0.75), ("SIGMA_SHADOW", 0.001, 0.999)]


@pytest.mark.skipif(not __import__("oracle.driver", fromlist=["x"]).ref_available(), reason="oracle/_ref/libnrdref.so not built (needs /root/reference: make -C oracle/ref -j8; the .so travels)")
@pytest.mark.parametrize("name,max_frac,min_exact", REF_TEXT_SEQUENCES)
def test_gpu_sequence_against_the_reference_shader_text(name, max_frac, min_exact):
    from oracle import driver as oracle_driver
    from raytracingdenoiser_amd import api

    w, h, frames = 192, 128, 8
    seq = parity.generate_sequence(name, w, h, frames, device="cpu")
    ref, hip = parity.OracleRun(name, w, h), parity.HipRun(name, w, h)
    ref_ex = oracle_driver.RefExecutor(ref.inst, w, h, api.FORMAT_BYTES)  # the reference's text instead of the restatement, over the same plane bindings
    ref_ex.user = ref.ex.user
    ref.ex = ref_ex
    stats = parity.ParityStats()
    for f, frame in enumerate(seq):
        cam, cam_prev = frame["camera"], seq[max(f - 1, 0)]["camera"]
        for run in (ref, hip):
            run.step(frame, parity.common_settings(cam, cam_prev, w, h, f), parity.denoiser_settings(name, frame, None))
        for rt in ref.outs:
            stats.add(rt.name, f, parity.error_stats(hip.output(rt), ref.output(rt)))
    out = _report("gpu_vs_reference_text_8f", name, (w, h), frames, stats)["outputs"]
    assert out["planes"] >= 1 and out["frac_gt_tol_vec"] <= max_frac and out["bit_exact_frac"] >= min_exact, out


# ---- the saturated-history regime at the BASELINE size: the protocol of SURVEY 8d times frames 32..95; held bit for bit here at frames 32..39 of the
# headline configuration. The first 32 frames run on the GPU alone -- the oracle does 1440p at ~0.6 frames / s --, then every plane of the GPU's state (user outputs, which double
# as history, and both pools) is handed to the oracle and the two run frames 32..39 side by side: all outputs and pool planes equal on every frame.
SATURATED = [("REBLUR_DIFFUSE_SPECULAR", 2560, 1440, 32, 8), ("RELAX_DIFFUSE_SPECULAR_SH", 3840, 2160, 32, 2),
             ("SIGMA_SHADOW", 1920, 1080, 32, 6), ("REBLUR_DIFFUSE", 2560, 1440, 32, 6)]  # BASELINE.json configs[1] and [2] (3 + 10 s); four more configurations: tools/parity_saturated.py


@pytest.mark.parametrize("name,width,height,warm,frames", SATURATED, ids=["%s_%dx%d_frames_%d_%d" % (c[0], c[1], c[2], c[3], c[3] + c[4] - 1) for c in SATURATED])
def test_bit_exact_at_baseline_size_with_a_saturated_history(name, width, height, warm, frames):
    worst = parity.run_parity_from_gpu_state(name, width, height, warm, frames)
    assert worst == 0.0, "the library differs from the oracle at %dx%d in frames %d..%d: max rel err %g" % (width, height, warm, warm + frames - 1, worst)


@pytest.mark.parametrize("name", ["REBLUR_DIFFUSE_SPECULAR", "RELAX_DIFFUSE_SPECULAR_SH", "SIGMA_SHADOW"])
def test_state_handover_to_the_oracle_is_exact(name):
    """the mechanism of the test above at a size where the oracle can also walk the whole way: frames 0..9 on the device alone, its state handed over, frames 10..12 side by side
    bit for bit -- and test_bit_exact_over_48_frames holds the same frames with the oracle running from frame 0"""
    worst = parity.run_parity_from_gpu_state(name, 192, 128, 10, 3, device="cpu")
    assert worst == 0.0, worst
