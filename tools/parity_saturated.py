"""One-off check (GPU box): bit-exactness in the saturated-history regime at BASELINE.json's sizes for the configurations the suite does not hold there -- the GPU runs `warm` frames
alone, its whole state (outputs, both pools) is handed to the oracle, then both run side by side (tests/parity.py run_parity_from_gpu_state; the suite has REBLUR_DIFFUSE_SPECULAR 1440p
and RELAX_DIFFUSE_SPECULAR_SH 4K: tests/test_deep_parity.py).  usage: python tools/parity_saturated.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import parity  # noqa: E402

CASES = [("SIGMA_SHADOW", 1920, 1080, 32, 6), ("REBLUR_DIFFUSE", 2560, 1440, 32, 6), ("RELAX_DIFFUSE_SPECULAR", 3840, 2160, 32, 2), ("REBLUR_DIFFUSE_SPECULAR_SH", 2560, 1440, 32, 3),
         ("REBLUR_DIFFUSE_SPECULAR_OCCLUSION", 2560, 1440, 32, 4), ("SIGMA_SHADOW_TRANSLUCENCY", 1920, 1080, 32, 4)]
bad = 0
for name, w, h, warm, frames in CASES:
    t0 = time.time()
    worst = parity.run_parity_from_gpu_state(name, w, h, warm, frames)
    print("%s %dx%d frames %d..%d: max rel err over every output and pool plane %g (%.0f s)" % (name, w, h, warm, warm + frames - 1, worst, time.time() - t0), flush=True)
    bad += worst != 0.0
sys.exit(1 if bad else 0)
