"""One-off check beyond BASELINE.json's sizes: GPU == oracle, bit for bit, at 7680 x 4320 (129 600 tiles of 32 x 8 pixels: more than 16 bits of tile indices, x beyond 4096).
usage (GPU box): python tools/parity_8k.py [denoiser ...]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import parity  # noqa: E402

names = sys.argv[1:] or ["SIGMA_SHADOW", "REBLUR_DIFFUSE_SPECULAR", "RELAX_DIFFUSE_SPECULAR_SH"]
bad = 0
for name in names:
    t0 = time.time()
    worst = parity.run_parity(name, 7680, 4320, 2, device="cuda")
    print("%s 7680x4320 x2: max rel err %g (%.0f s)" % (name, worst, time.time() - t0), flush=True)
    bad += worst != 0.0
sys.exit(1 if bad else 0)
