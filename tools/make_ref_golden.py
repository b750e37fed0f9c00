"""Records tests/golden/ref_text_<DENOISER>.npz: what the REFERENCE'S OWN SHADER TEXT (oracle/_ref: /root/reference/Shaders compiled as C++, oracle/ref/Makefile) writes,
pass by pass, when it is handed the inputs of a short frame sequence driven by the strict oracle -- so that the pinning of the oracle survives on a machine without the
reference tree (the GPU box, a fresh checkout): tests/test_ref_golden.py replays the sequence with the strict oracle alone, checks per dispatch that it is looking at the
same inputs (sha1 of constants + bound planes) and holds its outputs against the recorded ones with the floors of tests/test_ref_parity.py.

usage: python tools/make_ref_golden.py          (needs oracle/_ref/libnrdref.so, i.e. /root/reference; rewrites all fixtures)
The generating parameters live in tests/ref_golden.py (CASES) so that recorder and test cannot disagree."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import ref_golden  # noqa: E402

if __name__ == "__main__":
    for case in ref_golden.CASES:
        path, n, size = ref_golden.record(case)
        print("%s: %d dispatches, %.0f KB" % (os.path.relpath(path, ROOT), n, size / 1024.0))
