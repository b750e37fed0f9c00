"""Records tests/golden/ref_host_dispatches.json from the reference's OWN host code (oracle/_ref/libnrdhost.so: /root/reference/Source/*.cpp compiled by oracle/ref/host/Makefile), so that the
pinning of the product's dispatch compiler survives on a machine without the reference tree: tests/test_ref_host_golden.py replays the same sequences through the product's host alone.
usage: python tools/make_ref_host_golden.py        (parameters: tests/ref_host_golden.py)"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import ref_host_golden as G  # noqa: E402
from oracle import driver  # noqa: E402

if __name__ == "__main__":
    ref = driver.load_ref_host()
    out = {G.key(name, ov): G.stream(name, ov, lib=ref) for name, cases in G.CASES.items() for ov in cases}
    with open(G.PATH, "w") as fp:
        json.dump(out, fp, indent=0, sort_keys=True)
    print("%s: %d sequences, %d dispatches, %.0f KB" % (os.path.relpath(G.PATH, ROOT), len(out), sum(len(f) for v in out.values() for f in v[1]), os.path.getsize(G.PATH) / 1024.0))
