"""Debug aid: execute a frame sequence dispatch by dispatch on the oracle and on the GPU and report, after every pass,
which planes differ (test infrastructure; run on the GPU box: python tools/pass_diff.py REBLUR_DIFFUSE_SPECULAR 192 128 2)."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch

import parity
from raytracingdenoiser_amd import api

RT = api.ResourceType


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "REBLUR_DIFFUSE_SPECULAR"
    w, h, frames = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    overrides = eval(sys.argv[5]) if len(sys.argv) > 5 else None
    seq = parity.generate_sequence(name, w, h, frames)
    ora, hip = parity.OracleRun(name, w, h), parity.HipRun(name, w, h)
    for f, frame in enumerate(seq):
        cs = parity.common_settings(frame["camera"], seq[max(f - 1, 0)]["camera"], w, h, f)
        for run in (ora, hip):
            for rt, t, fmt in parity.user_planes(name, frame):
                run.ex.bind(rt, t.cuda().contiguous() if run is hip else np.ascontiguousarray(t.cpu().numpy()), fmt)
            run.inst.set_denoiser_settings(0, parity.denoiser_settings(name, frame, overrides))
            run.inst.set_common_settings(parity.common_settings(frame["camera"], seq[max(f - 1, 0)]["camera"], w, h, f))
        r, ods = ora.inst.get_compute_dispatches()
        r2, raw, num = hip.inst.get_compute_dispatches_raw()
        assert num == len(ods)
        for i, d in enumerate(ods):
            ora.ex.execute([d])
            hip.ex.execute_raw(C.byref(raw[i]), 1)
            if d.shader.startswith("Clear"):
                continue
            report = []
            for pool in (RT.PERMANENT_POOL, RT.TRANSIENT_POOL):
                descs = ora.inst.permanent_pool if pool == RT.PERMANENT_POOL else ora.inst.transient_pool
                for k in range(len(descs)):
                    o_raw, fmt, pw = ora.ex.pool_plane(pool, k)
                    h_raw, _, _ = hip.ex.read_pool_plane(pool, k)
                    want, got = parity.decode_plane(o_raw, fmt, pw), parity.decode_plane(h_raw, fmt, pw)
                    bad = np.argwhere(np.any(got != want, axis=-1))
                    if len(bad):
                        y, x = bad[0]
                        report.append("%s[%d] %s: %d texels, e.g. (x=%d,y=%d) got %s want %s" % (pool.name[:4], k, fmt.name, len(bad), x, y, got[y, x], want[y, x]))
            for rt in ora.outs:
                want, got = ora.output(rt), hip.output(rt)
                bad = np.argwhere(np.any(got != want, axis=-1))
                if len(bad):
                    y, x = bad[0]
                    report.append("%s: %d texels, e.g. (x=%d,y=%d) got %s want %s" % (rt.name, len(bad), x, y, got[y, x], want[y, x]))
            print("frame %d after %-50s %s" % (f, d.shader, "identical" if not report else ""))
            for line in report:
                print("      " + line)


if __name__ == "__main__":
    main()
