"""TEST INFRASTRUCTURE. Runs a frame sequence dispatch by dispatch on the oracle and on the CPU emulation of the device sources and reports, after every
pass, which planes differ. With --isolate (default) the emulation's planes are overwritten with the oracle's before every pass, so each pass is judged
on identical inputs and one faulty pass does not hide the next.
usage: python tests/emu/pass_diff.py DENOISER [W H FRAMES] [--no-isolate] [--settings "dict(...)"] [--cs "dict(...)"]"""
import ctypes as C
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, os.path.join(ROOT, "tests"), HERE):
    if p not in sys.path:
        sys.path.insert(0, p)
import numpy as np  # noqa: E402

import parity  # noqa: E402
from emu_run import EmuRun  # noqa: E402
from raytracingdenoiser_amd import api  # noqa: E402

RT = api.ResourceType


def pool_view(run, pool, k):
    d = api.HipPlaneDesc()
    run.ex._check(run.ex.lib.nrdHipGetPoolPlane(run.ex.handle, int(pool), k, C.byref(d)), "nrdHipGetPoolPlane")
    buf = (C.c_uint8 * (d.height * d.rowPitchBytes)).from_address(d.data)
    return np.frombuffer(buf, dtype=np.uint8).reshape(d.height, d.rowPitchBytes), api.Format(d.format), d.width


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    opts = sys.argv[1:]
    name = args[0] if args else "REBLUR_DIFFUSE_SPECULAR"
    w, h, frames = (int(args[1]), int(args[2]), int(args[3])) if len(args) > 3 else (192, 128, 2)
    isolate = "--no-isolate" not in opts
    overrides = eval(opts[opts.index("--settings") + 1]) if "--settings" in opts else None
    cs_kw = eval(opts[opts.index("--cs") + 1]) if "--cs" in opts else {}
    extra = tuple(eval(opts[opts.index("--want") + 1])) if "--want" in opts else ()
    seq = parity.generate_sequence(name, w, h, frames, extra_want=extra)
    ora, emu = parity.OracleRun(name, w, h), EmuRun(name, w, h)
    bad_passes = {}
    for f, frame in enumerate(seq):
        parity.tag_checkerboard(frame, overrides, f)
        for run in (ora, emu):
            for rt, t, fmt in parity.user_planes(name, frame):
                a = np.array(t.cpu().numpy(), copy=True, order="C")
                run.inputs[rt] = a
                run.ex.bind(rt, a, fmt)
            run.inst.set_denoiser_settings(0, parity.denoiser_settings(name, frame, overrides))
            run.inst.set_common_settings(parity.common_settings(frame["camera"], seq[max(f - 1, 0)]["camera"], w, h, f, **cs_kw))
        r, ods = ora.inst.get_compute_dispatches()
        r2, raw, num = emu.inst.get_compute_dispatches_raw()
        assert num == len(ods)
        pools = [(pool, k) for pool in (RT.PERMANENT_POOL, RT.TRANSIENT_POOL) for k in range(len(ora.inst.permanent_pool if pool == RT.PERMANENT_POOL else ora.inst.transient_pool))]
        for i, d in enumerate(ods):
            if isolate:  # identical inputs for this pass
                for pool, k in pools:
                    o_raw, _, _ = ora.ex.pool_plane(pool, k)
                    pool_view(emu, pool, k)[0][...] = o_raw
                for rt in ora.outs:
                    emu.outs[rt][0][...] = ora.outs[rt][0].view(emu.outs[rt][0].dtype)
                for rt in ora.inputs:
                    emu.inputs[rt][...] = ora.inputs[rt]
            ora.ex.execute([d])
            emu.ex.execute_raw(C.byref(raw[i]), 1)
            if d.shader.startswith("Clear"):
                continue
            report = []
            for pool, k in pools:
                o_raw, fmt, pw = ora.ex.pool_plane(pool, k)
                h_raw = pool_view(emu, pool, k)[0]
                if np.array_equal(o_raw, h_raw):
                    continue
                want, got = parity.decode_plane(o_raw, fmt, pw), parity.decode_plane(h_raw, fmt, pw)
                bad = np.argwhere(np.any(got != want, axis=-1))
                if len(bad):
                    y, x = bad[0]
                    report.append("%s[%d] %s: %d texels, e.g. (x=%d,y=%d) got %s want %s" % (pool.name[:4], k, fmt.name, len(bad), x, y, got[y, x], want[y, x]))
            for rt in ora.outs:
                want, got = ora.output(rt), emu.output(rt)
                bad = np.argwhere(np.any(got != want, axis=-1))
                if len(bad):
                    y, x = bad[0]
                    report.append("%s: %d texels, e.g. (x=%d,y=%d) got %s want %s" % (rt.name, len(bad), x, y, got[y, x], want[y, x]))
            print("frame %d after %-55s %s" % (f, d.shader, "identical" if not report else ""))
            for line in report:
                print("      " + line)
            if report:
                bad_passes[d.shader] = bad_passes.get(d.shader, 0) + 1
    print("passes with differences:", bad_passes or "none")


if __name__ == "__main__":
    main()
