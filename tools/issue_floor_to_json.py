"""Folds the kernel traces of one "floor" session (tools/session_r04c.sh: bench.py --uniform with the product, with the L1-resident A/B build and with the
arithmetic-only A/B build) into profiles/issue_floor.json: per kernel the measured time, the time with every load an L1 hit, and their ratio.
usage: python tools/issue_floor_to_json.py TAG        (reads profiles/<TAG>_<workload>_uniform_{product,l1,alu}_kernel_stats.txt)
What the numbers mean: --uniform makes every input plane constant, so the redirected loads of the L1-resident build return the values the
real loads return: same control flow, same arithmetic, same instruction stream plus two address instructions per load. Its time is the kernel's floor of VALU issue +
memory-instruction issue + L1-hit latency; measured / floor - 1 is what the memory system beyond the L1 still costs. The arithmetic-only build (every load from
texel (0, 0), so the compiler keeps one load per plane) is a LOWER bound only: identical taps collapse into one."""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def parse(path):
    out = {}
    for line in open(path):
        m = re.match(r"(\S.*?)\s+(\d+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)%", line)
        if m:
            name = re.sub(r"nrdhip::|\(anonymous namespace\)::|void ", "", m.group(1))
            out[re.sub(r"\(.*", "", name)] = float(m.group(3))
    return out


def main():
    tag = sys.argv[1]
    digest_path = os.path.join(ROOT, "raytracingdenoiser_amd", "lib", "libNRD_hip.so.digest")
    data = {"source": "profiles/%s_*_uniform_{product,l1[,alu]}_kernel_stats.txt (rocprofv3 --kernel-trace of bench.py --uniform --no-graph, every pixel denoised)" % tag,
            "library_digest": open(digest_path).read().strip() if os.path.exists(digest_path) else None,  # (the product library of the session: bench.py marks the floors stale when it times another)
            "workloads": {}}
    for w, size in (("reblur_ds", "REBLUR_DIFFUSE_SPECULAR 2560x1440"), ("relax_ds_sh", "RELAX_DIFFUSE_SPECULAR_SH 3840x2160")):
        t = {}
        for lib in ("product", "l1", "alu"):
            path = os.path.join(ROOT, "profiles", "%s_%s_uniform_%s_kernel_stats.txt" % (tag, w, lib))
            t[lib] = parse(path) if os.path.exists(path) else {}  # (the arithmetic-only build is optional)
        rows = {}
        for k, us in sorted(t["product"].items(), key=lambda kv: -kv[1]):
            if k in t["l1"] and not k.startswith(("CopyProbe", "ClearPlane")):
                rows[k] = {"measured_us": us, "l1_resident_us": t["l1"][k], "arithmetic_only_us": t["alu"].get(k), "measured_over_floor": round(us / t["l1"][k], 3)}
        data["workloads"][size] = rows
    json.dump(data, open(os.path.join(ROOT, "profiles", "issue_floor.json"), "w"), indent=1)
    for size, rows in data["workloads"].items():
        print(size)
        for k, r in rows.items():
            print("  %-78s %8.1f %8.1f  x%.2f" % (k[:78], r["measured_us"], r["l1_resident_us"], r["measured_over_floor"]))


if __name__ == "__main__":
    main()
