"""Which lines of a KERNEL BODY own its instructions: every VALU instruction is charged to the outermost frame of its inline stack (the line of the kernel / tile
function that -- through however many inlined helpers -- caused it), weighted by the measured issue cost (tools/isa_stats.py: 2.4 / 4.1 / 8.1 SIMD cycles).
tools/isa_by_source.py charges the INNERMOST location instead (which helper), this one answers "which statement of the pass is expensive". Static; no GPU.
usage: python tools/isa_by_root_line.py kernels_X.hip --kernel <substring of the mangled name> [--top N] [--depth D] [-D...]
  --under L: only the instructions caused by line L of the kernel's file, charged one frame further in (which helper line under that statement)
  --depth D: charge to the D-th frame from the outside (0 = the __global__ function, 1 = the function it inlined first, ...; default: the deepest frame that
             still lies in the kernel's own .hip file)"""
import collections
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from raytracingdenoiser_amd import build as b  # noqa: E402
import isa_stats  # noqa: E402

LLVM = "/opt/rocm/lib/llvm/bin/"


def main():
    args = sys.argv[1:]
    src = args.pop(0)
    kernel, top, depth, under, extra = "", 50, None, None, []
    while args:
        a = args.pop(0)
        if a == "--kernel":
            kernel = args.pop(0)
        elif a == "--top":
            top = int(args.pop(0))
        elif a == "--depth":
            depth = int(args.pop(0))
        elif a == "--under":
            under = args.pop(0)
        else:
            extra.append(a)
    src = src if os.path.exists(src) else os.path.join(ROOT, "raytracingdenoiser_amd", "csrc", "hip", src)
    obj, co = "/tmp/isa_root.o", "/tmp/isa_root.co"
    subprocess.run(["/opt/rocm/bin/hipcc"] + b._flags(src, extra) + ["-g", "-c", "--cuda-device-only", src, "-o", obj], check=True)
    subprocess.run([LLVM + "clang-offload-bundler", "--unbundle", "--type=o", "--input=" + obj, "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co], check=True)
    dis = subprocess.run([LLVM + "llvm-objdump", "-d", "--no-show-raw-insn", co], capture_output=True, text=True, check=True).stdout
    blocks = re.split(r"\n(?=[0-9a-f]{16} <)", dis)
    for blk in blocks:
        head = blk.split("\n", 1)[0]
        m = re.match(r"([0-9a-f]{16}) <([^>]*)>:", head)
        if not m or kernel not in m.group(2) or not m.group(2).startswith("_Z"):
            continue
        ins = []
        for l in blk.split("\n")[1:]:
            mm = re.match(r"\s+(\S+)\s.*//\s*([0-9A-Fa-f]+):", l) or re.match(r"\s+(\S+).*//\s*([0-9A-Fa-f]+):", l)
            if mm:
                ins.append((int(mm.group(2), 16), mm.group(1)))
        if not ins:
            continue
        q = "\n".join("0x%x" % a for a, _ in ins) + "\n"
        sym = subprocess.run([LLVM + "llvm-symbolizer", "--inlines", "--obj=" + co, "--output-style=LLVM", "--functions=short"], input=q, capture_output=True, text=True, check=True).stdout
        stacks = [s for s in sym.strip().split("\n\n")]
        assert len(stacks) == len(ins), (len(stacks), len(ins))
        cost, count, mem = collections.Counter(), collections.Counter(), collections.Counter()
        base_src = os.path.basename(src)
        for (addr, name), st in zip(ins, stacks):
            lines = st.strip().split("\n")
            frames = [(lines[i], lines[i + 1]) for i in range(0, len(lines) - 1, 2)]  # innermost first: (function, file:line:col)
            frames.reverse()
            if depth is not None:
                f = frames[min(depth, len(frames) - 1)]
            else:
                own = [k for k, fr in enumerate(frames) if os.path.basename(fr[1].rsplit(":", 2)[0]) == base_src]
                f = frames[own[-1]] if own else frames[0]
                if under is not None:
                    if not own or f[1].rsplit(":", 2)[1] != under:
                        continue
                    f = frames[min(own[-1] + 1, len(frames) - 1)]
            path, line = f[1].rsplit(":", 2)[0], f[1].rsplit(":", 2)[1]
            key = "%s:%s" % (os.path.basename(path), line)
            if name.startswith("v_"):
                base = name.replace("_e32", "").replace("_e64", "").replace("_sdwa", "").replace("_dpp", "")
                cost[key] += 8.1 if base.startswith(isa_stats.TRANS) else 2.4 if base in isa_stats.FULL_RATE else 4.1
                count[key] += 1
            elif name.startswith(("global_", "ds_", "buffer_", "scratch_")):
                mem[key] += 1
        total = sum(cost.values())
        text = {}
        print("%s\n  VALU %d instructions, %.0f SIMD cycles per wave (static)" % (m.group(2)[:150], sum(count.values()), total))
        for key, c in cost.most_common(top):
            p, ln = key.rsplit(":", 1)
            full = src if p == base_src else os.path.join(os.path.dirname(src), p)
            if full not in text:
                try:
                    text[full] = open(full, errors="replace").read().split("\n")
                except OSError:
                    text[full] = []
            t = text[full][int(ln) - 1].strip()[:110] if 0 < int(ln) <= len(text[full]) else ""
            print("  %7.0f cyc %5.1f%%  %5d instr %3d mem  %-28s %s" % (c, 100.0 * c / total, count[key], mem.get(key, 0), key, t))


if __name__ == "__main__":
    main()
