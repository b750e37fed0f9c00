"""Which source lines / functions own the instructions of a kernel. Developer tooling (static; no GPU).
usage: python tools/isa_by_source.py kernels_X.hip --kernel <substring of the mangled or demangled name> [--by line|func] [--top N] [-D...]
Compiles the file for gfx950 with the flags of the product build plus -gline-tables-only, walks the `.loc` directives of the listing and charges every
VALU instruction (weighted by the measured issue cost: 2.4 / 4.1 / 8.1 SIMD cycles, tools/isa_stats.py) to the innermost source location. `--by func`
maps a location to the function whose body contains the line (a brace-depth scan of the source, good enough for these files)."""
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

_func_cache = {}


def functions_of(path):
    """[(first line, last line, name)] of the function bodies at namespace scope of a C++ file (heuristic)"""
    if path in _func_cache:
        return _func_cache[path]
    out = []
    try:
        lines = open(path, errors="replace").read().split("\n")
    except OSError:
        _func_cache[path] = out
        return out
    depth, start, name = 0, None, None
    for i, l in enumerate(lines, 1):
        code = re.sub(r'"[^"]*"', '""', l.split("//")[0])
        if re.search(r"\b(namespace\b[^{;]*|extern\s*\"\")\s*\{", code) and depth == 0:
            code = code.replace("{", "", 1)  # namespaces are transparent
        elif depth == 0 and code.strip() == "}" and start is None:
            continue  # closes a namespace
        if depth == 0 and "{" in code and start is None:
            text = " ".join(x.split("//")[0] for x in lines[max(0, i - 8):i])
            text = text[text.rfind(";") + 1:] if ";" in text[: text.rfind("{")] else text
            m = [x for x in re.findall(r"([A-Za-z_][A-Za-z_0-9]*)\s*\(", text) if x not in ("if", "for", "while", "switch", "__launch_bounds__", "__attribute__", "defined", "sizeof", "__align__", "alignas")]
            if m:
                start, name = i, m[0]
            elif re.search(r"\b(struct|class|union|enum)\b", text):
                start, name = i, "struct " + (re.findall(r"\b(?:struct|class|union|enum)\s+([A-Za-z_0-9]+)", text) or ["?"])[-1]
        depth += code.count("{") - code.count("}")
        if depth <= 0:
            depth = 0
            if start is not None:
                out.append((start, i, name))
                start = None
    _func_cache[path] = out
    return out


def main():
    args = sys.argv[1:]
    src = args.pop(0)
    kernel, by, top, extra = "", "func", 40, []
    while args:
        a = args.pop(0)
        if a == "--kernel":
            kernel = args.pop(0)
        elif a == "--by":
            by = args.pop(0)
        elif a == "--top":
            top = int(args.pop(0))
        else:
            extra.append(a)
    src = src if os.path.exists(src) else os.path.join(ROOT, "raytracingdenoiser_amd", "csrc", "hip", src)
    out = "/tmp/isa_by_source.s"
    cmd = ["/opt/rocm/bin/hipcc"] + b._flags(src, extra) + ["-gline-tables-only", "-S", "--cuda-device-only", src, "-o", out]
    subprocess.run(cmd, check=True)
    txt = open(out).read()
    files = {}
    for m in re.finditer(r'\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', txt):
        files[int(m.group(1))] = os.path.join(m.group(2), m.group(3)) if m.group(3) else m.group(2)
    starts = [(m.start(), m.group(1)) for m in re.finditer(r"\n(_Z[^\n:]*):[^\n]*\n", txt)]
    names = isa_stats.demangle([n for _, n in starts])
    for i, (pos, name) in enumerate(starts):
        if kernel not in name and kernel not in names.get(name, ""):
            continue
        end = starts[i + 1][0] if i + 1 < len(starts) else len(txt)
        body = txt[pos:end].split(".Lfunc_end")[0]
        loc = (0, 0)
        cost, count = collections.Counter(), collections.Counter()
        for l in body.split("\n"):
            s = l.strip()
            if s.startswith(".loc"):
                p = s.split()
                loc = (int(p[1]), int(p[2]))
                continue
            if not l.startswith("\t") or s.startswith((".", ";")):
                continue
            ins = s.split()[0]
            path = files.get(loc[0], "?")
            if by == "line":
                key = "%s:%d" % (os.path.basename(path), loc[1])
            else:
                key = os.path.basename(path) + ":?"
                for a, z, fn in functions_of(path):
                    if a <= loc[1] <= z:
                        key = "%s:%s" % (os.path.basename(path), fn)
                        break
            if ins.startswith("v_"):
                base = ins.replace("_e32", "").replace("_e64", "").replace("_sdwa", "").replace("_dpp", "")
                cost[key] += 8.1 if base.startswith(isa_stats.TRANS) else 2.4 if base in isa_stats.FULL_RATE else 4.1
                count[key] += 1
            elif ins.startswith(("global_", "ds_", "buffer_", "scratch_")):
                count[key + " [mem]"] += 1
        total = sum(cost.values())
        print("%s\n  VALU %d instructions, %.0f SIMD cycles per wave (static)" % (names.get(name, name)[:160], sum(v for k, v in count.items() if not k.endswith("[mem]")), total))
        for key, c in cost.most_common(top):
            print("  %7.0f cyc %5.1f%%  %5d instr   %s" % (c, 100.0 * c / total, count[key], key))
        mem = [(k, v) for k, v in count.items() if k.endswith("[mem]")]
        print("  memory instructions: " + ", ".join("%s %d" % (k.replace(" [mem]", ""), v) for k, v in sorted(mem, key=lambda x: -x[1])[:12]))


if __name__ == "__main__":
    main()
