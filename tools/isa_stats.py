"""Static instruction statistics per kernel of a `hipcc -S --cuda-device-only` listing (optionally next to a second listing).
usage: python tools/isa_stats.py a.s [b.s] [--filter substr] [--top N]
Columns: VALU / SALU / VMEM / LDS instruction counts, VGPRs, occupancy (waves per SIMD), scratch bytes, LDS bytes.
The tap loops of the pass kernels are fully unrolled, so the static VALU count of the hot path is close to the dynamic one."""
import collections
import re
import subprocess
import sys


def demangle(names):
    try:
        out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
        return dict(zip(names, out))
    except OSError:
        return {n: n for n in names}


def parse(path):
    txt = open(path).read()
    starts = [(m.start(), m.group(1)) for m in re.finditer(r"\n(_Z[^\n:]*):[^\n]*\n", txt)]
    res = {}
    for i, (pos, name) in enumerate(starts):
        end = starts[i + 1][0] if i + 1 < len(starts) else len(txt)
        chunk = txt[pos:end]
        body = chunk.split(".Lfunc_end")[0]
        ins = [l.strip().split()[0] for l in body.split("\n") if l.startswith("\t") and not l.strip().startswith((".", ";"))]
        c = collections.Counter(ins)
        grp = lambda *ps: sum(v for k, v in c.items() if k.startswith(ps))
        meta = {k: int(v) for k, v in re.findall(r"; (NumVgprs|Occupancy|ScratchSize|LDSByteSize): (\d+)", chunk)}
        res[name] = dict(total=len(ins), valu=grp("v_"), salu=grp("s_"), vmem=grp("global_", "buffer_", "flat_"), lds=grp("ds_"), pk=grp("v_pk_"), trans=grp("v_rcp", "v_sqrt", "v_rsq", "v_exp", "v_log"),
                         vgpr=meta.get("NumVgprs", -1), occ=meta.get("Occupancy", -1), scratch=meta.get("ScratchSize", -1), ldsb=meta.get("LDSByteSize", -1), counter=c)
    return res


def main():
    args = [a for a in sys.argv[1:]]
    flt, top = "", 0
    if "--filter" in args:
        i = args.index("--filter")
        flt = args[i + 1]
        del args[i:i + 2]
    if "--top" in args:
        i = args.index("--top")
        top = int(args[i + 1])
        del args[i:i + 2]
    a = parse(args[0])
    b = parse(args[1]) if len(args) > 1 else None
    names = demangle(list(a))
    for name, s in a.items():
        d = names[name]
        if flt not in d and flt not in name:
            continue
        short = re.sub(r"\(.*", "", d.replace("(nrdhip::SpatialMode)", "MODE").replace("nrdhip::", "").replace("(anonymous namespace)::", "").replace("void ", ""))
        line = "%-90s valu %5d (pk %4d, trans %3d) salu %4d vmem %3d lds %3d | vgpr %3d occ %d scratch %d lds %dB" % (short[:90], s["valu"], s["pk"], s["trans"], s["salu"], s["vmem"], s["lds"], s["vgpr"], s["occ"], s["scratch"], s["ldsb"])
        if b and name in b:
            t = b[name]
            line += "  ||  valu %5d (%.2fx) vgpr %3d occ %d scratch %d" % (t["valu"], t["valu"] / max(s["valu"], 1), t["vgpr"], t["occ"], t["scratch"])
        print(line)
        if top:
            print("    " + "  ".join("%s %d" % kv for kv in s["counter"].most_common(top)))


if __name__ == "__main__":
    main()
