"""TEST INFRASTRUCTURE (debugging aid). Instruments a line range of a source file with TRACE lines (tests/emu/trace.h) after every scalar / vector
declaration or assignment, so that the device sources (in the CPU emulation) and the oracle print their intermediates at one pixel under the names
they share; the first name whose bits differ is where the two restatements of the HLSL part ways.
  python tests/emu/autotrace.py instrument FILE START END      (keeps FILE.untraced; the file must call TRACE_AT(px, py) itself -- add it by hand)
  python tests/emu/autotrace.py restore FILE
  python tests/emu/autotrace.py compare LOG                    (LOG = stderr of a run: first differing tag between the sides "dev" and "ora")"""
import os
import re
import shutil
import sys

DECL = re.compile(r"^\s*(?:const\s+)?(float|float2|float3|float4|int|bool|uint32_t)\s+(.*);\s*(?://.*)?$")
ASSIGN = re.compile(r"^\s*([A-Za-z_]\w*(?:\.[xyzw])?)\s*(?:[-+*]?=)\s*[^=].*;\s*(?://.*)?$")
MACRO = {"float": "TRACE", "int": "TRACE", "bool": "TRACE", "uint32_t": "TRACE", "float2": "TRACE2", "float3": "TRACE3", "float4": "TRACE4"}


def split_top(s):
    out, depth, cur = [], 0, ""
    for ch in s:
        if ch in "([{":
            depth += 1
        elif ch in ")]}":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur)
            cur = ""
        else:
            cur += ch
    out.append(cur)
    return out


def instrument(path, start, end):
    if not os.path.exists(path + ".untraced"):
        shutil.copy(path, path + ".untraced")
    lines = open(path).read().split("\n")
    types = {}
    out = []
    for i, line in enumerate(lines, 1):
        out.append(line)
        if not (start <= i <= end):
            continue
        prev = next((l.strip() for l in reversed(lines[: i - 1]) if l.strip()), "")
        if (prev.endswith(")") and not prev.endswith(");")) or prev.endswith("else") or prev.startswith("#pragma") or prev.startswith("for "):
            continue  # body of an un-braced control statement
        nxt = lines[i].strip() if i < len(lines) else ""
        if nxt.startswith("else"):
            continue
        indent = re.match(r"\s*", line).group(0)
        m = DECL.match(line)
        if m:
            typ = m.group(1)
            for part in split_top(m.group(2)):
                mm = re.match(r"\s*([A-Za-z_]\w*)\s*(=|$)", part)
                if mm:
                    types[mm.group(1)] = typ
                    if mm.group(2) == "=":
                        out.append('%s%s("%s", %s);' % (indent, MACRO[typ], mm.group(1), mm.group(1)))
            continue
        m = ASSIGN.match(line)
        if m and not line.strip().startswith(("return", "if", "for", "while")):
            name = m.group(1)
            base = name.split(".")[0]
            if base in types:
                macro = "TRACE" if "." in name else MACRO[types[base]]
                out.append('%s%s("%s", %s);' % (indent, macro, name, name))
    open(path, "w").write("\n".join(out))


def compare(log):
    vals = {"dev": {}, "ora": {}}
    order = []
    for line in open(log):
        if not line.startswith("TRACE "):
            continue
        _, side, tag, bits, val = line.split()
        d = vals[side].setdefault(tag, [])
        d.append((bits, val))
        if side == "ora" and tag not in order:
            order.append(tag)
    for tag in order:
        a, b = vals["dev"].get(tag), vals["ora"].get(tag)
        if a is None:
            continue
        n = min(len(a), len(b))
        for k in range(n):
            if a[k][0] != b[k][0]:
                print("DIFF %-36s #%d dev %s %s   ora %s %s" % (tag, k, a[k][0], a[k][1], b[k][0], b[k][1]))
                break
        else:
            if len(a) != len(b):
                print("count %-35s dev %d ora %d" % (tag, len(a), len(b)))
    print("tags: dev %d, ora %d, common %d" % (len(vals["dev"]), len(vals["ora"]), len(set(vals["dev"]) & set(vals["ora"]))))


if __name__ == "__main__":
    if sys.argv[1] == "instrument":
        instrument(sys.argv[2], int(sys.argv[3]), int(sys.argv[4]))
    elif sys.argv[1] == "restore":
        shutil.move(sys.argv[2] + ".untraced", sys.argv[2])
    else:
        compare(sys.argv[2])
