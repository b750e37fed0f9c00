"""One-off source rewriter of round 3 (kept for the record): turns the floating-point divisions `A / B` of the device sources and of the oracle into the
contract's `Div(A, B)` (= A * v_rcp_f32(B)) / `Rcp(B)`, with the C precedence of the multiplicative chain kept (`a * b / c` -> `Div(a * b, c)`).
Left alone: preprocessor lines, comments, divisions whose operands are both literals (compile-time constants, folded identically by every compiler),
divisions by an integer literal or by an ALL_CAPS / integer-looking name (reported for review), divisions by a power-of-two float literal (rewritten
as the exact multiplication). usage: python tools/div_rewrite.py [--apply] file..."""
import math
import re
import sys

IDENT = re.compile(r"[A-Za-z0-9_.:]")
FLOAT_LIT = re.compile(r"^-?(\d+\.\d*|\.\d+|\d+)([eE][-+]?\d+)?f$|^-?\d+\.\d*([eE][-+]?\d+)?$")
INT_LIT = re.compile(r"^\d+[uU]?$")
TYPES = {"float", "int", "uint32_t", "int32_t", "double", "uint", "unsigned"}


def mask(text):
    """same-length copy with comments, strings and preprocessor lines blanked"""
    out = list(text)
    i, n = 0, len(text)
    bol = True
    while i < n:
        c = text[i]
        if text.startswith("//", i):
            j = text.find("\n", i)
            j = n if j < 0 else j
            for k in range(i, j):
                out[k] = " "
            i = j
            continue
        if text.startswith("/*", i):
            j = text.find("*/", i)
            j = n if j < 0 else j + 2
            for k in range(i, j):
                if out[k] != "\n":
                    out[k] = " "
            i = j
            continue
        if c == '"':
            j = i + 1
            while j < n and text[j] != '"':
                j += 2 if text[j] == "\\" else 1
            for k in range(i, min(j + 1, n)):
                out[k] = " "
            i = j + 1
            continue
        if c == "#" and bol:
            j = i
            while True:
                e = text.find("\n", j)
                e = n if e < 0 else e
                if e > 0 and text[e - 1] == "\\":
                    j = e + 1
                    continue
                break
            for k in range(i, e):
                if out[k] != "\n":
                    out[k] = " "
            i = e
            continue
        if c == "\n":
            bol = True
        elif not c.isspace():
            bol = False
        i += 1
    return "".join(out)


def match_fwd(m, i):
    """index just past the bracket group opening at i"""
    pairs = {"(": ")", "[": "]"}
    depth, j = 0, i
    while j < len(m):
        if m[j] in "([":
            depth += 1
        elif m[j] in ")]":
            depth -= 1
            if depth == 0:
                return j + 1
        j += 1
    raise ValueError("unbalanced")


def match_bwd(m, i):
    """index of the bracket that opens the group closing at i"""
    depth, j = 0, i
    while j >= 0:
        if m[j] in ")]":
            depth += 1
        elif m[j] in "([":
            depth -= 1
            if depth == 0:
                return j
        j -= 1
    raise ValueError("unbalanced")


def primary_fwd(m, i):
    """end of the primary expression starting at or after i (unary sign, casts, calls, members, indices)"""
    j = i
    while m[j].isspace():
        j += 1
    if m[j] in "+-":
        j += 1
        while m[j].isspace():
            j += 1
    while True:
        if m[j] == "(":
            e = match_fwd(m, j)
            inner = m[j + 1:e - 1].strip()
            j = e
            if inner in TYPES:  # C cast: the operand follows
                while m[j].isspace():
                    j += 1
                continue
        else:
            if not IDENT.match(m[j]):
                raise ValueError("no primary at %d: %r" % (j, m[j:j + 20]))
            # a float literal may carry an exponent sign
            k = j
            while k < len(m) and (IDENT.match(m[k]) or (m[k] in "+-" and m[k - 1] in "eE" and re.match(r"[\d.]", m[j]))):
                k += 1
            j = k
        # postfix
        while j < len(m) and (m[j] in "([" or (m[j] == "." and IDENT.match(m[j + 1])) or (m[j] == "-" and m[j + 1] == ">")):
            if m[j] in "([":
                j = match_fwd(m, j)
            elif m[j] == ".":
                j += 1
                while IDENT.match(m[j]):
                    j += 1
            else:
                j += 2
                while IDENT.match(m[j]):
                    j += 1
        return j


def primary_bwd(m, i):
    """start of the primary expression ending just before i"""
    j = i - 1
    while m[j].isspace():
        j -= 1
    while True:
        if m[j] in ")]":
            j = match_bwd(m, j) - 1
        elif IDENT.match(m[j]):
            while j >= 0 and (IDENT.match(m[j]) or (m[j] in "+-" and m[j - 1] in "eE" and m[j - 2].isdigit())):
                j -= 1
            if m[j] == ">" and m[j - 1] == "-":  # a->b
                j -= 2
                continue
            return j + 1
        else:
            raise ValueError("no primary before %d: %r" % (i, m[max(0, i - 20):i]))
        # a group: is it preceded by a callee / array name?
        if j >= 0 and IDENT.match(m[j]):
            continue
        if j >= 1 and m[j] == ">" and m[j - 1] != "-":  # template call Foo<...>(...)
            depth, k = 0, j
            while k >= 0:
                if m[k] == ">":
                    depth += 1
                elif m[k] == "<":
                    depth -= 1
                    if depth == 0:
                        break
                k -= 1
            j = k - 1
            continue
        return j + 1


def chain_bwd(m, i):
    """start of the multiplicative chain that ends just before i"""
    s = primary_bwd(m, i)
    while True:
        k = s - 1
        while k >= 0 and m[k].isspace():
            k -= 1
        # a C cast in front of the primary belongs to it
        if k >= 0 and m[k] == ")":
            o = match_bwd(m, k)
            if m[o + 1:k].strip() in TYPES:
                s = o
                continue
        if k >= 0 and m[k] == "*" and m[k - 1] != "/":
            s = primary_bwd(m, k)
            continue
        return s


def pow2_reciprocal(lit):
    v = float(lit.rstrip("f"))
    if v > 0 and math.frexp(v)[0] == 0.5:
        r = 1.0 / v
        return (repr(r) if r != int(r) else "%d.0" % int(r)) + "f"
    return None


def strip_parens(e):
    """(a + b) -> a + b when the parentheses enclose the whole operand (it becomes a function argument)"""
    while e.startswith("(") and match_fwd(e, 0) == len(e) and e[1:-1].strip() not in TYPES:
        e = e[1:-1].strip()
    return e


def rewrite(text, report):
    changed = 0
    pos = 0
    while True:
        m = mask(text)
        i = m.find("/", pos)
        while i >= 0 and (m[i + 1] in "/=*" or m[i - 1] in "/*"):
            i = m.find("/", i + 1)
        if i < 0:
            return text, changed
        try:
            ls, re_ = chain_bwd(m, i), primary_fwd(m, i + 1)
        except (ValueError, IndexError) as e:
            report.append("SKIP (parse) %r: %s" % (text[max(0, i - 40):i + 40], e))
            pos = i + 1
            continue
        left, right = text[ls:i].strip(), text[i + 1:re_].strip()
        line = text.count("\n", 0, i) + 1
        lit_l, lit_r = bool(FLOAT_LIT.match(left)), bool(FLOAT_LIT.match(right))
        if (lit_l or INT_LIT.match(left)) and (lit_r or INT_LIT.match(right)):
            pos = i + 1  # constant expression
            continue
        if INT_LIT.match(right) or re.match(r"^[A-Z][A-Z0-9_]*$", right) or re.match(r"^\(?(unsigned|int|uint32_t)\)", right) or re.search(r"\b(tile[WH]|stripeTiles|TILE_[XY]|BUF_[XY]|tilesPerRow|stride)\b", right):
            report.append("INT? line %d: %s / %s" % (line, left, right))
            pos = i + 1
            continue
        if lit_r and pow2_reciprocal(right):
            new = "%s * %s" % (left, pow2_reciprocal(right))
            left = right = None
        elif left in ("1.0f", "1.0"):
            new = "Rcp(%s)" % strip_parens(right)
        elif left.startswith("sizeof"):
            pos = i + 1
            continue
        else:
            new = "Div(%s, %s)" % (strip_parens(left), strip_parens(right))
        text = text[:ls] + new + text[re_:]
        changed += 1
        pos = ls  # rescan: the new text may be the left operand of a following division


COMPOUND = re.compile(r"^(\s*)([A-Za-z_][\w.\[\]]*) /= (.*);(\s*(//.*)?)$")


def rewrite_compound(text):
    out, n = [], 0
    for line in text.split("\n"):
        m = COMPOUND.match(line)
        if m:
            line = "%s%s = Div(%s, %s);%s" % (m.group(1), m.group(2), m.group(2), m.group(3), m.group(4))
            n += 1
        out.append(line)
    return "\n".join(out), n


def main():
    apply = "--apply" in sys.argv
    for path in [a for a in sys.argv[1:] if not a.startswith("--")]:
        src = open(path).read()
        report = []
        out, n = rewrite(src, report)
        out, n2 = rewrite_compound(out)
        n += n2
        print("%s: %d divisions rewritten" % (path, n))
        for r in report:
            print("   ", r)
        if apply and n:
            open(path, "w").write(out)


if __name__ == "__main__":
    main()
