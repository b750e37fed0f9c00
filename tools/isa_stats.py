"""Static instruction statistics per kernel of a `hipcc -S --cuda-device-only` listing (optionally next to a second listing).
usage: python tools/isa_stats.py a.s [b.s] [--filter substr] [--top N]
Columns: VALU / SALU / VMEM / LDS instruction counts, VGPRs, occupancy (waves per SIMD), scratch bytes, LDS bytes.
--cost adds a static cost estimate from the measured gfx950 price lists (profiles/r02_b_valu_bench.txt, r02_c_gather_bench.txt): SIMD cycles of VALU issue per wave,
CU cycles of the L1 request path per wave and (conflict-free) LDS-array cycles per wave, and the time both bounds give for --pixels P (default 2560x1440) on 256 CUs at 2.4 GHz. Static = every
instruction of the kernel counted once (loops once, both sides of branches), so it is an upper bound for straight-line kernels and a rough guide otherwise.
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


FULL_RATE = ("v_fma_f32", "v_fmac_f32", "v_mul_f32", "v_add_f32", "v_sub_f32", "v_subrev_f32", "v_mov_b32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_add_u32", "v_sub_u32", "v_subrev_u32",
             "v_fmaak_f32", "v_fmamk_f32", "v_mad_f32", "v_mac_f32", "v_add_co_u32", "v_not_b32")
TRANS = ("v_rcp", "v_sqrt", "v_rsq", "v_exp", "v_log", "v_sin", "v_cos")
LOAD_COST = {"dwordx4": 39.6, "dwordx3": 30.0, "dwordx2": 18.4, "dword": 6.4, "ushort": 6.4, "ubyte": 6.4, "sbyte": 6.4, "sshort": 6.4, "short": 6.4, "byte": 6.4}


# LDS-array cycles per wave64 instruction, conflict-free (MI355X_MICROARCH.md "LDS": banking and cycles are per instruction). A dword read at a 16-byte lane
# stride (one component of a float4 texel) is a 4-way conflict on top: the pattern that bound RELAX HistoryClamping in round 3.
LDS_COST = {"ds_read_b32": 2, "ds_read_u8": 2, "ds_read_u16": 2, "ds_read_i8": 2, "ds_read_i16": 2, "ds_read_b64": 2, "ds_read_b96": 8, "ds_read_b128": 4, "ds_read2_b32": 4, "ds_read2st64_b32": 4,
            "ds_read2_b64": 8, "ds_read2st64_b64": 8, "ds_write_b8": 4, "ds_write_b16": 4, "ds_write_b32": 4, "ds_write_b64": 6, "ds_write2_b32": 6, "ds_write_b96": 10, "ds_write_b128": 13,
            "ds_write2_b64": 13, "ds_bpermute_b32": 2, "ds_swizzle_b32": 2}


def lds_cycles(counter):
    return sum(n * LDS_COST.get(ins, 4) for ins, n in counter.items() if ins.startswith("ds_"))


def cost(counter):
    """(VALU SIMD cycles per wave, L1 CU cycles per wave) by the price lists: full-rate ops 2.4, transcendentals 8.1, everything else on the VALU 4.1"""
    valu = l1 = 0.0
    for ins, n in counter.items():
        if ins.startswith("v_"):
            base = ins.replace("_e32", "").replace("_e64", "").replace("_sdwa", "").replace("_dpp", "")
            valu += n * (8.1 if base.startswith(TRANS) else 2.4 if base in FULL_RATE else 4.1)
        elif ins.startswith(("global_load_", "global_store_", "buffer_load_", "buffer_store_", "flat_load_", "flat_store_", "scratch_")):
            l1 += n * LOAD_COST.get(ins.rsplit("_", 1)[-1], 6.4)
    return valu, l1


def main():
    args = [a for a in sys.argv[1:]]
    flt, top = "", 0
    want_cost = "--cost" in args
    if want_cost:
        args.remove("--cost")
    pixels = 2560 * 1440
    if "--pixels" in args:
        i = args.index("--pixels")
        pixels = int(args[i + 1])
        del args[i:i + 2]
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
        if want_cost:
            valu, l1 = cost(s["counter"])
            waves = pixels / 64.0
            # 1024 SIMDs issue VALU, 256 CUs serve L1 requests
            lds = lds_cycles(s["counter"])
            print("    static cost per wave: VALU %.0f SIMD cycles, L1 path %.0f CU cycles, LDS array %.0f CU cycles  ->  %d px: VALU bound %.3f ms, L1 bound %.3f ms, LDS bound %.3f ms" % (
                valu, l1, lds, pixels, waves * valu / 1024 / 2.4e6, waves * l1 / 256 / 2.4e6, waves * lds / 256 / 2.4e6))
        if top:
            print("    " + "  ".join("%s %d" % kv for kv in s["counter"].most_common(top)))


if __name__ == "__main__":
    main()
