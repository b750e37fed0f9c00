"""Instruction mix of the kernels in a hipcc -S --cuda-device-only listing. usage: python tools/isa_count.py file.s [name filter]"""
import collections
import re
import sys

txt = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
starts = [(m.start(), m.group(1)) for m in re.finditer(r"\n(_Z[^\n:]*):[^\n]*\n", txt)]
for i, (pos, name) in enumerate(starts):
    if flt not in name:
        continue
    body = txt[pos:starts[i + 1][0] if i + 1 < len(starts) else len(txt)].split(".Lfunc_end")[0]
    ins = [l.strip().split()[0] for l in body.split("\n") if l.startswith("\t") and not l.strip().startswith((".", ";"))]
    c = collections.Counter(ins)
    grp = lambda p: sum(v for k, v in c.items() if k.startswith(p))
    print("%s\n  total %d  valu %d  salu %d  vmem %d  lds %d  scratch %d" % (name[:110], len(ins), grp("v_"), grp("s_"), grp("global_") + grp("buffer_"), grp("ds_"), grp("scratch_")))
    print("  " + "  ".join("%s %d" % (k.replace("v_", "").replace("_f32", ""), c.get(k, 0)) for k in
                         ("v_div_scale_f32", "v_div_fmas_f32", "v_div_fixup_f32", "v_rcp_f32", "v_sqrt_f32", "v_rsq_f32", "v_cndmask_b32", "v_fma_f32", "v_mul_f32", "v_add_f32", "v_cvt_f32_f16")))
