"""Developer tool (TEST INFRASTRUCTURE): where do the hand-written oracle and the reference's own shader text part ways at ONE texel of ONE pass?

Instruments out-of-tree copies of both -- the generated C++ of the reference shader (oracle/_ref/gen/<shader>.cpp) and a line range of the oracle source --
with a TRACE line after every scalar / vector declaration and assignment (tests/emu/autotrace.py, tests/emu/trace.h), builds them under /tmp/ref_trace, runs the
frame sequence with every pass on identical inputs (oracle.driver.ComparingExecutor) and prints the named intermediates of the selected frame whose bits
differ, in program order. The oracle keeps the identifiers of the HLSL it restates, so the first differing name IS the expression (or the threshold) that
separates the two.

usage: python tools/ref_trace.py DENOISER FRAMES FRAME X Y SHADER_SUBSTRING ORACLE_FILE:START:END [--contract | --device] [--head N]
  (default: the strict-IEEE oracle build; --contract: contraction on + a * rcp(b), IEEE transcendentals; --device: the arithmetic of the HIP library)
  e.g. python tools/ref_trace.py REBLUR_DIFFUSE_OCCLUSION 3 1 40 42 HistoryFix oracle/reblur_passes.cpp:1294:1515
  options of the run through the environment: NRD_TRACE_OVERRIDES='{"minMaterialForDiffuse": 0}' NRD_TRACE_CS='{"strandMaterialID": 1}' NRD_TRACE_WANT=materials
"""
import glob
import os
import re
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "emu"))
TMP = "/tmp/ref_trace"
CXX = "/opt/rocm/lib/llvm/bin/clang++"
TRACE_H = os.path.join(ROOT, "tests", "emu", "trace.h")


def build_oracle(spec, contract):
    import autotrace

    path, start, end = spec.split(":")
    dst = os.path.join(TMP, "oracle")
    shutil.rmtree(dst, ignore_errors=True)
    os.makedirs(dst)
    for f in glob.glob(os.path.join(ROOT, "oracle", "*")):
        if os.path.isfile(f) and f.endswith((".cpp", ".h", "Makefile")):
            shutil.copy(f, dst)
    target = os.path.join(dst, os.path.basename(path))
    autotrace.instrument(target, int(start), int(end))
    lines = open(target).read().split("\n")
    out = []
    for i, line in enumerate(lines):
        out.append(line)
        if re.match(r"\s*for \(int px\b.*\{\s*$", line):
            out.append("TRACE_AT(px, py);")
    open(target, "w").write("\n".join(out))
    lib = "liboracle.so" if contract else "liboracle_strict.so"
    subprocess.run(["make", "-C", dst, "-s", "-j8", lib, "EXTRA=-include %s -DNRD_TRACE_SIDE=\\\"ora\\\"" % TRACE_H], check=True)
    return os.path.join(dst, lib)


def build_ref(shader):
    import autotrace

    gen = os.path.join(ROOT, "oracle", "_ref", os.environ.get("NRD_TRACE_GEN", "gen"))  # NRD_TRACE_GEN=gen_vo: the viewport-offset build (oracle/ref/Makefile "vo")
    src = os.path.join(gen, shader[:-3] + ".cpp")  # "<name>.cs" -> "<name>.cpp"
    dst = os.path.join(TMP, "ref_traced.cpp")
    text = open(src).read().split("\n")
    # instrument from the first line of the pass's own text (the last #include of the entry file) to the end of the entry point
    first = max(i for i, l in enumerate(text) if l.startswith("# ") and "/Shaders/Include/" in l and not re.search(r"/(NRD|Common|Poisson|\w+_Config|\w+_Common)\.hlsli", l) and l.rstrip().endswith(' 1'))
    last = max(i for i, l in enumerate(text) if l.startswith("static void hlsl_thunk"))
    open(dst, "w").write("\n".join(text))
    autotrace.DECL = re.compile(r"^\s*(?:const\s+)?(float|float2|float3|float4|int|bool|uint)\s+(.*);\s*(?://.*)?$")
    autotrace.MACRO["uint"] = "TRACE"
    autotrace.instrument(dst, first + 1, last)
    lines = open(dst).read().split("\n")
    out = ['#define NRD_TRACE_SIDE "ref"', '#include "%s"' % TRACE_H]
    in_main = False
    for line in lines:
        out.append(line)
        if re.search(r"const int2 pixelPos = [^;]*;\s*$", line):
            out.append("TRACE_AT(pixelPos.x, pixelPos.y);")
            in_main = True
        elif in_main and "GroupMemoryBarrierWithGroupSync" in line and line.rstrip().endswith(";"):
            out.append("TRACE_AT(pixelPos.x, pixelPos.y);")  # the other threads of the group ran in between (fibers): select the texel again
    open(dst, "w").write("\n".join(out))
    obj = os.path.join(TMP, "ref_traced.o")
    subprocess.run([CXX, "-std=c++17", "-O1", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-fvisibility=hidden", "-I", os.path.join(ROOT, "oracle", "ref"), "-Wno-gnu-anonymous-struct",
                    "-Wno-nested-anon-types", "-Wno-constant-logical-operand", "-Wno-unused-value", "-c", dst, "-o", obj], check=True)
    others = [o for o in glob.glob(os.path.join(gen, "*.o")) if os.path.basename(o) != shader[:-3] + ".o"]
    if not any(os.path.basename(o) == "hlsl_rt.o" for o in others):
        others.append(os.path.join(ROOT, "oracle", "_ref", "gen", "hlsl_rt.o"))
    lib = os.path.join(TMP, "libnrdref_trace.so")
    subprocess.run([CXX, "-shared", "-fopenmp", "-Wl,-rpath,/opt/rocm/lib/llvm/lib", "-o", lib, obj] + others, check=True)
    return lib


def find_shader(name, substring):
    import parity
    from raytracingdenoiser_amd import api

    inst = api.Instance([(0, parity.DENOISERS[name][0])])
    seq = parity.generate_sequence(name, 64, 32, 2, device="cpu")
    for f in (0, 1):
        inst.set_denoiser_settings(0, parity.denoiser_settings(name, seq[f], None))
        inst.set_common_settings(parity.common_settings(seq[f]["camera"], seq[0]["camera"], 64, 32, f))
        _, ds = inst.get_compute_dispatches()
    hits = sorted({d.shader for d in ds if substring in d.shader})
    if len(hits) != 1:
        raise SystemExit("shader substring '%s' matches %s" % (substring, hits))
    return hits[0]


def run(name, frames, frame, x, y, shader, lib_oracle, lib_ref, contract, log, device=False):
    code = r'''
import os, sys
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "tests"))
import ctypes as C
import parity
from oracle import driver
from raytracingdenoiser_amd import api
driver.REF_LIB_PATH = %(lib_ref)r
traced = C.CDLL(%(lib_oracle)r)
traced.oracle_dispatch.argtypes = [C.c_char_p, C.c_void_p, C.c_uint32, C.POINTER(driver.OraclePlane), C.c_uint32]
traced.oracle_dispatch.restype = C.c_int
traced.oracle_set_ieee_mode.argtypes, traced.oracle_set_ieee_mode.restype = [C.c_int], C.c_int
traced.oracle_set_ieee_mode(0 if %(device)r else 1)
if %(contract)r:
    main = driver.load()  # loads the deviation tables; hand them to the traced copy
    traced.oracle_set_hw_tables.argtypes = [C.c_void_p] * 5
    traced.oracle_set_hw_tables(*[t.ctypes.data for t in driver._hw_tables])
name = %(name)r
import json
overrides = json.loads(os.environ.get("NRD_TRACE_OVERRIDES", "null"))  # settings_overrides of tests/ref_parity.run_per_pass, as JSON
cs_kw = json.loads(os.environ.get("NRD_TRACE_CS", "{}"))               # CommonSettings keywords, as JSON
want = tuple(w for w in os.environ.get("NRD_TRACE_WANT", "").split(",") if w)  # extra_want of the frame generator ("materials", "confidence", ...)
origin = tuple(int(v) for v in os.environ["NRD_TRACE_ORIGIN"].split(",")) if os.environ.get("NRD_TRACE_ORIGIN") else None  # rectOrigin: a 144x96 rect in 192x128 planes
RW, RH = (144, 96) if origin else (192, 128)
seq = parity.generate_sequence(name, RW, RH, %(frames)d, device="cpu", extra_want=want)
if origin:
    import ref_parity
    seq = [ref_parity.embed_guides_at(fr, (192, 128), origin) for fr in seq]
    cs_kw.update(resourceSize=(192, 128), resourceSizePrev=(192, 128), rectOrigin=origin)
run = parity.OracleRun(name, 192, 128)
ex = driver.ComparingExecutor(run.inst, 192, 128, api.FORMAT_BYTES, strict=False)
ex.lib = traced
ex.user = run.ex.user
run.ex = ex
for f, frame in enumerate(seq):
    os.write(2, ("MARK FRAME %%d\n" %% f).encode())
    cam, camp = frame["camera"], seq[max(f - 1, 0)]["camera"]
    run.step(frame, parity.common_settings(cam, camp, RW, RH, f, **cs_kw), parity.denoiser_settings(name, frame, overrides))
''' % dict(root=ROOT, lib_ref=lib_ref, lib_oracle=lib_oracle, contract=contract, device=device, name=name, frames=frames)
    env = dict(os.environ, NRD_TRACE_X=str(x), NRD_TRACE_Y=str(y), OMP_NUM_THREADS="1")
    with open(log, "w") as fp:
        subprocess.run([sys.executable, "-c", code], check=True, env=env, stderr=fp)


def compare(log, frame, head):
    vals = {"ref": {}, "ora": {}}
    order = []
    cur = -1
    for line in open(log):
        if line.startswith("MARK FRAME"):
            cur = int(line.split()[2])
        if cur != frame or not line.startswith("TRACE "):
            continue
        _, side, tag, bits, val = line.split()
        vals[side].setdefault(tag, []).append((bits, val))
        if side == "ora" and tag not in order:
            order.append(tag)
    shown = 0
    for tag in order:
        a, b = vals["ref"].get(tag), vals["ora"].get(tag)
        if a is None:
            continue
        for k in range(min(len(a), len(b))):
            if a[k][0] != b[k][0]:
                rel = abs(float(a[k][1]) - float(b[k][1])) / max(abs(float(a[k][1])), 1e-30)
                # the k-th occurrence of a name is compared with the k-th occurrence on the other side: only meaningful when both sides assign the name equally often
                # (a declaration with an initialiser on one side, a loop that runs a different number of times on the other: marked, read with care)
                note = "" if len(a) == len(b) else "   [occurrences: ref %d, ora %d -- possibly misaligned]" % (len(a), len(b))
                print("DIFF %-40s #%d ref %s %-16s ora %s %-16s rel %.2g%s" % (tag, k, a[k][0], a[k][1], b[k][0], b[k][1], rel, note))
                shown += 1
                break
        if shown >= head:
            break
    print("tags: ref %d, ora %d, common %d" % (len(vals["ref"]), len(vals["ora"]), len(set(vals["ref"]) & set(vals["ora"]))))


def main():
    args = sys.argv[1:]
    contract = "--contract" in args or "--device" in args
    device = "--device" in args
    for flag in ("--contract", "--device"):
        if flag in args:
            args.remove(flag)
    head = 40
    if "--head" in args:
        i = args.index("--head")
        head = int(args[i + 1])
        del args[i:i + 2]
    name, frames, frame, x, y, sub, spec = args
    os.makedirs(TMP, exist_ok=True)
    shader = find_shader(name, sub)
    print("shader:", shader)
    lib_o = build_oracle(spec, contract)
    lib_r = build_ref(shader)
    log = os.path.join(TMP, "trace.log")
    run(name, int(frames), int(frame), int(x), int(y), shader, lib_o, lib_r, contract, log, device)
    compare(log, int(frame), head)


if __name__ == "__main__":
    main()
