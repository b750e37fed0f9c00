#!/usr/bin/env python3
"""ORACLE/_ref -- TEST INFRASTRUCTURE, NOT PRODUCT CODE (see hlsl_shim.h).

Turns ONE reference shader entry (/root/reference/Shaders/Source/<name>.cs.hlsl) into a C++ translation unit under oracle/_ref/gen/ (git-ignored):

  1. the C preprocessor (`clang -E -x c -undef`) resolves the shader's own #include / #define / #if structure with the reference's include
     directories, oracle/ref/prelude.hlsli force-included in front (binding macros) and oracle/ref/ml.hlsli standing in for the absent MathLib;
  2. a handful of purely lexical rewrites make the preprocessed HLSL text valid C++ over oracle/ref/hlsl_shim.h -- no statement is added, removed or
     reordered, no expression is touched:
        [unroll] [branch] [flatten] [loop] [numthreads( x, y, z )]   removed (the group size is recorded for the runtime)
        `: SV_GroupThreadId` ... semantics of the entry's parameters  removed (recorded: the generated thunk passes the matching thread ids)
        `out T x` / `inout T x` / `in T x` parameters                 `T& x` / `T& x` / `T x`
        groupshared                                                   static thread_local (one group at a time per OS thread)
        1.0, 0.5e-3 (HLSL: float literals)                            1.0f, 0.5e-3f (C++ would make them double)
        s.x / s.xx / s.xxx / s.xxxx (s may be a scalar)               swz_x1( s ) / swz_x2( s ) / swz_x3( s ) / swz_x4( s )  (same meaning on vectors)
  3. the text is wrapped into namespace hlsl and registered under the name the dispatch list uses (<name>.cs).

usage: hlsl2cpp.py <entry.cs.hlsl> <out.cpp> [--reference /root/reference] [--keep-preprocessed]
"""
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CLANG = os.environ.get("NRDREF_CLANG", "/opt/rocm/lib/llvm/bin/clang")

SEMANTICS = {"SV_GroupThreadId": "groupThreadId", "SV_GroupId": "groupId", "SV_DispatchThreadId": "dispatchThreadId", "SV_GroupIndex": "groupIndex",
             "SV_GroupThreadID": "groupThreadId", "SV_GroupID": "groupId", "SV_DispatchThreadID": "dispatchThreadId"}


def preprocess(entry, reference, include_first=None, encoding=(2, 1), mathlib=None, defs=()):
    """include_first: a directory searched in front of the reference's own Include directory (the viewport-offset build of oracle/ref/Makefile puts a Common.hlsli there whose
    NRD_USE_VIEWPORT_OFFSET is 1 -- the reference makes that switch an edit of the file, Common.hlsli:64)"""
    shaders = os.path.join(reference, "Shaders")
    # mathlib: a directory holding NVIDIA-RTX/MathLib's ml.hlsli (the reference's un-vendored submodule, CMakeLists.txt:118-127), searched in front of the stand-in beside this file
    cmd = [CLANG, "-E", "-x", "c", "-undef", "-nostdinc", "-Wno-everything"] + (["-I", mathlib] if mathlib else []) + ["-I", HERE] + list(defs) + (["-I", include_first] if include_first else []) + ["-I", os.path.join(shaders, "Include"), "-I", os.path.join(shaders, "Resources"),
           "-include", os.path.join(HERE, "prelude.hlsli"), "-DNRD_NORMAL_ENCODING=%d" % encoding[0], "-DNRD_ROUGHNESS_ENCODING=%d" % encoding[1], entry]
    return subprocess.run(cmd, check=True, capture_output=True, text=True).stdout


def split_params(s):
    out, depth, cur = [], 0, ""
    for ch in s:
        if ch in "(<[":
            depth += 1
        elif ch in ")>]":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur)
            cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur)
    return out


def translate(text, shader_name):
    # ---- attributes
    m = re.search(r"\[\s*numthreads\s*\(\s*([^,\]]+?)\s*,\s*([^,\]]+?)\s*,\s*([^,\]]+?)\s*\)\s*\]", text)
    if not m:
        raise SystemExit("%s: no [numthreads]" % shader_name)
    group = [int(eval(v, {"__builtins__": {}})) for v in m.groups()]
    if group[2] != 1:
        raise SystemExit("%s: numthreads z != 1" % shader_name)
    text = text[:m.start()] + text[m.end():]
    text = re.sub(r"\[\s*(unroll|branch|flatten|loop)\s*(\(\s*\d+\s*\))?\s*\]", "", text)

    # ---- the entry point: record and strip the semantics
    m = re.search(r"\bvoid\s+hlsl_cs_main\s*\(([^)]*)\)", text)
    if not m:
        raise SystemExit("%s: entry point not found" % shader_name)
    params, thunk_args = [], []
    for p in split_params(m.group(1)):
        pm = re.match(r"\s*(\w+)\s+(\w+)\s*:\s*(\w+)\s*$", p)
        if not pm:
            raise SystemExit("%s: cannot parse entry parameter '%s'" % (shader_name, p))
        ptype, pname, sem = pm.groups()
        if sem not in SEMANTICS:
            raise SystemExit("%s: unknown semantic %s" % (shader_name, sem))
        params.append("%s %s" % (ptype, pname))
        thunk_args.append("%s( ids.%s )" % (ptype, SEMANTICS[sem]))
    text = text[:m.start()] + "void hlsl_cs_main( " + ", ".join(params) + " )" + text[m.end():]

    # ---- parameter qualifiers (only in front of a type + name, i.e. inside parameter lists)
    text = re.sub(r"(?<=[(,])(\s*)(?:inout|out)\s+((?:const\s+)?\w+)\s+(\w+)", r"\1\2& \3", text)
    text = re.sub(r"(?<=[(,])(\s*)in\s+((?:const\s+)?\w+\s+\w+)", r"\1\2", text)

    # ---- group-shared memory (+ the list of the arrays, for the per-group clear of hlsl_rt.cpp RunGroup)
    shared_names = re.findall(r"\bgroupshared\s+[\w:<>]+\s+(\w+)\s*(?:\[[^\]]*\]\s*)*;", text)
    text = re.sub(r"\bgroupshared\b", "static thread_local", text)

    # ---- swizzles of scalars (and the same spelling on vectors: identical meaning)
    def scalar_swizzle(mm):
        return "swz_x%d( %s )" % (len(mm.group(2)), mm.group(1))
    text = re.sub(r"(?<![\w.])(\d+\.\d*(?:[eE][+-]?\d+)?)\s*\.(x{2,4})\b", lambda mm: "swz_x%d( %sf )" % (len(mm.group(2)), mm.group(1)), text)
    text = re.sub(r"(?<![\w.])([A-Za-z_]\w*)\.(x{2,4})\b", scalar_swizzle, text)
    text = re.sub(r"(?<![\w.])([A-Za-z_]\w*)\.x\b", r"swz_x1( \1 )", text)

    # ---- float literals (not inside linemarkers / preprocessor leftovers: those carry no decimal point)
    text = re.sub(r"(?<![\w.])(\d+\.\d*(?:[eE][+-]?\d+)?|\.\d+(?:[eE][+-]?\d+)?|\d+[eE][+-]?\d+)(?![\w.])", r"\1f", text)

    uses_barrier = "GroupMemoryBarrier" in text  # (with or without GroupSync: hlsl_shim.h)
    head = ('// GENERATED by oracle/ref/hlsl2cpp.py from the reference shader entry %s -- never committed (oracle/_ref/ is git-ignored)\n'
            '#include "hlsl_shim.h"\n'
            'namespace hlsl { namespace {\n'
            'static hlsl_rt::ShaderTable* hlsl_table() { static hlsl_rt::ShaderTable* t = hlsl_rt::NewTable(); return t; }\n' % shader_name)
    clear = "".join(" memset( (void*)&%s, 0, sizeof( %s ) );" % (n, n) for n in shared_names)
    tail = ('\nstatic void hlsl_thunk( const hlsl_rt::ThreadIds& ids ) { hlsl_cs_main( %s ); }\n'
            'static void hlsl_clear_groupshared() {%s }\n'
            'static hlsl_rt::ShaderAdder hlsl_register( hlsl_table(), "%s", %d, %d, %s, hlsl_thunk );\n'
            'static int hlsl_clear_registered = ( hlsl_rt::SetGroupSharedClear( hlsl_table(), hlsl_clear_groupshared ), 0 );\n'
            '} }\n' % (", ".join(thunk_args), clear, shader_name, group[0], group[1], "true" if uses_barrier else "false"))
    return head + text + tail


def main():
    args = sys.argv[1:]
    reference = "/root/reference"
    if "--reference" in args:
        i = args.index("--reference")
        reference = args[i + 1]
        del args[i:i + 2]
    include_first = None
    if "--include-first" in args:
        i = args.index("--include-first")
        include_first = args[i + 1]
        del args[i:i + 2]
    encoding = [2, 1]  # the reference's CMake defaults (CMakeLists.txt:28-29); oracle/ref/Makefile "enc" builds others
    for k, flag in enumerate(("--normal-encoding", "--roughness-encoding")):
        if flag in args:
            i = args.index(flag)
            encoding[k] = int(args[i + 1])
            del args[i:i + 2]
    mathlib = None
    if "--mathlib" in args:
        i = args.index("--mathlib")
        mathlib = args[i + 1] or None
        del args[i:i + 2]
    defs = [a for a in args if a.startswith("-D")]
    args = [a for a in args if not a.startswith("-D")]
    keep = "--keep-preprocessed" in args
    if keep:
        args.remove("--keep-preprocessed")
    entry, out = args
    name = os.path.basename(entry)
    assert name.endswith(".cs.hlsl"), name
    shader_name = name[:-len(".hlsl")]
    pre = preprocess(entry, reference, include_first, tuple(encoding), mathlib, defs)
    if keep:
        with open(out + ".i", "w") as fp:
            fp.write(pre)
    with open(out, "w") as fp:
        fp.write(translate(pre, shader_name))


if __name__ == "__main__":
    main()
