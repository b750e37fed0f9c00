"""The drop-in boundary, held against the reference's OWN headers: every struct field and every enumerator our re-typed
include/NRD*.h declare is compiled twice -- against include/ and against /root/reference/Include -- and must come out with the same offset, size, default
bytes and value; and the C++ application of tests/cpp/integration_reference.cpp is built against the reference's headers (NRD.h, NRDDescs.h, NRDSettings.h
from /root/reference/Include; only the HIP executor's NRDHip.h / NRDIntegrationHip.hpp from include/) and linked with libNRD_hip.so.
Skipped where the reference tree is absent (the GPU box); the sizeof constants of include/*.h stay as static_asserts there."""
import os
import re
import subprocess

import pytest

from raytracingdenoiser_amd import build as native_build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OURS = os.path.join(ROOT, "include")
THEIRS = "/root/reference/Include"
OUT_DIR = os.path.join(ROOT, "tests", "cpp", "build")

pytestmark = pytest.mark.skipif(not os.path.exists(os.path.join(THEIRS, "NRD.h")), reason="/root/reference/Include not present")


def _strip_comments(text):
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return re.sub(r"//[^\n]*", "", text)


def _parse(header):
    """([(struct, [field, ...])], [(enum, [enumerator, ...])]) of the nrd namespace declarations of one of OUR headers"""
    text = _strip_comments(open(os.path.join(OURS, header)).read())
    structs, enums = [], []
    for m in re.finditer(r"\bstruct\s+(\w+)\s*\{(.*?)\n\};", text, flags=re.S):
        fields = []
        for line in m.group(2).split(";"):
            line = line.strip()
            if not line or "(" in line.split("=")[0] or line.startswith(("static", "using", "typedef")):
                continue
            decl = line.split("=")[0].strip()
            fm = re.search(r"(\w+)\s*(?:\[[^\]]*\])*$", decl)  # the declarator: the last identifier in front of the array extents
            if fm:
                fields.append(fm.group(1))
        if fields:
            structs.append((m.group(1), fields))
    for m in re.finditer(r"\benum\s+class\s+(\w+)\s*(?::\s*\w+\s*)?\{(.*?)\};", text, flags=re.S):
        names = [e.split("=")[0].strip() for e in m.group(2).split(",") if e.split("=")[0].strip()]
        enums.append((m.group(1), names))
    return structs, enums


def _program(structs, enums):
    lines = ['#include "NRD.h"', "#include <cstddef>", "#include <cstdio>", "#include <cstring>", "#include <type_traits>", "using namespace nrd;", "template <class T> void dump(const char* n) {",
             "    if constexpr (std::is_default_constructible<T>::value) { T v{}; const unsigned char* b = (const unsigned char*)&v; printf(\"default %s\", n);",
             "        unsigned h = 2166136261u; for (size_t i = 0; i < sizeof(T); i++) h = (h ^ b[i]) * 16777619u; printf(\" fnv %08x\\n\", h); }", "}", "int main() {"]
    for name, fields in structs:
        lines.append('    printf("sizeof %s %%zu align %%zu\\n", sizeof(%s), alignof(%s));' % (name, name, name))
        for f in fields:
            lines.append('    printf("field %s.%s offset %%zu size %%zu\\n", offsetof(%s, %s), sizeof(((%s*)nullptr)->%s));' % (name, f, name, f, name, f))
    for name, names in enums:
        for e in names:
            lines.append('    printf("enum %s::%s %%lld\\n", (long long)%s::%s);' % (name, e, name, e))
    # default member initialisers (settings structs are handed over by value: a changed default is a changed behaviour)
    for name in ("CommonSettings", "ReblurSettings", "RelaxSettings", "SigmaSettings", "ReferenceSettings", "HitDistanceParameters", "ReblurAntilagSettings", "RelaxAntilagSettings"):
        lines.append('    dump<%s>("%s");' % (name, name))
    lines += ["    return 0;", "}"]
    return "\n".join(lines)


def _run(include_dir, src, tag):
    exe = os.path.join(OUT_DIR, "abi_layout_" + tag)
    subprocess.run(["g++", "-std=c++17", "-O0", "-Wno-invalid-offsetof", "-I" + include_dir, src, "-o", exe], check=True, capture_output=True, text=True)
    return subprocess.run([exe], check=True, capture_output=True, text=True).stdout.strip().split("\n")


def test_every_field_and_enumerator_matches_the_reference_headers():
    structs, enums = [], []
    for header in ("NRDDescs.h", "NRDSettings.h"):
        s, e = _parse(header)
        structs += s
        enums += e
    names = {n for n, _ in structs}
    assert {"CommonSettings", "ReblurSettings", "RelaxSettings", "SigmaSettings", "ReferenceSettings", "DispatchDesc", "InstanceDesc", "PipelineDesc", "TextureDesc", "ResourceDesc",
            "DenoiserDesc", "InstanceCreationDesc", "LibraryDesc"} <= names, names
    assert sum(len(f) for _, f in structs) >= 150 and sum(len(e) for _, e in enums) >= 100
    os.makedirs(OUT_DIR, exist_ok=True)
    src = os.path.join(OUT_DIR, "abi_layout.cpp")
    with open(src, "w") as fp:
        fp.write(_program(structs, enums))
    ours, theirs = _run(OURS, src, "ours"), _run(THEIRS, src, "theirs")
    assert len(ours) == len(theirs) > 300
    diff = [(a, b) for a, b in zip(ours, theirs) if a != b]
    assert not diff, diff[:10]


def test_cpp_application_builds_against_the_reference_headers_and_runs():
    lib = native_build.build_product()
    os.makedirs(OUT_DIR, exist_ok=True)
    exe = os.path.join(OUT_DIR, "integration_reference_refhdr")
    src = os.path.join(ROOT, "tests", "cpp", "integration_reference.cpp")
    # the reference's directory first: NRD.h / NRDDescs.h / NRDSettings.h resolve there; include/ only contributes NRDHip.h and NRDIntegrationHip.hpp
    cmd = ["g++", "-std=c++17", "-O1", "-ffp-contract=off", "-Wno-attributes", "-I" + THEIRS, "-I" + OURS, "-I/opt/rocm/include", "-D__HIP_PLATFORM_AMD__", "-H", src, "-o", exe,
           "-L" + os.path.dirname(lib), "-lNRD_hip", "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath," + os.path.dirname(lib), "-Wl,-rpath,/opt/rocm/lib"]
    r = subprocess.run(cmd, check=True, capture_output=True, text=True)
    included = [l.strip(". ") for l in r.stderr.split("\n") if l.startswith(".")]
    assert any(p.startswith(THEIRS) and p.endswith("NRD.h") for p in included), included[:10]
    assert not any(p.startswith(OURS) and os.path.basename(p) in ("NRD.h", "NRDDescs.h", "NRDSettings.h") for p in included)
    run = subprocess.run([exe, "--no-gpu"], capture_output=True, text=True, timeout=120)
    assert run.returncode == 0, run.stdout + run.stderr
    assert "NRD 4.14.0" in run.stdout and "host-only OK" in run.stdout
