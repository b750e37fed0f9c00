"""The oracle is test infrastructure: nothing that ships may import, include, link or execute it (the contract of oracle/README and DESIGN.md section 4).

Checked statically, so the check itself needs neither a GPU nor the oracle:
  * no source file of the product (raytracingdenoiser_amd/, include/) names oracle/ or tests/ in an import or an #include;
  * bench.py touches the oracle only inside cpu_baseline() -- the CPU-baseline / parity leg that runs after the timed region;
  * __graft_entry__.py touches it only in build() (building the checker) and smoke() (the check);
  * lib/libNRD_hip.so does not depend on a library of oracle/;
  * the executor refuses to work without a GPU instead of falling back to anything.
"""
import ast
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _product_sources():
    for base in ("raytracingdenoiser_amd", "include"):
        for d, dirs, files in os.walk(os.path.join(ROOT, base)):
            dirs[:] = [x for x in dirs if x not in ("__pycache__", "lib")]
            for f in files:
                if f.endswith((".py", ".cpp", ".hip", ".h", ".hpp")):
                    yield os.path.join(d, f)


def test_no_product_source_imports_or_includes_the_oracle_or_the_tests():
    bad = []
    for path in _product_sources():
        text = open(path, errors="replace").read()
        for n, line in enumerate(text.split("\n"), 1):
            code = line.split("//")[0] if not path.endswith(".py") else line.split("#")[0]
            if re.search(r'#\s*include\s*[<"][^>"]*(oracle|tests)/', code) or re.search(r"^\s*(from|import)\s+(oracle|tests|emu|parity|ref_parity)\b", code):
                bad.append("%s:%d: %s" % (os.path.relpath(path, ROOT), n, line.strip()))
            if re.search(r"liboracle|libnrdref|oracle_dispatch|nrdref_dispatch", code):
                bad.append("%s:%d: %s" % (os.path.relpath(path, ROOT), n, line.strip()))
    # build.py builds the checker's libraries (build_oracle / build_ref) without loading them: the only file allowed to name them
    bad = [b for b in bad if not b.startswith(os.path.join("raytracingdenoiser_amd", "build.py"))]
    assert not bad, "\n".join(bad)


def _functions_naming(path, words):
    """{top-level function name or '<module>': [line, ...]} for every AST node of `path` that names one of `words` (attribute, name, import, string)"""
    tree = ast.parse(open(path).read())
    owner = {}
    for top in tree.body:
        name = top.name if isinstance(top, (ast.FunctionDef, ast.AsyncFunctionDef, ast.ClassDef)) else "<module>"
        for node in ast.walk(top):
            ids = []
            if isinstance(node, ast.Import):
                ids = [a.name for a in node.names]
            elif isinstance(node, ast.ImportFrom):
                ids = [node.module or ""] + [a.name for a in node.names]
            elif isinstance(node, ast.Name):
                ids = [node.id]
            elif isinstance(node, ast.Attribute):
                ids = [node.attr]
            if any(i.split(".")[0] in words for i in ids):
                owner.setdefault(name, []).append(node.lineno)
    return owner


def test_bench_touches_the_oracle_only_in_its_cpu_baseline_leg():
    owners = _functions_naming(os.path.join(ROOT, "bench.py"), {"oracle", "oracle_driver", "OracleRun", "parity"})
    assert set(owners) <= {"cpu_baseline"}, owners
    # ... and that leg runs after the timed region: its call site in main() comes after the last perf_counter() read
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert src.count("def cpu_baseline(") == 1 and src.rindex("cpu_baseline(") > src.rindex("elapsed = time.perf_counter() - t0")


def test_graft_entry_touches_the_oracle_only_to_build_it_and_to_check_against_it():
    owners = _functions_naming(os.path.join(ROOT, "__graft_entry__.py"), {"oracle", "build_oracle", "build_ref", "parity", "run_parity"})
    assert set(owners) <= {"build", "smoke"}, owners


def test_the_product_library_does_not_depend_on_the_oracle():
    lib = os.path.join(ROOT, "raytracingdenoiser_amd", "lib", "libNRD_hip.so")
    if not os.path.exists(lib):
        pytest.skip("library not built")
    needed = subprocess.run(["readelf", "-d", lib], capture_output=True, text=True).stdout
    libs = re.findall(r"\(NEEDED\)\s+Shared library: \[([^\]]+)\]", needed)
    assert libs and not [l for l in libs if "oracle" in l or "nrdref" in l], libs
    strings = subprocess.run(["strings", "-n", "8", lib], capture_output=True, text=True).stdout
    assert "liboracle" not in strings and "libnrdref" not in strings


def test_the_executor_refuses_to_run_without_a_gpu():
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from raytracingdenoiser_amd import api
    from raytracingdenoiser_amd.executor import HipExecutor

    inst = api.Instance([(0, api.Denoiser.REBLUR_DIFFUSE)])
    with pytest.raises(RuntimeError, match="needs a GPU"):
        HipExecutor(inst, 64, 32)
