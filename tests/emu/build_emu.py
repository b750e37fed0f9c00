"""TEST INFRASTRUCTURE -- NOT PRODUCT CODE. Builds tests/emu/libNRD_emu.so: the host dispatch compiler + the DEVICE SOURCES of the product
(raytracingdenoiser_amd/csrc/hip/*.hip) compiled as plain C++ for x86-64 over the shim headers in tests/emu/shim, with the arithmetic flags of
the product build (raytracingdenoiser_amd/build.py NUMERICS_FLAGS) so that clang's front end makes the same contraction decisions for both
targets. See shim/hip/hip_runtime.h for what this is for (and what it is not)."""
import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from raytracingdenoiser_amd import build as B  # noqa: E402

CLANG = os.environ.get("NRD_EMU_CXX", "/opt/rocm/lib/llvm/bin/clang++")
OBJ_DIR = os.path.join(HERE, "obj" + B.encoding_suffix())  # (one emulation library per G-buffer encoding: raytracingdenoiser_amd/build.py encoding())
OUT = os.path.join(HERE, "libNRD_emu%s.so" % B.encoding_suffix())


def flags():
    return ["-std=c++17", "-O1", "-fPIC", "-fopenmp", "-fdeclspec", "-fvisibility=hidden", "-Wno-return-type-c-linkage", "-Wno-unused-value", "-mfma", "-mf16c",
            "-I" + os.path.join(HERE, "shim"), "-I" + os.path.join(ROOT, "oracle"), "-I" + os.path.join(ROOT, "include")] + B.encoding_flags() + B.DEVICE_NUMERICS_FLAGS + os.environ.get("NRD_EMU_EXTRA_FLAGS", "").split()


def _digest(paths, extra):
    h = hashlib.sha1(extra.encode())
    for p in sorted(paths):
        with open(p, "rb") as fp:
            h.update(fp.read())
    return h.hexdigest()[:16]


def build(verbose=False):
    with B._BuildLock("emu"):  # pytest-xdist workers build at the same time otherwise
        return _build(verbose)


def _build(verbose):
    host, hip = B._sources()
    srcs = host + hip + [os.path.join(HERE, "emu_runtime.cpp")]
    hdrs = []
    for base in (os.path.join(ROOT, "include"), B.CSRC, os.path.join(HERE, "shim"), os.path.join(ROOT, "oracle")):
        for d, _, files in os.walk(base):
            hdrs += [os.path.join(d, f) for f in files if f.endswith(".h")]
    hdr_digest = _digest(hdrs, B._portable(flags()))
    os.makedirs(OBJ_DIR, exist_ok=True)

    def compile_one(src):
        with open(src, "rb") as fp:
            d = hashlib.sha1(fp.read() + hdr_digest.encode()).hexdigest()[:16]
        obj = os.path.join(OBJ_DIR, os.path.basename(src) + "." + d + ".o")
        if os.path.exists(obj):
            return obj
        for old in os.listdir(OBJ_DIR):
            if old.startswith(os.path.basename(src) + "."):
                os.remove(os.path.join(OBJ_DIR, old))
        # the host dispatch compiler keeps -ffp-contract=off as in the product build (raytracingdenoiser_amd/build.py): both sides must be handed the same constants
        cmd = [CLANG] + flags() + (["-ffp-contract=off"] if src.endswith(".cpp") and "csrc" in src else []) + ["-x", "c++", "-c", src, "-o", obj]
        if verbose:
            print("[emu]", " ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
        return obj

    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as pool:
        objs = list(pool.map(compile_one, srcs))
    stamp = _digest(objs, "")
    if os.path.exists(OUT) and os.path.exists(OUT + ".digest") and open(OUT + ".digest").read() == stamp:
        return OUT
    subprocess.run([CLANG, "-shared", "-fPIC", "-fopenmp"] + objs + ["-o", OUT], check=True)
    with open(OUT + ".digest", "w") as fp:
        fp.write(stamp)
    return OUT


if __name__ == "__main__":
    print(build(verbose="-v" in sys.argv))
