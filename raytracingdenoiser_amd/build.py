"""Builds the native pieces in-tree (no JIT cache): the product library and the test oracle.

  lib/libNRD_hip.so   host dispatch compiler + HIP kernels + HIP executor (hipcc, gfx950)  -- THE PRODUCT
  oracle/liboracle.so CPU restatement of the pass arithmetic (g++)                          -- TEST INFRASTRUCTURE ONLY

hipcc cross-compiles gfx950 without a GPU, so this runs in the build container; the .so files travel to the GPU box.
"""
import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "raytracingdenoiser_amd")
CSRC = os.path.join(PKG, "csrc")
LIB_DIR = os.path.join(PKG, "lib")
OBJ_DIR = os.path.join(PKG, "lib", "obj")
ORACLE_DIR = os.path.join(ROOT, "oracle")

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# -ffp-contract=off: no FMA contraction, so device arithmetic is bit-reproducible against the CPU oracle
COMMON_FLAGS = ["-std=c++17", "-O3", "-fPIC", "-ffp-contract=off", "-fvisibility=hidden", "-Wno-return-type-c-linkage", "-I" + os.path.join(ROOT, "include")]
HIP_FLAGS = ["--offload-arch=gfx950", "-fno-gpu-rdc", "-Wno-unused-result", "-Wno-return-type-c-linkage"]


def _sources():
    host = sorted(os.path.join(CSRC, "host", f) for f in os.listdir(os.path.join(CSRC, "host")) if f.endswith(".cpp"))
    hip = sorted(os.path.join(CSRC, "hip", f) for f in os.listdir(os.path.join(CSRC, "hip")) if f.endswith(".hip"))
    return host, hip


def _headers_digest():
    h = hashlib.sha1()
    for base in (os.path.join(ROOT, "include"), CSRC):
        for d, _, files in sorted(os.walk(base)):
            for f in sorted(files):
                if f.endswith(".h"):
                    with open(os.path.join(d, f), "rb") as fp:
                        h.update(fp.read())
    return h.hexdigest()


def _compile(src, hdr_digest, verbose):
    with open(src, "rb") as fp:
        digest = hashlib.sha1(fp.read() + hdr_digest.encode() + " ".join(COMMON_FLAGS + HIP_FLAGS).encode()).hexdigest()[:16]
    obj = os.path.join(OBJ_DIR, os.path.basename(src) + "." + digest + ".o")
    if os.path.exists(obj):
        return obj
    for old in os.listdir(OBJ_DIR):
        if old.startswith(os.path.basename(src) + "."):
            os.remove(os.path.join(OBJ_DIR, old))
    is_hip = src.endswith(".hip")
    cmd = [HIPCC] + COMMON_FLAGS + (HIP_FLAGS + ["-x", "hip"] if is_hip else ["-x", "c++"]) + ["-c", src, "-o", obj]
    if verbose:
        print("[build]", " ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return obj


def _global_digest(srcs, hdr_digest):
    h = hashlib.sha1((hdr_digest + " ".join(COMMON_FLAGS + HIP_FLAGS)).encode())
    for s in srcs:
        with open(s, "rb") as fp:
            h.update(fp.read())
    return h.hexdigest()


def build_product(verbose=False):
    """hipcc --offload-arch=gfx950 every HIP translation unit and link lib/libNRD_hip.so. Returns the path."""
    host, hip = _sources()
    hdr = _headers_digest()
    out = os.path.join(LIB_DIR, "libNRD_hip.so")
    whole = _global_digest(host + hip, hdr)
    if os.path.exists(out) and os.path.exists(out + ".digest") and open(out + ".digest").read() == whole:
        return out  # prebuilt (e.g. shipped to the GPU box) and up to date
    os.makedirs(OBJ_DIR, exist_ok=True)
    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as pool:
        objs = list(pool.map(lambda s: _compile(s, hdr, verbose), host + hip))
    cmd = [HIPCC, "-shared", "-fPIC", "--offload-arch=gfx950", "-fno-gpu-rdc"] + objs + ["-o", out]
    if verbose:
        print("[build]", " ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    with open(out + ".digest", "w") as fp:
        fp.write(whole)
    return out


def build_oracle(verbose=False):
    """g++ the CPU oracle (test infrastructure). Returns the path of oracle/liboracle.so."""
    cmd = ["make", "-C", ORACLE_DIR, "-j8"] + ([] if verbose else ["-s"])
    subprocess.run(cmd, check=True)
    return os.path.join(ORACLE_DIR, "liboracle.so")


if __name__ == "__main__":
    print(build_product(verbose=True))
    if "--no-oracle" not in sys.argv:
        print(build_oracle(verbose=True))
