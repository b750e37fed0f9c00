"""Builds the native pieces in-tree (no JIT cache): the product library (in its two numerics modes) and the test oracle.

  lib/libNRD_hip.so        host dispatch compiler + HIP kernels + HIP executor (hipcc, gfx950)  -- THE PRODUCT ("fast" numerics:
                           hardware rcp / exp2 / log2, FMA contraction, fp32 denormals flushed -- DESIGN.md "Numerics")
  lib/libNRD_hip_exact.so  the same sources with the pinned IEEE arithmetic (no contraction, correctly rounded division, polynomial
                           transcendentals): bit-identical to the CPU oracle -- the regression build of the parity suite
  oracle/liboracle.so      CPU restatement of the pass arithmetic (g++)                          -- TEST INFRASTRUCTURE ONLY

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
COMMON_FLAGS = ["-std=c++17", "-O3", "-fPIC", "-fvisibility=hidden", "-Wno-return-type-c-linkage", "-I" + os.path.join(ROOT, "include")]
# -fno-slp-vectorize: gfx950 issues v_pk_{fma,mul,add}_f32 at HALF the rate of the scalar forms (profiles/r02_a_valu_bench.txt: 4.5 vs 2.3-2.6 cycles per
# wave instruction), so packing two independent fp32 operations buys nothing, while the SLP vectoriser's register-pair shuffles cost v_movs and 30-50 VGPRs
# (REBLUR blur pass: 124 -> 78 VGPRs at an unchanged instruction count). It never changes a rounding, so both numerics builds take it.
HIP_FLAGS = ["--offload-arch=gfx950", "-fno-gpu-rdc", "-fno-slp-vectorize", "-Wno-unused-result", "-Wno-return-type-c-linkage"]
# exact: no FMA contraction, so device arithmetic is bit-reproducible against the CPU oracle (DESIGN.md "Numerics")
# fast : contraction on, a / b = a * v_rcp_f32(b), sqrt / exp2 / log2 as single hardware instructions (-fapprox-func together with flushed fp32
#        denormals is what makes hipcc emit them without range-scaling code); NaN / infinity semantics are kept (no -ffinite-math-only)
NUMERICS_FLAGS = {
    "exact": ["-ffp-contract=off"],
    "fast": ["-DNRD_FAST=1", "-ffp-contract=fast", "-fapprox-func", "-fgpu-flush-denormals-to-zero"],
}
# fast build, per translation unit: value-changing reassociation (a * b + a * c -> a * (b + c), x * 0 -> 0, hoisted reciprocals). NaN / infinity semantics
# stay (the passes test for both). Measured per pass (profiles/r02_j_reassoc_*.json): -2 % on the REBLUR frame and on RELAX temporal accumulation; the a-trous
# kernels lose a wave of occupancy with it (+13 %), so they keep the plain flags.
REASSOC_FLAGS = ["-fno-signed-zeros", "-freciprocal-math", "-fassociative-math", "-fno-trapping-math"]
FAST_EXTRA = {"kernels_reblur_ta.hip": REASSOC_FLAGS, "kernels_reblur_spatial.hip": REASSOC_FLAGS, "kernels_reblur_history.hip": REASSOC_FLAGS, "kernels_relax_ta.hip": REASSOC_FLAGS}
DEVICE_NUMERICS_FLAGS = ["-ffp-contract=on"]  # the arithmetic flags of the device sources (also read by tests/emu/build_emu.py)
LIB_NAMES = {"fast": "libNRD_hip.so", "exact": "libNRD_hip_exact.so"}
# translation units that keep the exact flags in both builds: the REFERENCE accumulator is specified bit-exact (BASELINE.json) and is a pure
# streaming kernel, the host dispatch compiler must hand identical constants to both builds
ALWAYS_EXACT = ("kernels_common.hip",)


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


def _flags(src, numerics):
    exact = numerics == "exact" or os.path.basename(src) in ALWAYS_EXACT or src.endswith(".cpp")
    return COMMON_FLAGS + NUMERICS_FLAGS["exact" if exact else "fast"] + ([] if exact else FAST_EXTRA.get(os.path.basename(src), [])) + ([] if numerics == "exact" else ["-DNRD_FAST_BUILD=1"])


def _compile(src, hdr_digest, verbose, numerics):
    flags = _flags(src, numerics)
    with open(src, "rb") as fp:
        digest = hashlib.sha1(fp.read() + hdr_digest.encode() + " ".join(flags + HIP_FLAGS).encode()).hexdigest()[:16]
    obj_dir = os.path.join(OBJ_DIR, numerics)
    os.makedirs(obj_dir, exist_ok=True)
    obj = os.path.join(obj_dir, os.path.basename(src) + "." + digest + ".o")
    if os.path.exists(obj):
        return obj
    for old in os.listdir(obj_dir):
        if old.startswith(os.path.basename(src) + "."):
            os.remove(os.path.join(obj_dir, old))
    is_hip = src.endswith(".hip")
    cmd = [HIPCC] + flags + (HIP_FLAGS + ["-x", "hip"] if is_hip else ["-x", "c++"]) + ["-c", src, "-o", obj]
    if verbose:
        print("[build]", " ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return obj


def _global_digest(srcs, hdr_digest, numerics):
    h = hashlib.sha1((hdr_digest + " ".join(COMMON_FLAGS + HIP_FLAGS + NUMERICS_FLAGS[numerics]) + numerics + repr(sorted(FAST_EXTRA.items()))).encode())
    for s in srcs:
        with open(s, "rb") as fp:
            h.update(fp.read())
    return h.hexdigest()


def build_product(verbose=False, numerics="fast"):
    """hipcc --offload-arch=gfx950 every HIP translation unit and link lib/libNRD_hip.so (numerics "fast") or lib/libNRD_hip_exact.so. Returns the path."""
    host, hip = _sources()
    hdr = _headers_digest()
    out = os.path.join(LIB_DIR, LIB_NAMES[numerics])
    whole = _global_digest(host + hip, hdr, numerics)
    if os.path.exists(out) and os.path.exists(out + ".digest") and open(out + ".digest").read() == whole:
        return out  # prebuilt (e.g. shipped to the GPU box) and up to date
    os.makedirs(OBJ_DIR, exist_ok=True)
    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as pool:
        objs = list(pool.map(lambda s: _compile(s, hdr, verbose, numerics), host + hip))
    cmd = [HIPCC, "-shared", "-fPIC", "--offload-arch=gfx950", "-fno-gpu-rdc"] + objs + ["-o", out]
    if verbose:
        print("[build]", " ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    with open(out + ".digest", "w") as fp:
        fp.write(whole)
    return out


def build_all(verbose=False):
    """both numerics modes of the product"""
    return [build_product(verbose, "fast"), build_product(verbose, "exact")]


def build_oracle(verbose=False):
    """g++ the CPU oracle (test infrastructure). Returns the path of oracle/liboracle.so."""
    cmd = ["make", "-C", ORACLE_DIR, "-j8"] + ([] if verbose else ["-s"])
    subprocess.run(cmd, check=True)
    return os.path.join(ORACLE_DIR, "liboracle.so")


if __name__ == "__main__":
    print(build_product(verbose=True, numerics="fast"))
    print(build_product(verbose=True, numerics="exact"))
    if "--no-oracle" not in sys.argv:
        print(build_oracle(verbose=True))
