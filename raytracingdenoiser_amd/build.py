"""Builds the native pieces in-tree (no JIT cache): the product library and the test oracle.

  lib/libNRD_hip.so    host dispatch compiler + HIP kernels + HIP executor (hipcc, gfx950)  -- THE PRODUCT. One library, one arithmetic
                       (DESIGN.md "Numerics"): the arithmetic the benchmark times is the arithmetic the parity suite holds against the oracle bit for bit.
  oracle/liboracle.so  CPU restatement of the pass arithmetic (ROCm's clang, x86-64)         -- TEST INFRASTRUCTURE ONLY
  oracle/liboracle_strict.so  the same sources without contraction / with true divisions     -- TEST INFRASTRUCTURE ONLY (held against oracle/_ref)
  oracle/_ref/libnrdref.so    the reference's own HLSL shaders compiled as C++ (oracle/ref/) -- TEST INFRASTRUCTURE ONLY, built where /root/reference exists

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
LIB_NAME = "libNRD_hip.so"

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
COMMON_FLAGS = ["-std=c++17", "-O3", "-fPIC", "-fvisibility=hidden", "-Wno-return-type-c-linkage", "-I" + os.path.join(ROOT, "include")]
# -fno-slp-vectorize: gfx950 issues v_pk_{fma,mul,add}_f32 at HALF the rate of the scalar forms (profiles/r02_a_valu_bench.txt: 4.5 vs 2.3-2.6 cycles per
# wave instruction), so packing two independent fp32 operations buys nothing, while the SLP vectoriser's register-pair shuffles cost v_movs and 30-50 VGPRs
# (REBLUR blur pass: 124 -> 78 VGPRs at an unchanged instruction count). It never changes a rounding.
HIP_FLAGS = ["--offload-arch=gfx950", "-fno-gpu-rdc", "-fno-slp-vectorize", "-Wno-unused-result", "-Wno-return-type-c-linkage"]
# The arithmetic of the device sources (csrc/hip/nrdmath.h states the contract). -ffp-contract=on is ISO C "FP_CONTRACT ON": `a * b + c` written as ONE
# expression becomes a single fused multiply-add -- decided by clang's front end from the source text, identically for gfx950 and for the x86-64 build of the
# oracle (oracle/Makefile) and of the CPU emulation of these sources (tests/emu) -- and nothing else is fused (no cross-statement contraction, no
# reassociation, no approximate functions, fp32 denormals kept). Division, sqrt, exp2, log2 never reach the compiler as such: the sources spell them
# as v_rcp_f32 / v_sqrt_f32 / v_rsq_f32 / v_exp_f32 / v_log_f32, which the oracle reproduces from measured tables.
DEVICE_NUMERICS_FLAGS = ["-ffp-contract=on"]
# The host dispatch compiler (csrc/host/*.cpp) computes the per-frame constants both sides are handed; it stays unfused IEEE arithmetic so that
# every build of it (hipcc's host compiler here, clang in tests/emu) emits the same bytes (tests/test_host_constants.py holds them against numpy).
HOST_NUMERICS_FLAGS = ["-ffp-contract=off"]


def encoding():
    """(NRD_NORMAL_ENCODING, NRD_ROUGHNESS_ENCODING) of this build: a BUILD configuration of the whole stack, as in the reference (CMakeLists.txt:28-29), taken from the environment
    variables of the same names; default (2, 1) = R10_G10_B10_A2_UNORM + LINEAR. Every native piece -- the product, the oracle, oracle/_ref, tests/emu -- is built per encoding into
    its own file (suffix below), so one checkout holds several and a process picks one through its environment."""
    n, r = int(os.environ.get("NRD_NORMAL_ENCODING", "2")), int(os.environ.get("NRD_ROUGHNESS_ENCODING", "1"))
    if not (0 <= n <= 4 and 0 <= r <= 2):
        raise ValueError("NRD_NORMAL_ENCODING must be 0..4 and NRD_ROUGHNESS_ENCODING 0..2 (nrd::NormalEncoding / nrd::RoughnessEncoding), got %d / %d" % (n, r))
    return n, r


def encoding_suffix():
    n, r = encoding()
    return "" if (n, r) == (2, 1) else "_enc%d%d" % (n, r)


def encoding_flags():
    n, r = encoding()
    return [] if (n, r) == (2, 1) else ["-DNRD_NORMAL_ENCODING=%d" % n, "-DNRD_ROUGHNESS_ENCODING=%d" % r]


def product_path():
    """lib/libNRD_hip.so, or lib/enc<N><R>/libNRD_hip.so for a non-default encoding (same file name: it is the library an application links)"""
    sfx = encoding_suffix()
    return os.path.join(LIB_DIR, sfx.lstrip("_"), LIB_NAME) if sfx else os.path.join(LIB_DIR, LIB_NAME)


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


def _flags(src, extra=()):
    """compiler flags of one translation unit; `extra` = A/B switches of tools/build_variant.py (device sources only)"""
    if src.endswith(".hip"):
        return COMMON_FLAGS + encoding_flags() + DEVICE_NUMERICS_FLAGS + HIP_FLAGS + list(extra) + ["-x", "hip"]
    return COMMON_FLAGS + encoding_flags() + HOST_NUMERICS_FLAGS + ["-x", "c++"]


def _portable(flags):
    """the flags as they enter a digest: without the absolute path of this checkout (the snapshot that travels to a GPU box lives under another path -- until round 5 the
    `-I<root>/include` flag made the shipped library look stale there, and every test session on the box recompiled it)"""
    return " ".join(flags).replace(ROOT, "<root>")


def _compile(src, hdr_digest, verbose, obj_dir, extra=()):
    flags = _flags(src, extra)
    with open(src, "rb") as fp:
        digest = hashlib.sha1(fp.read() + hdr_digest.encode() + _portable(flags).encode()).hexdigest()[:16]
    os.makedirs(obj_dir, exist_ok=True)
    obj = os.path.join(obj_dir, os.path.basename(src) + "." + digest + ".o")
    if os.path.exists(obj):
        return obj
    for old in os.listdir(obj_dir):
        if old.startswith(os.path.basename(src) + ".") and old.endswith(".o"):
            try:
                os.remove(os.path.join(obj_dir, old))
            except FileNotFoundError:
                pass
    cmd = [HIPCC] + flags + ["-c", src, "-o", obj + ".tmp%d" % os.getpid()]
    if verbose:
        print("[build]", " ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    os.replace(obj + ".tmp%d" % os.getpid(), obj)
    return obj


def _global_digest(srcs, hdr_digest, extra=()):
    h = hashlib.sha1((hdr_digest + _portable(COMMON_FLAGS + encoding_flags() + HIP_FLAGS + DEVICE_NUMERICS_FLAGS + HOST_NUMERICS_FLAGS + list(extra))).encode())
    for s in srcs:
        with open(s, "rb") as fp:
            h.update(fp.read())
    return h.hexdigest()


class _BuildLock:
    """one builder at a time (pytest-xdist starts several workers whose session fixtures all call the build functions): an advisory lock on a file next to the outputs"""

    def __init__(self, name):
        os.makedirs(LIB_DIR, exist_ok=True)
        self.path = os.path.join(LIB_DIR, "." + name + ".lock")

    def __enter__(self):
        import fcntl

        self.fp = open(self.path, "w")
        fcntl.flock(self.fp, fcntl.LOCK_EX)

    def __exit__(self, *exc):
        import fcntl

        fcntl.flock(self.fp, fcntl.LOCK_UN)
        self.fp.close()


def build_product(verbose=False, out=None, extra=(), obj_dir=None):
    """hipcc --offload-arch=gfx950 every HIP translation unit and link lib/libNRD_hip.so (or `out`, for the A/B variants of tools/build_variant.py).
    Returns the path."""
    with _BuildLock("product"):
        return _build_product(verbose, out, extra, obj_dir)


def _build_product(verbose, out, extra, obj_dir):
    host, hip = _sources()
    hdr = _headers_digest()
    out = out or product_path()
    whole = _global_digest(host + hip, hdr, extra)
    if os.path.exists(out) and os.path.exists(out + ".digest") and open(out + ".digest").read() == whole:
        return out  # prebuilt (e.g. shipped to the GPU box) and up to date
    os.makedirs(os.path.dirname(out), exist_ok=True)
    obj_dir = obj_dir or os.path.join(OBJ_DIR, "product" + encoding_suffix())
    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as pool:
        objs = list(pool.map(lambda s: _compile(s, hdr, verbose, obj_dir, extra), host + hip))
    cmd = [HIPCC, "-shared", "-fPIC", "--offload-arch=gfx950", "-fno-gpu-rdc"] + objs + ["-o", out]
    if verbose:
        print("[build]", " ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    with open(out + ".digest", "w") as fp:
        fp.write(whole)
    return out


def build_oracle(verbose=False):
    """ROCm's clang (x86-64) on the CPU oracle (test infrastructure). Returns the path of oracle/liboracle.so."""
    sfx = encoding_suffix()
    cmd = ["make", "-C", ORACLE_DIR, "-j8"] + ([] if verbose else ["-s"]) + (["SUFFIX=" + sfx, "EXTRA=" + " ".join(encoding_flags())] if sfx else [])
    with _BuildLock("oracle"):
        subprocess.run(cmd, check=True)
    return os.path.join(ORACLE_DIR, "liboracle%s.so" % sfx)


def build_ref(verbose=False, reference="/root/reference"):
    """oracle/_ref/libnrdref.so: the reference's own HLSL shaders compiled as C++ (oracle/ref/Makefile). Test infrastructure, like the oracle. Only possible
    where the reference tree is present (the build container); elsewhere the prebuilt library that travelled with the snapshot is used. Returns the path or None."""
    sfx = encoding_suffix()
    out = os.path.join(ORACLE_DIR, "_ref", sfx.lstrip("_"), "libnrdref.so") if sfx else os.path.join(ORACLE_DIR, "_ref", "libnrdref.so")
    if not os.path.isdir(os.path.join(reference, "Shaders", "Source")):
        return out if os.path.exists(out) else None
    with _BuildLock("ref"):
        if sfx:  # another encoding: the reference's host and one denoiser per family (oracle/ref/Makefile "enc")
            n, r = encoding()
            quiet = [] if verbose else ["-s"]
            subprocess.run(["make", "-C", os.path.join(ORACLE_DIR, "ref", "host"), "-j8", "REFERENCE=" + reference, "NE=%d" % n, "RE=%d" % r] + quiet, check=True)
            subprocess.run(["make", "-C", os.path.join(ORACLE_DIR, "ref"), "-j8", "REFERENCE=" + reference, "enc", "NE=%d" % n, "RE=%d" % r] + quiet, check=True)
            return out
        return _build_ref(verbose, reference, out)


def _build_ref(verbose, reference, out):
    # oracle/_ref/libnrdhost.so: the reference's own host sources over a MathLib stand-in (oracle/ref/host/Makefile)
    subprocess.run(["make", "-C", os.path.join(ORACLE_DIR, "ref", "host"), "-j8", "REFERENCE=" + reference] + ([] if verbose else ["-s"]), check=True)
    for target in ([], ["vo"]):  # "vo": one denoiser per family with NRD_USE_VIEWPORT_OFFSET = 1 (CommonSettings::rectOrigin; oracle/ref/Makefile)
        cmd = ["make", "-C", os.path.join(ORACLE_DIR, "ref"), "-j8", "REFERENCE=" + reference] + target + ([] if verbose else ["-s"])
        subprocess.run(cmd, check=True)
    return out


# the non-default encodings the test-suite exercises (tests/test_encodings.py): all four other normal encodings, both non-linear roughness encodings
TESTED_ENCODINGS = [(0, 0), (4, 2), (1, 1), (3, 1)]


def build_encoding_variants(encodings=TESTED_ENCODINGS, verbose=False, reference="/root/reference"):
    """product + oracle (+ oracle/_ref where the reference tree is present) of the given non-default encodings; each into its own files (encoding_suffix). Returns the product paths."""
    out, saved = [], {k: os.environ.get(k) for k in ("NRD_NORMAL_ENCODING", "NRD_ROUGHNESS_ENCODING")}
    try:
        for n, r in encodings:
            os.environ["NRD_NORMAL_ENCODING"], os.environ["NRD_ROUGHNESS_ENCODING"] = str(n), str(r)
            out.append(build_product(verbose=verbose))
            build_oracle(verbose=verbose)
            build_ref(verbose=verbose, reference=reference)
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    return out


if __name__ == "__main__":
    print(build_product(verbose=True))
    if "--no-oracle" not in sys.argv:
        print(build_oracle(verbose=True))
        print(build_ref(verbose=False))
