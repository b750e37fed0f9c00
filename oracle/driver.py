"""ORACLE -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Drives oracle/liboracle.so through a dispatch list obtained from the product's nrd::GetComputeDispatches, i.e. plays
for the CPU oracle the role the HIP executor plays for the GPU: owns the pool planes (numpy, host memory) and calls
one CPU pass per DispatchDesc. Importable only from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
"""
import ctypes as C
import os

import numpy as np

_DIR = os.path.dirname(os.path.abspath(__file__))
# the G-buffer encoding is a build configuration of the whole stack (raytracingdenoiser_amd/build.py encoding()): a non-default pair has its own oracle libraries
_NE, _RE = int(os.environ.get("NRD_NORMAL_ENCODING", "2")), int(os.environ.get("NRD_ROUGHNESS_ENCODING", "1"))
ENCODING_SUFFIX = "" if (_NE, _RE) == (2, 1) else "_enc%d%d" % (_NE, _RE)
LIB_PATH = os.path.join(_DIR, "liboracle%s.so" % ENCODING_SUFFIX)


class OraclePlane(C.Structure):
    _fields_ = [("data", C.c_void_p), ("rowPitchBytes", C.c_uint32), ("format", C.c_uint32), ("width", C.c_uint16), ("height", C.c_uint16)]


_lib = None
_hw_tables = None  # keeps the numpy arrays behind oracle_set_hw_tables alive


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("oracle not built: run `make -C oracle`")
        lib = C.CDLL(LIB_PATH)
        lib.oracle_dispatch.argtypes = [C.c_char_p, C.c_void_p, C.c_uint32, C.POINTER(OraclePlane), C.c_uint32]
        lib.oracle_dispatch.restype = C.c_int
        lib.oracle_set_threads.argtypes, lib.oracle_set_threads.restype = [C.c_int], C.c_int
        lib.oracle_f32tof16.argtypes, lib.oracle_f32tof16.restype = [C.c_float], C.c_uint32
        lib.oracle_f16tof32.argtypes, lib.oracle_f16tof32.restype = [C.c_uint32], C.c_float
        for name in ("oracle_exp2", "oracle_log2", "oracle_atan"):
            getattr(lib, name).argtypes, getattr(lib, name).restype = [C.c_float], C.c_float
        lib.oracle_pow.argtypes, lib.oracle_pow.restype = [C.c_float, C.c_float], C.c_float
        lib.oracle_pow01.argtypes, lib.oracle_pow01.restype = [C.c_float, C.c_float], C.c_float
        lib.oracle_eval_hw.argtypes, lib.oracle_eval_hw.restype = [C.c_int, C.c_void_p, C.c_void_p, C.c_int], None
        lib.oracle_frontend.argtypes, lib.oracle_frontend.restype = [C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int], None
        # rcp / sqrt / rsqrt / exp2 / log2 follow gfx950's instructions: per-mantissa deviation (in ulps) from the reference results of oracle/hw_ref.h,
        # measured on the device by tools/hw_tables.hip and committed next to the oracle (oracle/hw_math.h)
        import zlib

        global _hw_tables
        _hw_tables = []
        for name, size in (("hw_rcp", 1 << 23), ("hw_sqrt", 1 << 24), ("hw_rsq", 1 << 24), ("hw_exp2", (1 << 23) + 1), ("hw_log2", 1 << 23)):
            path = os.path.join(_DIR, name + ".i8.z")
            if not os.path.exists(path):
                raise RuntimeError("oracle: %s is missing (tools/hw_tables.hip on the GPU writes it)" % path)
            table = np.frombuffer(zlib.decompress(open(path, "rb").read()), dtype=np.int8)
            assert table.size == size and int(np.abs(table).max()) <= 1, name
            _hw_tables.append(np.ascontiguousarray(table))
        lib.oracle_set_hw_tables.argtypes, lib.oracle_set_hw_tables.restype = [C.c_void_p] * 5, None
        lib.oracle_set_hw_tables(*[t.ctypes.data for t in _hw_tables])
        neg = np.ascontiguousarray(np.frombuffer(zlib.decompress(open(os.path.join(_DIR, "hw_exp2neg.i8.z"), "rb").read()), dtype=np.int8))  # v_exp_f32 on [-2, -1] (round 5, tools/hw_exp_neg.hip)
        assert neg.size == (1 << 23) + 1 and int(np.abs(neg).max()) <= 1
        _hw_tables.append(neg)
        lib.oracle_set_hw_table_exp2neg.argtypes, lib.oracle_set_hw_table_exp2neg.restype = [C.c_void_p], None
        lib.oracle_set_hw_table_exp2neg(neg.ctypes.data)
        lib.oracle_set_ieee_mode.argtypes, lib.oracle_set_ieee_mode.restype = [C.c_int], C.c_int
        if os.environ.get("ORACLE_EXACT_SQRT", "0") not in ("", "0"):
            lib.oracle_set_ieee_mode(1)
        _lib = lib
    return _lib


def set_ieee_mode(on):
    """True: rcp / sqrt / rsqrt / exp2 / log2 are the reference results of oracle/hw_ref.h (the oracle knows nothing about the device); False: they
    emulate gfx950's instructions from the measured tables (bit-exact comparison with the GPU). Returns the previous setting."""
    return bool(load().oracle_set_ieee_mode(1 if on else 0))


def _pitch(width, bpt):
    return (width * bpt + 255) & ~255


class OracleExecutor:
    """CPU twin of raytracingdenoiser_amd.executor.HipExecutor."""

    def __init__(self, instance, width, height, format_bytes, threads=0, promote_fp16=False):
        """promote_fp16: create every fp16 pool plane as fp32 -- what the reference's integration layer offers as `promoteFloat16to32`
        (reference Integration/NRDIntegration.h:73-78); used by the per-pass comparison with oracle/_ref to look below the fp16 storage granularity"""
        from raytracingdenoiser_amd import api  # ctypes plumbing of the public NRD API only

        self.api = api
        self.lib = load()
        if threads:
            self.lib.oracle_set_threads(threads)
        self.instance = instance
        self.width, self.height = width, height
        self.pools = {}
        for pool_type, descs in ((api.ResourceType.PERMANENT_POOL, instance.permanent_pool), (api.ResourceType.TRANSIENT_POOL, instance.transient_pool)):
            planes = []
            for fmt, ds in descs:
                if promote_fp16:
                    fmt = {api.Format.RGBA16_SFLOAT: api.Format.RGBA32_SFLOAT, api.Format.R16_SFLOAT: api.Format.R32_SFLOAT}.get(fmt, fmt)
                w, h = (width + ds - 1) // ds, (height + ds - 1) // ds
                planes.append((np.zeros((h, _pitch(w, format_bytes[fmt])), dtype=np.uint8), fmt, w, h))
            self.pools[pool_type] = planes
        self.user = {}

    def bind(self, resource_type, array, fmt):
        """array: C-contiguous numpy array whose rows are plane rows ([H, W, C] or [H, W], any dtype)."""
        assert array.flags["C_CONTIGUOUS"]
        self.user[int(resource_type)] = (array, fmt, self.width, self.height)

    def _array(self, res):
        """(numpy array, format, width, height) behind one DispatchDesc resource"""
        _, rtype, index = res
        api = self.api
        if rtype in (api.ResourceType.PERMANENT_POOL, api.ResourceType.TRANSIENT_POOL):
            return self.pools[rtype][index]
        return self.user[int(rtype)]

    def _plane(self, res):
        arr, fmt, w, h = self._array(res)
        pitch = arr.strides[0]
        return OraclePlane(arr.ctypes.data, pitch, int(fmt), w, h)

    def _run(self, d, constants, planes):
        """one pass on the CPU: oracle/liboracle.so, the hand-written restatement (RefExecutor: oracle/_ref/libnrdref.so, the reference's own shader text)"""
        buf = C.create_string_buffer(constants, len(constants)) if constants else None
        rc = self.lib.oracle_dispatch(d.shader.encode(), buf, len(constants), planes, len(d.resources))
        if rc != 0:
            raise RuntimeError("oracle has no pass '%s'" % d.shader)

    # byte offsets of (gRectOrigin, gRectOffset) in the shared constant blocks and their sizes (reference REBLUR / RELAX / SIGMA _SHARED_CONSTANTS)
    _ORIGIN_FIELDS = {"REBLUR_": (648, 616, 832), "RELAX_": (472, 416, 704), "SIGMA_": (448, 432, 516)}
    _GUIDE_INPUTS = ("IN_MV", "IN_NORMAL_ROUGHNESS", "IN_VIEWZ", "IN_DIFF_CONFIDENCE", "IN_SPEC_CONFIDENCE", "IN_DISOCCLUSION_THRESHOLD_MIX", "IN_BASECOLOR_METALNESS")

    def _rect_origin(self, dispatches):
        import struct

        for d in dispatches:
            for prefix, (origin, _, size) in self._ORIGIN_FIELDS.items():
                if d.shader.startswith(prefix) and len(d.constants) >= size:
                    return struct.unpack_from("<II", d.constants, origin)
        return 0, 0

    def execute(self, dispatches):
        # CommonSettings::rectOrigin: the reference addresses its guide inputs at rectOrigin + pixel (Common.hlsli:200-206 and the WithRectOrigin /
        # WithRectOffset call sites) and nothing else. As in the HIP executor, the passes are handed rect-at-origin copies of those inputs and constant
        # blocks whose gRectOrigin / gRectOffset are zero.
        ox, oy = self._rect_origin(dispatches)
        saved = {}
        self._origin_of_frame = (ox, oy)
        self._unshifted = saved  # (ComparingExecutor hands the reference's viewport-offset build the planes and constants as the application passed them)
        if ox or oy:
            guide = {int(getattr(self.api.ResourceType, n)) for n in self._GUIDE_INPUTS}
            for key in [k for k in self.user if k in guide]:
                arr, fmt, w, h = self.user[key]
                twin = np.zeros_like(arr)
                twin[: arr.shape[0] - oy, : arr.shape[1] - ox] = arr[oy:, ox:]
                saved[key] = self.user[key]
                self.user[key] = (twin, fmt, w, h)
        try:
            for d in dispatches:
                planes = (OraclePlane * len(d.resources))(*[self._plane(r) for r in d.resources])
                constants = d.constants
                if (ox or oy) and constants:
                    for prefix, (origin, offset, size) in self._ORIGIN_FIELDS.items():
                        if d.shader.startswith(prefix) and len(constants) >= size:
                            b = bytearray(constants)
                            b[origin:origin + 8] = bytes(8)
                            b[offset:offset + 8] = bytes(8)
                            constants = bytes(b)
                self._run(d, constants, planes)
        finally:
            mv = int(self.api.ResourceType.IN_MV)
            for key, original in saved.items():
                if key == mv:  # in/out plane: REBLUR's specular MV modification writes it
                    arr, twin = original[0], self.user[key][0]
                    arr[oy:, ox:] = twin[: arr.shape[0] - oy, : arr.shape[1] - ox]
                self.user[key] = original

    def pool_plane(self, pool, index):
        arr, fmt, w, _ = self.pools[pool][index]
        return arr, fmt, w


# ---- oracle/_ref: the reference's own HLSL shaders compiled as C++ (oracle/ref/Makefile) ------------------------------------------------------------
_REF_DIR = os.path.join(_DIR, "_ref", ENCODING_SUFFIX.lstrip("_")) if ENCODING_SUFFIX else os.path.join(_DIR, "_ref")  # (oracle/ref/Makefile "enc": one denoiser per family + RELAX SH)
REF_LIB_PATH = os.path.join(_REF_DIR, "libnrdref.so")
REF_VO_LIB_PATH = os.path.join(_DIR, "_ref", "libnrdref_vo.so")  # the NRD_USE_VIEWPORT_OFFSET = 1 build of one denoiser per family (oracle/ref/Makefile "vo")
REF_HOST_LIB_PATH = os.path.join(_REF_DIR, "libnrdhost.so")  # the reference's own HOST sources (Source/*.cpp) over a MathLib stand-in (oracle/ref/host/Makefile)
_ref_libs = {}


def load_ref_host():
    """oracle/_ref/libnrdhost.so with the prototypes of the reference's C entry points (NRD.h) -- the same ctypes structures raytracingdenoiser_amd.api binds the product with, so
    api.Instance(denoisers, lib=load_ref_host()) IS the reference's nrd::Instance"""
    from raytracingdenoiser_amd import api

    if REF_HOST_LIB_PATH in _ref_libs:
        return _ref_libs[REF_HOST_LIB_PATH]
    lib = C.CDLL(REF_HOST_LIB_PATH)
    P = C.POINTER
    lib.CreateInstance.argtypes, lib.CreateInstance.restype = [P(api.InstanceCreationDesc), P(C.c_void_p)], C.c_uint32
    lib.DestroyInstance.argtypes, lib.DestroyInstance.restype = [C.c_void_p], None
    lib.GetLibraryDesc.argtypes, lib.GetLibraryDesc.restype = [], P(api.LibraryDesc)
    lib.GetInstanceDesc.argtypes, lib.GetInstanceDesc.restype = [C.c_void_p], P(api.InstanceDesc)
    lib.SetCommonSettings.argtypes, lib.SetCommonSettings.restype = [C.c_void_p, P(api.CommonSettings)], C.c_uint32
    lib.SetDenoiserSettings.argtypes, lib.SetDenoiserSettings.restype = [C.c_void_p, C.c_uint32, C.c_void_p], C.c_uint32
    lib.GetComputeDispatches.argtypes = [C.c_void_p, P(C.c_uint32), C.c_uint32, P(P(api.DispatchDesc)), P(C.c_uint32)]
    lib.GetComputeDispatches.restype = C.c_uint32
    lib.GetResourceTypeString.argtypes, lib.GetResourceTypeString.restype = [C.c_uint32], C.c_char_p
    lib.GetDenoiserString.argtypes, lib.GetDenoiserString.restype = [C.c_uint32], C.c_char_p
    _ref_libs[REF_HOST_LIB_PATH] = lib
    return lib


def ref_available(path=None):
    return os.path.exists(path or REF_LIB_PATH)


def load_ref(path=None):
    path = path or REF_LIB_PATH
    if path not in _ref_libs:
        if not os.path.exists(path):
            raise RuntimeError("%s not built: run `make -C oracle/ref -j8 [vo]` (needs /root/reference)" % os.path.relpath(path, os.path.dirname(_DIR)))
        lib = C.CDLL(path)
        lib.nrdref_dispatch.argtypes = [C.c_char_p, C.c_void_p, C.c_uint32, C.POINTER(OraclePlane), C.c_uint32, C.c_uint32, C.c_uint32]
        lib.nrdref_dispatch.restype = C.c_int
        lib.nrdref_has.argtypes, lib.nrdref_has.restype = [C.c_char_p], C.c_int
        lib.nrdref_count.argtypes, lib.nrdref_count.restype = [], C.c_int
        lib.nrdref_name.argtypes, lib.nrdref_name.restype = [C.c_int], C.c_char_p
        lib.nrdref_set_threads.argtypes, lib.nrdref_set_threads.restype = [C.c_int], C.c_int
        _ref_libs[path] = lib
    return _ref_libs[path]


def ref_shaders():
    lib = load_ref()
    return sorted(lib.nrdref_name(i).decode() for i in range(lib.nrdref_count()))


class RefExecutor(OracleExecutor):
    """The same driver over oracle/_ref/libnrdref.so: every pass is the reference's own shader text, executed in plain IEEE arithmetic."""

    def __init__(self, *a, **kw):
        super().__init__(*a, **kw)
        self.ref = load_ref()

    def _run(self, d, constants, planes):
        buf = C.create_string_buffer(constants, len(constants)) if constants else None
        rc = self.ref.nrdref_dispatch(d.shader.encode(), buf, len(constants), planes, len(d.resources), d.grid[0], d.grid[1])
        if rc != 0:
            raise RuntimeError("oracle/_ref cannot run '%s' (code %d)" % (d.shader, rc))


def _digest(constants, arrays):
    """sha1 over the constant block and every bound plane as a pass finds them: identifies "the same pass on the same inputs" across runs"""
    import hashlib

    h = hashlib.sha1(constants or b"")
    for a in arrays:
        h.update(np.ascontiguousarray(a).view(np.uint8).tobytes())
    return h.hexdigest()


class StrictRecordingExecutor(OracleExecutor):
    """The strict oracle alone; `on_pass(dispatch, inputs_digest, [(resource, fmt, width, array), ...] for every bound slot)` after every pass (the arrays are
    live views: use them inside the callback): the replay side of the golden fixtures of tests/golden/ref_text_*.npz (recorded with ComparingExecutor from
    the reference's compiled shader text)."""

    def __init__(self, *a, on_pass=None, **kw):
        super().__init__(*a, **kw)
        self.lib = load_strict()
        self.on_pass = on_pass

    def _run(self, d, constants, planes):
        arrays = [self._array(r) for r in d.resources]
        digest = _digest(constants, [a[0] for a in arrays])
        super()._run(d, constants, planes)
        if self.on_pass:
            self.on_pass(d, digest, [(res, a[1], a[2], a[0]) for res, a in zip(d.resources, arrays)])


_strict_lib = None


def load_strict():
    """oracle/liboracle_strict.so: the oracle's sources without contraction and with true divisions, always in IEEE mode (oracle/Makefile)"""
    global _strict_lib
    if _strict_lib is None:
        path = os.path.join(_DIR, "liboracle_strict%s.so" % ENCODING_SUFFIX)
        if not os.path.exists(path):
            raise RuntimeError("oracle/liboracle_strict.so not built: run `make -C oracle all`")
        lib = C.CDLL(path)
        lib.oracle_dispatch.argtypes = [C.c_char_p, C.c_void_p, C.c_uint32, C.POINTER(OraclePlane), C.c_uint32]
        lib.oracle_dispatch.restype = C.c_int
        lib.oracle_set_ieee_mode.argtypes, lib.oracle_set_ieee_mode.restype = [C.c_int], C.c_int
        lib.oracle_set_ieee_mode(1)
        _strict_lib = lib
    return _strict_lib


class ComparingExecutor(OracleExecutor):
    """Runs every pass TWICE on identical inputs -- the hand-written oracle and the reference's own shader (oracle/_ref) -- and hands both sets of
    output planes to `on_pass(dispatch, [(resource, fmt, width, oracle_array, ref_array, [oracle arrays in other arithmetics]), ...])`. The sequence continues on the oracle's results, so there
    is no recurrence in the comparison: every difference is the difference of ONE pass."""

    def __init__(self, *a, on_pass=None, strict=True, sensitivity=False, ref_lib_path=None, **kw):
        """strict: "the oracle" is liboracle_strict.so (no contraction, true divisions) -- the arithmetic of the reference text; False: liboracle.so in
        whatever mode set_ieee_mode selected (the arithmetic contract the HIP library is held against)"""
        super().__init__(*a, **kw)
        if strict:
            self.lib = load_strict()
        self.ref = load_ref(ref_lib_path)
        self.on_pass = on_pass
        self.sensitivity = sensitivity

    def _run(self, d, constants, planes):
        arrays = [self._array(r) for r in d.resources]
        before = [a[0].copy() for a in arrays]
        self.last_inputs_digest = _digest(constants, before)
        buf = C.create_string_buffer(constants, len(constants)) if constants else None

        def restore():
            for a, b in zip(arrays, before):
                a[0][...] = b

        # the same pass in the oracle's OTHER arithmetics (same source text, different roundings): where these move a texel, the texel sits on a
        # discontinuity of the pass (a snapped tap, a threshold, an ill-conditioned quotient) -- the attribution of the outliers
        alts = []
        if self.sensitivity:
            main = load()
            for ieee in (1, 0):
                prev = main.oracle_set_ieee_mode(ieee)
                try:
                    rc = main.oracle_dispatch(d.shader.encode(), buf, len(constants), planes, len(d.resources))
                finally:
                    main.oracle_set_ieee_mode(prev)
                if rc != 0:
                    raise RuntimeError("oracle has no pass '%s'" % d.shader)
                alts.append([a[0].copy() for a in arrays])
                restore()
        ref_buf, ref_size, ref_planes = buf, len(constants), planes
        unshifted = getattr(self, "_unshifted", None)
        if unshifted:
            # CommonSettings::rectOrigin != 0: the oracle's passes run on rect-at-origin twins of the guide inputs with gRectOrigin / gRectOffset zeroed (execute()); the reference
            # -- its NRD_USE_VIEWPORT_OFFSET = 1 build, REF_VO_LIB_PATH -- gets what the application passed: the planes as bound and the constant block of the dispatch
            ref_buf, ref_size = C.create_string_buffer(d.constants, len(d.constants)), len(d.constants)
            def plane_of(res):
                arr, fmt, w, h = unshifted.get(int(res[1]), None) or self._array(res)
                return OraclePlane(arr.ctypes.data, arr.strides[0], int(fmt), w, h)
            ref_planes = (OraclePlane * len(d.resources))(*[plane_of(r) for r in d.resources])
            written = {i: unshifted[int(r[1])][0] for i, r in enumerate(d.resources) if r[0] == self.api.DescriptorType.STORAGE_TEXTURE and int(r[1]) in unshifted}
            written_before = {i: a.copy() for i, a in written.items()}
        rc = self.ref.nrdref_dispatch(d.shader.encode(), ref_buf, ref_size, ref_planes, len(d.resources), d.grid[0], d.grid[1])
        if rc != 0:
            raise RuntimeError("oracle/_ref cannot run '%s' (code %d)" % (d.shader, rc))
        theirs_all = [a[0].copy() for a in arrays]
        if unshifted:
            # an application plane the pass WRITES (IN_MV: Clear_Float on restart, REBLUR's specular motion-vector patch): the reference wrote the plane as bound; what is
            # compared with the oracle's rect-at-origin twin is the same window of it
            ox, oy = self._origin_of_frame
            for i, arr in written.items():
                theirs_all[i] = np.zeros_like(arr)
                theirs_all[i][: arr.shape[0] - oy, : arr.shape[1] - ox] = arr[oy:, ox:]
                arr[...] = written_before[i]
        restore()
        super()._run(d, constants, planes)  # last: the sequence continues on these results
        if unshifted:
            for i, arr in written.items():  # ... in the application's plane too (the following passes of the frame read it at rectOrigin + pixel)
                twin = arrays[i][0]
                arr[oy:, ox:] = twin[: arr.shape[0] - oy, : arr.shape[1] - ox]
        report = []
        for i, (res, a, b, theirs) in enumerate(zip(d.resources, arrays, before, theirs_all)):
            m = a[0]
            if res[0] == self.api.DescriptorType.STORAGE_TEXTURE or not np.array_equal(b, m) or not np.array_equal(b, theirs):
                report.append((res, a[1], a[2], m.copy(), theirs, [alt[i] for alt in alts]))
        if self.on_pass:
            self.on_pass(d, report)
