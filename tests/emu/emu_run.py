"""TEST INFRASTRUCTURE -- NOT PRODUCT CODE. Drives tests/emu/libNRD_emu.so (the device sources compiled for the CPU, shim/hip/hip_runtime.h) through
the same C-ABI the GPU tests use, with numpy arrays as planes. `EmuRun` has the interface of tests/parity.py's HipRun, so
parity.run_parity(..., backend="emu") holds the emulated device code against the oracle on a machine without a GPU."""
import ctypes as C
import os
import sys
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, os.path.join(ROOT, "tests"), HERE):
    if p not in sys.path:
        sys.path.insert(0, p)

from raytracingdenoiser_amd import api  # noqa: E402  (ctypes prototypes of the public NRD / NRDHip API only)

_lib = None
_tables = None


def hw_tables():
    """the five deviation tables of oracle/hw_math.h as contiguous int8 arrays: (rcp, sqrt, rsq, exp2, log2)"""
    global _tables
    if _tables is None:
        out = []
        for name, size in (("hw_rcp", 1 << 23), ("hw_sqrt", 1 << 24), ("hw_rsq", 1 << 24), ("hw_exp2", (1 << 23) + 1), ("hw_log2", 1 << 23)):
            path = os.path.join(ROOT, "oracle", name + ".i8.z")
            t = np.frombuffer(zlib.decompress(open(path, "rb").read()), dtype=np.int8)
            assert t.size == size and int(np.abs(t).max()) <= 1, name
            out.append(np.ascontiguousarray(t))
        _tables = out
    return _tables


def load():
    global _lib
    if _lib is None:
        import build_emu

        lib = api.load_library(path=build_emu.build())
        lib.nrdEmuSetHwTables.argtypes, lib.nrdEmuSetHwTables.restype = [C.c_void_p] * 5, None
        lib.nrdEmuSetThreads.argtypes, lib.nrdEmuSetThreads.restype = [C.c_int], C.c_int
        lib.nrdEmuSetHwTables(*[t.ctypes.data for t in hw_tables()])
        global _neg_table
        _neg_table = np.ascontiguousarray(np.frombuffer(zlib.decompress(open(os.path.join(ROOT, "oracle", "hw_exp2neg.i8.z"), "rb").read()), dtype=np.int8))
        lib.nrdEmuSetHwTableExp2Neg.argtypes, lib.nrdEmuSetHwTableExp2Neg.restype = [C.c_void_p], None
        lib.nrdEmuSetHwTableExp2Neg(_neg_table.ctypes.data)
        _lib = lib
    return _lib


class EmuExecutor:
    """the HipExecutor interface (raytracingdenoiser_amd/executor.py) over host memory"""

    def __init__(self, instance, width, height):
        self.instance, self.lib = instance, instance.lib
        self.width, self.height = width, height
        size = self.lib.nrdHipGetArenaSize(instance.handle, width, height)
        raw = np.zeros(max(size, 256) + 4096, dtype=np.uint8)
        off = (-raw.ctypes.data) % 4096
        self.arena = raw[off:off + max(size, 256)]
        handle = C.c_void_p()
        r = api.Result(self.lib.nrdHipCreateExecutorWithArena(instance.handle, width, height, None, self.arena.ctypes.data, size, C.byref(handle)))
        if r != api.Result.SUCCESS:
            raise RuntimeError("nrdHipCreateExecutorWithArena (emulation) failed: %s" % r.name)
        self.handle = handle
        self._bound = {}

    def _check(self, code, what):
        r = api.Result(code)
        if r != api.Result.SUCCESS:
            raise RuntimeError("%s failed: %s (%s)" % (what, r.name, self.lib.nrdHipGetLastError(self.handle).decode()))

    def bind(self, resource_type, array, fmt):
        assert array.strides[-1] == array.itemsize
        desc = api.HipPlaneDesc(array.ctypes.data, array.strides[0], int(fmt), self.width, self.height)
        self._check(self.lib.nrdHipBindResource(self.handle, int(resource_type), C.byref(desc)), "nrdHipBindResource(%s)" % api.ResourceType(resource_type).name)
        self._bound[int(resource_type)] = array

    def denoise(self, identifiers=None):
        ids = identifiers if identifiers is not None else self.instance.identifiers
        arr = (C.c_uint32 * len(ids))(*ids)
        self._check(self.lib.nrdHipDenoise(self.handle, arr, len(ids)), "nrdHipDenoise")

    def execute_raw(self, dispatch_ptr, num):
        self._check(self.lib.nrdHipExecuteDispatches(self.handle, C.cast(dispatch_ptr, C.c_void_p), num), "nrdHipExecuteDispatches")

    def read_pool_plane(self, pool, index):
        d = api.HipPlaneDesc()
        self._check(self.lib.nrdHipGetPoolPlane(self.handle, int(pool), index, C.byref(d)), "nrdHipGetPoolPlane")
        buf = (C.c_uint8 * (d.height * d.rowPitchBytes)).from_address(d.data)
        return np.frombuffer(buf, dtype=np.uint8).reshape(d.height, d.rowPitchBytes).copy(), api.Format(d.format), d.width

    def tile_fallback_stats(self):
        a, b = C.c_uint32(), C.c_uint32()
        self._check(self.lib.nrdHipGetTileFallbackStats(self.handle, C.byref(a), C.byref(b)), "nrdHipGetTileFallbackStats")
        return a.value, b.value

    def measure_motion_rows(self, ptr, n, row_begin=0, row_end=0xFFFFFFFF):
        out = C.c_float()
        self._check(self.lib.nrdHipMeasureMotionRows(self.handle, C.cast(ptr, C.c_void_p), n, row_begin, min(row_end, 0xFFFFFFFF), C.byref(out)), "nrdHipMeasureMotionRows")
        return out.value

    def set_history_reach_word(self, word):
        """word: a 1-element float32 CPU tensor / array (the emulated kernels run on the host) or None -- raytracingdenoiser_amd/executor.py set_history_reach_word"""
        ptr = 0 if word is None else (word.data_ptr() if hasattr(word, "data_ptr") else word.ctypes.data)
        self._check(self.lib.nrdHipSetHistoryReachWord(self.handle, C.c_void_p(ptr)), "nrdHipSetHistoryReachWord")

    def set_graph_mode(self, enable):
        pass  # graphs are a launch mechanism of the real runtime (graph == eager is a GPU test, tests/test_executor.py); the emulation always launches eagerly

    def destroy(self):
        if self.handle:
            self.lib.nrdHipDestroyExecutor(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


class EmuTorchExecutor(EmuExecutor):
    """EmuExecutor with the tensor-facing side of raytracingdenoiser_amd.executor.HipExecutor (planes as torch CPU tensors that alias host memory), so that the
    multi-GPU plumbing of raytracingdenoiser_amd/sharding.py -- which only ever talks to an executor through bind / _bound / pool_plane_tensor / execute_range --
    runs unmodified on the CPU emulation of the kernels: real processes, real gloo messages, emulated passes (tests/test_sharding.py)."""

    def __init__(self, instance, width, height):
        import torch

        super().__init__(instance, width, height)
        self._arena_tensor = torch.from_numpy(self.arena)  # shares the memory of the numpy arena

    def bind(self, resource_type, tensor, fmt):
        import torch

        if not isinstance(tensor, torch.Tensor):
            tensor = torch.from_numpy(tensor)
        assert not tensor.is_cuda and tensor[0].is_contiguous()
        desc = api.HipPlaneDesc(tensor.data_ptr(), tensor.stride(0) * tensor.element_size(), int(fmt), self.width, self.height)
        self._check(self.lib.nrdHipBindResource(self.handle, int(resource_type), C.byref(desc)), "nrdHipBindResource(%s)" % api.ResourceType(resource_type).name)
        self._bound[int(resource_type)] = tensor

    def pool_plane_desc(self, pool, index):
        d = api.HipPlaneDesc()
        self._check(self.lib.nrdHipGetPoolPlane(self.handle, int(pool), index, C.byref(d)), "nrdHipGetPoolPlane")
        return d

    def pool_plane_tensor(self, pool, index):
        d = self.pool_plane_desc(pool, index)
        off = d.data - self._arena_tensor.data_ptr()
        return self._arena_tensor[off: off + d.height * d.rowPitchBytes].view(d.height, d.rowPitchBytes)

    def set_owned_rows(self, row_begin, row_end):
        self._check(self.lib.nrdHipSetOwnedRows(self.handle, row_begin, row_end), "nrdHipSetOwnedRows")

    def execute_range(self, dispatch_ptr, num, first, count, row_begin=None, row_end=None):
        rb = row_begin if row_begin is None or isinstance(row_begin, C.Array) else (C.c_int32 * num)(*row_begin)
        re = row_end if row_end is None or isinstance(row_end, C.Array) else (C.c_int32 * num)(*row_end)
        self._check(self.lib.nrdHipExecuteDispatchRange(self.handle, C.cast(dispatch_ptr, C.c_void_p), num, first, count, rb, re), "nrdHipExecuteDispatchRange")


class _HostTensor:
    """what parity.py needs from a CUDA tensor, over a numpy array"""

    def __init__(self, a):
        self.a = a

    def cpu(self):
        return self

    def numpy(self):
        return self.a


class EmuRun:
    """tests/parity.py HipRun over the emulation"""

    def __init__(self, name, width, height, pad=0, validation=False):
        import parity
        import torch

        self.name, self.width, self.height, self.pad = name, width, height, pad
        self.inst = api.Instance([(0, parity.DENOISERS[name][0])], lib=load())
        self.ex = EmuExecutor(self.inst, width, height)
        self.outs = {}
        np_dtype = {torch.float16: np.float16, torch.int16: np.int16, torch.uint8: np.uint8}
        for rt, dtype, ch, fmt in parity.output_planes(name, width, height, validation):
            a = self._alloc((height, width, ch), np_dtype[dtype])
            self.outs[rt] = (a, fmt)
            self.ex.bind(rt, a, fmt)
        self.inputs = {}

    def _alloc(self, shape, dtype, fill=0):
        if not self.pad:
            return np.full(shape, fill, dtype=dtype)
        big = np.full((shape[0] + 1, shape[1] + self.pad) + tuple(shape[2:]), 77, dtype=dtype)
        view = big[1:, : shape[1]]
        view[...] = fill
        return view

    def step(self, frame, cs, settings=None):
        import parity

        for rt, t, fmt in parity.user_planes(self.name, frame):
            src = t.cpu().numpy()
            a = self._alloc(src.shape, src.dtype)
            a[...] = src
            self.inputs[rt] = _HostTensor(a)
            self.ex.bind(rt, a, fmt)
        if settings is not None:
            assert self.inst.set_denoiser_settings(0, settings) == api.Result.SUCCESS
        assert self.inst.set_common_settings(cs) == api.Result.SUCCESS
        self.ex.denoise()

    def output(self, rt):
        a, fmt = self.outs[rt]
        return (a.view(np.uint16) if a.dtype == np.int16 and fmt == api.Format.R16_UNORM else a).astype(np.float32)
