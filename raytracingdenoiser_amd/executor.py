"""Plumbing over include/NRDHip.h: device memory comes from torch (ROCm), everything else is the C-ABI.

There is no CPU path here: creating an executor without a visible GPU raises.
"""
import ctypes as C

import numpy as np

from . import api

_hip = None


def _hip_runtime():
    global _hip
    if _hip is None:
        _hip = C.CDLL("libamdhip64.so")
        _hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        _hip.hipMemcpy.restype = C.c_int
    return _hip


def texel_bytes(fmt):
    return api.FORMAT_BYTES[api.Format(fmt)]


class HipExecutor:
    """One nrd::Instance bound to pool planes in HBM; launches the pass chain on `stream` (a torch.cuda.Stream or None)."""

    def __init__(self, instance, width, height, stream=None):
        import torch

        if not torch.cuda.is_available():
            raise RuntimeError("NRD HIP executor needs a GPU (no CPU fallback exists)")
        self.instance = instance
        self.lib = instance.lib
        self.width, self.height = width, height
        self.stream = stream
        handle = C.c_void_p()
        stream_ptr = C.c_void_p(stream.cuda_stream) if stream is not None else C.c_void_p(torch.cuda.current_stream().cuda_stream)
        # the pool arena is a torch allocation, so pool planes can be aliased as tensors (needed for the RCCL all-gather)
        size = self.lib.nrdHipGetArenaSize(instance.handle, width, height)
        self.arena = torch.zeros(max(size, 256), dtype=torch.uint8, device="cuda")
        assert self.arena.data_ptr() % 256 == 0
        r = api.Result(self.lib.nrdHipCreateExecutorWithArena(instance.handle, width, height, stream_ptr, self.arena.data_ptr(), size, C.byref(handle)))
        if r != api.Result.SUCCESS:
            raise RuntimeError("nrdHipCreateExecutorWithArena failed: %s" % r.name)
        self.handle = handle
        self._bound = {}  # keeps tensors alive

    def _check(self, code, what):
        r = api.Result(code)
        if r != api.Result.SUCCESS:
            raise RuntimeError("%s failed: %s (%s)" % (what, r.name, self.lib.nrdHipGetLastError(self.handle).decode()))

    def bind(self, resource_type, tensor, fmt):
        """tensor: CUDA tensor whose rows are the plane rows (any dtype; [H, W, C] or [H, W]); rows must be dense, the row pitch may exceed
        the row size (e.g. a [:, :W] view of a wider allocation)."""
        assert tensor.is_cuda and tensor[0].is_contiguous()
        pitch = tensor.stride(0) * tensor.element_size()
        desc = api.HipPlaneDesc(tensor.data_ptr(), pitch, int(fmt), self.width, self.height)
        self._check(self.lib.nrdHipBindResource(self.handle, int(resource_type), C.byref(desc)), "nrdHipBindResource(%s)" % api.ResourceType(resource_type).name)
        self._bound[int(resource_type)] = tensor

    def denoise(self, identifiers=None):
        ids = identifiers if identifiers is not None else self.instance.identifiers
        arr = (C.c_uint32 * len(ids))(*ids)
        self._check(self.lib.nrdHipDenoise(self.handle, arr, len(ids)), "nrdHipDenoise")

    def execute_raw(self, dispatch_ptr, num):
        self._check(self.lib.nrdHipExecuteDispatches(self.handle, C.cast(dispatch_ptr, C.c_void_p), num), "nrdHipExecuteDispatches")

    def pool_plane_desc(self, pool, index):
        desc = api.HipPlaneDesc()
        self._check(self.lib.nrdHipGetPoolPlane(self.handle, int(pool), index, C.byref(desc)), "nrdHipGetPoolPlane")
        return desc

    def read_pool_plane(self, pool, index):
        """Synchronous device->host copy of a pool plane: returns (uint8 ndarray [h, pitch], Format, width)."""
        import torch

        torch.cuda.synchronize()
        d = self.pool_plane_desc(pool, index)
        host = np.empty((d.height, d.rowPitchBytes), dtype=np.uint8)
        err = _hip_runtime().hipMemcpy(host.ctypes.data_as(C.c_void_p), C.c_void_p(d.data), host.nbytes, 2)
        if err != 0:
            raise RuntimeError("hipMemcpy D2H failed: %d" % err)
        return host, api.Format(d.format), d.width

    def pool_plane_tensor(self, pool, index):
        """uint8 tensor view [h, pitch] aliasing a pool plane inside the arena (no copy)."""
        d = self.pool_plane_desc(pool, index)
        off = d.data - self.arena.data_ptr()
        return self.arena[off : off + d.height * d.rowPitchBytes].view(d.height, d.rowPitchBytes)

    def execute_range(self, dispatch_ptr, num, first, count, row_begin=None, row_end=None):
        """dispatches [first, first + count) of the list; row_begin / row_end: per-dispatch rows to produce (lists of length num) or None"""
        # lists are converted per call; a caller on a hot path passes ready-made ctypes arrays (sharding.HaloPlan.c_rows)
        rb = row_begin if row_begin is None or isinstance(row_begin, C.Array) else (C.c_int32 * num)(*row_begin)
        re = row_end if row_end is None or isinstance(row_end, C.Array) else (C.c_int32 * num)(*row_end)
        self._check(self.lib.nrdHipExecuteDispatchRange(self.handle, C.cast(dispatch_ptr, C.c_void_p), num, first, count, rb, re), "nrdHipExecuteDispatchRange")

    def set_owned_rows(self, row_begin, row_end):
        self._check(self.lib.nrdHipSetOwnedRows(self.handle, row_begin, row_end), "nrdHipSetOwnedRows")

    def set_graph_mode(self, enable):
        """one hipGraph launch per dispatch range instead of one launch per pass (include/NRDHip.h nrdHipSetGraphMode)"""
        self._check(self.lib.nrdHipSetGraphMode(self.handle, 1 if enable else 0), "nrdHipSetGraphMode")

    def graph_stats(self):
        """(graph launches, graphs built, node-parameter updates)"""
        a, b, c = C.c_uint64(), C.c_uint64(), C.c_uint64()
        self._check(self.lib.nrdHipGetGraphStats(self.handle, C.byref(a), C.byref(b), C.byref(c)), "nrdHipGetGraphStats")
        return a.value, b.value, c.value

    def tile_fallback_stats(self):
        """(tiles of the last frame left to a fallback kernel, tiles in total) -- include/NRDHip.h nrdHipGetTileFallbackStats"""
        a, b = C.c_uint32(), C.c_uint32()
        self._check(self.lib.nrdHipGetTileFallbackStats(self.handle, C.byref(a), C.byref(b)), "nrdHipGetTileFallbackStats")
        return a.value, b.value

    def measure_motion_rows(self, ptr, n, row_begin=0, row_end=0xFFFFFFFF):
        """largest vertical surface-motion reprojection distance (rows) over rows [row_begin, row_end) of this frame's rect -- include/NRDHip.h nrdHipMeasureMotionRows"""
        out = C.c_float()
        self._check(self.lib.nrdHipMeasureMotionRows(self.handle, C.cast(ptr, C.c_void_p), n, row_begin, min(row_end, 0xFFFFFFFF), C.byref(out)), "nrdHipMeasureMotionRows")
        return out.value

    def measure_motion_rows_async(self, ptr, n, row_begin, row_end, device_word):
        """the same measurement enqueued on the executor's stream: the result lands in device_word (a 1-element float32 CUDA tensor) in stream order, no host synchronisation
        -- include/NRDHip.h nrdHipMeasureMotionRowsAsync. The caller all-reduces the tensor and reads it once."""
        assert device_word.is_cuda and device_word.numel() == 1 and device_word.element_size() == 4
        self._check(self.lib.nrdHipMeasureMotionRowsAsync(self.handle, C.cast(ptr, C.c_void_p), n, row_begin, min(row_end, 0xFFFFFFFF), C.c_void_p(device_word.data_ptr())),
                    "nrdHipMeasureMotionRowsAsync")

    def set_history_reach_word(self, device_word):
        """device_word: a 1-element float32 CUDA tensor (kept alive by the caller) the temporal passes report their history reach into (rows; atomicMax) -- or None to switch the
        tracking off. include/NRDHip.h nrdHipSetHistoryReachWord."""
        assert device_word is None or (device_word.numel() == 1 and device_word.element_size() == 4)
        self._check(self.lib.nrdHipSetHistoryReachWord(self.handle, C.c_void_p(device_word.data_ptr() if device_word is not None else 0)), "nrdHipSetHistoryReachWord")

    def set_profiling(self, enable):
        self._check(self.lib.nrdHipSetProfiling(self.handle, 1 if enable else 0), "nrdHipSetProfiling")

    def collect_pass_timings(self):
        """{shaderFileName: (total_ms, launches)} for all dispatches since the last collect (synchronises the stream)."""
        n = len(self.instance.pipelines) + 1  # + the per-frame guide preparation (decode / shift kernels in front of the first pass), reported under GUIDE_PREPARATION
        idx, ms, cnt, written = (C.c_uint32 * n)(), (C.c_double * n)(), (C.c_uint32 * n)(), C.c_uint32()
        self._check(self.lib.nrdHipCollectPassTimings(self.handle, idx, ms, cnt, n, C.byref(written)), "nrdHipCollectPassTimings")
        names = list(self.instance.pipelines) + [self.GUIDE_PREPARATION]
        return {names[idx[i]]: (ms[i], cnt[i]) for i in range(written.value)}

    # absent (0 launches) when the tile-classification kernel of the list writes the guide planes itself: single-GPU REBLUR / RELAX lists (include/NRDHip.h)
    GUIDE_PREPARATION = "(guide planes: DecodeGuidesKernel)"

    def pool_memory(self):
        p, t = C.c_uint64(), C.c_uint64()
        self._check(self.lib.nrdHipGetPoolMemoryUsage(self.handle, C.byref(p), C.byref(t)), "nrdHipGetPoolMemoryUsage")
        return p.value, t.value

    def destroy(self):
        if self.handle:
            self.lib.nrdHipDestroyExecutor(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass
